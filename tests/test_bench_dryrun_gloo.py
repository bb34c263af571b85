"""CPU suite: bench.py's multi-rank control flow (warm-up, barrier-bracketed timed region, variable-length detection
gather, MAX-over-ranks time, per-rank report) with world_size 2 on gloo and a stand-in rig -- so that the first real
N-GPU launch cannot fail on plumbing.  The stand-in replaces the model only; `timed_run`, `detections_rows` and
`dagr_amd.parallel.gather_detections` are the shipped code."""
import json
import os
import socket
import sys

import torch
import torch.distributed as dist
import torch.multiprocessing as mp

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)


def _free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    p = s.getsockname()[1]
    s.close()
    return p


class _FakeEngine:
    def check_status(self):
        pass


class _FakeRig:
    """Deterministic per-(rank, step) detections: image b of step i on rank r keeps (r + i + b) % 4 rows."""
    B, A = 3, 8

    def __init__(self, rank):
        self.rank, self.dev = rank, torch.device("cpu")
        self.engines = [_FakeEngine()]
        self.streams = []

    def step(self, i, slots):
        det = torch.zeros((self.B, self.A, 6))
        n = torch.tensor([(self.rank + i + b) % 4 for b in range(self.B)], dtype=torch.int32)
        for b in range(self.B):
            for k in range(int(n[b])):
                det[b, k] = torch.tensor([1.0 * k, 2.0, 3.0 + k, 4.0, 0.5, float(self.rank)])
        return det, n

    def drain(self):
        pass


def _worker(rank, world, port, out_dir):
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port))
    dist.init_process_group("gloo", rank=rank, world_size=world)
    import bench
    run = bench.timed_run(_FakeRig(rank), None, steps=5, warmup=2, dist=dist, world=world, rank=rank)
    json.dump(run, open(os.path.join(out_dir, f"rank{rank}.json"), "w"))
    dist.barrier()
    dist.destroy_process_group()


def test_timed_run_control_flow_world2(tmp_path):
    world = 2
    mp.spawn(_worker, args=(world, _free_port(), str(tmp_path)), nprocs=world, join=True)
    runs = [json.load(open(os.path.join(tmp_path, f"rank{r}.json"))) for r in range(world)]
    want = sum((r + i + b) % 4 for r in range(world) for i in range(5) for b in range(3))
    for run in runs:
        assert run["n_detections"] == want                      # every rank holds the whole job's detections
        assert len(run["per_rank"]) == world and all(len(p) == 2 for p in run["per_rank"])
        assert run["elapsed"] >= max(p[0] for p in run["per_rank"]) - 1e-3
    assert runs[0]["elapsed"] == runs[1]["elapsed"]              # MAX over ranks, identical everywhere


def test_detections_rows_cut_and_window_ids():
    import bench
    rig = _FakeRig(1)
    res = [rig.step(i, None) for i in range(4)]
    rows = bench.detections_rows(res, rank=1, B=rig.B)
    assert rows.shape == (sum((1 + i + b) % 4 for i in range(4) for b in range(3)), 7)
    # window id = rank*K*B + step*B + image
    wid = rows[:, 0].long()
    assert wid.min() >= 12 and wid.max() < 24
    for i in range(4):
        for b in range(3):
            assert int((wid == 12 + i * 3 + b).sum()) == (1 + i + b) % 4
