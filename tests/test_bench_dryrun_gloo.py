"""CPU suite: bench.py's multi-rank control flow (warm-up, barrier-bracketed timed region, variable-length detection
gather, MAX-over-ranks time, per-rank report) with world_size 2 on gloo and a stand-in rig -- so that the first real
N-GPU launch cannot fail on plumbing.  The stand-in replaces the model only; `timed_run`, `detections_rows` and
`dagr_amd.parallel.gather_detections` are the shipped code."""
import json
import os
import socket
import subprocess
import sys

import torch
import torch.distributed as dist
import torch.multiprocessing as mp

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)

import bench  # noqa: E402

_FakeRig = bench.DryRunRig     # the stand-in `python bench.py --dry-run-gloo` runs


def _free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    p = s.getsockname()[1]
    s.close()
    return p



def _worker(rank, world, port, out_dir):
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port))
    dist.init_process_group("gloo", rank=rank, world_size=world)
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


def _bench(*flags):
    env = {k: v for k, v in os.environ.items() if k not in ("WORLD_SIZE", "RANK", "LOCAL_RANK", "MASTER_ADDR", "MASTER_PORT")}
    return subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), *flags], capture_output=True, text=True,
                          env=env, timeout=600)


def test_gpus_flag_launches_the_ranks_itself():
    """`python bench.py --gpus 2` without a launcher's environment starts two ranks (torch.distributed.run on
    127.0.0.1) of the same command line: rendezvous, timed region, gather and the per-rank report all run (gloo)."""
    p = _bench("--gpus", "2", "--dry-run-gloo", "--steps", "4", "--warmup", "1")
    assert p.returncode == 0, p.stderr[-2000:]
    line = json.loads([l for l in p.stdout.splitlines() if l.startswith("{")][-1])
    assert line["n_gpus"] == 2 and line["world"] == 2 and line["ranks_seen"] == 2
    assert len(line["per_rank"]) == 2 and line["steps"] == 4
    assert line["gather"]["detections"] == sum((r + i + b) % 4 for r in range(2) for i in range(4) for b in range(3))


def test_gpus_flag_refuses_fewer_devices_than_ranks():
    """No silent single-rank run: on a box with fewer devices than --gpus (here: none) the command fails loudly."""
    p = _bench("--gpus", "2", "--steps", "1", "--warmup", "0", "--no-latency", "--no-cpu-baseline")
    assert p.returncode != 0
    assert "2 ranks requested" in p.stderr and "device(s) visible" in p.stderr


def test_world_size_must_match_gpus_flag():
    env = dict(os.environ, WORLD_SIZE="2", RANK="0", LOCAL_RANK="0", MASTER_ADDR="127.0.0.1", MASTER_PORT=str(_free_port()))
    p = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "1"], capture_output=True, text=True,
                       env=env, timeout=120)
    assert p.returncode != 0 and "--gpus 1 but the launcher started 2" in p.stderr


def test_eight_ranks_as_the_driver_launches_them():
    """The driver's scaling run: `python -m torch.distributed.run --nnodes=1 --nproc-per-node 8 --master-addr 127.0.0.1
    --master-port P bench.py --gpus 8 ...` -- here with the stand-in rig on gloo: eight ranks rendezvous, every rank takes
    part in the collectives, rank 0 prints ONE line whose value is the whole job's."""
    env = {k: v for k, v in os.environ.items() if k not in ("RANK", "WORLD_SIZE", "LOCAL_RANK", "MASTER_ADDR", "MASTER_PORT")}
    p = subprocess.run([sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", "8",
                        "--master-addr", "127.0.0.1", "--master-port", str(_free_port()), os.path.join(ROOT, "bench.py"),
                        "--gpus", "8", "--dry-run-gloo", "--steps", "3", "--warmup", "1"],
                       capture_output=True, text=True, env=env, timeout=900)
    assert p.returncode == 0, p.stderr[-2000:]
    lines = [l for l in p.stdout.splitlines() if l.startswith("{")]
    assert len(lines) == 1, lines
    line = json.loads(lines[0])
    assert line["n_gpus"] == 8 and line["world"] == 8 and line["ranks_seen"] == 8 and len(line["per_rank"]) == 8
    assert line["scaling"] == "weak" and line["steps"] == 3
    assert all(set(r) == {"compute_ms", "gather_ms"} for r in line["per_rank"])
    assert line["gather"]["detections"] == sum((r + i + b) % 4 for r in range(8) for i in range(3) for b in range(3))
    # every rank set itself up apart from the others before touching anything: its own MIOpen user db (eight processes
    # running the benchmark-mode find at start-up would otherwise write one file) and its own slice of the host cores
    envs = sorted(line["rank_env"], key=lambda e: e["rank"])
    assert [e["rank"] for e in envs] == list(range(8))
    assert len({e["miopen_db"] for e in envs}) == 8 and all(e["miopen_db"].endswith(f"miopen_rank{e['rank']}") for e in envs)
    cores = [tuple(e["cores"]) for e in envs]
    n_host = len(os.sched_getaffinity(0))
    assert all(len(c) == max(1, n_host // 8) for c in cores)
    if n_host >= 8:
        assert len(set(c for cs in cores for c in cs)) == sum(len(c) for c in cores)      # disjoint slices
    # start-up (interpreter + imports + environment, to the first collective) of eight ranks started together
    assert max(e["startup_s"] for e in envs) < 120


def test_rank_environment_is_a_no_op_for_one_rank(monkeypatch):
    from dagr_amd import parallel
    for k in ("LOCAL_RANK", "LOCAL_WORLD_SIZE", "WORLD_SIZE"):
        monkeypatch.delenv(k, raising=False)
    before = os.sched_getaffinity(0)
    out = parallel.rank_environment()
    assert out["miopen_db"] is None and out["cores"] is None and os.sched_getaffinity(0) == before
    plan = parallel.rank_environment(local_rank=3, local_world=8, apply=False)
    assert plan["miopen_db"].endswith("miopen_rank3") and len(plan["cores"]) == max(1, len(before) // 8)
    assert os.sched_getaffinity(0) == before
