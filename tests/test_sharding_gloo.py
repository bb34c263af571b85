"""CPU suite: the N>1 path (window sharding + detection gather) with world_size 2 on gloo."""
import os
import socket

import torch
import torch.distributed as dist
import torch.multiprocessing as mp

from dagr_amd import parallel


def _free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    p = s.getsockname()[1]
    s.close()
    return p


def _worker(rank, world, port, num_windows, out_dir):
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    mine = parallel.shard_indices(num_windows, rank, world)
    rows = []
    for w in mine:  # window w yields (w % 3) detections with recognisable content
        for k in range(w % 3):
            rows.append([float(w), float(k), float(w * 10 + k), 1.0, 2.0, 0.5, float(w % 2)])
    rows = torch.tensor(rows, dtype=torch.float32).reshape(-1, 7)
    allrows = parallel.restore_window_order(parallel.gather_detections(rows))
    torch.save(allrows, os.path.join(out_dir, f"rank{rank}.pt"))
    dist.barrier()
    dist.destroy_process_group()


def test_shard_and_gather_world2(tmp_path):
    world, num_windows = 2, 11
    port = _free_port()
    mp.spawn(_worker, args=(world, port, num_windows, str(tmp_path)), nprocs=world, join=True)
    got = [torch.load(os.path.join(tmp_path, f"rank{r}.pt")) for r in range(world)]
    want = []
    for w in range(num_windows):
        for k in range(w % 3):
            want.append([float(w), float(k), float(w * 10 + k), 1.0, 2.0, 0.5, float(w % 2)])
    want = torch.tensor(want, dtype=torch.float32)
    for g in got:
        assert torch.equal(g, want)


def test_shard_indices_cover_everything():
    for world in (1, 2, 4, 8):
        seen = sorted(i for r in range(world) for i in parallel.shard_indices(37, r, world))
        assert seen == list(range(37))
