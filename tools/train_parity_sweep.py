#!/usr/bin/env python
"""One-off randomized training-parity sweep on the GPU box: HIP training forward + backward vs the CPU oracle
(tests/test_training_gpu.py helpers) over sensor sizes, batch sizes, event counts, model widths, head scales and seeds
beyond the fixed cases of the test-suite.  usage: python tools/train_parity_sweep.py [n_cases] [first_seed]"""
import os
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))
import numpy as np
import torch

import test_training_gpu as T
from oracle import train as otr
from dagr_amd.utils.buffers import format_data

n_cases = int(sys.argv[1]) if len(sys.argv) > 1 else 10
seed0 = int(sys.argv[2]) if len(sys.argv) > 2 else 500
rng = np.random.default_rng(seed0)
geoms = [(240, 180), (320, 215), (346, 260)]
t0 = time.time()
bad = 0
for k in range(n_cases):
    W, H = geoms[int(rng.integers(0, len(geoms)))]
    B = int(rng.integers(1, 5))
    n = int(rng.integers(800, 5000))
    over = {}
    kind = int(rng.integers(0, 4))
    if kind == 1:
        over = dict(num_scales=1, dataset="ncaltech101")
    elif kind == 2:
        over = dict(net_stem_width=1.0, yolo_stem_width=1.0, num_scales=1, dataset="ncaltech101")
    elif kind == 3:
        over = dict(net_stem_width=0.25, yolo_stem_width=0.25)
    seed = seed0 + k
    args, model, sd, batch, ev, b = T._training_case(W, H, B, n, seed, **over)
    ref = otr.training_losses(sd, args, H, W, ev[0], ev[1], ev[2], ev[3], b, B, batch.bbox, batch.bbox_batch)
    ref[0].backward()
    out = model(format_data(batch.cuda()))
    out["total_loss"].backward()
    dl = abs(float(out["total_loss"]) - float(ref[0])) / max(1.0, abs(float(ref[0])))
    params = dict(model.named_parameters())
    worst, cnt = (0.0, ""), 0
    for name, v in sd.items():
        if v.requires_grad and v.grad is not None and float(v.grad.abs().max()) > 0:
            worst = max(worst, (T._rel(params[name].grad, v.grad), name))
            cnt += 1
    # typical agreement is ~1e-5; the bar is 2e-3 of every tensor's own scale (the backward is deterministic).  A tensor
    # beyond it is put to the oracle itself: its gradient is recomputed with the weights perturbed by 1e-6 relative noise
    # (one fp32 rounding); if the ORACLE's own gradient of that tensor moves by as much, the case sits on a discrete
    # switch (a max-pool arg-max / ReLU within rounding distance) and the difference is the problem's, not the kernels'
    ok = dl < 5e-4 and worst[0] < 2e-3 and out["num_fg"] == ref[5]
    if not ok and dl < 5e-4 and out["num_fg"] == ref[5] and worst[0] < 5e-2:
        gen = torch.Generator().manual_seed(1)
        sd2 = {kk: ((v.detach() * (1 + 1e-6 * torch.randn(v.shape, generator=gen))).requires_grad_(True)
                    if v.requires_grad else v.detach().clone()) for kk, v in sd.items()}
        ref2 = otr.training_losses(sd2, args, H, W, ev[0], ev[1], ev[2], ev[3], b, B, batch.bbox, batch.bbox_batch)
        ref2[0].backward()
        own = T._rel(sd2[worst[1]].grad, sd[worst[1]].grad)
        print(f"   case {k}: {worst[1]} differs by {worst[0]:.1e}; the oracle's own gradient of that tensor moves by "
              f"{own:.1e} under 1e-6 relative weight noise -> {'a discrete switch within rounding distance' if own >= 0.5 * worst[0] else 'NOT explained'}")
        ok = own >= 0.5 * worst[0]
    bad += not ok
    print(f"case {k}: {W}x{H} B={B} n={n}/sample kind={kind} seed={seed}: loss rel {dl:.1e}, worst grad {worst[0]:.1e} "
          f"({worst[1]}), {cnt} tensors: {'ok' if ok else 'MISMATCH'} ({time.time() - t0:.0f} s)", flush=True)
print("sweep ok" if not bad else f"sweep: {bad} mismatching cases")
