#!/usr/bin/env python
"""tools/train_probe.py -- training-step timing of BASELINE config 5's per-GPU share (dagr-l, N-Caltech101 settings: one
head scale, 240 x 180, per-GPU batch = 64 / 8 GPUs) on synthetic labelled samples: forward (module by module over
libdagr_hip with autograd) + loss + backward + clip + AdamW step + EMA.  Builder tool; prints one JSON line.
usage: python tools/train_probe.py [per_gpu_batch] [events_per_sample] [steps]"""
import json
import os
import sys
import time

import numpy as np
import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from dagr_amd.data import DataLoader  # noqa: E402
from dagr_amd.data.synthetic_data import SyntheticObjects  # noqa: E402
from dagr_amd.model.networks.dagr import DAGR  # noqa: E402
from dagr_amd.model.networks.ema import ModelEMA  # noqa: E402
from dagr_amd.utils.args import model_args  # noqa: E402
from dagr_amd.utils.buffers import format_data  # noqa: E402

B = int(sys.argv[1]) if len(sys.argv) > 1 else 8
N = int(sys.argv[2]) if len(sys.argv) > 2 else 50000
STEPS = int(sys.argv[3]) if len(sys.argv) > 3 else 10
dev = torch.device("cuda:0")
torch.manual_seed(0)
args = model_args("dagr-l", dataset="ncaltech101", num_scales=1, batch_size=B, n_nodes=N)
ds = SyntheticObjects(B * 4, N, seed=3)
model = DAGR(args, height=ds.height, width=ds.width).to(dev)
model.cache_luts(width=ds.width, height=ds.height, radius=args.radius)
ema = ModelEMA(model)
opt = torch.optim.AdamW(model.parameters(), lr=1e-3 * np.sqrt(64) / np.sqrt(64), weight_decay=1e-5, fused=True)
batches = [b.to(dev) for b in DataLoader(ds, batch_size=B, follow_batch=["bbox"])]
model.train()


def step(batch):
    data = format_data(batch.clone())
    opt.zero_grad(set_to_none=True)
    out = model(data)
    out["total_loss"].backward()
    torch.nn.utils.clip_grad_value_(model.parameters(), 0.1)
    opt.step()
    ema.update(model)
    return out


for k in range(3):
    out = step(batches[k % len(batches)])
torch.cuda.synchronize()
t0 = time.perf_counter()
losses = []
for k in range(STEPS):
    losses.append(step(batches[k % len(batches)])["total_loss"].detach())
torch.cuda.synchronize()
dt = (time.perf_counter() - t0) / STEPS
fwd = []
for k in range(3):
    torch.cuda.synchronize()
    t1 = time.perf_counter()
    with torch.no_grad():
        model(format_data(batches[k % len(batches)].clone()))
    torch.cuda.synchronize()
    fwd.append(time.perf_counter() - t1)
n_events = int(np.mean([b.pos.shape[0] for b in batches]))
print(json.dumps({"workload": f"dagr-l ncaltech (1 scale), 240x180, per-GPU batch {B} x {N} events, synthetic objects",
                  "ms_per_step": round(dt * 1e3, 2), "samples_per_s": round(B / dt, 1), "events_per_s": round(n_events / dt),
                  "forward_only_ms": round(min(fwd) * 1e3, 2), "loss_first_last": [float(losses[0]), float(losses[-1])],
                  "max_mem_GB": round(torch.cuda.max_memory_allocated() / 1e9, 2),
                  "params": sum(p.numel() for p in model.parameters())}), flush=True)
