#!/usr/bin/env python
"""tools/train_profile.py -- torch.profiler over a few training steps (same workload as tools/train_probe.py): operators by
device time and by call count, forward and backward apart.  Builder tool."""
import os
import sys

import numpy as np
import torch
from torch.profiler import ProfilerActivity, profile, record_function

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from dagr_amd.data import DataLoader  # noqa: E402
from dagr_amd.data.synthetic_data import SyntheticObjects  # noqa: E402
from dagr_amd.model.networks.dagr import DAGR  # noqa: E402
from dagr_amd.model.networks.ema import ModelEMA  # noqa: E402
from dagr_amd.utils.args import model_args  # noqa: E402
from dagr_amd.utils.buffers import format_data  # noqa: E402

B, N = 8, 50000
dev = torch.device("cuda:0")
torch.manual_seed(0)
args = model_args("dagr-l", dataset="ncaltech101", num_scales=1, batch_size=B, n_nodes=N)
ds = SyntheticObjects(B * 4, N, seed=3)
model = DAGR(args, height=ds.height, width=ds.width).to(dev)
model.cache_luts(width=ds.width, height=ds.height, radius=args.radius)
ema = ModelEMA(model)
opt = torch.optim.AdamW(model.parameters(), lr=1e-3, weight_decay=1e-5, fused=True)
batches = [b.to(dev) for b in DataLoader(ds, batch_size=B, follow_batch=["bbox"])]
model.train()


def step(batch):
    with record_function("S.format"):
        data = format_data(batch.clone())
    opt.zero_grad(set_to_none=True)
    with record_function("S.forward"):
        out = model(data)
    with record_function("S.backward"):
        out["total_loss"].backward()
    with record_function("S.optim"):
        torch.nn.utils.clip_grad_value_(model.parameters(), 0.1)
        opt.step()
        ema.update(model)


for k in range(3):
    step(batches[k % 4])
torch.cuda.synchronize()
with profile(activities=[ProfilerActivity.CPU, ProfilerActivity.CUDA]) as prof:
    for k in range(3):
        step(batches[k % 4])
    torch.cuda.synchronize()
ka = prof.key_averages()
print(ka.table(sort_by="self_cuda_time_total", row_limit=28, max_name_column_width=60))
print(ka.table(sort_by="self_cpu_time_total", row_limit=28, max_name_column_width=60))
print(ka.table(sort_by="count", row_limit=20, max_name_column_width=60))
