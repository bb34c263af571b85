#!/usr/bin/env python
"""tools/train_syncs.py -- where a training step synchronises the host with the device and where its launches come from
(same workload as tools/train_probe.py): torch's sync-debug warnings aggregated by source line, then torch.profiler's
operator counts aggregated by the repository line that issued them.  Builder tool."""
import collections
import os
import sys
import traceback
import warnings

import torch
from torch.profiler import ProfilerActivity, profile

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
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
    data = format_data(batch.clone())
    opt.zero_grad(set_to_none=True)
    out = model(data)
    out["total_loss"].backward()
    torch.nn.utils.clip_grad_value_(model.parameters(), 0.1)
    opt.step()
    ema.update(model)


for k in range(3):
    step(batches[k % 4])
torch.cuda.synchronize()

# ---- 1. synchronising calls, by the repository frame that made them
sites = collections.Counter()
orig = warnings.showwarning


def show(message, category, filename, lineno, file=None, line=None):
    if "synchroniz" not in str(message):
        return
    frames = [f for f in traceback.extract_stack() if f.filename.startswith(ROOT) and "train_syncs" not in f.filename]
    key = " <- ".join(f"{os.path.relpath(f.filename, ROOT)}:{f.lineno}" for f in frames[-2:][::-1]) or f"{filename}:{lineno}"
    sites[key] += 1


warnings.showwarning = show
warnings.simplefilter("always")
torch.cuda.set_sync_debug_mode(1)
step(batches[0])
torch.cuda.set_sync_debug_mode(0)
warnings.showwarning = orig
torch.cuda.synchronize()
print(f"# synchronising torch calls in one step: {sum(sites.values())}")
for k, v in sites.most_common(60):
    print(f"{v:4d}  {k}")

# ---- 2. operators by issuing repository line
with profile(activities=[ProfilerActivity.CPU, ProfilerActivity.CUDA], with_stack=True) as prof:
    step(batches[1])
    torch.cuda.synchronize()
lines = collections.Counter()
cpu_us = collections.Counter()
n_launch = 0
for ev in prof.events():
    if ev.name in ("hipLaunchKernel", "hipExtModuleLaunchKernel", "hipMemcpyAsync", "hipMemcpyWithStream", "hipMemsetAsync"):
        n_launch += 1
    if not ev.name.startswith("aten::") or ev.cpu_parent is not None and ev.cpu_parent.name.startswith("aten::"):
        continue
    st = [s for s in (ev.stack or []) if ROOT in s and "train_syncs" not in s]
    key = st[0].replace(ROOT + "/", "") if st else "(outside the repository: autograd / optimizer)"
    lines[key] += 1
    cpu_us[key] += ev.cpu_time_total
print(f"\n# driver calls (launches, copies, memsets) in one step: {n_launch}")
print("# top-level aten operators by issuing line: count, host us")
for k, v in sorted(cpu_us.items(), key=lambda kv: -kv[1])[:70]:
    print(f"{lines[k]:4d} {v:9.0f}  {k}")
