#!/usr/bin/env python
"""tools/train_sections.py -- where a training step's time goes: synchronised wall time (host + device) per module of the
forward, the loss, backward, clipping, optimizer and EMA.  Same workload as tools/train_probe.py.  Builder tool.
usage: python tools/train_sections.py [per_gpu_batch] [events_per_sample] [steps]"""
import collections
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
STEPS = int(sys.argv[3]) if len(sys.argv) > 3 else 6
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

acc = collections.OrderedDict()
stack = []


def tic(name):
    torch.cuda.synchronize()
    stack.append((name, time.perf_counter()))


def toc():
    torch.cuda.synchronize()
    name, t0 = stack.pop()
    acc[name] = acc.get(name, 0.0) + (time.perf_counter() - t0)


def hook(mod, name):
    mod.register_forward_pre_hook(lambda m, i: tic(name))
    mod.register_forward_hook(lambda m, i, o: toc())


bb = model.backbone
hook(bb.events_to_graph, "fwd.events_to_graph")
hook(bb.edge_attrs, "fwd.cartesian")
for n in bb.LAYER_NAMES:
    hook(getattr(bb, n), "fwd." + n)
for k in range(1, 5):
    hook(getattr(bb, f"pool{k}"), f"fwd.pool{k}")
hook(model.head, "fwd.head+loss")


def step(batch, timed):
    T = tic if timed else (lambda n: None)
    E = toc if timed else (lambda: None)
    T("format_data"); data = format_data(batch.clone()); E()
    opt.zero_grad(set_to_none=True)
    T("forward(total)"); out = model(data); E()
    T("backward"); out["total_loss"].backward(); E()
    T("clip"); torch.nn.utils.clip_grad_value_(model.parameters(), 0.1); E()
    T("opt.step"); opt.step(); E()
    T("ema"); ema.update(model); E()
    return out


for k in range(3):
    step(batches[k % len(batches)], False)
acc.clear()
for k in range(STEPS):
    step(batches[k % len(batches)], True)
out = {k: round(v / STEPS * 1e3, 3) for k, v in acc.items()}
out["sum_top_level_ms"] = round(sum(out[k] for k in ("format_data", "forward(total)", "backward", "clip", "opt.step", "ema")), 2)
print(json.dumps(out), flush=True)
