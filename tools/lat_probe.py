#!/usr/bin/env python
"""tools/lat_probe.py -- bench.py's per-window latency table for a few sizes (events-only and, with IMG=1, with the image branch)
and the asynchronous-update leg."""
import json
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import bench  # noqa: E402

dev = torch.device("cuda:0")
torch.backends.cudnn.benchmark = True
Ns = [int(v) for v in (sys.argv[1] if len(sys.argv) > 1 else "25000,100000").split(",")]
with torch.no_grad():
    out = {"events_only": bench.latency_sweep(640, 480, False, "resnet50", dev, Ns, 20, 100)}
    if os.environ.get("IMG", "0") == "1":
        out["image"] = bench.latency_sweep(640, 480, True, "resnet50", dev, Ns[:1], 10, 40)
    if os.environ.get("ASYNC", "1") == "1":
        out["async_update"] = bench.async_update_leg(640, 480, dev)
print(json.dumps(out))
