#!/usr/bin/env python
"""tools/stage_probe.py -- per-stage timings (bench.py:stage_timings) of one engine for a list of (stream, B, N)
workloads; prints one JSON line each.  Builder tool: python tools/stage_probe.py uniform:8:200000 edges:8:100000 ..."""
import json
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import bench  # noqa: E402
from dagr_amd.utils import synthetic as syn  # noqa: E402

use_image = os.environ.get("PROBE_IMAGE", "0") == "1"
dev = torch.device("cuda:0")
torch.cuda.set_device(0)
with torch.no_grad():
    for spec in sys.argv[1:]:
        stream, B, N = spec.split(":")
        B, N = int(B), int(N)
        rig = bench.Rig(640, 480, B, use_image, "resnet50", 1, dev)
        slots = rig.make_slots(syn.uniform_window if stream == "uniform" else syn.edges_window, N, 1, seed=4234)
        st = bench.stage_timings(rig, slots, B * N)
        print(json.dumps({"spec": spec, "edges": st["edges_per_step"], "levels": st["levels"],
                          "stages_ms": {k: v["ms"] for k, v in st["stages"].items()}}), flush=True)
        del rig, slots
        torch.cuda.empty_cache()
