#!/usr/bin/env python
"""tools/config_probe.py -- throughput of the other BASELINE.json configurations through bench.py's rig (three engines on
three streams, detections on the device, inputs resident): events/s and ms per step of B windows.  Builder tool.
usage: python tools/config_probe.py [model:use_image:stream:B:N[:max_neighbors] ...]   (default: config 4 =
dagr-l:1:uniform:8:100000; a sixth field sets the YAML key max_neighbors -- != 16 takes the generic level-0 kernel)"""
import json
import os
import sys
import time

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import bench  # noqa: E402
from dagr_amd.utils import synthetic as syn  # noqa: E402

dev = torch.device("cuda:0")
torch.cuda.set_device(0)
STEPS, WARM = int(os.environ.get("PROBE_STEPS", "30")), 8
with torch.no_grad():
    for spec in (sys.argv[1:] or ["dagr-l:1:uniform:8:100000"]):
        name, img, stream, B, N, *rest = spec.split(":")
        B, N, img = int(B), int(N), img == "1"
        over = dict(max_neighbors=int(rest[0])) if rest else {}
        rig = bench.Rig(640, 480, B, img, "resnet50", 3, dev, model_name=name, **over)
        slots = rig.make_slots(syn.uniform_window if stream == "uniform" else syn.edges_window, N, 3, seed=4234)
        for i in range(WARM):
            rig.step(i, slots)
        rig.drain()
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        for i in range(STEPS):
            rig.step(i, slots)
        rig.drain()
        torch.cuda.synchronize()
        ms = (time.perf_counter() - t0) * 1e3 / STEPS
        eng = rig.engines[0]
        eng.check_status()
        print(json.dumps({"spec": spec, "ms_per_step": round(ms, 4), "events_per_s": round(B * N / ms * 1e3, 1),
                          "levels": [(int(l.counts[0]), int(l.counts[1])) if hasattr(l, "counts") else None
                                     for l in getattr(eng, "levels", [])][:4]}), flush=True)
        del rig, slots
        torch.cuda.empty_cache()
