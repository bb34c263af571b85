#!/usr/bin/env python
"""tools/pool_probe.py -- pool1's kernels alone (accumulate / whole stage) for (stream, B, N[, image]) workloads."""
import json
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import bench  # noqa: E402
from dagr_amd.utils import synthetic as syn  # noqa: E402

dev = torch.device("cuda:0")
with torch.no_grad():
    for spec in sys.argv[1:]:
        stream, B, N, img = spec.split(":")
        B, N, img = int(B), int(N), img == "1"
        rig = bench.Rig(640, 480, B, img, "resnet50", 1, dev)
        eng = rig.engines[0]
        pos, feat, batch, image = rig.make_slots(syn.uniform_window if stream == "uniform" else syn.edges_window, N, 1, seed=4234)[0]
        eng.forward_raw(pos, feat, batch, image=image)
        acc = bench.time_gpu(eng.pool1_accumulate_again, 30)
        eng.stage_pool1()
        full = bench.time_gpu(eng.stage_pool1, 30)
        print(json.dumps({"spec": spec, "exp": os.environ.get("DAGR_POOL_EXP", "0"), "grid_mult": os.environ.get("DAGR_POOL_GRID_MULT", "4"),
                          "accumulate_us": round(1e3 * acc, 1), "pool1_us": round(1e3 * full, 1)}), flush=True)
        del rig, eng
        torch.cuda.empty_cache()
