#!/usr/bin/env python
"""tools/img_branch_probe.py -- the image branch alone (B, resnet50, 640x480): ms per forward, eager and as a captured graph."""
import json
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import bench  # noqa: E402

dev = torch.device("cuda:0")
torch.backends.cudnn.benchmark = True
out = {}
with torch.no_grad():
    for B in (8, 1):
        rig = bench.Rig(640, 480, B, True, "resnet50", 1, dev)
        eng = rig.engines[0]
        img = (torch.randint(0, 256, (B, 3, 480, 640), dtype=torch.uint8).float() / 255).to(dev)
        out[f"B{B}_eager_ms"] = round(bench.time_gpu(lambda: eng.stage_image(img), 10, warm=3), 4)
        g = torch.cuda.CUDAGraph()
        static = img.clone()
        with torch.cuda.graph(g):
            eng.stage_image(static)
        out[f"B{B}_graph_ms"] = round(bench.time_gpu(g.replay, 10, warm=2), 4)
        del rig, eng, g
        torch.cuda.empty_cache()
print(json.dumps(dict(lt_residual=os.environ.get("DAGR_LT_RESIDUAL", "1"), **out)))
