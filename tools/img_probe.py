#!/usr/bin/env python
"""tools/img_probe.py -- steady-state time of the image branch alone (ResNet-50 trunk + CNN head of the bench workload:
B = 8 frames of 640 x 480, fp32, channels-last inference copy) under whatever MIOpen settings the environment carries.
Builder tool for A/B-ing MIOpen solver choices (PyTorch-ROCm side of the path).  Prints one JSON line.
usage: [MIOPEN_...=..] python tools/img_probe.py [tag] [benchmark:0|1]"""
import json
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch  # noqa: E402

from dagr_amd.model.networks.dagr import DAGR  # noqa: E402
from dagr_amd.utils.args import model_args  # noqa: E402
from dagr_amd.utils.testing_weights import randomize_  # noqa: E402

tag = sys.argv[1] if len(sys.argv) > 1 else "default"
torch.backends.cudnn.benchmark = (sys.argv[2] if len(sys.argv) > 2 else "1") == "1"
a = model_args("dagr-s", batch_size=8, use_image=True, img_net="resnet50")
m = randomize_(DAGR(a, height=480, width=640)).eval().cuda()
eng = m.engine()
img = torch.rand(8, 3, 480, 640, device="cuda")
with torch.no_grad():
    for _ in range(4):
        eng._image_branch(img)
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    ts = []
    for _ in range(5):
        e0.record()
        for _ in range(10):
            eng._image_branch(img)
        e1.record()
        torch.cuda.synchronize()
        ts.append(e0.elapsed_time(e1) / 10)
print(json.dumps({"tag": tag, "ms": round(min(ts), 4), "ms_all": [round(t, 3) for t in ts],
                  "env": {k: v for k, v in os.environ.items() if k.startswith("MIOPEN")}}), flush=True)
