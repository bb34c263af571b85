#!/usr/bin/env python
"""tools/latency_trace.py -- a handful of single-window latency-mode runs (events-only, B and N from argv) with a marker
kernel before each timed window, for a rocprofv3 kernel trace: the per-window GPU timeline (busy vs gaps between
launches) tells whether a window is bound by the host issuing launches or by the kernels.  Builder tool.
usage: python tools/latency_trace.py [B] [N] [windows]"""
import os
import sys
import time

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import bench  # noqa: E402
from dagr_amd.utils import synthetic as syn  # noqa: E402

B = int(sys.argv[1]) if len(sys.argv) > 1 else 1
N = int(sys.argv[2]) if len(sys.argv) > 2 else 25000
WIN = int(sys.argv[3]) if len(sys.argv) > 3 else 6
dev = torch.device("cuda:0")
torch.cuda.set_device(0)
with torch.no_grad():
    rig = bench.Rig(640, 480, B, os.environ.get("PROBE_IMAGE", "0") == "1", "resnet50", 1, dev, low_latency=True)
    eng = rig.engines[0]
    slots = rig.make_slots(syn.uniform_window, N, 2, seed=4234)

    def window(i):
        pos, feat, batch, image = slots[i % 2]
        return eng.forward_detections(pos, feat, batch, image=image)     # as bench.py's latency sweep runs a window
    for i in range(20):
        window(i)
    torch.cuda.synchronize()
    host = []
    for i in range(WIN):
        torch.cuda.synchronize()
        torch.zeros(7, device=dev).cumsum(0)          # marker: a kernel that appears nowhere else
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        window(i)
        host.append(time.perf_counter() - t0)          # host time to ISSUE the window (no sync inside)
        torch.cuda.synchronize()
    print("host issue time per window (us):", [round(h * 1e6) for h in host])

