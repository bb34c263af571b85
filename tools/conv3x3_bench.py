#!/usr/bin/env python
"""tools/conv3x3_bench.py -- the 3x3 convolutions of ResNet-50 at 640x480 (fp32): MIOpen through torch in channels-last
(what the inference copy runs) against contiguous NCHW (where MIOpen's Winograd solvers live).  Builder tool."""
import json
import sys

import torch

torch.backends.cudnn.benchmark = True
dev = torch.device("cuda:0")
B = int(sys.argv[1]) if len(sys.argv) > 1 else 8


def timed(fn, n=20):
    for _ in range(5):
        fn()
    torch.cuda.synchronize()
    a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    a.record()
    for _ in range(n):
        fn()
    b.record()
    torch.cuda.synchronize()
    return a.elapsed_time(b) / n * 1e3


with torch.no_grad():
    for C, H, W, s in ((64, 120, 160, 1), (128, 120, 160, 2), (128, 60, 80, 1), (256, 60, 80, 2), (256, 30, 40, 1),
                       (512, 30, 40, 2), (512, 15, 20, 1)):
        x = torch.randn(B, C, H, W, device=dev)
        w = torch.randn(C, C, 3, 3, device=dev) * 0.05
        xl, wl = x.contiguous(memory_format=torch.channels_last), w.contiguous(memory_format=torch.channels_last)
        gf = 2.0 * B * (H // s) * (W // s) * 9 * C * C / 1e9
        t_cl = timed(lambda: torch.nn.functional.conv2d(xl, wl, None, s, 1))
        t_nc = timed(lambda: torch.nn.functional.conv2d(x, w, None, s, 1))
        t_tr = timed(lambda: xl.contiguous())            # an NHWC -> NCHW copy of the input, for scale
        print(json.dumps(dict(B=B, C=C, H=H, W=W, stride=s, gflop=round(gf, 2), nhwc_us=round(t_cl, 1), nchw_us=round(t_nc, 1),
                              nhwc_tf=round(gf / t_cl * 1e3, 1), nchw_tf=round(gf / t_nc * 1e3, 1), layout_copy_us=round(t_tr, 1))),
              flush=True)
