#!/usr/bin/env python
"""Known-byte-count kernels for FETCH_SIZE / WRITE_SIZE calibration (run under rocprofv3 --pmc)."""
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from dagr_amd import _lib  # noqa: E402

L = _lib.lib()
n = 1 << 28  # 1 GiB of floats: past the 256 MiB Infinity Cache
buf = torch.ones(n, dtype=torch.float32, device="cuda")
sink = torch.zeros(1, dtype=torch.float32, device="cuda")
st = _lib.cur_stream()
for mode in (0, 1, 2, 3):
    for _ in range(3):
        _lib.check(L.dagr_debug_calibrate(mode, _lib.ptr(buf), n, 1 << 24, _lib.ptr(sink), st))
torch.cuda.synchronize()
print("bytes: read4/read16/write4 =", n * 4, " gather64 =", (1 << 24) * 64)
