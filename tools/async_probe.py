#!/usr/bin/env python
"""tools/async_probe.py -- bench.py's f3 leg alone: microseconds per asynchronous update vs re-evaluating the window."""
import json
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import bench  # noqa: E402

with torch.no_grad():
    nw = int(sys.argv[1]) if len(sys.argv) > 1 else 25000
    print(json.dumps(bench.async_update_leg(640, 480, torch.device("cuda:0"), n_window=nw)))
