#!/usr/bin/env python
"""tools/tail_probe.py -- isolated timings of every launch group after pool1 (layer2..5 conv pairs, pool2..4, the head
convs) on one engine, so that a kernel variant can be A/B-ed without a profiler.  Builder tool.
usage: python tools/tail_probe.py [stream:B:N ...]   (default uniform:8:100000 uniform:1:25000)"""
import ctypes
import json
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import bench  # noqa: E402
from dagr_amd import _lib  # noqa: E402
from dagr_amd.utils import synthetic as syn  # noqa: E402

dev = torch.device("cuda:0")
torch.cuda.set_device(0)
REPS = int(os.environ.get("PROBE_REPS", "30"))


def timed(fn):
    for _ in range(3):
        fn()
    torch.cuda.synchronize()
    a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    a.record()
    for _ in range(REPS):
        fn()
    b.record()
    torch.cuda.synchronize()
    return round(a.elapsed_time(b) / REPS * 1e3, 1)      # us


with torch.no_grad():
    for spec in (sys.argv[1:] or ["uniform:8:100000", "uniform:1:25000"]):
        stream, B, N = spec.split(":")
        B, N = int(B), int(N)
        rig = bench.Rig(640, 480, B, False, "resnet50", 1, dev)
        slots = rig.make_slots(syn.uniform_window if stream == "uniform" else syn.edges_window, N, 1, seed=4234)
        eng = rig.engines[0]
        pos, feat, batch, _ = slots[0]
        eng.forward_raw(pos, feat, batch)
        torch.cuda.synchronize()
        P = _lib.ptr
        st = _lib.cur_stream(dev)
        out = {"spec": spec, "levels": [[int(v) for v in l.counts.tolist()] for l in eng.levels]}
        for k in range(4):
            lvl = eng.levels[k]
            c1, c2 = eng.packs[k]
            dom = eng.dom[k + 1]
            ldx = lvl.x.shape[1]
            ldh = lvl.hp.shape[1]
            out[f"L{k + 1}c1_{c1.cin}->{c1.N}"] = timed(lambda: eng._conv_generic(lvl, c1, P(lvl.x), ldx, None, 0, P(lvl.h1), c1.N, dom, st))
            out[f"L{k + 1}c2_{c2.cin}+{c2.cskip}->{c2.N}"] = timed(lambda: eng._conv_generic(lvl, c2, P(lvl.h1), c1.N, P(lvl.x), ldx, P(lvl.hp), ldh, dom, st))
            if k < 3:
                out[f"pool{k + 2}"] = timed(lambda: eng._stage_pool(k))
        for i in range(len(eng.head_levels)):
            out[f"head{i + 1}"] = timed(lambda: eng._stage_head_scale(i))
        out["tail"] = timed(eng.stage_tail)
        out["tail_head"] = timed(lambda: eng._tail_and_head())
        print(json.dumps(out), flush=True)
        del rig, slots, eng
        torch.cuda.empty_cache()
