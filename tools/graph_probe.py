#!/usr/bin/env python
"""tools/graph_probe.py -- the window graph builder alone: time per build (HIP events) for a list of (stream, B, N)
workloads and, with PROBE_CHECK=1, a digest of the event-ordered edge_index + offset codes (to A/B two builds of the
library: DAGR_HIP_LIB=<path>; tools/graph_vs_oracle.py checks a build against the C oracle at full size).
Builder tool: python tools/graph_probe.py uniform:8:100000 edges:8:100000 ..."""
import hashlib
import json
import os
import sys

import numpy as np
import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from dagr_amd.graph.ev_graph import WindowGraphBuilder  # noqa: E402
from dagr_amd.utils import synthetic as syn  # noqa: E402

REPS = int(os.environ.get("PROBE_REPS", "20"))
W, H = int(os.environ.get("PROBE_W", "640")), int(os.environ.get("PROBE_H", "480"))
R = int(0.01 * W + 1)


def digest(g, nbr_src, nbr_code, deg):
    """event-ordered edge_index + offset codes (slot numbers of events beyond a pixel's FIFO depth are not stable)"""
    ei, rowptr = g.edge_index(nbr_src, deg)
    _, event_slot = g.node_order(int(deg.shape[0]))
    K = nbr_src.shape[1]
    es = event_slot.long().clamp(min=0)
    valid = torch.arange(K, device=deg.device)[None, :] < deg[es][:, None]
    h = hashlib.sha256()
    h.update(ei.cpu().numpy().tobytes())
    h.update(torch.where(valid, nbr_code[es], -1).cpu().numpy().tobytes())
    return h.hexdigest()[:16]


def main():
    dev = torch.device("cuda:0")
    out = []
    for spec in sys.argv[1:]:
        stream, B, N = spec.split(":")
        B, N = int(B), int(N)
        x, y, t, p, b = syn.batch_windows(syn.uniform_window if stream == "uniform" else syn.edges_window, N, B, W, H,
                                          seed=4234)
        pos = torch.from_numpy(syn.format_data_np(x, y, t, W, H)).to(dev)
        batch = torch.from_numpy(b).to(dev)
        g = WindowGraphBuilder(W, H, B, 16, 128, R, 10000, max_events=B * N, device=dev)
        outb = g.build(pos, batch)
        ne, flags = g.status()
        for _ in range(3):
            g.build(pos, batch, out=outb)
        torch.cuda.synchronize()
        a, c = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        a.record()
        for _ in range(REPS):
            g.build(pos, batch, out=outb)
        c.record()
        torch.cuda.synchronize()
        rec = {"spec": spec, "edges": ne, "flags": flags,
               "build_us": round(a.elapsed_time(c) / REPS * 1e3, 1)}
        try:        # device-side path counters of the last build (a library without the call: skipped)
            import ctypes
            from dagr_amd import _lib
            gc = (ctypes.c_int32 * 8)()
            _lib.check(_lib.lib().dagr_graph_counters(ctypes.byref(g.desc), _lib.ptr(g.workspace),
                                                      ctypes.cast(gc, ctypes.c_void_p), _lib.cur_stream(dev)), "counters")
            rec["deferred"], rec["ring_limited"] = int(gc[5]), int(gc[7])
        except Exception:
            pass
        if os.environ.get("PROBE_CHECK") == "1":
            rec["digest"] = digest(g, *outb)
        print(json.dumps(rec), flush=True)
        out.append(rec)
        del g, pos, batch, outb
    return out


if __name__ == "__main__":
    main()
