#!/usr/bin/env python
"""tools/graph_vs_oracle.py -- full-size window batches: edge_index of the HIP builder against the C oracle, sample by
sample (the oracle builds a 100 k-event window in ~1.5 s).  Prints the first differing destination.
usage: python tools/graph_vs_oracle.py edges:8:100000 [uniform:8:100000 ...]"""
import os
import sys

import numpy as np
import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from oracle import graph as og  # noqa: E402
from dagr_amd.graph.ev_graph import WindowGraphBuilder  # noqa: E402
from dagr_amd.utils import synthetic as syn  # noqa: E402

W, H = 640, 480
R = 7
dev = torch.device("cuda:0")
bad = 0
for spec in sys.argv[1:]:
    stream, B, N = spec.split(":")
    B, N = int(B), int(N)
    seed = int(os.environ.get("PROBE_SEED", "4234"))
    x, y, t, p, b = syn.batch_windows(syn.uniform_window if stream == "uniform" else syn.edges_window, N, B, W, H, seed=seed)
    pos = torch.from_numpy(np.stack([x, y, t], -1).astype(np.int32)).to(dev)
    batch = torch.from_numpy(b).to(dev)
    g = WindowGraphBuilder(W, H, B, 16, 128, R, 10000, max_events=B * N, device=dev)
    nbr_src, nbr_code, deg = g.build(pos, batch)
    ei, rowptr = g.edge_index(nbr_src, deg)
    ei, rowptr = ei.cpu().numpy(), rowptr.cpu().numpy().astype(np.int64)
    for s in range(B):
        sel = np.nonzero(b == s)[0]
        lo, hi = sel[0], sel[-1] + 1
        ref = og.build_window_graph(x[sel], y[sel], t[sel], np.zeros(len(sel), np.int32), W, H, 1, R, 10000, K=16, Q=128) + lo
        part = ei[:, rowptr[lo]:rowptr[hi]]
        ok = part.shape == ref.shape and (part == ref).all()
        print(f"{spec} sample {s}: {'ok' if ok else 'MISMATCH'} E={ref.shape[1]}", flush=True)
        if not ok:
            bad += 1
            # first differing destination
            rp_ref = np.searchsorted(ref[1], np.arange(lo, hi + 1))
            for e in range(lo, hi):
                a = part[0, rowptr[e] - rowptr[lo]:rowptr[e + 1] - rowptr[lo]]
                c = ref[0, rp_ref[e - lo]:rp_ref[e - lo + 1]]
                if len(a) != len(c) or (a != c).any():
                    print(" dest", e, "xyt", x[e], y[e], t[e], "\n  hip", a.tolist(), "\n  ref", c.tolist())
                    for nm, lst in (("hip", a), ("ref", c)):
                        print("  ", nm, [(int(x[i] - x[e]), int(y[i] - y[e]), int(t[e] - t[i])) for i in lst])
                    px = np.nonzero((x[lo:hi] == x[e]) & (y[lo:hi] == y[e]))[0]
                    print("   events on the destination's pixel:", len(px))
                    break
sys.exit(1 if bad else 0)
