#!/usr/bin/env python
"""tools/l0_class_stats.py -- CPU analysis for the level-0 conv's phase 1 (walk a node's in-edges grouped by
the y tap class, so that only the two live tap rows are multiplied).  For a synthetic window: every node's in-edges by y class
(k0 of the degree-1 spline along y on the 15-offset domain -> 4 classes on the 3 x 5 tap window), nodes grouped into the
kernel's tiles (16 consecutive nodes in slot order = (sample, y, x, id)), and per tile the iterations a class-by-class walk
takes -- sum over classes of the LARGEST class count among the tile's 16 nodes (the lanes of a wave run in lockstep) -- against
the iterations of the kernel as it is (the largest degree of the tile).  Builder tool, no GPU."""
import sys, os, numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from dagr_amd.utils import synthetic as syn
from oracle import graph as og
W, H, r, DT = 640, 480, 7, 10000
for name, gen, n in (("uniform 100k", syn.uniform_window, 100000), ("edges 100k", syn.edges_window, 100000), ("uniform 25k", syn.uniform_window, 25000)):
    x, y, t, p = gen(n, W, H, 4234)
    ei = og.build_window_graph(x, y, t, np.zeros(n, np.int32), W, H, 1, r, DT, K=16, Q=128)
    src, dst = ei[0], ei[1]
    keep = src != dst                                  # (the self loop's offset (0, 0) is one more edge of class of dy = 0)
    dy = (y[src] - y[dst]).astype(np.int64)            # source - destination, -r .. r
    # level-0 pseudo coordinate: (dy / (2 M H) + 0.5) * 4 with M = 2 * int(0.01 W + 2) / W  (net.py:72-73): den = 2 M H px
    den = 2.0 * (2 * int(0.01 * W + 2) / W) * H
    k0 = np.floor(((dy / den) + 0.5) * 4).astype(np.int64)
    cls = k0 - k0.min()
    ncls = int(cls.max()) + 1
    order = np.lexsort((np.arange(n), x, y))           # slot order of one sample
    slot_of = np.empty(n, np.int64); slot_of[order] = np.arange(n)
    cnt = np.zeros((n, ncls), np.int64)
    np.add.at(cnt, (slot_of[dst], cls), 1)
    deg = cnt.sum(1)
    T = n // 16
    c = cnt[:T * 16].reshape(T, 16, ncls)
    d = deg[:T * 16].reshape(T, 16)
    now = d.max(1)                                     # iterations of the kernel as it is (8 / 16 batches aside)
    by_class = c.max(1).sum(1)                         # class-by-class walk
    print(f"== {name}: {ncls} y classes; mean degree {deg.mean():.1f}; per tile: iterations now {now.mean():.1f}, class walk "
          f"{by_class.mean():.1f}; packed FMAs per tile-lane now {30 * now.mean():.0f} (30 per edge), class walk "
          f"{12 * by_class.mean():.0f} (12 per edge) -> x{12 * by_class.mean() / (30 * now.mean()):.2f}; useful "
          f"{4 * 2 * deg.mean():.0f} (4 taps x 2 packed per edge)")
