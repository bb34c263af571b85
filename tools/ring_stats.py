#!/usr/bin/env python
"""tools/ring_stats.py -- CPU analysis (numpy + scipy k-d tree): for sampled destinations of a synthetic window, the candidates in
the 15 x 15 neighbourhood, the admissible ones, and the spiral ring at which K - 1 admissible sources are reached (what a
ring-limited search would have to examine).  Builder tool, no GPU."""
import sys, numpy as np
sys.path.insert(0, __import__('os').path.dirname(__import__('os').path.dirname(__import__('os').path.abspath(__file__))))
from dagr_amd.utils import synthetic as syn
from scipy.spatial import cKDTree
W, H, r, DT, K = 640, 480, 7, 10000, 16
def stats(gen, n, seed=4234, nd=20000):
    x, y, t, p = gen(n, W, H, seed)
    # time_window normalisation: t in us already relative (0..window) shifted to end at 1e6? use as is
    order = np.argsort(t, kind='stable'); x, y, t = x[order], y[order], t[order]
    ids = np.arange(n)
    tree = cKDTree(np.stack([x, y], 1).astype(np.float64))
    rng = np.random.default_rng(1)
    dests = rng.choice(n, min(nd, n), replace=False)
    nb = tree.query_ball_point(np.stack([x[dests], y[dests]], 1).astype(np.float64), r + 0.5, p=np.inf)
    C = np.zeros(len(dests), int); rho_star = np.full(len(dests), -1); Cin = np.zeros(len(dests), int); V = np.zeros(len(dests), int)
    cum_by_ring = np.zeros((len(dests), r + 1), int); cand_by_ring = np.zeros((len(dests), r + 1), int)
    for i, (d, lst) in enumerate(zip(dests, nb)):
        lst = np.asarray(lst)
        C[i] = len(lst)
        ring = np.maximum(np.abs(x[lst] - x[d]), np.abs(y[lst] - y[d]))
        adm = (lst < d) & (t[d] - t[lst] <= DT)
        V[i] = adm.sum()
        a = np.bincount(ring[adm], minlength=r + 1).cumsum(); c = np.bincount(ring, minlength=r + 1).cumsum()
        cum_by_ring[i] = a; cand_by_ring[i] = c
        w = np.nonzero(a >= K - 1)[0]
        if len(w): rho_star[i] = w[0]; Cin[i] = c[w[0]]
    return C, V, rho_star, Cin, cum_by_ring, cand_by_ring
for name, gen, n in (("edges 100k", syn.edges_window, 100000), ("edges 200k", syn.edges_window, 200000), ("uniform 100k", syn.uniform_window, 100000), ("uniform 400k", syn.uniform_window, 400000)):
    C, V, rs, Cin, cum, cand = stats(gen, n)
    print(f"== {name}: mean C {C.mean():.0f}  median {np.median(C):.0f}  p90 {np.percentile(C,90):.0f}  C>320: {(C>320).mean():.2f}  C>96: {(C>96).mean():.2f}; admissible mean {V.mean():.1f}, >= K-1: {(V>=K-1).mean():.2f}")
    full = rs >= 0
    print("   ring at which K-1 admissible are reached (share of all dests):", {int(k): round(float((rs==k).mean()),3) for k in range(r+1)}, "never:", round(float((rs<0).mean()),3))
    for lo, hi in ((0, 96), (96, 320), (320, 10**9)):
        m = (C > lo) & (C <= hi)
        if m.sum() == 0: continue
        mm = m & full
        print(f"   C in ({lo},{hi}]: share {m.mean():.2f}; reach K-1: {mm.sum()/max(1,m.sum()):.2f}; mean C {C[m].mean():.0f}; mean candidates inside the ring that fills K-1: {Cin[mm].mean() if mm.sum() else 0:.0f}; mean ring {rs[mm].mean() if mm.sum() else 0:.1f}")
    # total candidates examined: now vs ring-limited (fallback to full when never reached)
    now = np.minimum(C, 10**9).sum(); ringed = np.where(full, Cin, C).sum()
    print(f"   candidates examined: all {now/len(C):.0f} per dest; ring-limited {ringed/len(C):.0f} per dest ({ringed/now:.2f})")
