"""CPU suite: the graph oracle against (i) the golden vectors produced by the reference's own
kernels on an MI355X (tests/golden/graph_ref_small.npz, made by tests/make_golden_graph.py),
(ii) its pure-Python twin, (iii) invariants the reference states (ev_tgn.py:52-54)."""
import os

import numpy as np
import pytest

from oracle import graph as og
from tests.graph_cases import small_cases

GOLDEN = os.path.join(os.path.dirname(__file__), "golden", "graph_ref_small.npz")


def test_spiral_matches_reference_order():
    # spiral.h:8-15 written out by hand for the first two rings (SURVEY QUIRK-4)
    want = [(0, 0), (1, 0), (1, 1), (0, 1), (-1, 1), (-1, 0), (-1, -1), (0, -1), (1, -1), (2, -1), (2, 0), (2, 1),
            (2, 2), (1, 2), (0, 2), (-1, 2), (-2, 2), (-2, 1), (-2, 0), (-2, -1), (-2, -2), (-1, -2), (0, -2),
            (1, -2), (2, -2), (3, -2)]
    dx, dy = og.spiral_offsets(len(want))
    assert list(zip(dx.tolist(), dy.tolist())) == want
    assert og.spiral_offsets_py(len(want)) == want


def test_oracle_matches_reference_golden():
    z = np.load(GOLDEN)
    names = sorted({k.split("/")[0] for k in z.files})
    assert len(names) >= 15
    for n in names:
        W, H, B, r, dt, K, Q = [int(v) for v in z[f"{n}/params"]]
        e = og.build_window_graph(z[f"{n}/x"], z[f"{n}/y"], z[f"{n}/t"], z[f"{n}/b"], W, H, B, r, dt, K=K, Q=Q)
        ref = z[f"{n}/edge_index"].astype(np.int64)
        assert e.shape == ref.shape, n
        assert (e == ref).all(), n


def test_golden_covers_current_cases():
    """The committed fixture was generated from the same inputs the suites use today."""
    z = np.load(GOLDEN)
    for c in small_cases():
        assert (z[f"{c['name']}/x"] == c["x"]).all() and (z[f"{c['name']}/t"] == c["t"]).all(), c["name"]


@pytest.mark.parametrize("name", ["dt_boundary", "hot_pixel", "border", "unsorted_t", "k4"])
def test_c_kernels_match_python_twin(name):
    c = [c for c in small_cases() if c["name"] == name][0]
    N = len(c["x"])
    g = og.SlidingWindowGraph(c["W"], c["H"], c["B"], c["K"], c["Q"], c["r"], c["dt"])
    g.initialize(N)
    pos = np.stack([c["x"], c["y"], c["t"]], -1).astype(np.int32)
    ind = np.arange(N, dtype=np.int32)
    q = og.insert_events_into_queue(c["b"], pos, ind, g.event_queue)
    eb = np.full((2, c["K"] * N), -1, np.int64)
    og.fill_edges_py(c["b"], pos, pos[:, 2].copy(), ind, q, eb, c["r"], c["dt"], c["K"], 0)
    e_py = eb[:, eb[1] >= 0]
    e_c = og.build_window_graph(c["x"], c["y"], c["t"], c["b"], c["W"], c["H"], c["B"], c["r"], c["dt"], K=c["K"], Q=c["Q"])
    assert (e_py == e_c).all()


def test_dt_boundary_hand_computed():
    # events at t=989999, 990000, 1000000 on adjacent pixels, delta=10000:
    # e1 <- e0 (dt=1), e2 <- e1 (dt=10000, admitted), e2 <-/- e0 (dt=10001, rejected)
    c = [c for c in small_cases() if c["name"] == "dt_boundary"][0]
    e = og.build_window_graph(c["x"], c["y"], c["t"], c["b"], c["W"], c["H"], c["B"], c["r"], c["dt"])
    assert e.T.tolist() == [[0, 0], [1, 1], [0, 1], [2, 2], [1, 2]]


def test_invariants_and_fifo_overflow():
    c = [c for c in small_cases() if c["name"] == "hot_pixel"][0]
    e = og.build_window_graph(c["x"], c["y"], c["t"], c["b"], c["W"], c["H"], c["B"], c["r"], c["dt"])
    assert (e[0] <= e[1]).all() and (np.diff(e[1]) >= 0).all()
    deg = np.bincount(e[1], minlength=len(c["x"]))
    assert deg.max() <= 16 and deg.min() >= 1
    first = np.r_[0, np.cumsum(deg)[:-1]]
    assert (e[0, first] == e[1, first]).all()  # self loop first


def test_denormalize_roundtrip_matches_raw_pixels():
    """x,y survive format_data -> denormalize_pos exactly for every geometry we use; t may not."""
    from dagr_amd.utils.synthetic import format_data_np
    for W, H in [(640, 480), (320, 215), (240, 180), (64, 48), (80, 60)]:
        x = np.arange(W); y = np.arange(W) % H
        pos = og.denormalize_pos(format_data_np(x, y, np.zeros(W), W, H), W, H, 1000000)
        assert (pos[:, 0] == x).all() and (pos[:, 1] == y).all()


@pytest.mark.parametrize("rho", [1, 2, 3])
def test_a_destination_filled_by_its_inner_rings_needs_nothing_outside_them(rho):
    """What the row search's ring limit relies on (csrc/graph_build.hip, k_search_rows): the walk takes sources in spiral
    order, the spiral runs ring by ring (spiral.h:1-15), and it stops at K entries (ev_graph.cu:48-78) -- so a destination
    whose in-edges at search radius rho already number K has exactly the same in-edges at any larger radius.  Checked on the
    oracle: dense windows, radius rho against radius 7."""
    from dagr_amd.utils import synthetic as syn
    W, H, K, Q, dt = 96, 72, 16, 128, 10000
    x, y, t, p, b = syn.batch_windows(syn.edges_window, 6000, 1, W, H, seed=91 + rho)
    t = (t - t.max() + 1000000).astype(np.int64)     # windows end at the normaliser (dsec_data.py:145)
    full = og.build_window_graph(x, y, t, b, W, H, 1, 7, dt, K=K, Q=Q)
    inner = og.build_window_graph(x, y, t, b, W, H, 1, rho, dt, K=K, Q=Q)

    def rows(e):
        order = np.argsort(e[1], kind="stable")
        src, dst = e[0][order], e[1][order]
        cut = np.flatnonzero(np.diff(dst)) + 1
        return dict(zip(dst[np.r_[0, cut]].tolist(), np.split(src, cut)))
    rf, ri = rows(full), rows(inner)
    filled = [d for d, s in ri.items() if len(s) == K]
    assert len(filled) > 500, "the window is not dense enough to exercise the claim"
    for d in filled:
        assert (rf[d] == ri[d]).all(), f"destination {d}: radius {rho} fills K, radius 7 gives another list"
    # and the converse bound: a destination NOT filled at rho keeps those sources as a prefix at radius 7
    for d, s in ri.items():
        if len(s) < K:
            assert (rf[d][:len(s)] == s).all(), f"destination {d}: the inner sources are not a prefix of the full list"
