"""Shared graph-builder test cases (inputs only) for the CPU and GPU suites and the golden maker."""
import numpy as np

from dagr_amd.utils import synthetic as syn


def _case(name, x, y, t, b, W, H, B, r, dt, K=16, Q=128):
    return dict(name=name, x=np.asarray(x, np.int32), y=np.asarray(y, np.int32), t=np.asarray(t, np.int32),
                b=np.asarray(b, np.int32), W=W, H=H, B=B, r=r, dt=dt, K=K, Q=Q)


def small_cases():
    """Edge cases of SURVEY.md section 4 / 8d (S-edge-cases) + small random windows."""
    cases = []
    W, H = 64, 48
    cases.append(_case("empty", [], [], [], [], W, H, 1, 4, 10000))
    cases.append(_case("single", [5], [7], [1000000], [0], W, H, 1, 4, 10000))
    cases.append(_case("single_b1", [5], [7], [1000000], [1], W, H, 2, 4, 10000))  # QUIRK-5
    cases.append(_case("two_same_pixel", [5, 5], [7, 7], [999000, 1000000], [0, 0], W, H, 1, 4, 10000))
    # dt boundary: 10000 is an edge, 10001 is not (ev_graph.cu:69)
    cases.append(_case("dt_boundary", [10, 11, 12], [10, 10, 10], [989999, 990000, 1000000], [0, 0, 0], W, H, 1, 4, 10000))
    # identical timestamps
    rng = np.random.default_rng(3)
    n = 200
    cases.append(_case("same_t", rng.integers(0, W, n), rng.integers(0, H, n), np.full(n, 1000000), np.zeros(n), W, H, 1, 4, 10000))
    # hot pixel: 300 events on one pixel (> Q=128) + neighbours around it
    n = 300
    t = np.sort(rng.integers(990000, 1000001, n + 40))
    x = np.concatenate([np.full(n, 20), rng.integers(17, 24, 40)])
    y = np.concatenate([np.full(n, 20), rng.integers(17, 24, 40)])
    perm = rng.permutation(n + 40)
    cases.append(_case("hot_pixel", x[perm], y[perm], t, np.zeros(n + 40), W, H, 1, 4, 10000))
    # hot pixel with small FIFO (Q=8) and mid-size segments (exercise long path with Q < 64 < n)
    cases.append(_case("hot_pixel_q8", x[perm], y[perm], t, np.zeros(n + 40), W, H, 1, 4, 10000, Q=8))
    # border events
    bx = np.array([0, 0, W - 1, W - 1, 0, W - 1, 1, W - 2] * 5)
    by = np.array([0, H - 1, 0, H - 1, 1, H - 2, 0, H - 1] * 5)
    cases.append(_case("border", bx, by, np.sort(rng.integers(995000, 1000001, len(bx))), np.zeros(len(bx)), W, H, 1, 4, 10000))
    # dense blob: saturates K=16 in the first ring
    n = 1500
    cases.append(_case("dense_blob", rng.integers(28, 36, n), rng.integers(20, 28, n),
                       np.sort(rng.integers(990000, 1000001, n)), np.zeros(n), W, H, 1, 4, 10000))
    # unsorted timestamps (negative dt is admitted by the reference's `dt > delta` test)
    n = 400
    cases.append(_case("unsorted_t", rng.integers(0, W, n), rng.integers(0, H, n), rng.integers(950000, 1000001, n),
                       np.zeros(n), W, H, 1, 4, 10000))
    # small K, radius 0 and radius 1
    n = 500
    xx, yy, tt = rng.integers(0, 16, n), rng.integers(0, 16, n), np.sort(rng.integers(950000, 1000001, n))
    cases.append(_case("k4", xx, yy, tt, np.zeros(n), 16, 16, 1, 2, 20000, K=4))
    cases.append(_case("r0", xx, yy, tt, np.zeros(n), 16, 16, 1, 0, 20000))
    cases.append(_case("r1", xx, yy, tt, np.zeros(n), 16, 16, 1, 1, 20000))
    # batched random windows (B=3), DSEC-like geometry scaled down
    x, y, t, p, b = syn.batch_windows(syn.uniform_window, 1500, 3, 80, 60, seed=11)
    cases.append(_case("uniform_b3", x, y, t, b, 80, 60, 3, 4, 10000))
    x, y, t, p, b = syn.batch_windows(syn.edges_window, 3000, 2, 80, 60, seed=21)
    cases.append(_case("edges_b2", x, y, t, b, 80, 60, 2, 4, 10000))
    return cases


def medium_cases():
    cases = []
    x, y, t, p, b = syn.batch_windows(syn.uniform_window, 20000, 2, 320, 215, seed=101)
    cases.append(_case("dsec_uniform_b2", x, y, t, b, 320, 215, 2, 4, 10000))
    x, y, t, p, b = syn.batch_windows(syn.edges_window, 30000, 1, 320, 215, seed=201)
    cases.append(_case("dsec_edges_b1", x, y, t, b, 320, 215, 1, 4, 10000))
    x, y, t, p, b = syn.batch_windows(syn.edges_window, 50000, 1, 640, 480, seed=301)
    cases.append(_case("vga_edges_b1", x, y, t, b, 640, 480, 1, 7, 10000))
    return cases


def wide_radius_cases():
    """Search radii beyond 7 pixels (a sensor wider than 700 px at radius 0.01): the builder's generic search form
    (csrc/graph_build.hip:k_search_dense, every node; the row kernel's 16-lane layout stops at 2r + 1 = 15 rows)."""
    cases = []
    rng = np.random.default_rng(17)
    x, y, t, p, b = syn.batch_windows(syn.uniform_window, 2500, 2, 96, 72, seed=31)
    cases.append(_case("r9_uniform_b2", x, y, t, b, 96, 72, 2, 9, 10000))
    x, y, t, p, b = syn.batch_windows(syn.edges_window, 4000, 1, 96, 72, seed=41)
    cases.append(_case("r12_edges_b1", x, y, t, b, 96, 72, 1, 12, 10000, K=16, Q=16))
    n = 900
    cases.append(_case("r8_unsorted_k24", rng.integers(0, 64, n), rng.integers(0, 48, n), rng.integers(960000, 1000001, n),
                       np.zeros(n), 64, 48, 1, 8, 10000, K=24, Q=8))
    return cases


def heavy_cases():
    """Event-dense neighbourhoods (csrc/graph_build.hip: the row kernel's ring-limited passes for 200 .. 320 candidates, the
    position-centric walk of k_search_dense beyond).  GPU suites only: the
    expected graphs come from the oracle and, where it is built, from the reference's own kernels (oracle/_ref)."""
    cases = []
    rng = np.random.default_rng(29)
    W, H = 96, 64

    def around(cx, cy, n_core, core, n_halo, halo, t_lo, t_hi=1000001):
        x = np.concatenate([rng.integers(cx, cx + core, n_core), rng.integers(cx - halo, cx + core + halo, n_halo)])
        y = np.concatenate([rng.integers(cy, cy + core, n_core), rng.integers(cy - halo, cy + core + halo, n_halo)])
        t = np.sort(rng.integers(t_lo, t_hi, n_core + n_halo))
        perm = rng.permutation(n_core + n_halo)
        return x[perm], y[perm], t
    # 2400 events on 2 x 2 pixels + a halo: ring 0 alone overflows the key list -> position-centric walk; FIFO depth 128
    x, y, t = around(40, 30, 2400, 2, 300, 4, 985000)
    cases.append(_case("walk_2x2_q128", x, y, t, np.zeros(len(x)), W, H, 1, 4, 10000))
    # the same without any pixel beyond the FIFO depth (every slot visible: no visibility searches)
    cases.append(_case("walk_2x2_q1024", x, y, t, np.zeros(len(x)), W, H, 1, 4, 10000, Q=1024))
    # at the sensor's corner: clipped windows, clamped offsets
    x, y, t = around(0, 0, 1800, 2, 300, 3, 985000)
    cases.append(_case("walk_corner", np.clip(x, 0, W - 1), np.clip(y, 0, H - 1), t, np.zeros(len(x)), W, H, 1, 4, 10000, Q=64))
    # dense but mostly stale: few admissible sources per ring, the passes grow to the full window (and fall short of K - 1)
    n = 6000
    cases.append(_case("stale_block", rng.integers(20, 50, n), rng.integers(10, 40, n),
                       np.sort(rng.integers(700000, 1000001, n)), np.zeros(n), W, H, 1, 4, 10000))
    # radius 7 (15 x 15 windows), moving edges at a density where most destinations are heavy; K = 12 and K = 16
    x, y, t, p, b = syn.batch_windows(syn.edges_window, 20000, 2, 160, 120, seed=51)
    cases.append(_case("edges_r7_b2", x, y, t, b, 160, 120, 2, 7, 10000))
    cases.append(_case("edges_r7_b2_k12_q16", x, y, t, b, 160, 120, 2, 7, 10000, K=12, Q=16))
    # radius 5 and 6 (row matrices narrower than 16 columns), uniform at 6 events per pixel
    x, y, t, p, b = syn.batch_windows(syn.uniform_window, 30000, 1, 80, 60, seed=61)
    cases.append(_case("uniform_dense_r5", x, y, t, b, 80, 60, 1, 5, 10000))
    cases.append(_case("uniform_dense_r6_k24", x, y, t, b, 80, 60, 1, 6, 10000, K=24))
    # K = 64 (the interface's maximum) on a dense blob: more sources wanted than one round of candidates holds
    n = 5000
    cases.append(_case("blob_k64", rng.integers(30, 44, n), rng.integers(20, 34, n),
                       np.sort(rng.integers(990000, 1000001, n)), np.zeros(n), W, H, 1, 4, 10000, K=64))
    return cases
