"""Kernel-level parity of the pooled-level SplineConv entry points on random CSR graphs:
dagr_spline_conv_fused and dagr_spline_tap_aggregate + dagr_gemm_bias_act against a float64 evaluation
built on the oracle's torch_spline_conv basis (oracle/ops.py:spline_basis).  Tolerance 1e-4 relative to the
output scale (fp32 accumulation over K <= 6700 terms)."""
import numpy as np
import pytest
import torch

from oracle import ops as oo

pytestmark = pytest.mark.gpu


def _pack_wq(Wm):
    K, N = Wm.shape
    K16, N16 = (K + 15) // 16 * 16, (N + 15) // 16 * 16
    Wp = torch.zeros((K16, N16), dtype=Wm.dtype, device=Wm.device)
    Wp[:K, :N] = Wm
    return Wp.view(K16 // 16, 4, 4, N16 // 16, 16).permute(3, 0, 2, 4, 1).contiguous()


def _reference(rowptr, col, code, x, xs, Wm, bias, relu, r, den):
    T = len(rowptr) - 1
    cin = x.shape[1]
    dst = np.repeat(np.arange(T), np.diff(rowptr))
    ix, iy = code & 0xFFFF, code >> 16
    pseudo = torch.stack([torch.from_numpy((ix - r).astype(np.float32)) / np.float32(den[0]) + 0.5,
                          torch.from_numpy((iy - r).astype(np.float32)) / np.float32(den[1]) + 0.5], 1)
    basis, index = oo.spline_basis(pseudo)
    A = np.zeros((T, 25, cin), dtype=np.float64)
    xj = x[col].astype(np.float64)
    for s in range(4):
        np.add.at(A, (dst, index[:, s].numpy()), basis[:, s].numpy().astype(np.float64)[:, None] * xj)
    full = [A.reshape(T, 25 * cin), x.astype(np.float64)]
    if xs is not None:
        full.append(xs.astype(np.float64))
    out = np.concatenate(full, 1) @ Wm.astype(np.float64) + bias.astype(np.float64)
    return np.maximum(out, 0) if relu else out


CASES = [  # T, cin, cskip, N, max_deg, relu
    (700, 32, 0, 64, 9, True),
    (1000, 64, 32, 64, 12, True),
    (37, 64, 64, 128, 70, False),     # ragged tile, two column blocks, > 64 edges on a node
    (300, 16, 0, 21, 5, True),        # narrow, odd N (column-split mode)
    (5000, 82, 0, 64, 8, True),       # cin > 64 (two channel chunks), K = 2132
    (16, 3, 5, 7, 0, True),           # no edges at all
    (200, 64, 0, 200, 6, False),      # N > 128: four column blocks
    (6000, 18, 0, 32, 9, True),       # level 1 of a B = 8 batch: more workgroups than CUs, two tiles per CU (62-register form)
    (6000, 32, 18, 32, 9, True),
    (20000, 18, 0, 64, 9, True),      # level 1 of a B = 8 batch at its size: 80-node tiles, three nodes per wave and pass (k_conv_fused_narrow<5>)
    (13000, 3, 0, 40, 20, False),     # the same form on 3 input channels, ragged N, a last tile of 40 rows
    (12500, 12, 0, 96, 6, True),      # ... and its 48-node form (N > 64: two column blocks over one A tile)
    # rows beyond the LDS tile: K cut at tap boundaries into passes (accumulators stay in registers across them)
    (300, 128, 0, 256, 9, True),      # head conv pair of dagr-s (cls_conv | reg_conv): 2 passes of 13 / 12 + root
    (1200, 130, 130, 64, 7, True),    # --use_image level >= 2: K = 3510, tap split must be a multiple of 8
    (500, 98, 98, 64, 8, True),       # dagr-m level: K = 2646
    (150, 256, 0, 7, 70, False),      # merged predictor of a 128-wide head: 4 passes, > 64 edges on a node
    (64, 66, 2000, 64, 5, True),      # a skip row longer than 128 channels, root + skip alone in the last pass
]


@pytest.mark.parametrize("T,cin,cskip,N,max_deg,relu", CASES)
def test_fused_and_unfused_match_float64(T, cin, cskip, N, max_deg, relu):
    from dagr_amd import _lib
    L = _lib.lib()
    dev = torch.device("cuda:0")
    rng = np.random.default_rng(T * 131 + cin)
    r, den = 7, (14.0, 17.5)      # den >= 2 r keeps the pseudo-coordinate inside [0, 1] (cartesian(), oracle/ops.py:17)
    deg = rng.integers(0, max_deg + 1, size=T)
    rowptr = np.concatenate([[0], np.cumsum(deg)]).astype(np.int32)
    E = int(rowptr[-1])
    col = rng.integers(0, T, size=E).astype(np.int32)
    code = (rng.integers(0, 2 * r + 1, size=E) | (rng.integers(0, 2 * r + 1, size=E) << 16)).astype(np.int32)
    x = rng.standard_normal((T, cin)).astype(np.float32)
    xs = rng.standard_normal((T, cskip)).astype(np.float32) if cskip else None
    K = 26 * cin + cskip
    Wm = (rng.standard_normal((K, N)) / np.sqrt(K)).astype(np.float32)
    bias = rng.standard_normal(N).astype(np.float32)
    want = _reference(rowptr, col, code, x, xs, Wm, bias, relu, r, den)
    scale = max(1.0, float(np.abs(want).max()))

    t = lambda a: None if a is None else torch.from_numpy(a).to(dev)
    d_rowptr, d_col, d_code, d_x, d_xs, d_bias = t(rowptr), t(col), t(code), t(x), t(xs), t(bias)
    if E == 0:
        d_col = torch.zeros(1, dtype=torch.int32, device=dev)
        d_code = torch.zeros(1, dtype=torch.int32, device=dev)
    n_ptr = torch.tensor([T], dtype=torch.int32, device=dev)
    P, S = _lib.ptr, _lib.cur_stream(dev)
    ldw = (N + 7) // 8 * 8
    Wpad = torch.zeros((K, ldw), dtype=torch.float32, device=dev)
    Wpad[:, :N] = t(Wm)

    # unfused: A in HBM
    lda = (K + 3) // 4 * 4
    A = torch.zeros((T, lda), dtype=torch.float32, device=dev)
    out_u = torch.full((T, N), 7.0, dtype=torch.float32, device=dev)
    _lib.check(L.dagr_spline_tap_aggregate(P(n_ptr), T, P(d_rowptr), P(d_col), P(d_code), P(d_x), cin, cin, P(d_xs),
                                           cskip, cskip, r, r, den[0], den[1], P(A), lda, S), "tap_aggregate")
    _lib.check(L.dagr_gemm_bias_act(P(n_ptr), T, P(A), lda, P(Wpad), ldw, P(d_bias), P(out_u), N, K, N, int(relu), S),
               "gemm")
    assert np.abs(out_u.cpu().numpy() - want).max() <= 1e-4 * scale

    # fused: A tile in LDS
    assert L.dagr_spline_conv_fused_lds_bytes(cin, cskip) <= 160 * 1024
    Wq = _pack_wq(t(Wm))
    out_f = torch.full((T, N), 7.0, dtype=torch.float32, device=dev)
    _lib.check(L.dagr_spline_conv_fused(P(n_ptr), T, P(d_rowptr), P(d_col), P(d_code), P(d_x), cin, cin, P(d_xs), cskip,
                                        cskip, r, r, den[0], den[1], P(Wq), P(d_bias), P(out_f), N, N, int(relu), S),
               "spline_conv_fused")
    torch.cuda.synchronize()
    assert np.abs(out_f.cpu().numpy() - want).max() <= 1e-4 * scale

    # device-side node count below the static bound: rows past *n_ptr stay untouched
    if T > 20:
        n_ptr.fill_(T - 19)
        out_f.fill_(7.0)
        _lib.check(L.dagr_spline_conv_fused(P(n_ptr), T, P(d_rowptr), P(d_col), P(d_code), P(d_x), cin, cin, P(d_xs),
                                            cskip, cskip, r, r, den[0], den[1], P(Wq), P(d_bias), P(out_f), N, N,
                                            int(relu), S), "spline_conv_fused")
        got = out_f.cpu().numpy()
        assert np.abs(got[: T - 19] - want[: T - 19]).max() <= 1e-4 * scale
        assert (got[T - 19:] == 7.0).all()


def test_fused_rejects_oversized_k():
    from dagr_amd import _lib
    L = _lib.lib()
    assert L.dagr_spline_conv_fused_lds_bytes(130, 0) <= 160 * 1024      # two passes
    assert L.dagr_spline_conv_fused_lds_bytes(2400, 0) > 160 * 1024      # not even one tap fits the tile
    dev = torch.device("cuda:0")
    z = torch.zeros(64, dtype=torch.int32, device=dev)
    f = torch.zeros(4096, dtype=torch.float32, device=dev)
    rc = L.dagr_spline_conv_fused(None, 1, _lib.ptr(z), _lib.ptr(z), _lib.ptr(z), _lib.ptr(f), 2400, 2400, None, 0, 0, 7, 7,
                                  14.0, 14.0, _lib.ptr(f), None, _lib.ptr(f), 64, 64, 1, _lib.cur_stream(dev))
    assert rc != 0 and b"does not fit" in L.dagr_last_error()


def _random_job(rng, dev, T, cin, cskip, N, max_deg, relu, live=None):
    """Device buffers + the dagr_conv_job of one random conv; returns (job, out tensor, keep-alive list)."""
    from dagr_amd import _lib
    r, den = 7, (14.0, 17.5)
    deg = rng.integers(0, max_deg + 1, size=T)
    rowptr = np.concatenate([[0], np.cumsum(deg)]).astype(np.int32)
    E = max(1, int(rowptr[-1]))
    t = lambda a: torch.from_numpy(a).to(dev)
    col = t(rng.integers(0, T, size=E).astype(np.int32))
    code = t((rng.integers(0, 2 * r + 1, size=E) | (rng.integers(0, 2 * r + 1, size=E) << 16)).astype(np.int32))
    x = t(rng.standard_normal((T, cin)).astype(np.float32))
    xs = t(rng.standard_normal((T, cskip)).astype(np.float32)) if cskip else None
    K = 26 * cin + cskip
    Wq = _pack_wq(t((rng.standard_normal((K, N)) / np.sqrt(K)).astype(np.float32)))
    bias = t(rng.standard_normal(N).astype(np.float32))
    n_ptr = torch.tensor([T if live is None else live], dtype=torch.int32, device=dev)
    d_rowptr = t(rowptr)
    out = torch.full((T, N), 7.0, dtype=torch.float32, device=dev)
    job = _lib.ConvJob(n_nodes_ptr=n_ptr.data_ptr(), n_nodes_max=T, rowptr=d_rowptr.data_ptr(), col=col.data_ptr(),
                       code=code.data_ptr(), x=x.data_ptr(), ldx=cin, cin=cin, xskip=xs.data_ptr() if cskip else None,
                       ldskip=cskip, cskip=cskip, rx=r, ry=r, den_x=den[0], den_y=den[1], Wq=Wq.data_ptr(),
                       bias=bias.data_ptr(), C=out.data_ptr(), ldc=N, N=N, relu=int(relu))
    return job, out, [n_ptr, d_rowptr, col, code, x, xs, Wq, bias]


MULTI = [  # jobs of one launch: (T, cin, cskip, N, max_deg, relu, live rows or None)
    [(70, 66, 0, 64, 6, True, 36), (280, 64, 0, 64, 7, True, 141)],                               # layer5.conv1 | stem_1
    [(70, 64, 66, 64, 6, True, 36), (280, 64, 0, 128, 7, True, 141)],                             # layer5.conv2 | cls,reg conv_1
    [(70, 64, 0, 64, 6, True, None), (280, 64, 0, 5, 7, False, 141), (280, 64, 0, 2, 7, False, 141)],   # stem_2 | preds_1
    [(1300, 32, 0, 200, 9, True, 1122), (16, 3, 5, 7, 0, True, None), (37, 64, 64, 128, 70, False, None),
     (5000, 82, 0, 64, 8, True, 4482)],                                                           # four very different jobs
    [(300, 128, 0, 256, 9, True, None), (150, 256, 0, 7, 70, False, None)],                       # pass form for all jobs
]


@pytest.mark.parametrize("specs", MULTI)
def test_multi_job_launch_equals_single_launches_bit_for_bit(specs):
    """dagr_spline_conv_fused_multi: every job's output equals what dagr_spline_conv_fused writes for it alone (same
    column split, same K partial order), rows past a job's device-side count untouched."""
    from dagr_amd import _lib
    L = _lib.lib()
    dev = torch.device("cuda:0")
    rng = np.random.default_rng(len(specs) * 977 + specs[0][0])
    S = _lib.cur_stream(dev)
    jobs, outs, keep = [], [], []
    for sp in specs:
        j, o, k = _random_job(rng, dev, *sp)
        jobs.append(j); outs.append(o); keep.append(k)
    arr = (_lib.ConvJob * len(jobs))(*jobs)
    _lib.check(L.dagr_spline_conv_fused_multi(arr, len(jobs), S), "multi")
    torch.cuda.synchronize()
    got = [o.clone() for o in outs]
    for j, o, sp in zip(jobs, outs, specs):
        o.fill_(7.0)
        _lib.check(L.dagr_spline_conv_fused(j.n_nodes_ptr, j.n_nodes_max, j.rowptr, j.col, j.code, j.x, j.ldx, j.cin, j.xskip,
                                            j.ldskip, j.cskip, j.rx, j.ry, j.den_x, j.den_y, j.Wq, j.bias, j.C, j.ldc, j.N,
                                            j.relu, S), "single")
    torch.cuda.synchronize()
    for g, o, sp in zip(got, outs, specs):
        assert torch.equal(g, o), sp
        live = sp[0] if sp[6] is None else sp[6]
        assert bool((g[live:] == 7.0).all()) and not bool((g[:live] == 7.0).all()), sp


def test_multi_job_launch_rejects_mixed_forms():
    from dagr_amd import _lib
    L = _lib.lib()
    dev = torch.device("cuda:0")
    rng = np.random.default_rng(5)
    a = _random_job(rng, dev, 64, 64, 0, 64, 5, True)
    b = _random_job(rng, dev, 64, 128, 0, 64, 5, True)          # needs passes
    arr = (_lib.ConvJob * 2)(a[0], b[0])
    rc = L.dagr_spline_conv_fused_multi(arr, 2, _lib.cur_stream(dev))
    assert rc != 0 and b"cannot share a launch" in L.dagr_last_error()
