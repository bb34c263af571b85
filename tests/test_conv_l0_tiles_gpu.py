"""Kernel-level parity of the tiled level-0 SplineConv (dagr_spline_conv_l0_tiles, csrc/conv_l0_tiles.hip) on random
fixed-stride neighbour lists, through the C ABI, against a float64 evaluation built on the oracle's torch_spline_conv
basis (oracle/ops.py:spline_basis) and the full 5x5 weight tensor.  Tolerance 1e-4 relative to the output scale."""
import ctypes

import numpy as np
import pytest
import torch

from oracle import ops as oo

pytestmark = pytest.mark.gpu

CASES = [  # N, (cmain, cextra), cskip, r, (den_x, den_y), relu
    (5000, (0, 3), 0, 7, (32.0, 24.0), True),      # events-only first conv, 640x480 geometry: 3x5 tap window
    (5000, (16, 0), 3, 7, (32.0, 24.0), True),     # events-only second conv + skip Linear(3)
    (3001, (16, 3), 0, 7, (32.0, 24.0), True),     # --use_image first conv (19 inputs), ragged last tile
    (3001, (16, 0), 19, 7, (32.0, 24.0), False),   # --use_image second conv + skip Linear(19)
    (777, (16, 3), 0, 4, (20.0, 20.0), True),      # square-ish domain: 3x3 window
    (17, (0, 3), 0, 4, (20.0, 13.4375), True),     # 320x215 geometry: r = 4, den = (20, 13.4375): 3x5
    (1, (16, 0), 19, 4, (20.0, 13.4375), True),
]


def _window(L, r, den):
    lo, cnt = ctypes.c_int32(0), ctypes.c_int32(0)
    assert L.dagr_spline_tap_window(r, den, ctypes.byref(lo), ctypes.byref(cnt)) == 0
    return (min(lo.value, 2), 3) if cnt.value <= 3 else (0, 5)


@pytest.mark.parametrize("N,cm_ce,cskip,r,den,relu", CASES)
def test_tiles_match_float64(N, cm_ce, cskip, r, den, relu):
    from dagr_amd import _lib
    L, P = _lib.lib(), _lib.ptr
    dev = torch.device("cuda:0")
    cm, ce = cm_ce
    cin = cm + ce
    ldx = (cin + 3) // 4 * 4
    lds = (cskip + 3) // 4 * 4 if cskip else 0
    rng = np.random.default_rng(N * 7 + cin + cskip)
    K, S = 16, 2 * r + 1
    deg = rng.integers(1, K + 1, size=N).astype(np.int32)
    deg[rng.integers(0, N, size=max(1, N // 10))] = K
    src = rng.integers(0, N, size=(N, K)).astype(np.int32)
    src[:, 0] = np.arange(N)                                   # slot 0 = the self loop
    code = rng.integers(0, S * S, size=(N, K)).astype(np.int16)
    code[:, 0] = r * S + r
    x = rng.standard_normal((N, ldx)).astype(np.float32)
    xs = rng.standard_normal((N, max(lds, 1))).astype(np.float32)
    W = (rng.standard_normal((25, cin, 16)) * 0.3).astype(np.float32)
    root = (rng.standard_normal((cin, 16)) * 0.3).astype(np.float32)
    wskip = (rng.standard_normal((cskip, 16)) * 0.3).astype(np.float32)
    shift = rng.standard_normal(16).astype(np.float32)
    (wx0, tx), (wy0, ty) = _window(L, r, den[0]), _window(L, r, den[1])
    rows = [W[(wx0 + a) + 5 * (wy0 + b)] for b in range(ty) for a in range(tx)] + [root] + ([wskip] if cskip else [])
    wpack = np.concatenate(rows, 0).astype(np.float32)
    # float64 reference over the valid slots, all 25 taps
    dst = np.repeat(np.arange(N), deg)
    slot = np.concatenate([np.arange(d) for d in deg])
    e_src, e_code = src[dst, slot], code[dst, slot].astype(np.int64)
    ix, iy = e_code // S, e_code % S
    pseudo = torch.stack([torch.from_numpy((ix - r).astype(np.float32)) / np.float32(den[0]) + 0.5,
                          torch.from_numpy((iy - r).astype(np.float32)) / np.float32(den[1]) + 0.5], 1)
    basis, index = oo.spline_basis(pseudo)
    A = np.zeros((N, 25, cin), dtype=np.float64)
    xj = x[e_src, :cin].astype(np.float64)
    for s in range(4):
        np.add.at(A, (dst, index[:, s].numpy()), basis[:, s].numpy().astype(np.float64)[:, None] * xj)
    want = A.reshape(N, -1) @ W.reshape(25 * cin, 16).astype(np.float64) + x[:, :cin].astype(np.float64) @ root
    if cskip:
        want = want + xs[:, :cskip].astype(np.float64) @ wskip
    want = want + shift
    if relu:
        want = np.maximum(want, 0)
    T = lambda a: torch.from_numpy(np.ascontiguousarray(a)).to(dev)
    d_src, d_code, d_deg, d_x, d_xs, d_w, d_s = T(src), T(code), T(deg), T(x), T(xs), T(wpack), T(shift)
    out = torch.full((N, 16), float("nan"), device=dev)
    _lib.check(L.dagr_spline_conv_l0_tiles(cm, ce, cskip, wx0, tx, wy0, ty, r, r, den[0], den[1], N, K, P(d_src),
                                           P(d_code), P(d_deg), P(d_x), ldx, P(d_xs) if cskip else None, lds, P(d_w),
                                           P(d_s), 1 if relu else 0, P(out), 16, None, _lib.cur_stream(dev)), "l0_tiles")
    torch.cuda.synchronize()
    got = out.cpu().numpy().astype(np.float64)
    assert np.isfinite(got).all()
    err = np.abs(got - want).max() / max(1.0, np.abs(want).max())
    assert err < 1e-4, err
