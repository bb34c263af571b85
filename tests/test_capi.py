"""CPU suite: the C-ABI library loads and exports every symbol include/dagr_hip.h declares
(no compute calls without a GPU), argument validation works, the spiral helper matches the oracle."""
import ctypes
import os
import re

import numpy as np

from dagr_amd import _lib
from oracle import graph as og

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _declared_functions():
    src = open(os.path.join(ROOT, "include", "dagr_hip.h")).read()
    src = re.sub(r"/\*.*?\*/", "", src, flags=re.S)
    return sorted(set(re.findall(r"\b(dagr_[A-Za-z0-9_]+)\s*\(", src)))


def test_header_symbols_exported_and_bound():
    names = _declared_functions()
    assert len(names) >= 10
    L = _lib.lib()
    for n in names:
        assert hasattr(L, n), f"{n} declared in dagr_hip.h but not exported"
        assert n in _lib.SIGNATURES, f"{n} declared in dagr_hip.h but not bound in dagr_amd/_lib.py"
    for n in _lib.SIGNATURES:
        assert n in names, f"{n} bound but not declared in the header"


def test_version_and_error_string():
    L = _lib.lib()
    assert L.dagr_version() >= 100
    d = _lib.GraphDesc(width=0, height=10, batch_size=1, max_neighbors=16, queue_size=128, radius=4,
                       delta_t_us=10000, time_window=1000000, max_events=100)
    assert L.dagr_graph_workspace_bytes(ctypes.byref(d)) == 0
    assert b"width" in L.dagr_last_error()
    d.width = 64
    assert L.dagr_graph_workspace_bytes(ctypes.byref(d)) > 64 * 10 * 4


def test_spiral_helper_matches_oracle():
    n = 63 * 63
    dx = np.zeros(n, np.int32); dy = np.zeros(n, np.int32)
    assert _lib.lib().dagr_spiral_offsets(n, dx.ctypes.data, dy.ctypes.data) == 0
    ox, oy = og.spiral_offsets(n)
    assert (dx == ox).all() and (dy == oy).all()


def test_argument_validation_of_the_conv_and_elementwise_entry_points():
    """These calls are rejected on the host before any device work, so they run without a GPU."""
    L = _lib.lib()
    # fused pooled-level conv: the 16-node LDS tile holds K = 26*cin + cskip, or -- cut at tap boundaries -- one pass of it
    assert 0 < L.dagr_spline_conv_fused_lds_bytes(64, 64) <= 160 * 1024
    assert 0 < L.dagr_spline_conv_fused_lds_bytes(82, 0) <= 160 * 1024
    assert 0 < L.dagr_spline_conv_fused_lds_bytes(130, 0) <= 160 * 1024       # two passes
    assert 0 < L.dagr_spline_conv_fused_lds_bytes(256, 0) <= 160 * 1024       # four
    assert L.dagr_spline_conv_fused_lds_bytes(2400, 0) > 160 * 1024           # a single tap is wider than the tile
    one = ctypes.c_void_p(16)     # non-NULL, 16-byte aligned, never dereferenced
    rc = L.dagr_spline_conv_fused(None, 1, one, one, one, one, 2400, 2400, None, 0, 0, 7, 7, 14.0, 14.0, one, None, one,
                                  64, 64, 1, None)
    assert rc != 0 and b"does not fit" in L.dagr_last_error()
    assert L.dagr_spline_conv_fused(None, 1, None, one, one, one, 64, 64, None, 0, 0, 7, 7, 14.0, 14.0, one, None, one,
                                    64, 64, 1, None) != 0
    assert b"NULL" in L.dagr_last_error()
    # image-branch elementwise kernels
    assert L.dagr_add_relu(None, one, 8, None) != 0 and b"NULL" in L.dagr_last_error()
    assert L.dagr_add_relu(ctypes.c_void_p(20), one, 8, None) != 0 and b"aligned" in L.dagr_last_error()
    assert L.dagr_add_relu(one, one, 0, None) == 0
    assert L.dagr_bias_relu(one, one, 30, 6, None) != 0 and b"multiple of 4" in L.dagr_last_error()
    assert L.dagr_bias_relu(one, one, 30, 8, None) != 0          # C does not divide n
    assert L.dagr_bias_silu(one, one, 32, 6, None) != 0 and b"multiple of 4" in L.dagr_last_error()
    assert L.dagr_bias_silu(one, one, 0, 8, None) == 0
    assert L.dagr_bn_relu_maxpool(one, 1, 8, 8, 6, one, one, one, None) != 0
    assert b"multiple of 4" in L.dagr_last_error()
