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
    return sorted(set(re.findall(r"\b(dagr_[a-z0-9_]+)\s*\(", src)))


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
