"""Run the reference's OWN graph-builder kernels (oracle/_ref/libev_graph_ref.so, compiled from
/root/reference/src/dagr/graph/ev_graph.cu by oracle/Makefile) on the GPU box.

TEST INFRASTRUCTURE ONLY.  The host preparation around the kernels restates
``src/dagr/graph/utils.py:6-23`` and ``ev_graph.py:52-103`` with torch ops on the device, exactly as
the reference does (stable sort, unique_consecutive, cumsum, -1-filled int64 buffer, mask).
"""
import ctypes
import os

import torch

_HERE = os.path.dirname(os.path.abspath(__file__))
REF_LIB = os.path.join(_HERE, "_ref", "libev_graph_ref.so")
REF_ASY_LIB = os.path.join(_HERE, "_ref", "libasy_tools_ref.so")
_lib = None
_asy = None


def asy_available():
    return os.path.exists(REF_ASY_LIB) and torch.cuda.is_available()


def asy_lib():
    """The reference's own asy_tools kernels (src/dagr/asynchronous/asy_tools/main.cu via oracle/ref_driver_asy.hip)."""
    global _asy
    if _asy is None:
        _asy = ctypes.CDLL(REF_ASY_LIB)
        vp, ci, cf = ctypes.c_void_p, ctypes.c_int, ctypes.c_float
        _asy.ref_masked_lin.argtypes = [vp, ci, vp, vp, ci, vp, vp, ci, ci, ci]
        _asy.ref_masked_lin_no_bias.argtypes = [vp, ci, vp, vp, ci, vp, ci, ci, ci]
        _asy.ref_masked_isdiff.argtypes = [vp, ci, vp, vp, ci, ci, cf, cf]
        _asy.ref_masked_inplace_BN.argtypes = [vp, ci, vp, vp, ci, ci, vp, vp, vp, vp, cf]
    return _asy


def available():
    return os.path.exists(REF_LIB) and torch.cuda.is_available()


def lib():
    global _lib
    if _lib is None:
        _lib = ctypes.CDLL(REF_LIB)
        vp, ci, cl, cf = ctypes.c_void_p, ctypes.c_int, ctypes.c_int64, ctypes.c_float
        _lib.ref_fill_edges.argtypes = [vp, vp, vp, cl, vp, vp, ci, cf, cf, vp, cl, ci, ci, ci, ci, ci, ci]
        _lib.ref_insert_in_queue.argtypes = [vp, ci, vp, vp, ci, vp, ci, ci, ci, ci]
        _lib.ref_insert_in_queue_single.argtypes = [vp, vp, vp, ci, ci, ci, ci]
    return _lib


def _p(t):
    return ctypes.c_void_p(t.data_ptr())


def reference_window_graph(x, y, t, b, W, H, B, r, dt, K=16, Q=128, device="cuda:0"):
    """One reset=True window through the reference kernels; returns int64[2,E] (numpy)."""
    dev = torch.device(device)
    N = len(x)
    if N == 0:
        return torch.zeros((2, 0), dtype=torch.int64).numpy()
    L = lib()
    batch = torch.as_tensor(b, dtype=torch.int32, device=dev).contiguous()
    pos = torch.stack([torch.as_tensor(x, dtype=torch.int32), torch.as_tensor(y, dtype=torch.int32),
                       torch.as_tensor(t, dtype=torch.int32)], dim=-1).to(dev).contiguous()
    queue = torch.full((B, Q, H, W), -1, dtype=torch.int32, device=dev)            # ev_graph.py:50
    all_ts = pos[:, 2].contiguous()                                                  # ev_graph.py:75
    indices = torch.arange(N, dtype=torch.int32, device=dev)                        # ev_graph.py:82
    if N > 1:                                                                        # graph/utils.py:7-14
        lin = pos[:, 0] + W * pos[:, 1] + W * H * batch
        sorted_lin, sort_index = torch.sort(lin, stable=True, descending=False)
        sorted_indices = indices[sort_index].int().contiguous()
        uniq, counts = torch.unique_consecutive(sorted_lin, return_counts=True)
        cumsum = torch.cumsum(counts, dim=0).int().contiguous()
        uniq = uniq.int().contiguous()
        rc = L.ref_insert_in_queue(_p(sorted_indices), N, _p(uniq), _p(cumsum), len(uniq), _p(queue), B, Q, H, W)
    else:                                                                            # graph/utils.py:15-16
        rc = L.ref_insert_in_queue_single(_p(indices), _p(pos), _p(queue), B, Q, H, W)
    assert rc == 0
    edges = torch.full((2, K * N), -1, dtype=torch.int64, device=dev)               # ev_graph.py:49,89
    rc = L.ref_fill_edges(_p(batch), _p(pos), _p(all_ts), N, _p(queue), _p(indices), K, float(r), float(dt),
                          _p(edges), K * N, 0, N, B, Q, H, W)
    assert rc == 0
    edges = edges[:, edges[1] >= 0]                                                  # graph/utils.py:22
    return edges.cpu().numpy()
