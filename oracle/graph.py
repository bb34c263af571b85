"""CPU oracle for the event-graph builder (numpy + oracle/graph_oracle.c).

TEST INFRASTRUCTURE ONLY -- see oracle/__init__.py.

Restates, op for op (paths relative to /root/reference/src/dagr/):
  * ``graph/utils.py:6-18``  ``_insert_events_into_queue`` (stable sort by linear
    pixel, unique_consecutive, cumsum, then the insert kernel; N==1 takes the
    single-event kernel)
  * ``graph/utils.py:20-23`` ``_search_for_edges`` (fill kernel + mask compaction)
  * ``graph/ev_graph.py:18-103``  ``AsyncGraph`` state machine
  * ``graph/ev_graph.py:106-166`` ``SlidingWindowGraph`` (delete_nodes, forward)
  * ``model/layers/ev_tgn.py:11-16`` ``denormalize_pos``

The C kernels are literal per-thread emulations (oracle/graph_oracle.c).  A
pure-Python emulation of the same kernels (``*_py``) exists for tiny inputs so
the C build itself is cross-checked.
"""
import ctypes
import os
import subprocess

import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))
_LIB_PATH = os.path.join(_HERE, "liboracle_graph.so")
_lib = None

_i32p = ctypes.POINTER(ctypes.c_int32)
_i64p = ctypes.POINTER(ctypes.c_int64)


def build(force=False):
    """gcc-compile oracle/graph_oracle.c -> oracle/liboracle_graph.so."""
    src = os.path.join(_HERE, "graph_oracle.c")
    if force or not os.path.exists(_LIB_PATH) or os.path.getmtime(_LIB_PATH) < os.path.getmtime(src):
        subprocess.check_call(["gcc", "-O2", "-fPIC", "-shared", "-o", _LIB_PATH, src, "-lm"])
    return _LIB_PATH


def lib():
    global _lib
    if _lib is None:
        build()
        _lib = ctypes.CDLL(_LIB_PATH)
        _lib.oracle_spiral_offsets.argtypes = [ctypes.c_int, _i32p, _i32p]
        _lib.oracle_insert_in_queue.argtypes = [_i32p, _i32p, _i32p, _i32p] + [ctypes.c_int] * 5
        _lib.oracle_insert_in_queue_single.argtypes = [_i32p, _i32p, _i32p] + [ctypes.c_int] * 4
        _lib.oracle_fill_edges.argtypes = [_i32p, _i32p, _i32p, _i32p, _i32p, _i64p,
                                           ctypes.c_int, ctypes.c_int, ctypes.c_int, ctypes.c_int,
                                           ctypes.c_int, ctypes.c_int64, ctypes.c_float, ctypes.c_float,
                                           ctypes.c_int, ctypes.c_int]
    return _lib


def _p32(a):
    assert a.dtype == np.int32 and a.flags.c_contiguous
    return a.ctypes.data_as(_i32p)


def spiral_offsets(n):
    """First n offsets of SpiralOut (spiral.h:1-15)."""
    dx = np.zeros(n, np.int32)
    dy = np.zeros(n, np.int32)
    lib().oracle_spiral_offsets(n, _p32(dx), _p32(dy))
    return dx, dy


# --------------------------------------------------------------------------- kernels (python twins, tiny inputs)
def spiral_offsets_py(n):
    layer, leg, x, y = 1, 0, 0, 0
    out = []
    for _ in range(n):
        out.append((x, y))
        if leg == 0:
            x += 1
            if x == layer:
                leg += 1
        elif leg == 1:
            y += 1
            if y == layer:
                leg += 1
        elif leg == 2:
            x -= 1
            if -x == layer:
                leg += 1
        else:
            y -= 1
            if -y == layer:
                leg = 0
                layer += 1
    return out


def fill_edges_py(batch, pos, all_timestamps, indices, queue, edges, radius, delta_t_us, K_nb, min_index):
    """Pure-Python twin of ev_graph.cu:15-80 (tiny inputs only)."""
    B, Q, H, W = queue.shape
    N = len(batch)
    K = edges.shape[1]
    r = int(radius)
    sp = spiral_offsets_py((2 * r + 1) ** 2)
    dt_lim = np.float32(delta_t_us)
    for e in range(N):
        nn = 0
        off = e * K_nb
        b, x, y, ts = int(batch[e]), int(pos[e, 0]), int(pos[e, 1]), int(pos[e, 2])
        edges[0, off] = indices[e] - min_index
        edges[1, off] = indices[e] - min_index
        nn = 1
        for (sx, sy) in sp:
            if nn >= K_nb:
                break
            for q in range(Q):
                xn, yn = x + sx, y + sy
                if not (0 <= xn < W and 0 <= yn < H):
                    break
                idx = int(queue[b, q, yn, xn])
                if idx < min_index:
                    break
                if indices[e] > idx:
                    dt = np.int32(ts - int(all_timestamps[idx - min_index]))
                    if np.float32(dt) > dt_lim:
                        continue
                    edges[0, off + nn] = idx - min_index
                    edges[1, off + nn] = indices[e] - min_index
                    nn += 1
                    if nn >= K_nb:
                        break
    return edges


# --------------------------------------------------------------------------- graph/utils.py
def insert_events_into_queue(batch, pos, indices, queue):
    """graph/utils.py:6-18."""
    B, Q, H, W = queue.shape
    if len(batch) > 1:
        lin = pos[:, 0].astype(np.int64) + W * pos[:, 1].astype(np.int64) + W * H * batch.astype(np.int64)
        sort_index = np.argsort(lin, kind="stable")
        sorted_lin = lin[sort_index]
        sorted_indices = np.ascontiguousarray(indices[sort_index].astype(np.int32))
        uniq, counts = np.unique(sorted_lin, return_counts=True)  # sorted input => == unique_consecutive
        cumsum = np.ascontiguousarray(np.cumsum(counts).astype(np.int32))
        uniq = np.ascontiguousarray(uniq.astype(np.int32))
        lib().oracle_insert_in_queue(_p32(sorted_indices), _p32(uniq), _p32(cumsum), _p32(queue),
                                     B, Q, H, W, len(uniq))
    else:
        ind = np.ascontiguousarray(indices.astype(np.int32))
        ev = np.ascontiguousarray(pos.astype(np.int32))
        lib().oracle_insert_in_queue_single(_p32(ind), _p32(ev), _p32(queue), B, Q, H, W)
    return queue


def search_for_edges(batch, pos, all_timestamps, queue, indices, max_num_neighbors, radius, delta_t_us,
                     edges, min_index):
    """graph/utils.py:20-23."""
    B, Q, H, W = queue.shape
    N = len(batch)
    lib().oracle_fill_edges(_p32(batch), _p32(pos), _p32(all_timestamps), _p32(indices), _p32(queue),
                            edges.ctypes.data_as(_i64p), B, Q, H, W, N, edges.shape[1],
                            float(radius), float(delta_t_us), int(max_num_neighbors), int(min_index))
    return edges[:, edges[1] >= 0]


# --------------------------------------------------------------------------- graph/ev_graph.py
class AsyncGraph:
    """graph/ev_graph.py:18-103."""

    def __init__(self, width=640, height=480, batch_size=1, max_num_neighbors=16, max_queue_size=512,
                 radius=7, delta_t_us=600000):
        self.radius = radius
        self.delta_t_us = delta_t_us
        self.max_index = 0
        self.min_index = 0
        self.max_queue_size = max_queue_size
        self.max_num_neighbors = max_num_neighbors
        self.width = width
        self.height = height
        self.batch_size = batch_size
        self.initialized = False
        self.edges = np.zeros((2, 0), np.int64)
        self.all_timestamps = np.zeros((0,), np.int32)
        self.new_indices = None
        self.edge_buffer = None
        self.event_queue = None

    def initialize(self, n_ev):  # :45-50
        self.edges = np.zeros((2, 0), np.int64)
        self.all_timestamps = np.zeros((0,), np.int32)
        self.new_indices = np.arange(n_ev, dtype=np.int32)
        self.edge_buffer = np.full((2, self.max_num_neighbors * n_ev), -1, np.int64)
        self.event_queue = np.full((self.batch_size, self.max_queue_size, self.height, self.width), -1, np.int32)
        self.initialized = True

    def reset(self):  # :52-60
        self.edges = np.zeros((2, 0), np.int64)
        self.all_timestamps = np.zeros((0,), np.int32)
        self.max_index = 0
        self.min_index = 0
        if self.edge_buffer is not None:
            self.edge_buffer.fill(-1)
        if self.event_queue is not None:
            self.event_queue.fill(-1)

    def _forward(self, batch, pos, collect_edges=True):  # :63-103
        n_ev = len(batch)
        if not self.initialized:
            self.initialize(n_ev)
        if n_ev == 0:
            return np.zeros((2, 0), np.int32)
        assert batch.dtype == np.int32
        batch = np.ascontiguousarray(batch)
        pos = np.ascontiguousarray(pos.astype(np.int32))
        self.all_timestamps = np.concatenate([self.all_timestamps, pos[:, 2]])
        if n_ev > len(self.new_indices):
            self.new_indices = np.arange(0, n_ev, dtype=np.int32)
            self.edge_buffer = np.full((2, self.max_num_neighbors * n_ev), -1, np.int64)
        indices = np.ascontiguousarray((self.max_index + self.new_indices[:n_ev]).astype(np.int32))
        self.max_index += n_ev
        self.event_queue = insert_events_into_queue(batch, pos, indices, self.event_queue)
        self.edge_buffer.fill(-1)
        edge_indices = search_for_edges(batch, pos, np.ascontiguousarray(self.all_timestamps), self.event_queue,
                                        indices, self.max_num_neighbors, self.radius, self.delta_t_us,
                                        self.edge_buffer, self.min_index)
        if collect_edges:
            self.edges = np.concatenate([self.edges, edge_indices], axis=-1)
        return edge_indices


class SlidingWindowGraph(AsyncGraph):
    """graph/ev_graph.py:106-166."""

    def __init__(self, width=640, height=480, batch_size=1, max_num_neighbors=16, max_queue_size=1024,
                 radius=7, delta_t_us=600000):
        AsyncGraph.__init__(self, width, height, batch_size, max_num_neighbors, max_queue_size, radius,
                            delta_t_us)

    @property
    def init(self):
        return len(self.all_timestamps) > 0

    def delete_nodes(self, n_delete, delete_edges=True, return_edges=True):  # :121-137
        self.all_timestamps = self.all_timestamps[n_delete:]
        self.min_index += n_delete
        deleted = None
        if delete_edges:
            mask = (self.edges[0] < n_delete) | (self.edges[1] < n_delete)
            deleted = self.edges[:, mask].copy()
            self.edges = self.edges[:, ~mask]
        self.edges = self.edges - n_delete
        if delete_edges and return_edges:
            return deleted

    def forward(self, batch, pos, return_node_counts=False, return_total_edges=False, delete_nodes=True,
                collect_edges=True):  # :139-166
        n_delete = len(batch) if self.init else 0
        edges = AsyncGraph._forward(self, batch, pos, collect_edges=collect_edges)
        ret = [edges]
        if return_total_edges:
            total_edges = self.edges.copy()
        if return_node_counts:
            tot_nodes = len(self.all_timestamps)
        if delete_nodes:
            ret.append(self.delete_nodes(n_delete))
        if return_total_edges:
            ret.append(total_edges)
        if return_node_counts:
            ret.append([n_delete, len(batch), tot_nodes])
        return ret[0] if len(ret) == 1 else ret


# --------------------------------------------------------------------------- model/layers/ev_tgn.py
def denormalize_pos(pos_norm, width, height, time_window):
    """ev_tgn.py:11-16: ``(denorm * pos + 1e-3).int()`` -- an int64 tensor times fp32
    promotes to fp32; truncation toward zero."""
    denorm = np.array([width, height, time_window], dtype=np.float32).reshape(1, 3)
    return (denorm * pos_norm.astype(np.float32) + np.float32(1e-3)).astype(np.int32)


def graph_params(radius_frac, width, time_window):
    """ev_tgn.py:27-29: delta_t_us and pixel radius derived from args.radius."""
    delta_t_us = int(radius_frac * time_window)
    radius = int(radius_frac * width + 1)
    return radius, delta_t_us


def build_window_graph(x, y, t, b, width, height, batch_size, radius, delta_t_us, K=16, Q=128):
    """One reset=True window through a fresh SlidingWindowGraph
    (ev_tgn.py:39-58 with reset semantics): returns int64[2,E]."""
    g = SlidingWindowGraph(width=width, height=height, batch_size=batch_size, max_num_neighbors=K,
                           max_queue_size=Q, radius=radius, delta_t_us=delta_t_us)
    pos = np.stack([x, y, t], axis=-1).astype(np.int32)
    e = g.forward(np.ascontiguousarray(b.astype(np.int32)), pos, delete_nodes=False, collect_edges=True)
    return e.astype(np.int64)
