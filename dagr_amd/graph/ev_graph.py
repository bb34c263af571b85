"""Host-side mirror of the reference's graph builder classes over libdagr_hip.

Mirrors ``src/dagr/graph/ev_graph.py`` (``AsyncGraph`` :18-103, ``SlidingWindowGraph`` :106-166):
same constructor arguments, ``forward`` / ``reset`` / ``delete_nodes`` names and return shapes
(``int64[2, E]`` in the order of ``edges[:, edges[1] >= 0]``, graph/utils.py:22).

Two implementations sit behind these names:
  * ``AsyncGraph`` / ``SlidingWindowGraph`` are the reference's state machines line for line (persistent
    ``B x Q x H x W`` FIFO volume, timestamp log, growing index, ``delete_nodes``), on top of the 1:1
    replacements of ``ev_graph_cuda`` (``graph/utils.py`` -> csrc/queue_compat.hip).  They cover
    ``reset=False`` incremental use exactly like the reference.
  * ``WindowGraphBuilder`` is the fast path for the ``reset=True`` windows every evaluation script uses
    (``model/networks/dagr.py:74,90``, ``model/layers/ev_tgn.py:45-49``): one fused device pipeline
    (csrc/graph_build.hip) without the FIFO volume; ``EV_TGN`` / the engine use it.
"""
import ctypes

import torch

from .. import _lib
from .utils import _insert_events_into_queue, _search_for_edges


class WindowGraphBuilder:
    """Thin owner of a ``dagr_graph_desc`` + device workspace; neighbour-list output.

    ``build(pos, batch)`` -> ``(nbr_src int32[N,K], nbr_code int16[N,K], deg int32[N])`` on the
    current stream, no host synchronisation.  The lists are in *node (slot) order*: node n is the n-th
    event in (sample, y, x, time) order; ``node_order()`` returns the permutation and ``edge_index``
    the reference-shaped, event-ordered ``int64[2,E]``.
    """

    def __init__(self, width, height, batch_size, max_num_neighbors, max_queue_size, radius, delta_t_us,
                 time_window=1000000, max_events=1 << 16, device="cuda"):
        self.device = torch.device(device)
        self.params = dict(width=int(width), height=int(height), batch_size=int(batch_size),
                           max_neighbors=int(max_num_neighbors), queue_size=int(max_queue_size),
                           radius=int(radius), delta_t_us=int(delta_t_us), time_window=int(time_window))
        self.desc = None
        self.workspace = None
        self._alloc(int(max_events))

    def _alloc(self, max_events):
        L = _lib.lib()
        self.desc = _lib.GraphDesc(max_events=max_events, **self.params)
        nbytes = L.dagr_graph_workspace_bytes(ctypes.byref(self.desc))
        if nbytes == 0:
            raise RuntimeError("libdagr_hip: " + L.dagr_last_error().decode())
        self.workspace = torch.empty(nbytes, dtype=torch.uint8, device=self.device)
        _lib.check(L.dagr_graph_workspace_init(ctypes.byref(self.desc), _lib.ptr(self.workspace), nbytes,
                                               _lib.cur_stream(self.device)), "graph_workspace_init")

    @property
    def K(self):
        return self.params["max_neighbors"]

    def build(self, pos, batch, out=None):
        N = int(pos.shape[0])
        assert pos.is_cuda and pos.is_contiguous() and pos.shape[1] == 3
        if pos.dtype not in (torch.float32, torch.int32):
            raise RuntimeError(f"pos must be float32 (normalised) or int32 (x,y,t_us), got {pos.dtype}")
        if batch.dtype not in (torch.int32, torch.int64):
            raise RuntimeError(f"batch must be int32 or int64, got {batch.dtype}")
        assert batch.is_cuda and batch.is_contiguous() and batch.shape[0] == N
        if N > self.desc.max_events:  # grow like ev_graph.py:78-80
            self._alloc(max(N, 2 * int(self.desc.max_events)))
        K = self.K
        if out is None:
            nbr_src = torch.empty((N, K), dtype=torch.int32, device=self.device)
            nbr_code = torch.empty((N, K), dtype=torch.int16, device=self.device)
            deg = torch.empty((N,), dtype=torch.int32, device=self.device)
        else:
            nbr_src, nbr_code, deg = out
        L = _lib.lib()
        _lib.check(L.dagr_graph_build_window(ctypes.byref(self.desc), _lib.ptr(self.workspace), _lib.ptr(pos),
                                             1 if pos.dtype == torch.int32 else 0, _lib.ptr(batch),
                                             1 if batch.dtype == torch.int64 else 0, N, _lib.ptr(nbr_src),
                                             _lib.ptr(nbr_code), _lib.ptr(deg), _lib.cur_stream(self.device)),
                   "graph_build_window")
        return nbr_src, nbr_code, deg

    def status(self):
        """(num_edges, flags) of the last build; synchronises the current stream."""
        ne, fl = ctypes.c_int64(0), ctypes.c_int32(0)
        _lib.check(_lib.lib().dagr_graph_status(ctypes.byref(self.desc), _lib.ptr(self.workspace),
                                                ctypes.byref(ne), ctypes.byref(fl), _lib.cur_stream(self.device)),
                   "graph_status")
        return ne.value, fl.value

    def node_order(self, N):
        """(slot_event int32[N], event_slot int32[N]): event id of every node / node of every event."""
        slot_event = torch.empty((N,), dtype=torch.int32, device=self.device)
        event_slot = torch.empty((N,), dtype=torch.int32, device=self.device)
        _lib.check(_lib.lib().dagr_graph_node_order(ctypes.byref(self.desc), _lib.ptr(self.workspace), N,
                                                    _lib.ptr(slot_event), _lib.ptr(event_slot),
                                                    _lib.cur_stream(self.device)), "graph_node_order")
        return slot_event, event_slot

    def edge_index(self, nbr_src, deg):
        """Reference-shaped ``int64[2,E]`` in event ids (synchronises to learn E) + rowptr int32[N+1]."""
        N = int(deg.shape[0])
        if N == 0:
            return torch.zeros((2, 0), dtype=torch.int64, device=self.device), \
                torch.zeros((1,), dtype=torch.int32, device=self.device)
        L = _lib.lib()
        rowptr = torch.empty((N + 1,), dtype=torch.int32, device=self.device)
        scratch = torch.empty((L.dagr_scan_scratch_elems(N + 1),), dtype=torch.int32, device=self.device)
        stream = _lib.cur_stream(self.device)
        d, w = ctypes.byref(self.desc), _lib.ptr(self.workspace)
        _lib.check(L.dagr_graph_edge_index(d, w, _lib.ptr(nbr_src), _lib.ptr(deg), N, _lib.ptr(rowptr),
                                           _lib.ptr(scratch), None, 0, stream), "graph_edge_index(rowptr)")
        E = int(rowptr[-1].item())
        edge_index = torch.empty((2, E), dtype=torch.int64, device=self.device)
        if E > 0:
            _lib.check(L.dagr_graph_edge_index(d, w, _lib.ptr(nbr_src), _lib.ptr(deg), N, _lib.ptr(rowptr),
                                               _lib.ptr(scratch), _lib.ptr(edge_index), E, stream),
                       "graph_edge_index")
        return edge_index, rowptr


class AsyncGraph:
    """Mirror of ``ev_graph.py:18-103``."""

    def __init__(self, width=640, height=480, batch_size=1, max_num_neighbors=16, max_queue_size=512, radius=7,
                 delta_t_us=600000):
        self.radius = radius
        self.delta_t_us = delta_t_us
        self.event_queue = None
        self.max_index = 0
        self.min_index = 0
        self.max_queue_size = max_queue_size
        self.max_num_neighbors = max_num_neighbors
        self.width = width
        self.height = height
        self.batch_size = batch_size
        self.device = None
        self.edges = torch.zeros((2, 0), dtype=torch.long)
        self.all_timestamps = torch.zeros((0,), dtype=torch.int32)
        self.new_indices = None
        self.edge_buffer = None

    def initialize(self, n_ev, device):  # :45-50
        self.edges = torch.zeros((2, 0), dtype=torch.long, device=device)
        self.all_timestamps = torch.zeros((0,), dtype=torch.int32, device=device)
        self.new_indices = torch.arange(n_ev, dtype=torch.int32, device=device)
        self.edge_buffer = torch.full((2, self.max_num_neighbors * n_ev), dtype=torch.int64, fill_value=-1,
                                      device=device)
        self.event_queue = torch.full((self.batch_size, self.max_queue_size, self.height, self.width), fill_value=-1,
                                      device=device, dtype=torch.int32)

    def reset(self):  # :52-60
        self.edges = torch.zeros((2, 0), dtype=torch.long, device=self.device)
        self.all_timestamps = torch.zeros((0,), dtype=torch.int32, device=self.device)
        self.max_index = 0
        self.min_index = 0
        if self.edge_buffer is not None:
            self.edge_buffer.fill_(-1)
        if self.event_queue is not None:
            self.event_queue.fill_(-1)

    def _forward(self, batch, pos, collect_edges=True):  # :63-103
        n_ev = len(batch)
        if not batch.is_cuda:  # the reference's CPU shim never triggers (ev_graph.py:7-8); its kernels assert CUDA
            raise RuntimeError("batch must be a CUDA tensor")
        if self.device is None:
            self.device = batch.device
            self.initialize(n_ev, self.device)
        if len(batch) == 0:
            return torch.zeros((2, 0), device=self.device, dtype=torch.int32)
        assert type(batch) is torch.Tensor and batch.dtype == torch.int32, [type(batch), batch.dtype]
        pos = pos.int().contiguous()
        batch = batch.contiguous()
        self.all_timestamps = torch.cat([self.all_timestamps, pos[:, 2]])
        if n_ev > len(self.new_indices):
            self.new_indices = torch.arange(0, n_ev, dtype=torch.int32, device=self.device)
            self.edge_buffer = torch.full((2, self.max_num_neighbors * n_ev), dtype=torch.int64, fill_value=-1,
                                          device=self.device)
        indices = (self.max_index + self.new_indices[:n_ev]).contiguous()
        self.max_index += n_ev
        self.event_queue = _insert_events_into_queue(batch, pos, indices=indices, queue=self.event_queue)
        self.edge_buffer.fill_(-1)
        edge_indices = _search_for_edges(batch, pos, all_timestamps=self.all_timestamps.contiguous(), indices=indices,
                                         queue=self.event_queue, max_num_neighbors=self.max_num_neighbors,
                                         radius=self.radius, delta_t_us=self.delta_t_us, edges=self.edge_buffer,
                                         min_index=self.min_index)
        if collect_edges:
            self.edges = torch.cat([self.edges, edge_indices], dim=-1)
        return edge_indices

    def forward(self, batch, pos, collect_edges=True):
        return self._forward(batch, pos, collect_edges=collect_edges)


class SlidingWindowGraph(AsyncGraph):
    """``ev_graph.py:106-166``: ``AsyncGraph`` plus dropping the oldest nodes after every call."""

    def __init__(self, width=640, height=480, batch_size=1, max_num_neighbors=16, max_queue_size=1024, radius=7,
                 delta_t_us=600000):
        super().__init__(width, height, batch_size, max_num_neighbors, max_queue_size, radius, delta_t_us)

    @property
    def init(self):
        return self.all_timestamps.numel() > 0

    def delete_nodes(self, n_delete, delete_edges=True, return_edges=True):
        """Forget the n oldest nodes: shift the index origin; edges touching them are removed (and returned)."""
        self.all_timestamps = self.all_timestamps[n_delete:]
        self.min_index += n_delete
        removed = None
        if delete_edges:
            touches_old = (self.edges < n_delete).any(dim=0)
            removed = self.edges[:, touches_old].clone()
            self.edges = self.edges[:, ~touches_old]
        self.edges.add_(-n_delete)
        return removed if (delete_edges and return_edges) else None

    def forward(self, batch, pos, return_node_counts=False, return_total_edges=False, delete_nodes=True,
                collect_edges=True):
        n_old = len(batch) if self.init else 0
        out = [self._forward(batch, pos, collect_edges=collect_edges)]
        snapshot = self.edges.clone() if return_total_edges else None
        n_total = len(self.all_timestamps)
        if delete_nodes:
            out.append(self.delete_nodes(n_old))
        if return_total_edges:
            out.append(snapshot)
        if return_node_counts:
            out.append([n_old, len(batch), n_total])
        return out[0] if len(out) == 1 else out
