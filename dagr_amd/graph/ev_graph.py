"""Host-side mirror of the reference's graph builder classes over libdagr_hip.

Mirrors ``src/dagr/graph/ev_graph.py`` (``AsyncGraph`` :18-103, ``SlidingWindowGraph`` :106-166):
same constructor arguments, ``forward`` / ``reset`` / ``delete_nodes`` names and return shapes
(``int64[2, E]`` in the order of ``edges[:, edges[1] >= 0]``, graph/utils.py:22).

Two implementations sit behind these names:
  * ``AsyncGraph`` / ``SlidingWindowGraph`` keep the reference's interface and semantics (persistent
    ``B x Q x H x W`` FIFO volume, ever-growing node ids, ``delete_nodes``) for ``reset=False`` incremental use;
    their state is one ``EventQueueState`` object over the 1:1 replacements of ``ev_graph_cuda``
    (``graph/utils.py`` -> csrc/queue_compat.hip).
  * ``WindowGraphBuilder`` is the fast path for the ``reset=True`` windows every evaluation script uses
    (``model/networks/dagr.py:74,90``, ``model/layers/ev_tgn.py:45-49``): one fused device pipeline
    (csrc/graph_build.hip) without the FIFO volume; ``EV_TGN`` / the engine use it.
"""
import ctypes

import torch

from .. import _lib
from .utils import push_events, connect_events


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

    def node_count_ptr(self):
        """Device address of the node count of the last build (the ``n_ptr`` of the kernels that follow it in a captured
        window), as a ctypes void pointer."""
        import ctypes as _c
        return _c.c_void_p(_lib.lib().dagr_graph_node_count_ptr(ctypes.byref(self.desc), _lib.ptr(self.workspace)))

    def build(self, pos, batch, out=None, n_dev=None, inputs=None):
        """``n_dev`` (int32[1] on the device): the event count lives in device memory -- ``pos`` / ``batch`` are capacity-sized
        buffers and every launch is bounded by ``*n_dev`` on the device (a window captured once as a HIP graph then serves
        windows of any size)."""
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
        if inputs is not None:
            # ``inputs`` (_lib.L0Inputs): the node-ordered level-0 inputs come out of the build's last launch
            _lib.check(L.dagr_graph_build_window_inputs(ctypes.byref(self.desc), _lib.ptr(self.workspace), _lib.ptr(pos),
                                                        _lib.ptr(batch), 1 if batch.dtype == torch.int64 else 0, N,
                                                        _lib.ptr(n_dev) if n_dev is not None else None, _lib.ptr(nbr_src),
                                                        _lib.ptr(nbr_code), _lib.ptr(deg), ctypes.byref(inputs),
                                                        _lib.cur_stream(self.device)), "graph_build_window_inputs")
            return nbr_src, nbr_code, deg
        if n_dev is not None:
            _lib.check(L.dagr_graph_build_window_dev(ctypes.byref(self.desc), _lib.ptr(self.workspace), _lib.ptr(pos),
                                                     1 if pos.dtype == torch.int32 else 0, _lib.ptr(batch),
                                                     1 if batch.dtype == torch.int64 else 0, N, _lib.ptr(n_dev),
                                                     _lib.ptr(nbr_src), _lib.ptr(nbr_code), _lib.ptr(deg),
                                                     _lib.cur_stream(self.device)), "graph_build_window_dev")
            return nbr_src, nbr_code, deg
        _lib.check(L.dagr_graph_build_window(ctypes.byref(self.desc), _lib.ptr(self.workspace), _lib.ptr(pos),
                                             1 if pos.dtype == torch.int32 else 0, _lib.ptr(batch),
                                             1 if batch.dtype == torch.int64 else 0, N, _lib.ptr(nbr_src),
                                             _lib.ptr(nbr_code), _lib.ptr(deg), _lib.cur_stream(self.device)),
                   "graph_build_window")
        return nbr_src, nbr_code, deg

    def search_again(self, out):
        """The neighbour search alone on the pixel index of the last ``build`` (measurement: bench.py)."""
        nbr_src, nbr_code, deg = out
        _lib.check(_lib.lib().dagr_graph_search_window(ctypes.byref(self.desc), _lib.ptr(self.workspace), int(deg.shape[0]),
                                                       _lib.ptr(nbr_src), _lib.ptr(nbr_code), _lib.ptr(deg),
                                                       _lib.cur_stream(self.device)), "graph_search_window")

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

    def csr_codes(self, nbr_src, nbr_code, deg, code_bias):
        """(rowptr int32[N+1] by destination event, col int32[N*K] source event ids, code int32[N*K] offset codes
        (dx + code_bias) | (dy + code_bias) << 16): the graph as the convolutions consume it, no host synchronisation
        (entries past rowptr[N] are unused capacity)."""
        N = int(deg.shape[0])
        L = _lib.lib()
        rowptr = torch.empty((N + 1,), dtype=torch.int32, device=self.device)
        cap = max(1, N * self.K)
        col = torch.empty((cap,), dtype=torch.int32, device=self.device)
        code = torch.empty((cap,), dtype=torch.int32, device=self.device)
        scratch = torch.empty((L.dagr_scan_scratch_elems(N + 1),), dtype=torch.int32, device=self.device)
        _lib.check(L.dagr_graph_csr_codes(ctypes.byref(self.desc), _lib.ptr(self.workspace), _lib.ptr(nbr_src),
                                          _lib.ptr(nbr_code), _lib.ptr(deg), N, int(code_bias), _lib.ptr(rowptr),
                                          _lib.ptr(scratch), _lib.ptr(col), _lib.ptr(code), cap,
                                          _lib.cur_stream(self.device)), "graph_csr_codes")
        return rowptr, col, code

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


class EventQueueState:
    """Device-resident state of a running event graph: the per-pixel FIFO volume ``int32[B,Q,H,W]`` (-1 = empty, slot 0
    newest), the log of node timestamps and the id window ``[origin, next_id)`` of the nodes still alive.  Owns the calls
    into ``libdagr_hip``'s queue entry points (``graph/utils.py``).

    Unlike a tensor that is re-concatenated per call, the timestamp log and the list of collected edges live in
    capacity-doubling buffers with a fill mark, so a long ``reset=False`` stream costs amortised O(1) copies per
    event; ``timestamps`` / ``edges`` are views of the live part."""

    def __init__(self, shape, neighbors, radius, delta_t_us):
        self.shape = tuple(int(v) for v in shape)          # (B, Q, H, W)
        self.neighbors, self.radius, self.delta_t_us = int(neighbors), radius, delta_t_us
        self.device = None
        self.volume = None
        self.origin = 0                                   # id of the oldest node still alive
        self.next_id = 0                                  # id the next inserted event receives
        self._ts, self._ts_lo, self._ts_hi = None, 0, 0   # log buffer; live part = [_ts_lo, _ts_hi)
        self._edges, self._n_edges = None, 0
        self._ids = None                                  # arange scratch
        self._scratch = None                              # -1-filled int64[2, K*n] the search kernel writes into

    # -- storage ---------------------------------------------------------------------------------------------------
    def attach(self, device):
        if self.device is None:
            self.device = device
            self.volume = torch.full(self.shape, -1, dtype=torch.int32, device=device)
            self._ts = torch.empty((1024,), dtype=torch.int32, device=device)
            self._edges = torch.empty((2, 1024), dtype=torch.int64, device=device)

    def clear(self):
        self.origin = self.next_id = 0
        self._ts_lo = self._ts_hi = self._n_edges = 0
        if self.volume is not None:
            self.volume.fill_(-1)

    @property
    def timestamps(self):
        if self._ts is None:
            return torch.zeros((0,), dtype=torch.int32)
        return self._ts[self._ts_lo:self._ts_hi]

    @property
    def edges(self):
        if self._edges is None:
            return torch.zeros((2, 0), dtype=torch.long)
        return self._edges[:, :self._n_edges]

    def _log_timestamps(self, t):
        n = t.numel()
        if self._ts_hi + n > self._ts.numel():            # compact the dead prefix away, then grow if still needed
            live = self._ts[self._ts_lo:self._ts_hi].clone()
            if live.numel() + n > self._ts.numel():
                self._ts = torch.empty((2 * (live.numel() + n),), dtype=torch.int32, device=self.device)
            self._ts[:live.numel()] = live
            self._ts_lo, self._ts_hi = 0, live.numel()
        self._ts[self._ts_hi:self._ts_hi + n] = t
        self._ts_hi += n

    def _collect(self, e):
        n = e.shape[1]
        if self._n_edges + n > self._edges.shape[1]:
            grown = torch.empty((2, 2 * (self._n_edges + n)), dtype=torch.int64, device=self.device)
            grown[:, :self._n_edges] = self._edges[:, :self._n_edges]
            self._edges = grown
        self._edges[:, self._n_edges:self._n_edges + n] = e
        self._n_edges += n

    # -- one call: insert the events, connect them ---------------------------------------------------------------------
    def append(self, batch, pos, collect):
        """Push ``n`` events (ids ``next_id ..``) into the FIFO volume, then search their neighbourhoods.
        Returns ``int64[2, E]`` with source / destination ids relative to ``origin``-free numbering as the native
        module writes them (global ids)."""
        n = int(batch.shape[0])
        if self._ids is None or self._ids.numel() < n:
            self._ids = torch.arange(n, dtype=torch.int32, device=self.device)
            self._scratch = torch.empty((2, self.neighbors * n), dtype=torch.int64, device=self.device)
        ids = (self._ids[:n] + self.next_id).contiguous()
        self.next_id += n
        self._log_timestamps(pos[:, 2])
        push_events(self.volume, batch, pos, ids)
        self._scratch.fill_(-1)
        found = connect_events(self.volume, batch, pos, self.timestamps.contiguous(), ids, self.neighbors, self.radius,
                               self.delta_t_us, self._scratch, self.origin)
        if collect:
            self._collect(found)
        return found

    def forget_oldest(self, n, drop_edges):
        """The ``n`` oldest nodes leave the graph: ids are renumbered from the new origin.  Collected edges that touch
        a forgotten node are cut out (and returned) when ``drop_edges``."""
        self._ts_lo = min(self._ts_lo + n, self._ts_hi)
        self.origin += n
        cut = None
        live = self.edges
        if drop_edges:
            dead = live.min(dim=0).values < n if live.shape[1] else torch.zeros((0,), dtype=torch.bool, device=live.device)
            cut = live[:, dead].clone()
            kept = live[:, ~dead]
            self._n_edges = kept.shape[1]
            self._edges[:, :self._n_edges] = kept
        self.edges.sub_(n)
        return cut


class AsyncGraph:
    """Same constructor / ``forward`` / ``reset`` as ``ev_graph.py:18-103``; the state lives in ``EventQueueState``."""

    def __init__(self, width=640, height=480, batch_size=1, max_num_neighbors=16, max_queue_size=512, radius=7,
                 delta_t_us=600000):
        self.width, self.height, self.batch_size = width, height, batch_size
        self.max_num_neighbors, self.max_queue_size = max_num_neighbors, max_queue_size
        self.radius, self.delta_t_us = radius, delta_t_us
        self.state = EventQueueState((batch_size, max_queue_size, height, width), max_num_neighbors, radius, delta_t_us)

    # the reference's attribute names, as read-only views of the state
    device = property(lambda self: self.state.device)
    event_queue = property(lambda self: self.state.volume)
    all_timestamps = property(lambda self: self.state.timestamps)
    edges = property(lambda self: self.state.edges)
    min_index = property(lambda self: self.state.origin)
    max_index = property(lambda self: self.state.next_id)

    def reset(self):
        self.state.clear()

    def forward(self, batch, pos, collect_edges=True):
        if not batch.is_cuda:   # the native kernels assert CUDA tensors (ev_graph.cu:9); there is no CPU path
            raise RuntimeError("batch must be a CUDA tensor")
        self.state.attach(batch.device)
        if batch.numel() == 0:
            return torch.zeros((2, 0), device=batch.device, dtype=torch.int32)
        if batch.dtype != torch.int32:
            raise AssertionError([type(batch), batch.dtype])
        return self.state.append(batch.contiguous(), pos.int().contiguous(), collect_edges)


class SlidingWindowGraph(AsyncGraph):
    """``ev_graph.py:106-166``: after every call the window slides -- as many of the oldest nodes leave as new ones
    arrived (unless ``delete_nodes=False``)."""

    def __init__(self, width=640, height=480, batch_size=1, max_num_neighbors=16, max_queue_size=1024, radius=7,
                 delta_t_us=600000):
        super().__init__(width, height, batch_size, max_num_neighbors, max_queue_size, radius, delta_t_us)

    @property
    def init(self):
        return self.state.timestamps.numel() > 0

    def delete_nodes(self, n_delete, delete_edges=True, return_edges=True):
        cut = self.state.forget_oldest(int(n_delete), delete_edges)
        return cut if (delete_edges and return_edges) else None

    def forward(self, batch, pos, return_node_counts=False, return_total_edges=False, delete_nodes=True,
                collect_edges=True):
        leaving = int(batch.shape[0]) if self.init else 0
        result = [AsyncGraph.forward(self, batch, pos, collect_edges=collect_edges)]
        everything = self.edges.clone() if return_total_edges else None
        alive = int(self.state.timestamps.numel())
        if delete_nodes:
            result.append(self.delete_nodes(leaving))
        if return_total_edges:
            result.append(everything)
        if return_node_counts:
            result.append([leaving, int(batch.shape[0]), alive])
        return result if len(result) > 1 else result[0]
