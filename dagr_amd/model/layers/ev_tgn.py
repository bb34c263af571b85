"""Mirror of ``src/dagr/model/layers/ev_tgn.py`` (EV_TGN :19-58).

``forward(events, reset)`` is the reference's: it lazily creates the stateful ``SlidingWindowGraph`` from the first
batch's (width, height, time_window, num_graphs), resets it on ``reset=True`` calls, and writes ``events.edge_index``
(int64[2, E], event ids, destinations ascending, self loop first) -- including ``reset=False`` calls, whose new nodes
attach to the running graph.  ``DAGR.forward`` on whole ``reset=True`` windows does not go through it: the engine asks
``window_builder()`` for the fused single-window builder (csrc/graph_build.hip), which produces the same edge set as
fixed-stride neighbour lists without the FIFO volume."""
import os

import torch

from ...graph.ev_graph import SlidingWindowGraph, WindowGraphBuilder


def _get_value_as_int(obj, key):
    geo = getattr(obj, "_geometry", None)        # (width, height, time_window) as read once by format_data
    if geo is not None and key in ("width", "height", "time_window"):
        return geo[("width", "height", "time_window").index(key)]
    val = getattr(obj, key)
    return int(val) if isinstance(val, (int, float)) else int(val[0])


def denormalize_pos(events):
    """``ev_tgn.py:11-16``: int(pos * [W, H, T] + 1e-3)."""
    if hasattr(events, "pos_denorm"):
        return events.pos_denorm
    denorm = torch.tensor([_get_value_as_int(events, "width"), _get_value_as_int(events, "height"),
                           _get_value_as_int(events, "time_window")], device=events.pos.device)
    return (denorm.view(1, -1) * events.pos + 1e-3).int()


class EV_TGN(torch.nn.Module):
    def __init__(self, args):
        super().__init__()
        self.radius = args.radius
        self.max_neighbors = args.max_neighbors
        self.max_queue_size = 128  # ev_tgn.py:24
        self.graph_creators = None
        self._builder = None

    def _geometry(self, width, time_window):
        return int(self.radius * width + 1), int(self.radius * time_window)     # ev_tgn.py:29, :28

    def init_graph_creator(self, data):
        width, height = _get_value_as_int(data, "width"), _get_value_as_int(data, "height")
        radius, delta_t_us = self._geometry(width, _get_value_as_int(data, "time_window"))
        self.graph_creators = SlidingWindowGraph(width=width, height=height, max_num_neighbors=self.max_neighbors,
                                                 max_queue_size=self.max_queue_size, batch_size=data.num_graphs,
                                                 radius=radius, delta_t_us=delta_t_us)

    def window_builder(self, width, height, time_window, batch_size, device, max_events=1 << 16):
        radius, delta_t_us = self._geometry(width, time_window)
        self._builder = WindowGraphBuilder(width, height, batch_size, self.max_neighbors, self.max_queue_size, radius,
                                           delta_t_us, time_window=time_window, max_events=max_events, device=device)
        return self._builder

    def _training_graph(self, events):
        """Training windows are always independent (``reset=True``, train_ncaltech101.py:56): their graph comes from the
        single-window builder (csrc/graph_build.hip) instead of the FIFO volume -- same edge set, emitted grouped by
        destination event together with its row pointer, so the convolutions' CSR needs no sort either."""
        width, height = _get_value_as_int(events, "width"), _get_value_as_int(events, "height")
        tw = _get_value_as_int(events, "time_window")
        key = (width, height, tw, int(events.num_graphs), str(events.pos.device))
        if getattr(self, "_train_key", None) != key:
            self._train_builder = WindowGraphBuilder(width, height, events.num_graphs, self.max_neighbors,
                                                     self.max_queue_size, *self._geometry(width, tw), time_window=tw,
                                                     max_events=max(1 << 16, int(events.pos.shape[0])),
                                                     device=events.pos.device)
            self._train_key = key
        b = self._train_builder
        from . import _ops
        nbr_src, nbr_code, deg = b.build(events.pos.float().contiguous(), events.batch.contiguous())
        # the graph as the convolutions consume it -- CSR by destination + the integer pixel offset of every edge (what
        # its Cartesian attribute encodes) -- straight from the builder, without a host round trip; the reference-shaped
        # ``edge_index`` is built from it on first access (nothing in the training forward reads it)
        rowptr, col, code = b.csr_codes(nbr_src, nbr_code, deg, _ops.EXACT_R)
        n = int(events.pos.shape[0])
        events._dagr_csr = (rowptr, col, None, ("csr", n))
        events._dagr_pixel_codes = (code, width, height)
        events.set_lazy("edge_index", _ops.edge_index_from_csr)
        return events

    def forward(self, events, reset=True):
        if getattr(events, "batch", None) is None:
            events.batch = torch.zeros(events.pos.shape[0], dtype=torch.long, device=events.pos.device)
        if self.training and reset and events.pos.is_cuda and os.environ.get("DAGR_TRAIN_FAST_GRAPH", "1") != "0":
            return self._training_graph(events)
        if self.graph_creators is None:
            self.init_graph_creator(events)
        elif reset:
            self.graph_creators.reset()
        pos = denormalize_pos(events)
        events.edge_index = self.graph_creators.forward(events.batch.int(), pos, delete_nodes=False,
                                                        collect_edges=reset).long()
        return events
