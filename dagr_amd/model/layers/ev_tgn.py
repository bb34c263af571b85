"""Mirror of ``src/dagr/model/layers/ev_tgn.py`` (EV_TGN :19-58): lazily creates the graph builder
from the first batch's (width, height, time_window, num_graphs); ``reset`` semantics as the
reference (every eval call resets)."""
import torch

from ...graph.ev_graph import WindowGraphBuilder


def _get_value_as_int(obj, key):
    val = getattr(obj, key)
    return int(val) if isinstance(val, (int, float)) else int(val[0])


class EV_TGN(torch.nn.Module):
    def __init__(self, args):
        super().__init__()
        self.radius = args.radius
        self.max_neighbors = args.max_neighbors
        self.max_queue_size = 128  # ev_tgn.py:24
        self.graph_creators = None

    def init_graph_creator(self, width, height, time_window, batch_size, device, max_events=1 << 16):
        delta_t_us = int(self.radius * time_window)   # ev_tgn.py:28
        radius = int(self.radius * width + 1)          # ev_tgn.py:29
        self.graph_creators = WindowGraphBuilder(width, height, batch_size, self.max_neighbors,
                                                 self.max_queue_size, radius, delta_t_us,
                                                 time_window=time_window, max_events=max_events, device=device)
        return self.graph_creators
