"""Mirror of ``src/dagr/graph/utils.py`` (``_insert_events_into_queue`` :6-18, ``_search_for_edges`` :20-23)
over the 1:1 replacements of ``ev_graph_cuda`` in libdagr_hip (csrc/queue_compat.hip).  Same host
preparation as the reference (stable sort by linear pixel, unique_consecutive, cumsum; boolean-mask
compaction of the -1-filled edge buffer)."""
import torch

from .. import _lib


def _check(t, name):
    if not t.is_cuda:
        raise RuntimeError(f"{name} must be a CUDA tensor")          # ev_graph.cu:9
    if not t.is_contiguous():
        raise RuntimeError(f"{name} must be contiguous")             # ev_graph.cu:10


def _insert_events_into_queue(batch, pos, indices, queue):
    B, Q, H, W = queue.shape
    L, P = _lib.lib(), _lib.ptr
    stream = _lib.cur_stream(queue.device)
    if len(batch) > 1:
        lin_coords = pos[:, 0] + W * pos[:, 1] + W * H * batch
        sorted_lin_coords, sort_index = torch.sort(lin_coords, stable=True, descending=False)
        sorted_indices = indices[sort_index].int().contiguous()
        unique_coords, unique_counter = torch.unique_consecutive(sorted_lin_coords, return_counts=True)
        cumsum_counter = torch.cumsum(unique_counter, dim=0).int().contiguous()
        unique_coords = unique_coords.int().contiguous()
        for t, n in ((sorted_indices, "indices"), (unique_coords, "unique_coords"), (cumsum_counter, "cumsum_counts"),
                     (queue, "queue")):
            _check(t, n)
        _lib.check(L.dagr_insert_in_queue(P(sorted_indices), P(unique_coords), P(cumsum_counter), len(unique_coords),
                                          P(queue), B, Q, H, W, stream), "insert_in_queue")
    else:
        ind = indices.int().contiguous()
        ev = pos.int().contiguous()
        _lib.check(L.dagr_insert_in_queue_single(P(ind), P(ev), P(queue), B, Q, H, W, stream), "insert_in_queue_single")
    return queue


def _search_for_edges(batch, pos, all_timestamps, queue, indices, max_num_neighbors, radius, delta_t_us, edges,
                      min_index):
    B, Q, H, W = queue.shape
    for t, n in ((batch, "batch"), (pos, "pos"), (queue, "event_queue"), (all_timestamps, "all_timestamps"),
                 (edges, "edges"), (indices, "indices")):
        _check(t, n)
    _lib.check(_lib.lib().dagr_fill_edges(_lib.ptr(batch), _lib.ptr(pos), _lib.ptr(all_timestamps), _lib.ptr(queue),
                                          _lib.ptr(indices), int(max_num_neighbors), float(radius), float(delta_t_us),
                                          _lib.ptr(edges), edges.shape[1], int(min_index), len(batch), B, Q, H, W,
                                          _lib.cur_stream(queue.device)), "fill_edges")
    return edges[:, (edges[1] >= 0)]
