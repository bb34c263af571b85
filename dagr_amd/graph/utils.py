"""Host side of the two queue entry points of libdagr_hip (csrc/queue_compat.hip), the 1:1 replacements of the
reference's native module ``ev_graph_cuda`` (call sites in the reference: ``src/dagr/graph/utils.py:6-23``).

The native insert kernel consumes events *grouped by pixel*: ids sorted by linear pixel coordinate (stable, so ids
stay time-ordered inside a pixel), the distinct coordinates, and the running end offset of every group -- that input
contract is the reference kernel's (ev_graph.cu:169-212) and is prepared here with three torch ops."""
import torch

from .. import _lib


def _require_device(**tensors):
    """The reference's AT_ASSERTM checks (ev_graph.cu:9-12): RuntimeError for host or strided tensors."""
    for name, t in tensors.items():
        if not t.is_cuda:
            raise RuntimeError(f"{name} must be a CUDA tensor")
        if not t.is_contiguous():
            raise RuntimeError(f"{name} must be contiguous")


def group_by_pixel(batch, pos, ids, height, width):
    """(ids ordered by pixel, distinct linear pixel coordinates, end offset of every pixel's run), all int32."""
    key = (batch * height + pos[:, 1]) * width + pos[:, 0]
    key, order = torch.sort(key, stable=True)
    pixels, run = torch.unique_consecutive(key, return_counts=True)
    return ids[order].int().contiguous(), pixels.int().contiguous(), run.cumsum(0).int().contiguous()


def push_events(volume, batch, pos, ids):
    """Shift the FIFO columns of the touched pixels and write the new ids (slot 0 = newest), in place."""
    B, Q, H, W = volume.shape
    L, P = _lib.lib(), _lib.ptr
    stream = _lib.cur_stream(volume.device)
    if batch.numel() == 1:      # the reference routes a lone event to its own kernel (ev_graph.cu:130-166)
        one, xy = ids.int().contiguous(), pos.int().contiguous()
        _require_device(indices=one, events=xy, queue=volume)
        _lib.check(L.dagr_insert_in_queue_single(P(one), P(xy), P(volume), B, Q, H, W, stream), "insert_in_queue_single")
        return volume
    by_pixel, pixels, ends = group_by_pixel(batch, pos, ids, H, W)
    _require_device(indices=by_pixel, unique_coords=pixels, cumsum_counts=ends, queue=volume)
    _lib.check(L.dagr_insert_in_queue(P(by_pixel), P(pixels), P(ends), pixels.numel(), P(volume), B, Q, H, W, stream),
               "insert_in_queue")
    return volume


def connect_events(volume, batch, pos, timestamps, ids, max_num_neighbors, radius, delta_t_us, scratch, origin):
    """Spiral search of every new event's neighbourhood in the FIFO volume; ``scratch`` (-1-filled int64[2, K*n]) receives
    (source, destination) columns, the filled ones are returned in order."""
    B, Q, H, W = volume.shape
    _require_device(batch=batch, pos=pos, event_queue=volume, all_timestamps=timestamps, edges=scratch, indices=ids)
    _lib.check(_lib.lib().dagr_fill_edges(_lib.ptr(batch), _lib.ptr(pos), _lib.ptr(timestamps), _lib.ptr(volume),
                                          _lib.ptr(ids), int(max_num_neighbors), float(radius), float(delta_t_us),
                                          _lib.ptr(scratch), scratch.shape[1], int(origin), batch.numel(), B, Q, H, W,
                                          _lib.cur_stream(volume.device)), "fill_edges")
    used = scratch[:, :max_num_neighbors * batch.numel()]
    return used[:, used[1] >= 0]
