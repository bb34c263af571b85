"""Event-stream downsampling on the device: ``downsample_events`` of ``scripts/downsample_events.py:91-106`` (the step
that turns DSEC's 640x480 ``events.h5`` into the half-resolution ``events_2x.h5`` the reference trains and tests on).
Same contract: events as a dict of arrays (here device tensors ``x, y`` integer, ``t``, ``p`` in {-1, +1}), the
``change_map`` integrator state carried from chunk to chunk, returns the surviving events with ``x // fx``, ``y // fy``."""
import torch

from .. import _lib


def downsample_events(events, input_height, input_width, output_height, output_width, change_map=None):
    x, y, p = events["x"], events["y"], events["p"]
    dev = x.device
    if not x.is_cuda:
        raise RuntimeError("downsample_events runs on the device: the event tensors must be CUDA tensors")
    fx, fy = int(input_width / output_width), int(input_height / output_height)
    if change_map is None:
        change_map = torch.zeros((output_height, output_width), dtype=torch.float32, device=dev)
    n = x.shape[0]
    if n == 0:
        return {k: v for k, v in events.items()}, change_map
    cell = (torch.div(y.long(), fy, rounding_mode="floor") * output_width + torch.div(x.long(), fx, rounding_mode="floor"))
    key, order = torch.sort(cell, stable=True)
    cells, run = torch.unique_consecutive(key, return_counts=True)
    keep = torch.empty((n,), dtype=torch.uint8, device=dev)
    P = _lib.ptr
    order32, cells32, ends32 = order.int().contiguous(), cells.int().contiguous(), run.cumsum(0).int().contiguous()
    pol = p.reshape(-1).to(torch.int8).contiguous()
    _lib.check(_lib.lib().dagr_downsample_events(P(order32), P(cells32), P(ends32), cells32.shape[0], P(pol), fx, fy,
                                                 P(change_map), P(keep), _lib.cur_stream(dev)), "downsample_events")
    mask = keep.bool()
    out = {k: v[mask] for k, v in events.items()}
    out["x"] = torch.div(out["x"], fx, rounding_mode="floor").to(torch.int32)
    out["y"] = torch.div(out["y"], fy, rounding_mode="floor").to(torch.int32)
    return out, change_map
