"""Mirror of ``format_data`` (``src/dagr/utils/buffers.py:33-44``): the input contract of the hot path.
On device tensors the normalisation runs through ``dagr_format_events`` (libdagr_hip) when the batch
carries the dataset's raw dtypes (pos int16[N,2], t int32[N], x int8[N,1]); any other dtype takes the
same arithmetic through torch ops (true division in fp32)."""
import ctypes

import torch

from .. import _lib


def format_data(data, normalizer=None):
    W, H, T = int(data.width[0]), int(data.height[0]), int(data.time_window[0])
    if hasattr(data, "image"):
        data.image = data.image.float() / 255.0
    pos, t, x = data.pos, data.t, data.x
    if (normalizer is None and pos.is_cuda and pos.dtype == torch.int16 and t.dtype == torch.int32
            and x.dtype == torch.int8 and pos.is_contiguous() and t.is_contiguous() and x.is_contiguous()):
        N = pos.shape[0]
        pos_out = torch.empty((N, 3), dtype=torch.float32, device=pos.device)
        feat = torch.empty((N, 1), dtype=torch.float32, device=pos.device)
        _lib.check(_lib.lib().dagr_format_events(_lib.ptr(pos), _lib.ptr(t), _lib.ptr(x), N, W, H, T,
                                                 _lib.ptr(pos_out), _lib.ptr(feat), _lib.cur_stream(pos.device)),
                   "format_events")
        data.pos, data.x = pos_out, feat
    else:
        if normalizer is None:
            normalizer = torch.tensor([W, H, T], device=pos.device)
        data.pos = torch.cat([pos, t.view((-1, 1))], dim=-1) / normalizer
        data.x = x.float()
    data.t = None
    return data
