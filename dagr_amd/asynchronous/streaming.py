"""The running event window behind ``DAGR.forward(x, reset=False)``."""
import torch


class StreamingWindow:
    """Events of the current window (since the last ``reset=True`` call), resident on the device in arrival order.
    ``push`` appends a micro-batch (format_data'd: pos fp32[n,3] normalised, x fp32[n,1], batch); ``tensors`` returns the
    whole window ordered by sample -- the layout a single ``reset=True`` call on the same events would have produced
    (PyG collation concatenates sample after sample), so the engine output is bit-identical to that call."""

    def __init__(self):
        self.pos = self.x = self.batch = None
        self.image = None
        self._seed = None

    def reset(self):
        self.pos = self.x = self.batch = self.image = None
        self._seed = None

    def __len__(self):
        if self._seed is not None:
            return int(self._seed.pos.shape[0])
        return 0 if self.pos is None else int(self.pos.shape[0])

    def seed(self, data):
        """The batch of a plain ``reset=True`` call: only remembered -- the conversions of ``push`` are paid by the first
        ``reset=False`` call that continues from it, not by every window of an evaluation run."""
        self.reset()
        self._seed = data

    def push(self, data):
        if self._seed is not None:
            first, self._seed = self._seed, None
            self.push(first)
        batch = data.batch if getattr(data, "batch", None) is not None else \
            torch.zeros(data.pos.shape[0], dtype=torch.int64, device=data.pos.device)
        pos, x, batch = data.pos.float(), data.x.float().view(-1, 1), batch.long()
        if self.pos is None:
            self.pos, self.x, self.batch = pos, x, batch
        else:
            self.pos = torch.cat([self.pos, pos])
            self.x = torch.cat([self.x, x])
            self.batch = torch.cat([self.batch, batch])
        if getattr(data, "image", None) is not None:
            self.image = data.image          # the frame the window follows (dsec_data.py:154-155)

    def tensors(self):
        # stable by sample: events keep their arrival (= time) order inside a sample
        order = torch.argsort(self.batch, stable=True)
        return self.pos[order].contiguous(), self.x[order].contiguous(), self.batch[order].contiguous()
