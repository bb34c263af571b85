"""``ModelEMA`` with the reference's interface (``src/dagr/model/networks/ema.py:6-51``): evaluation loads the
checkpoint's ``ema`` state_dict into ``ModelEMA(model).ema`` (``scripts/run_test.py:54-58``).  ``update`` is the
standard exponential moving average over the floating-point state (training itself is out of scope)."""
import copy
import math

import torch


class ModelEMA:
    def __init__(self, model, decay=0.9999, updates=0):
        plan, model._engine = getattr(model, "_engine", None), None   # device-side plans are rebuilt, not copied
        self.ema = copy.deepcopy(model).eval().requires_grad_(False)
        model._engine = plan
        self.updates = updates
        self._decay = decay

    def decay(self, step):
        return self._decay * (1.0 - math.exp(-step / 2000.0))   # ramp: small decay during the first epochs

    @torch.no_grad()
    def update(self, model):
        """avg = d * avg + (1 - d) * value for every floating-point entry of the state_dict (ema.py:33-45), as two
        multi-tensor launches instead of two per entry (~600 for dagr-l: 2 ms of a 30 ms training step)."""
        self.updates += 1
        d = self.decay(self.updates)
        pairs = self.__dict__.get("_pairs")
        probe = next(model.parameters(), None)
        mine = next(self.ema.parameters(), None)
        stamp = (id(model), None if probe is None else probe.data_ptr(), None if mine is None else mine.data_ptr())
        if pairs is None or pairs[0] != stamp:
            source = model.state_dict()
            avgs, vals = [], []
            for name, avg in self.ema.state_dict().items():
                if avg.dtype.is_floating_point:
                    avgs.append(avg)
                    vals.append(source[name].detach())
            # parameters and buffers are updated in place by the optimizer / BatchNorm, so the tensor lists stay valid
            # while the model keeps its storage (the stamp catches .to() / .cuda(), which re-allocate it)
            pairs = self._pairs = (stamp, avgs, vals)
        _, avgs, vals = pairs
        if avgs:
            torch._foreach_mul_(avgs, d)
            torch._foreach_add_(avgs, vals, alpha=1.0 - d)
        self.ema._engine = None
