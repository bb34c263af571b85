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
        self.updates += 1
        d = self.decay(self.updates)
        source = model.state_dict()
        for name, avg in self.ema.state_dict().items():
            if avg.dtype.is_floating_point:
                avg.mul_(d).add_(source[name].detach(), alpha=1.0 - d)
        self.ema._engine = None
