"""Mirror of ``src/dagr/model/networks/ema.py:6-51`` (``ModelEMA``): evaluation loads the checkpoint's
``ema`` state_dict into ``ModelEMA(model).ema`` (``scripts/run_test.py:54-58``).  ``update`` is kept
for completeness; training itself is out of scope."""
import math
from copy import deepcopy

import torch


class ModelEMA:
    def __init__(self, model, decay=0.9999, updates=0):
        engine, model._engine = getattr(model, "_engine", None), None  # device plans are not deep-copied
        self.ema = deepcopy(model).eval()
        model._engine = engine
        self.updates = updates
        self.decay = lambda x: decay * (1 - math.exp(-x / 2000))
        for p in self.ema.parameters():
            p.requires_grad_(False)

    def update(self, model):
        with torch.no_grad():
            self.updates += 1
            d = self.decay(self.updates)
            msd = model.state_dict()
            for k, v in self.ema.state_dict().items():
                if v.dtype.is_floating_point:
                    v *= d
                    v += (1.0 - d) * msd[k].detach()
        self.ema._engine = None
