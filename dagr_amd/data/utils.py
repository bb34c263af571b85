"""``to_data`` (``src/dagr/data/utils.py:6-19``): raw event arrays -> the per-sample ``Data`` the loaders emit
(``pos int16[N,2]``, ``t int32[N]``, ``x = p[N,1]``; every ``bbox*`` array becomes a tensor)."""
import numpy as np
import torch

from . import Data


def to_data(**fields):
    xy = np.stack([fields.pop("x"), fields.pop("y")], axis=-1).astype("int16")
    out = {k: (torch.from_numpy(v) if k.startswith("bbox") else v) for k, v in fields.items()}
    out["x"] = torch.from_numpy(np.ascontiguousarray(out.pop("p")).reshape((-1, 1)))
    out["pos"] = torch.from_numpy(xy)
    out["t"] = torch.from_numpy(np.asarray(out["t"]).astype("int32"))
    return Data(**out)
