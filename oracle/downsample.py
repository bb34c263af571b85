"""CPU restatement of ``scripts/downsample_events.py:91-124`` (``downsample_events`` / ``_filter_events_resize``).

TEST INFRASTRUCTURE ONLY -- see oracle/__init__.py.  The numba-compiled loop of the reference written as the same plain
Python loop over a float32 ``change_map`` (numba is absent; the arithmetic -- a float64 product added into a float32
cell, ``abs(.) >= 1``, subtract the polarity -- is identical)."""
import numpy as np


def downsample_events(events, input_height, input_width, output_height, output_width, change_map=None):
    if change_map is None:
        change_map = np.zeros((output_height, output_width), dtype="float32")
    fx, fy = int(input_width / output_width), int(input_height / output_height)
    x, y, p = events["x"], events["y"], events["p"]
    mask = np.zeros(len(events["t"]), dtype=bool)
    for i in range(len(x)):                                    # :113-122
        x_l, y_l = x[i] // fx, y[i] // fy
        change_map[y_l, x_l] += p[i] * 1.0 / (fx * fy)
        if np.abs(change_map[y_l, x_l]) >= 1:
            mask[i] = True
            change_map[y_l, x_l] -= p[i]
    out = {k: v[mask] for k, v in events.items()}
    out["x"] = (out["x"] / fx).astype("uint16")
    out["y"] = (out["y"] / fy).astype("uint16")
    return out, change_map
