"""Golden for the subsampling branch of the reference's RandomZoom (``src/dagr/data/augment.py:146-198`` with
``subsample=True`` and ``zoom < 1``), which the shipped configs never take (``zoom`` in [1, aug_zoom]): the REFERENCE code
runs here on CPU over the stand-ins of tests/make_golden_refpy_data.py and its outputs are stored in
tests/golden/ref_py_zoom_subsample.npz; plus RandomCrop on a sample with a frame (the reference's ``_crop_image``
indexing).  Run where /root/reference exists: python tests/make_golden_refpy_zoom.py"""
import os
import sys

import numpy as np
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))
import make_golden_refpy_data as gen  # noqa: E402
import refpy_fakes  # noqa: E402


def base_sample(seed=5, n=4000):
    rng = np.random.default_rng(seed)
    # events clustered on a few pixels so that the integrate-and-fire accumulators do cross their threshold
    hot = rng.random(n) < 0.6
    x = np.where(hot, rng.integers(100, 116, n), rng.integers(0, 240, n))
    y = np.where(hot, rng.integers(80, 92, n), rng.integers(0, 180, n))
    return dict(x=x, y=y, t=np.sort(rng.integers(0, 50000, n)), p=rng.choice(np.array([-1, 1], dtype=np.int8), n, p=[0.3, 0.7]),
                bbox=np.array([[50., 40, 60, 50, 3, 1], [120., 30, 90, 120, 1, 1]], dtype=np.float32))


def main():
    gen.install_stubs()
    refpy_fakes.use_reference_package("/root/reference/src")
    raug = gen.import_ref("dagr.data.augment")
    from dagr_amd.data.utils import to_data
    base = base_sample()
    out = {f"base_{k}": v for k, v in base.items()}
    for seed in range(4):
        zoom = raug.RandomZoom(zoom=[0.5, 0.9], subsample=True)
        zoom.init(180, 240)
        d = to_data(**{k: v.copy() for k, v in base.items()}, width=240, height=180, time_window=1000000)
        d = refpy_fakes.Data(**d.__dict__)
        torch.manual_seed(seed)
        o = zoom(d)
        for k in ("pos", "x", "t", "bbox"):
            out[f"zoom{seed}_{k}"] = getattr(o, k).numpy()
    # ---- RandomCrop on a sample that carries a frame: _crop_image (augment.py:51-58) indexes the first two dimensions of
    # the [1, 3, H, W] tensor -- pinned as the reference behaves, not as the name suggests
    rng = np.random.default_rng(11)
    frame = rng.integers(1, 255, (1, 3, 180, 240), dtype=np.uint8)
    out["crop_frame"] = frame
    for seed in range(6):
        crop = raug.RandomCrop([0.75, 0.75], p=1.0)
        crop.init(180, 240)
        d = to_data(**{k: v.copy() for k, v in base.items()}, width=240, height=180, time_window=1000000)
        d = refpy_fakes.Data(**d.__dict__)
        d.image = torch.from_numpy(frame.copy())
        torch.manual_seed(seed)
        o = crop(d)
        for k in ("pos", "x", "t", "bbox"):
            out[f"crop{seed}_{k}"] = getattr(o, k).numpy()
        out[f"crop{seed}_channel_sums"] = o.image.numpy().astype(np.int64).sum(axis=(0, 2, 3))
        out[f"crop{seed}_grid"] = o.image.numpy()[..., ::12, ::12]
    np.savez_compressed(os.path.join(ROOT, "tests", "golden", "ref_py_zoom_subsample.npz"), **out)
    print({k: v.shape for k, v in out.items() if k.startswith("zoom")})
    print({k: v.tolist() for k, v in out.items() if k.endswith("channel_sums")})


if __name__ == "__main__":
    main()
