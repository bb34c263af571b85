"""A dataset with the interface of the reference's ``DSEC`` (``src/dagr/data/dsec_data.py:58-184``: ``height``, ``width``,
``classes``, ``time_window``, ``set_num_us``, ``__getitem__`` -> per-sample ``Data``) over the benchmark's synthetic
event streams (``dagr_amd/utils/synthetic.py``): what the test scripts run on where the DSEC files (h5 + blosc) cannot
be read."""
import numpy as np
import torch

from ..utils import synthetic as syn
from .utils import to_data


class SyntheticWindows:
    classes = ["car", "pedestrian"]         # dsec_data.py:78 (two remapped classes)

    def __init__(self, n_windows, n_events, width, height, stream="uniform", use_image=False, seed=1234, transform=None,
                 windows_per_sequence=100):
        self.n, self.n_events, self.width, self.height = int(n_windows), int(n_events), int(width), int(height)
        self.gen = syn.uniform_window if stream == "uniform" else syn.edges_window
        self.use_image, self.seed, self.transform = bool(use_image), int(seed), transform
        self.time_window = 1000000           # dsec_data.py:89
        self.num_us = -1                     # dsec_data.py:91: -1 = the whole 50 ms between two frames
        self.per_seq = int(windows_per_sequence)

    def set_num_us(self, num_us):            # dsec_data.py:114-115
        self.num_us = int(num_us)

    def sequence_names(self):
        """Names of the synthetic recordings (``windows_per_sequence`` consecutive windows each)."""
        return [f"synthetic{q:03d}" for q in range((self.n + self.per_seq - 1) // self.per_seq)]

    def __len__(self):
        return self.n

    def __getitem__(self, w):
        x, y, t, p = self.gen(self.n_events, self.width, self.height, seed=self.seed + int(w))
        t0 = 50000 * int(w)                  # frame timestamp (us) of window w; frames are 50 ms apart
        t1 = t0 + 50000
        if self.num_us >= 0:
            # dsec_data.py:159-161: only the events of the first num_us microseconds after the frame; then
            # preprocess_events (:141-147) shifts the kept events so that the last one sits at time_window
            rel = t - (self.time_window - 50000)             # event time relative to the frame, 0 .. 50000
            keep = rel < self.num_us
            x, y, t, p = x[keep], y[keep], t[keep], p[keep]
            if len(t):
                t = (self.time_window + t - t[-1]).astype(np.int32)
            t1 = t0 + self.num_us
        d = to_data(x=x, y=y, t=t, p=p, t0=t0, t1=t1, width=self.width, height=self.height,
                    time_window=self.time_window, sequence=f"synthetic{int(w) // self.per_seq:03d}", window=int(w))
        if self.use_image:
            d.image = torch.randint(0, 256, (1, 3, self.height, self.width), dtype=torch.uint8,
                                    generator=torch.Generator().manual_seed(self.seed + int(w)))
        return self.transform(d) if self.transform is not None else d


class SyntheticObjects:
    """Labelled synthetic samples with the interface of the reference's ``NCaltech101`` (``height``, ``width``,
    ``classes``, ``__getitem__`` -> ``Data`` with one ``bbox`` row (x, y, w, h, class, 1)): a rectangle whose outline
    (class 0) or diagonals + outline (class 1) fires events with pixel jitter over the last 50 ms, plus uniform noise.
    What the training script runs on where the N-Caltech101 files (h5) cannot be read: enough structure for the loss
    to fall, none of the real data's statistics."""
    classes = ["outline", "cross"]

    def __init__(self, n_samples, n_events=20000, width=240, height=180, seed=7, transform=None, noise=0.2,
                 use_image=False):
        self.n, self.n_events, self.width, self.height = int(n_samples), int(n_events), int(width), int(height)
        self.seed, self.transform, self.noise = int(seed), transform, float(noise)
        self.use_image = bool(use_image)       # DSEC-style samples: a frame with the object drawn in + bbox0
        self.num_classes = len(self.classes)
        self.time_window = 1000000
        self.num_us = -1
        if transform is not None and hasattr(transform, "transforms"):
            from .augment import init_transforms
            init_transforms(transform.transforms, self.height, self.width)

    def __len__(self):
        return self.n

    def set_num_us(self, num_us):            # dsec_data.py:114-115 (interframe evaluation)
        self.num_us = int(num_us)

    def sequence_names(self):
        return [f"objects{int(i):05d}" for i in range(self.n)]

    def __getitem__(self, i):
        rng = np.random.default_rng(self.seed + int(i))
        W, H, N = self.width, self.height, self.n_events
        cls = int(rng.integers(0, 2))
        w, h = rng.uniform(0.25, 0.5) * W, rng.uniform(0.25, 0.5) * H
        x0, y0 = rng.uniform(2, W - w - 2), rng.uniform(2, H - h - 2)
        n_obj = int(N * (1 - self.noise))
        u = rng.uniform(0, 1, n_obj)
        side = rng.integers(0, 4 if cls == 0 else 6, n_obj)
        px = np.where(side == 0, x0 + u * w, np.where(side == 1, x0 + u * w, np.where(side == 2, x0, np.where(
            side == 3, x0 + w, x0 + u * w))))
        py = np.where(side == 0, y0, np.where(side == 1, y0 + h, np.where(side == 2, y0 + u * h, np.where(
            side == 3, y0 + u * h, np.where(side == 4, y0 + u * h, y0 + (1 - u) * h)))))
        px = np.concatenate([px + rng.normal(0, 1.0, n_obj), rng.uniform(0, W, N - n_obj)])
        py = np.concatenate([py + rng.normal(0, 1.0, n_obj), rng.uniform(0, H, N - n_obj)])
        ok = (px >= 0) & (px < W) & (py >= 0) & (py < H)
        px, py = px[ok].astype(np.int16), py[ok].astype(np.int16)
        t = np.sort(rng.integers(0, 50000, len(px))).astype(np.int64)
        t = (self.time_window - 1 + t - t[-1]).astype(np.int32)
        p = rng.choice(np.array([-1, 1], dtype=np.int8), len(px))
        order = rng.permutation(len(px))
        px, py = px[order], py[order]                     # positions are not correlated with time
        if self.num_us >= 0:                              # dsec_data.py:159-161: the first num_us microseconds only
            keep = (t - (self.time_window - 50000)) < self.num_us
            px, py, t, p = px[keep], py[keep], t[keep], p[keep]
            if len(t):
                t = (self.time_window - 1 + t - t[-1]).astype(np.int32)
            else:                                         # keep one event so that the sample stays a graph
                px, py, t, p = (np.array([v], dtype=a.dtype) for v, a in ((x0, px), (y0, py), (self.time_window - 1, t), (1, p)))
        bbox = np.array([[x0, y0, w, h, cls, 1]], dtype=np.float32)
        extra = {}
        if self.use_image:
            frame = rng.integers(0, 40, (3, H, W)).astype(np.uint8)
            frame[cls, int(y0):int(y0 + h), int(x0):int(x0 + w)] += 150       # the object, brighter in its class channel
            extra = dict(bbox0=bbox.copy(), image=torch.from_numpy(frame)[None])
        d = to_data(x=px, y=py, t=t, p=p, bbox=bbox, t0=int(t[0]), t1=int(t[-1]), width=W, height=H,
                    time_window=self.time_window, sequence=f"objects{int(i):05d}", **extra)
        return self.transform(d) if self.transform is not None else d
