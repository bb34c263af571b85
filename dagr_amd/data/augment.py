"""The event / frame / box transforms of ``src/dagr/data/augment.py``: the test-time ``Crop([0, 0], [1, 1])`` (:115-145,
``Augmentations.transform_testing`` :282-285) and the training chain of ``Augmentations.__init__`` (:287-294): horizontal
flip (:90-112), random crop with probability 0.2 (:201-243), random zoom (:148-198), random translation (:246-279), then
the crop to the sensor.  All of them act on the per-sample ``Data`` BEFORE the graph exists (host side, integer pixel
coordinates), exactly where the reference applies them (the loaders' ``transform``)."""
import numpy as np
import torch


class Compose:
    def __init__(self, transforms):
        self.transforms = list(transforms)

    def __call__(self, data):
        for t in self.transforms:
            data = t(data)
        return data


class Crop:
    """Keep what lies inside [min, max) (fractions of the sensor size until ``init`` turns them into pixels)."""

    def __init__(self, min, max):
        self.min, self.max = torch.as_tensor(min), torch.as_tensor(max)
        self._pixels = False

    def init(self, height, width):
        size = torch.tensor([width, height], dtype=torch.float32)
        self.min = (self.min.float() * size).clamp(min=0).int()
        self.max = torch.minimum((self.max.float() * size), size).int()
        self._pixels = True

    def __call__(self, data):
        if not self._pixels:
            self.init(int(data.height), int(data.width))
        x0, y0 = int(self.min[0]), int(self.min[1])
        x1, y1 = int(self.max[0]), int(self.max[1])
        keep = (data.pos[:, 0] >= x0) & (data.pos[:, 0] < x1) & (data.pos[:, 1] >= y0) & (data.pos[:, 1] < y1)
        if not bool(keep.all()):
            for name in ("pos", "x", "t"):
                v = getattr(data, name, None)
                if torch.is_tensor(v) and v.shape[0] == keep.shape[0]:
                    setattr(data, name, v[keep])
        if (x0, y0) != (0, 0):
            data.pos = data.pos - torch.tensor([x0, y0], dtype=data.pos.dtype)
        if hasattr(data, "image"):
            data.image = data.image[..., y0:y1, x0:x1]
        for name in ("bbox", "bbox0"):
            b = getattr(data, name, None)
            if b is not None and len(b):
                b = b.clone()
                bx1 = (b[:, 0] + b[:, 2]).clamp(x0, x1 - 1)
                by1 = (b[:, 1] + b[:, 3]).clamp(y0, y1 - 1)
                b[:, 0] = b[:, 0].clamp(x0, x1 - 1)
                b[:, 1] = b[:, 1].clamp(y0, y1 - 1)
                b[:, 2], b[:, 3] = bx1 - b[:, 0], by1 - b[:, 1]
                b[:, 0] -= x0
                b[:, 1] -= y0
                setattr(data, name, b)
        return data


def init_transforms(transforms, height, width):
    for t in transforms:
        if hasattr(t, "init"):
            t.init(height=height, width=width)


def _each_box_field(data):
    for name in ("bbox", "bbox0"):
        b = getattr(data, name, None)
        if b is not None and len(b):
            yield name, b


class RandomHFlip:
    """Mirror the sample about the vertical axis with probability p (augment.py:90-112)."""

    def __init__(self, p):
        self.p = float(p)

    def __call__(self, data):
        if float(torch.rand(1)) > self.p:
            return data
        W = int(data.width)
        data.pos = data.pos.clone()
        data.pos[:, 0] = W - 1 - data.pos[:, 0]
        if hasattr(data, "image"):
            data.image = torch.flip(data.image, dims=[-1]).contiguous()
        for name, b in _each_box_field(data):
            b = b.clone()
            b[:, 0] = W - 1 - (b[:, 0] + b[:, 2])
            setattr(data, name, b)
        return data


REFERENCE_FRAME_CROP = True      # see _keep_window


def _keep_window(data, lo, hi):
    """Events inside [lo, hi] (inclusive, as augment.py:39-51 keeps them), frame cropped as the reference does it (below),
    boxes clamped."""
    keep = ((data.pos >= lo) & (data.pos <= hi)).all(dim=1)
    for name in ("pos", "x", "t"):
        v = getattr(data, name, None)
        if torch.is_tensor(v) and v.shape[0] == keep.shape[0]:
            setattr(data, name, v[keep])
    x0, y0, x1, y1 = int(lo[0]), int(lo[1]), int(hi[0]), int(hi[1])
    if hasattr(data, "image"):
        img = data.image.clone()
        if REFERENCE_FRAME_CROP:
            # augment.py:51-58 as written: the four assignments index the FIRST TWO dimensions of the [1, 3, H, W] frame
            # (batch, channel), not rows and columns -- any window that does not start in row 0 blanks the whole frame,
            # one that starts in column c > 0 of row 0 blanks the first min(c, 3) channels.  Kept: a model trained here
            # sees the frames the reference's training loop shows it.  REFERENCE_FRAME_CROP = False blanks outside the
            # window, which is what the function's name promises.
            img[:y0, :] = 0
            img[y1:, :] = 0
            img[:, :x0] = 0
            img[:, x1:] = 0
        else:
            img[..., :y0, :] = 0
            img[..., y1:, :] = 0
            img[..., :, :x0] = 0
            img[..., :, x1:] = 0
        data.image = img
    for name, b in _each_box_field(data):
        b = b.clone()
        far = b[:, :2] + b[:, 2:4]
        near = torch.minimum(torch.maximum(b[:, :2], lo.to(b.dtype)), hi.to(b.dtype))
        far = torch.minimum(torch.maximum(far, lo.to(b.dtype)), hi.to(b.dtype))
        b[:, :2], b[:, 2:4] = near, far - near
        setattr(data, name, b)
    return data


class RandomCrop:
    """With probability p keep a random window of ``size`` (fractions of the sensor) and drop the rest (augment.py:201-243);
    coordinates are NOT shifted: the window stays where it was."""

    def __init__(self, size=(0.75, 0.75), dim=(0, 1), p=0.5):
        self.size, self.dim, self.p = torch.as_tensor(size, dtype=torch.float32), list(dim), float(p)
        self.left_max = None

    def init(self, height, width):
        full = torch.tensor([width, height], dtype=torch.float32)
        self.size = torch.minimum((self.size * full), full - 1).clamp(min=0).int()
        self.left_max = full.int() - self.size

    def __call__(self, data):
        if self.left_max is None:
            self.init(int(data.height), int(data.width))
        if float(torch.rand(1)) > self.p:
            return data
        left = (torch.rand(len(self.dim)) * self.left_max).to(torch.int16)
        return _keep_window(data, left, left + self.size.to(torch.int16))


def subsample_events(pos, polarity, zoom):
    """Events of a shrunk sample (zoom < 1): each event spreads p * bilinear weights over the 4 surrounding pixels of an
    accumulator; a pixel emits an event whenever its accumulated magnitude exceeds 1 / zoom^2 (augment.py:13-36, a
    sequential integrate-and-fire: inherently ordered, a plain loop here where the reference uses numba)."""
    thr = 1.0 / float(zoom) ** 2
    H = int(pos[:, 1].max()) + 2 if len(pos) else 1
    W = int(pos[:, 0].max()) + 2 if len(pos) else 1
    acc = np.zeros((H, W), dtype=np.float32)
    keep = np.zeros(len(pos), dtype=bool)
    out = np.array(pos, dtype=np.float32, copy=True)
    for i in range(len(pos)):
        x, y = float(pos[i, 0]), float(pos[i, 1])
        for xl, yl in ((int(x), int(y)), (int(x + 1), int(y)), (int(x), int(y + 1)), (int(x + 1), int(y + 1))):
            acc[yl, xl] += np.float32(float(polarity[i]) * (1 - abs(x - xl)) * (1 - abs(y - yl)))
            sign = 1 if acc[yl, xl] > 0 else -1
            if sign * acc[yl, xl] > thr:
                acc[yl, xl] -= sign * thr
                keep[i] = True
                out[i, 0], out[i, 1] = xl, yl
    return out, keep


class RandomZoom:
    """Scale the sample about the sensor centre by a factor drawn from ``zoom`` = [lo, hi] (augment.py:148-198): events
    move to ((p - c) * z + c) truncated to int16 (events pushed out of the sensor are removed by the final Crop), boxes
    scale with them; the frame is resized (nearest) and centre-cropped / centre-padded to the sensor."""

    def __init__(self, zoom, subsample=False):
        self.zoom, self.subsample = list(zoom), bool(subsample)

    def init(self, height, width):
        pass

    def __call__(self, data):
        # the factor is a float32 tensor expression in the reference (augment.py:174): same roundings here
        z = float(torch.rand(1) * (self.zoom[1] - self.zoom[0]) + self.zoom[0])
        W, H = int(data.width), int(data.height)
        cx, cy = W // 2, H // 2
        pos = data.pos.float()
        zx = ((pos[:, 0] - cx) * z + cx)
        zy = ((pos[:, 1] - cy) * z + cy)
        if self.subsample and z < 1:
            # augment.py:178-182: the zoomed coordinates are int16 before the integrate-and-fire pass sees them (the
            # bilinear weights collapse onto the truncated pixel; the other three pixels still get their threshold check)
            p, keep = subsample_events(torch.stack([zx, zy], 1).to(torch.int16).numpy().astype(np.float32),
                                       data.x.reshape(-1).numpy(), z)
            data.pos = torch.from_numpy(p[keep].astype("int16"))
            data.x = data.x[torch.from_numpy(keep)]
            if hasattr(data, "t") and torch.is_tensor(data.t):
                data.t = data.t[torch.from_numpy(keep)]
        else:
            data.pos = torch.stack([zx, zy], 1).to(torch.int16)
        if hasattr(data, "image"):
            w2, h2 = int(np.ceil(W * z)), int(np.ceil(H * z))
            img = torch.nn.functional.interpolate(data.image.float(), size=(h2, w2), mode="nearest").to(data.image.dtype)
            px, py = (w2 - W) // 2, (h2 - H) // 2
            if px >= 0:
                data.image = img[..., py:py + H, px:px + W].contiguous()
            else:
                bg = torch.zeros_like(data.image)
                bg[..., -py:-py + h2, -px:-px + w2] = img
                data.image = bg
        for name, b in _each_box_field(data):
            b = b.clone()
            b[:, 2:4] *= z
            b[:, 0] = (b[:, 0] - cx) * z + cx
            b[:, 1] = (b[:, 1] - cy) * z + cy
            setattr(data, name, b)
        return data


class RandomTranslate:
    """Shift the sample by a random whole number of pixels within +-size (fractions of the sensor, augment.py:246-279);
    what leaves the sensor is removed by the final Crop, the frame is padded with black."""

    def __init__(self, size):
        self.size = torch.as_tensor(size, dtype=torch.float32)[:2]
        self.px = None

    def init(self, height, width):
        full = torch.tensor([width, height], dtype=torch.float32)
        self.px = torch.minimum(self.size * full, full - 1).clamp(min=0).int()

    def __call__(self, data):
        if self.px is None:
            self.init(int(data.height), int(data.width))
        move = (self.px * (torch.rand(2) * 2 - 1)).to(torch.int16)
        data.pos = data.pos + move
        if hasattr(data, "image"):
            H, W = data.image.shape[-2:]
            mx, my = int(move[0]), int(move[1])
            out = torch.zeros_like(data.image)
            xs0, xs1 = max(0, -mx), min(W, W - mx)
            ys0, ys1 = max(0, -my), min(H, H - my)
            if xs1 > xs0 and ys1 > ys0:
                out[..., ys0 + my:ys1 + my, xs0 + mx:xs1 + mx] = data.image[..., ys0:ys1, xs0:xs1]
            data.image = out
        for name, b in _each_box_field(data):
            b = b.clone()
            b[:, :2] += move.to(b.dtype)
            setattr(data, name, b)
        return data


class Augmentations:
    transform_testing = Compose([Crop([0, 0], [1, 1])])

    def __init__(self, args):
        """augment.py:287-294: the training chain, parameterised by ``aug_p_flip`` / ``aug_zoom`` / ``aug_trans``."""
        self.transform_training = Compose([
            RandomHFlip(p=args.aug_p_flip),
            RandomCrop([0.75, 0.75], p=0.2),
            RandomZoom(zoom=[1, args.aug_zoom], subsample=True),
            RandomTranslate([args.aug_trans, args.aug_trans, 0]),
            Crop([0, 0], [1, 1]),
        ])
