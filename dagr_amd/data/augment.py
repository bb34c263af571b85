"""The test-time transform of ``src/dagr/data/augment.py`` (``Augmentations.transform_testing`` :282-285 = ``Crop([0, 0],
[1, 1])`` :115-145): events outside the sensor are dropped, the frame is cut and the boxes clipped to it.  The training
augmentations (flip / zoom / translate / random crop, :90-280) belong to the training path, which is outside this stack."""
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


class Augmentations:
    transform_testing = Compose([Crop([0, 0], [1, 1])])

    def __init__(self, args):
        raise NotImplementedError("training augmentations are part of the training path (outside this stack)")
