"""``NCaltech101`` (``src/dagr/data/ncaltech101_data.py:14-82``): per-sample event files ``<root>/<split>/<class>/*.h5``
(group ``events`` with ``x``, ``y``, ``t``, ``p``; the LAST ``num_events`` are used, :75-82), one box per sample from
``<root>/annotations/<class>/annotation_*.bin`` (int16 words 2..9 = the four corners, :57-72), 240 x 180 sensor, class id
= index of the class directory in sorted order.

Reading needs h5py (+ hdf5plugin for the blosc filter), neither of which is part of this image: they are imported when
the first file is opened, and the reader can be replaced (``reader=``: path -> dict of arrays), which is also how
the tests exercise the class on ``.npz`` stand-ins."""
from pathlib import Path

import numpy as np
import torch

from .augment import init_transforms
from .utils import to_data


def load_events(path, num_events, reader=None):
    """The last ``num_events`` events of a file as numpy arrays (ncaltech101_data.py:75-82)."""
    if reader is not None:
        ev = reader(path)
        return {k: np.asarray(ev[k])[-num_events:] for k in ("x", "y", "t", "p")}
    try:
        import hdf5plugin  # noqa: F401
    except ImportError:
        pass
    try:
        import h5py
    except ImportError as e:
        raise RuntimeError("reading N-Caltech101 .h5 event files needs h5py (and hdf5plugin): not installed here; "
                           "pass reader= to NCaltech101 to plug another file format") from e
    with h5py.File(str(path)) as fh:
        ev = fh["events"]
        return {k: ev[k][-num_events:] for k in ("x", "y", "t", "p")}


def read_annotation(path, class_id):
    """[[x, y, w, h, class, 1]] float32 from the int16 annotation file (ncaltech101_data.py:57-72)."""
    words = np.fromfile(str(path), dtype=np.int16)[2:10]
    return np.array([words[0], words[1], words[2] - words[0], words[5] - words[1], class_id, 1],
                    dtype="float32").reshape((1, -1))


class NCaltech101(torch.utils.data.Dataset):
    def __init__(self, root, split, transform=None, num_events=50000, reader=None, suffix=".h5"):
        super().__init__()
        self.load_dir = Path(root) / split
        self.classes = sorted(d.name for d in self.load_dir.glob("*"))
        self.num_classes = len(self.classes)
        self.files = sorted(self.load_dir.rglob("*" + suffix))
        self.height, self.width = 180, 240
        if transform is not None and hasattr(transform, "transforms"):
            init_transforms(transform.transforms, self.height, self.width)
        self.transform = transform
        self.time_window = 1000000
        self.num_events = num_events
        self.reader, self.suffix = reader, suffix

    def __len__(self):
        return len(self.files)

    def load_bboxes(self, raw_file, class_id):
        rel = str(Path(raw_file).relative_to(self.load_dir)).replace("image_", "annotation_").replace(self.suffix, ".bin")
        return read_annotation(self.load_dir / "../annotations" / rel, class_id)

    def __getitem__(self, idx):
        path = self.files[idx]
        target = self.classes.index(str(path.parent.name))
        ev = load_events(path, self.num_events, self.reader)
        data = to_data(**ev, bbox=self.load_bboxes(path, target), t0=ev["t"][0], t1=ev["t"], width=self.width,
                       height=self.height, time_window=self.time_window)
        data.t = data.t - (data.t[-1] - self.time_window + 1)       # preprocess (:36-38): newest event at T - 1
        data = self.transform(data) if self.transform is not None else data
        if not hasattr(data, "t") or data.t is None:
            data.t = data.pos[:, -1:]
            data.pos = data.pos[:, :2].type(torch.int16)
        return data
