"""In-memory stand-in for ``dsec_det.dataset.DSECDet`` (third-party, absent): two short synthetic "recordings" with the
interface the reference's ``DSEC`` uses -- ``directories[name].{tracks.tracks, images.timestamps, events.root, root}``,
``subsequence_directories``, ``width``, ``classes``, ``get_tracks`` / ``get_image`` / ``get_events``.  Built from a seed, so
the golden generator (which feeds it to the REFERENCE's DSEC class) and the tests (which feed it to this repository's)
see the same recordings.  TEST INFRASTRUCTURE."""
import types
from pathlib import Path

import numpy as np

TRACK_DTYPE = np.dtype([("t", "<u8"), ("x", "<f8"), ("y", "<f8"), ("h", "<f8"), ("w", "<f8"), ("class_id", "u1"),
                        ("class_confidence", "<f4"), ("track_id", "<u4")])
CLASSES = ("pedestrian", "rider", "car", "bus", "truck", "bicycle", "motorcycle", "train")   # dsec-det's class list


def _recording(name, seed, n_frames, events_per_frame, width=640, height=480):
    rng = np.random.default_rng(seed)
    stamps = (50_000_000 + seed * 1000 + 50_000 * np.arange(n_frames)).astype(np.int64)
    rows = []
    n_tracks = 5
    base = np.stack([rng.uniform(0, width - 150, n_tracks), rng.uniform(0, height - 150, n_tracks),
                     rng.uniform(8, 140, n_tracks), rng.uniform(8, 160, n_tracks)], 1)
    vel = rng.uniform(-6, 6, (n_tracks, 2))
    cls = rng.integers(0, len(CLASSES), n_tracks)
    for f, t in enumerate(stamps):
        for k in range(n_tracks):
            if (f + k + seed) % 7 == 0:
                continue                                   # this track is not labelled in this frame
            x, y = base[k, 0] + vel[k, 0] * f, base[k, 1] + vel[k, 1] * f
            rows.append((t, x, y, base[k, 3] * (1 + 0.01 * f), base[k, 2] * (1 + 0.01 * f), cls[k], 1.0, k))
    tracks = np.array(rows, dtype=TRACK_DTYPE)
    events = []
    for f in range(n_frames):
        n = events_per_frame + 37 * f
        t = np.sort(rng.integers(stamps[f], stamps[f] + 50_000, n)).astype(np.int64)
        events.append(dict(x=rng.integers(0, width // 2, n).astype(np.uint16),
                           y=rng.integers(0, height // 2, n).astype(np.uint16), t=t,
                           p=rng.integers(0, 2, n).astype(np.uint8)))
    images = [rng.integers(0, 256, (height, width, 3)).astype(np.uint8) for _ in range(n_frames)]
    root = Path("/dsec") / name
    return types.SimpleNamespace(root=root, name=name, tracks=types.SimpleNamespace(tracks=tracks),
                                 images=types.SimpleNamespace(timestamps=stamps, frames=images),
                                 events=types.SimpleNamespace(root=root / "events", windows=events))


class FakeDSECDet:
    def __init__(self, root=None, split=None, sync="back", debug=False, split_config=None):
        recs = [_recording("zurich_city_12_a", 1, 9, 900), _recording("thun_01_a", 2, 7, 600)]
        self.directories = {r.name: r for r in recs}
        self.subsequence_directories = [r.root for r in recs]
        self.width, self.height = 640, 480
        self.classes = CLASSES

    def get_tracks(self, index, mask=None, directory_name=None):
        d = self.directories[directory_name]
        sel = d.tracks.tracks["t"] == d.images.timestamps[index]
        if mask is not None:
            sel = sel & mask
        return d.tracks.tracks[sel]

    def get_image(self, index, directory_name=None):
        return self.directories[directory_name].images.frames[index]

    def get_events(self, index, directory_name=None):
        # the windows live on the directory object the reference may have replaced (EventDirectory): keep our own table
        return {k: v.copy() for k, v in self._windows[directory_name][index].items()}

    @property
    def _windows(self):
        if not hasattr(self, "_w"):
            self._w = {n: _recording(n, s, f, e).events.windows
                       for n, s, f, e in (("zurich_city_12_a", 1, 9, 900), ("thun_01_a", 2, 7, 600))}
        return self._w


def nearest_resize_hwc(image, size, interpolation=None):
    """Stand-in for ``cv2.resize(image, (width, height), ...)`` (frame content is third-party arithmetic): nearest rows / columns."""
    w, h = size
    ys = (np.arange(h) * image.shape[0] / h).astype(int)
    xs = (np.arange(w) * image.shape[1] / w).astype(int)
    return image[ys][:, xs]
