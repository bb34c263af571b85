"""``DSEC`` (``src/dagr/data/dsec_data.py:58-205``): one sample = the events between two consecutive labelled frames of a
sequence (half-resolution ``left/events_2x.h5``), the earlier frame, and the boxes of both frames.

What this class owns is the reference's own logic: which frame pairs qualify (``dsec_utils.filter_tracks``), the class
remap to (car, pedestrian), rescaling / clipping of the boxes, the event crop to the kept rows, the shift of the
timestamps so that the newest event sits at ``time_window``, polarity {0, 1} -> {-1, +1}, the interframe cut
(``set_num_us``: only the first ``num_us`` microseconds after the earlier frame, boxes linearly interpolated to that
instant, :159-165), the sample ``Data`` (``to_data``), the transform and the small-box filter.

What it does NOT own is the file access: the reference delegates it to the third-party ``dsec_det.dataset.DSECDet``
(directory layout, image decoding, blosc-HDF5 event reads).  ``source`` is that object: by default a
``DSECDet(root, split, sync="back", split_config=...)`` built here when ``dsec_det`` is importable (it is not part of this
image -- the constructor then raises with the list of missing packages); tests pass an in-memory stand-in with the same
interface (``directories[name].{tracks.tracks, images.timestamps, events}``, ``subsequence_directories``, ``width``,
``classes``, ``get_tracks`` / ``get_image`` / ``get_events``)."""
import numpy as np
import torch

from .augment import init_transforms
from .dsec_utils import (compute_class_mapping, crop_tracks, filter_small_bboxes, filter_tracks, map_classes,
                         rescale_tracks)
from .utils import to_data

MAPPING = dict(pedestrian="pedestrian", rider=None, car="car", bus="car", truck="car", bicycle=None, motorcycle=None,
               train=None)


# Sequence names per split (the content of the reference's data/dsec_split.yaml, which follows the DSEC-Detection
# release: 41 training, 6 validation, 13 test recordings); handed to DSECDet as split_config (dsec_data.py:74-79).
DSEC_SPLIT = {
    "train": [
        "interlaken_00_c", "interlaken_00_d", "interlaken_00_e", "interlaken_00_f", "interlaken_00_g",
        "thun_00_a", "zurich_city_00_a", "zurich_city_00_b", "zurich_city_01_a", "zurich_city_01_b",
        "zurich_city_01_c", "zurich_city_01_d", "zurich_city_01_e", "zurich_city_01_f", "zurich_city_02_a",
        "zurich_city_02_b", "zurich_city_02_c", "zurich_city_02_d", "zurich_city_02_e", "zurich_city_03_a",
        "zurich_city_04_a", "zurich_city_04_b", "zurich_city_04_c", "zurich_city_04_d", "zurich_city_04_e",
        "zurich_city_04_f", "zurich_city_05_a", "zurich_city_05_b", "zurich_city_06_a", "zurich_city_07_a",
        "zurich_city_08_a", "zurich_city_09_a", "zurich_city_09_b", "zurich_city_09_c", "zurich_city_09_d",
        "zurich_city_09_e", "zurich_city_10_a", "zurich_city_10_b", "zurich_city_11_a", "zurich_city_11_b",
        "zurich_city_11_c"
    ],
    "val": [
        "zurich_city_16_a", "zurich_city_17_a", "zurich_city_18_a", "zurich_city_19_a", "zurich_city_20_a",
        "zurich_city_21_a"
    ],
    "test": [
        "interlaken_00_a", "interlaken_00_b", "interlaken_01_a", "thun_01_a", "thun_01_b", "thun_02_a",
        "zurich_city_12_a", "zurich_city_13_a", "zurich_city_13_b", "zurich_city_14_a", "zurich_city_14_b",
        "zurich_city_14_c", "zurich_city_15_a"
    ],
}


def tracks_to_array(tracks):
    return np.stack([tracks["x"], tracks["y"], tracks["w"], tracks["h"], tracks["class_id"]], axis=1)


def interpolate_tracks(detections_0, detections_1, t):
    """Boxes at time t between two frames holding the same tracks: linear in x, y, w, h, matched by track id
    (dsec_data.py:29-49).  Everything else (class, id, t) is the earlier frame's."""
    assert len(detections_1) == len(detections_0)
    if len(detections_0) == 0:
        return detections_1
    t0, t1 = detections_0["t"][0], detections_1["t"][0]
    assert t0 < t1
    d0 = detections_0[detections_0["track_id"].argsort()]
    d1 = detections_1[detections_1["track_id"].argsort()]
    r = (t - t0) / (t1 - t0)
    out = d0.copy()
    for k in "xywh":
        out[k] = d0[k] * (1 - r) + d1[k] * r
    return out


def _default_source(root, split, demo, debug):
    missing = []
    for name in ("dsec_det", "h5py", "hdf5plugin"):
        try:
            __import__(name)
        except ImportError:
            missing.append(name)
    if missing:
        raise RuntimeError("the DSEC reader needs " + ", ".join(missing) + " (third-party directory / blosc-HDF5 access, "
                           "dsec_data.py:12-16): not installed here.  Pass source= (an object with DSECDet's interface), "
                           "or use dagr.data.synthetic_data.SyntheticWindows")
    from dsec_det.dataset import DSECDet
    split_config = None
    if not demo:
        split_config = DSEC_SPLIT
        assert split in split_config.keys(), f"'{split}' not in {list(split_config.keys())}"
    src = DSECDet(root=root, split=split, sync="back", debug=debug, split_config=split_config)
    from dsec_det.directory import BaseDirectory

    class EventDirectory(BaseDirectory):           # dsec_data.py:51-55: the 2x-downsampled event file
        @property
        def event_file(self):
            return self.root / "left/events_2x.h5"
    for directory in src.directories.values():
        directory.events = EventDirectory(directory.events.root)
    return src


def _resize_area_free(image, width, height):
    """Frame resize for ``preprocess_image`` (the reference calls cv2.resize(..., INTER_CUBIC), dsec_data.py:149-154):
    torch's bicubic interpolation, antialias off, rounded back to uint8."""
    t = torch.from_numpy(np.ascontiguousarray(image)).permute(2, 0, 1)[None].float()
    t = torch.nn.functional.interpolate(t, size=(height, width), mode="bicubic", align_corners=False)
    return t.round().clamp(0, 255).to(torch.uint8)


class DSEC(torch.utils.data.Dataset):
    MAPPING = MAPPING

    def __init__(self, root=None, split="test", transform=None, debug=False, min_bbox_diag=0, min_bbox_height=0, scale=2,
                 cropped_height=430, only_perfect_tracks=False, demo=False, no_eval=False, source=None, resize=None):
        super().__init__()
        self.dataset = source if source is not None else _default_source(root, split, demo, debug)
        self.scale = scale
        self.width = self.dataset.width // scale
        self.height = cropped_height // scale
        self.classes = ("car", "pedestrian")
        self.time_window = 1000000
        self.min_bbox_height, self.min_bbox_diag = min_bbox_height, min_bbox_diag
        self.debug = debug
        self.num_us = -1
        self.class_remapping = compute_class_mapping(self.classes, self.dataset.classes, self.MAPPING)
        if transform is not None and hasattr(transform, "transforms"):
            init_transforms(transform.transforms, self.height, self.width)
        self.transform = transform
        self.no_eval = no_eval
        self._resize = resize or _resize_area_free
        self.image_index_pairs, self.track_masks = filter_tracks(
            dataset=self.dataset, image_width=self.width, image_height=self.height, class_remapping=self.class_remapping,
            min_bbox_height=min_bbox_height, min_bbox_diag=min_bbox_diag,
            only_perfect_tracks=only_perfect_tracks and not no_eval, scale=scale)

    def set_num_us(self, num_us):                  # dsec_data.py:114-115
        self.num_us = num_us

    def __len__(self):
        return sum(len(d) for d in self.image_index_pairs.values())

    def rel_index(self, idx):
        """Global sample index -> (sequence directory, its frame pairs, its track mask, index inside the sequence)."""
        for folder in self.dataset.subsequence_directories:
            pairs = self.image_index_pairs[folder.name]
            if idx < len(pairs):
                return self.dataset.directories[folder.name], pairs, self.track_masks[folder.name], idx
            idx -= len(pairs)
        raise IndexError(idx)

    def preprocess_detections(self, detections):
        detections = crop_tracks(rescale_tracks(detections, self.scale), self.width, self.height)
        detections["class_id"], _ = map_classes(detections["class_id"], self.class_remapping)
        return detections

    def preprocess_events(self, events):
        keep = events["y"] < self.height
        events = {k: v[keep] for k, v in events.items()}
        if len(events["t"]) > 0:
            events["t"] = self.time_window + events["t"] - events["t"][-1]
        events["p"] = 2 * events["p"].reshape((-1, 1)).astype("int8") - 1
        return events

    def preprocess_image(self, image):
        return self._resize(image[:self.scale * self.height], self.width, self.height)

    def __getitem__(self, idx):
        directory, pairs, track_mask, idx = self.rel_index(idx)
        name = directory.root.name
        i0, i1 = pairs[idx]
        ts0, ts1 = directory.images.timestamps[[i0, i1]]
        det0 = self.preprocess_detections(self.dataset.get_tracks(i0, mask=track_mask, directory_name=name))
        det1 = self.preprocess_detections(self.dataset.get_tracks(i1, mask=track_mask, directory_name=name))
        image0 = self.preprocess_image(self.dataset.get_image(i0, directory_name=name))
        events = self.dataset.get_events(i0, directory_name=name)
        if self.num_us >= 0:
            ts1 = ts0 + self.num_us
            events = {k: v[events["t"] < ts1] for k, v in events.items()}
            if not self.no_eval:
                det1 = interpolate_tracks(det0, det1, ts1)
        events = self.preprocess_events(events)
        data = to_data(**events, bbox=tracks_to_array(det1), bbox0=tracks_to_array(det0), t0=ts0, t1=ts1,
                       width=self.width, height=self.height, time_window=self.time_window, image=image0,
                       sequence=str(name))
        if self.transform is not None:
            data = self.transform(data)
        for key in ("bbox", "bbox0"):                   # boxes the transform shrank to nothing (:188-192)
            b = getattr(data, key)
            keep = filter_small_bboxes(b[:, 2].numpy(), b[:, 3].numpy(), self.min_bbox_height, self.min_bbox_diag)
            setattr(data, key, b[torch.from_numpy(keep)])
        return data
