"""Track and event-window helpers of the DSEC reader (``src/dagr/data/dsec_utils.py``): box rescaling / clipping / class
remapping / size filter (:15-49), the choice of consecutive labelled frame pairs (``filter_tracks`` :51-80,
``construct_pairs`` :5-13, the optional "perfect tracks" filter :129-190) and the millisecond-indexed event-window read
of ``events_2x.h5`` (``_load_events`` :82-126).  Tracks are numpy structured arrays with fields
``t, x, y, w, h, class_id, track_id`` (the dsec-det format).  Pure host-side numpy; held to the reference's functions by
tests/golden/ref_py_data.npz."""
import numpy as np


def construct_pairs(indices, n=2):
    """Rows [i, i+1, ..., i+n-1] for every run of n consecutive values in ``indices``."""
    idx = np.sort(np.asarray(indices))
    if len(idx) < n:
        return np.zeros((0, n), dtype=idx.dtype)
    windows = np.stack([idx[k:len(idx) - n + 1 + k] for k in range(n)], axis=1)
    consecutive = (np.diff(windows, axis=1) == 1).all(axis=1)
    return windows[consecutive]


def rescale_tracks(tracks, scale):
    out = tracks.copy()
    for k in "xywh":
        out[k] /= scale
    return out


def crop_tracks(tracks, width, height):
    out = tracks.copy()
    x1, y1 = np.clip(out["x"], 0, width - 1), np.clip(out["y"], 0, height - 1)
    x2 = np.clip(tracks["x"] + tracks["w"], 0, width - 1)
    y2 = np.clip(tracks["y"] + tracks["h"], 0, height - 1)
    out["x"], out["y"], out["w"], out["h"] = x1, y1, x2 - x1, y2 - y1
    return out


def map_classes(class_ids, old_to_new_mapping):
    new_ids = old_to_new_mapping[class_ids]
    return new_ids, new_ids > -1


def filter_small_bboxes(w, h, bbox_height=20, bbox_diag=30):
    diag = np.sqrt(h ** 2 + w ** 2)
    return (diag > bbox_diag) & (w > bbox_height) & (h > bbox_height)


def compute_class_mapping(classes, all_classes, mapping):
    """Index of every dataset class in ``classes`` after ``mapping`` (class name -> kept class name or None); -1 = dropped."""
    classes = list(classes)
    return np.array([classes.index(mapping[c]) if mapping[c] in classes else -1 for c in all_classes])


def box_iou(a, b):
    """Element-wise IoU of two equally long track arrays (dsec_utils.py:150-170)."""
    ax2, ay2 = a["x"] + a["w"], a["y"] + a["h"]
    bx2, by2 = b["x"] + b["w"], b["y"] + b["h"]
    ix1, iy1 = np.maximum(a["x"], b["x"]), np.maximum(a["y"], b["y"])
    ix2, iy2 = np.minimum(ax2, bx2), np.minimum(ay2, by2)
    inter = np.zeros_like(a["x"])
    ok = (iy2 > iy1) & (ix2 > ix1)
    inter[ok] = (ix2[ok] - ix1[ok]) * (iy2[ok] - iy1[ok])
    union = a["w"] * a["h"] + b["w"] * b["h"] - inter + 1e-9
    return inter / union


def _frame_slices(t):
    """[start, end) of every run of equal timestamps in the (time-sorted) track array, and the run's timestamp."""
    stamps, counts = np.unique(t, return_counts=True)
    ends = counts.cumsum()
    return stamps, ends - counts, ends


def perfect_track_mask(tracks, frame_pairs_t, tracks_mask=None, min_iou=0.10):
    """True for frame pairs (t0, t1) whose surviving tracks are the same set of track ids in both frames and overlap by
    at least ``min_iou`` each (dsec_utils.py:129-148)."""
    stamps, starts, ends = _frame_slices(tracks["t"])
    where = {int(s): (int(a), int(b)) for s, a, b in zip(stamps, starts, ends)}
    keep = np.ones(len(frame_pairs_t), dtype=bool)
    for i, (t0, t1) in enumerate(frame_pairs_t):
        frames = []
        for t in (t0, t1):
            a, b = where[int(t)]
            fr = tracks[a:b]
            if tracks_mask is not None:
                fr = fr[tracks_mask[a:b]]
            frames.append(fr[fr["track_id"].argsort()])
        f0, f1 = frames
        if len(f0) != len(f1) or not bool((f0["track_id"] == f1["track_id"]).all()):
            keep[i] = False
        else:
            # (pairs are built from frames that hold surviving tracks, so the frames are not empty)
            keep[i] = len(f0) == 0 or float(np.min(box_iou(f0, f1))) >= min_iou
    return keep


def filter_tracks(dataset, image_width, image_height, class_remapping, min_bbox_height=0, min_bbox_diag=0, scale=1,
                  only_perfect_tracks=False):
    """Per sequence: the (i, i+1) image-index pairs whose frames both carry at least one track that survives rescaling,
    clipping, the class remap and the size filter -- and the per-track survival mask."""
    image_index_pairs, track_masks = {}, {}
    for directory_path in dataset.subsequence_directories:
        name = directory_path.name
        tracks = dataset.directories[name].tracks.tracks
        stamps = dataset.directories[name].images.timestamps
        scaled = crop_tracks(rescale_tracks(tracks, scale), image_width, image_height)
        _, class_ok = map_classes(scaled["class_id"], class_remapping)
        keep = filter_small_bboxes(scaled["w"], scaled["h"], min_bbox_height, min_bbox_diag) & class_ok
        valid_images = np.unique(np.nonzero(np.isin(stamps, scaled[keep]["t"]))[0])
        pairs = construct_pairs(valid_images, 2)
        if only_perfect_tracks:
            pairs = pairs[perfect_track_mask(scaled, stamps[pairs], tracks_mask=keep)]
        image_index_pairs[name] = pairs
        track_masks[name] = keep
    return image_index_pairs, track_masks


def load_event_window(f, t0, num_events=None, num_us=None, height=None, time_window=None):
    """Events of an ``events_2x.h5``-layout file (``events/{x,y,t,p}``, ``t_offset``, ``ms_to_idx``) starting at the
    millisecond of absolute time ``t0``: ``num_events`` events or ``num_us`` microseconds (negative = backwards).
    ``f``: an open h5py.File or anything indexable the same way.  Returns ((xy int16[n,2], t int32[n,1], p int8[n,1]), tq)
    with t shifted so that the newest event sits at ``time_window`` (dsec_utils.py:82-126)."""
    t_offset = f["t_offset"][()]
    ms = int((t0 - t_offset) / 1e3)
    i0 = int(f["ms_to_idx"][ms])
    if num_events is not None:
        i1 = i0 + num_events
    if num_us is not None:
        i1 = int(f["ms_to_idx"][ms + int(num_us / 1e3)])
    i0, i1 = sorted([i0, i1])
    i0, i1 = max(i0, 0), max(i1, 0)
    ev = {k: np.asarray(f[f"events/{k}"][i0:i1]) for k in "xytp"}
    tq = ev["t"][-1] if i1 > i0 else f["events/t"][max(i1 - 1, i0)]
    p = 2 * ev["p"][..., None].astype("int8") - 1
    t_ev = ev["t"][..., None]
    xy = np.stack([ev["x"], ev["y"]], axis=-1).astype("int16")
    t = (time_window - tq + t_ev).astype("int32") if time_window is not None else np.array(tq)
    tq = np.int64(tq + t_offset)
    keep = t[:, 0] > 0
    if height is not None:
        keep &= xy[:, 1] < height
    return (xy[keep], t[keep], p[keep]), tq
