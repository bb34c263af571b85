"""Mirror of ``format_data`` (``src/dagr/utils/buffers.py:33-44``): the input contract of the hot path.
On device tensors the normalisation runs through ``dagr_format_events`` (libdagr_hip) when the batch
carries the dataset's raw dtypes (pos int16[N,2], t int32[N], x int8[N,1]); any other dtype takes the
same arithmetic through torch ops (true division in fp32)."""
import ctypes

import torch

from .. import _lib


def format_data(data, normalizer=None):
    geo = getattr(data, "_geometry", None)
    if geo is None:
        # one read-back for the three scalars (they sit on the device once the batch does); kept on the batch for the
        # layers that ask again (EV_TGN)
        w, h, tw = data.width, data.height, data.time_window
        if torch.is_tensor(w) and w.is_cuda:
            geo = tuple(int(v) for v in torch.stack((w.reshape(-1)[0], h.reshape(-1)[0], tw.reshape(-1)[0])).tolist())
        else:
            geo = (int(w[0]), int(h[0]), int(tw[0]))
        data._geometry = geo
    W, H, T = geo
    if hasattr(data, "image"):
        data.image = data.image.float() / 255.0
    pos, t, x = data.pos, data.t, data.x
    if (normalizer is None and pos.is_cuda and pos.dtype == torch.int16 and t.dtype == torch.int32
            and x.dtype == torch.int8 and pos.is_contiguous() and t.is_contiguous() and x.is_contiguous()):
        N = pos.shape[0]
        pos_out = torch.empty((N, 3), dtype=torch.float32, device=pos.device)
        feat = torch.empty((N, 1), dtype=torch.float32, device=pos.device)
        _lib.check(_lib.lib().dagr_format_events(_lib.ptr(pos), _lib.ptr(t), _lib.ptr(x), N, W, H, T,
                                                 _lib.ptr(pos_out), _lib.ptr(feat), _lib.cur_stream(pos.device)),
                   "format_events")
        data.pos, data.x = pos_out, feat
    else:
        if normalizer is None:
            normalizer = torch.tensor([W, H, T], device=pos.device)
        data.pos = torch.cat([pos, t.view((-1, 1))], dim=-1) / normalizer
        data.x = x.float()
    data.t = None
    return data


# ---------------------------------------------------------------------------------------------------------------
# Detection records and the evaluation buffer (src/dagr/utils/buffers.py:46-122).
DETECTION_DTYPE = [("t", "<u8"), ("x", "<f4"), ("y", "<f4"), ("w", "<f4"), ("h", "<f4"), ("class_id", "u1"),
                   ("class_confidence", "<f4")]


def detections_to_records(det, t):
    """One image's detections ``{boxes[x1,y1,x2,y2], labels[, scores]}`` -> the structured array the reference writes
    (``bbox_t_to_ndarray`` buffers.py:46-66 / ``to_npy`` run_test_interframe.py:21-32).  Ground-truth dicts carry no
    scores: the confidence column is dropped for them, as the reference does."""
    import numpy as np
    boxes = np.asarray(det["boxes"].cpu() if torch.is_tensor(det["boxes"]) else det["boxes"], dtype=np.float32)
    labels = np.asarray(det["labels"].cpu() if torch.is_tensor(det["labels"]) else det["labels"])
    has_scores = "scores" in det
    rec = np.zeros((len(boxes),), dtype=DETECTION_DTYPE if has_scores else DETECTION_DTYPE[:-1])
    rec["t"] = t
    if len(boxes):
        rec["x"], rec["y"] = boxes[:, 0], boxes[:, 1]
        rec["w"], rec["h"] = boxes[:, 2] - boxes[:, 0], boxes[:, 3] - boxes[:, 1]
        rec["class_id"] = labels
        if has_scores:
            rec["class_confidence"] = np.asarray(det["scores"].cpu() if torch.is_tensor(det["scores"]) else det["scores"])
    return rec


def bbox_t_to_ndarray(bbox, t):
    """The reference's name for ``detections_to_records`` (buffers.py:46-66)."""
    return detections_to_records(bbox, t)


def compile(detections, sequences, timestamps):        # noqa: A001  (the reference's module-level name, buffers.py:69-80)
    """Per-sequence concatenation of the per-image record arrays."""
    import numpy as np
    out = {}
    for det, seq, t in zip(detections, sequences, timestamps):
        out.setdefault(seq, []).append(detections_to_records(det, t))
    return {k: np.concatenate(v) for k, v in out.items() if len(v) > 0}


class DictBuffer:
    """Running mean of dictionaries with the same keys (buffers.py:124-146; the FLOP script's accumulator)."""

    def __init__(self):
        self.running_mean = None
        self.n = 0

    def update(self, dictionary):
        if self.running_mean is None:
            self.running_mean = {k: 0 for k in dictionary}
        self.running_mean = {k: self.n / (self.n + 1) * self.running_mean[k] + dictionary[k] / (self.n + 1)
                             for k in dictionary}
        self.n += 1

    def save(self, path):
        torch.save(self.running_mean, path)

    def compute(self):
        return self.running_mean


class DetectionBuffer:
    """Collects detections / ground truth of a test run on the host (buffers.py:100-122).  ``compute`` hands them to the
    COCO-protocol evaluation of ``utils/coco_eval.py`` (pycocotools / detectron2 in the reference; restated in numpy here)."""

    def __init__(self, height, width, classes):
        self.height, self.width, self.classes = height, width, classes
        self.detections, self.ground_truth, self.image_ids = [], [], []

    def update(self, detections, groundtruth, dataset=None, height=None, width=None, image_ids=None):
        """``image_ids``: the GLOBAL index of every image of the batch in the run (sharded runs: the images of a rank are
        a subset); default: a running count, i.e. the order of arrival."""
        n0 = len(self.detections)
        if image_ids is not None and len(image_ids) != len(detections):
            raise ValueError(f"DetectionBuffer.update: {len(image_ids)} image ids for {len(detections)} images")
        self.detections.extend({k: v.cpu() for k, v in d.items()} for d in detections)
        self.ground_truth.extend({k: v.cpu() for k, v in d.items()} for d in groundtruth)
        self.image_ids.extend(image_ids if image_ids is not None else range(n0, n0 + len(detections)))

    def compile(self, sequences, timestamps):
        def by_sequence(items):
            import numpy as np
            out = {}
            for det, seq, t in zip(items, sequences, timestamps):
                out.setdefault(seq, []).append(detections_to_records(det, t))
            return {k: np.concatenate(v) for k, v in out.items()}
        return by_sequence(self.detections), by_sequence(self.ground_truth)

    def compute(self, gather=True, group=None):
        """mAP & co over everything collected since the last call (buffers.py:113-122).  Under a process group (window
        batches sharded over the GPUs of a node) the ranks' images are gathered first -- detections AND ground truth, in
        global image order -- and every rank evaluates the whole run: ONE mAP, the number the reference's single process
        prints (run_test.py:61-65).  That makes this call a COLLECTIVE over ``group`` (default group when None): every
        rank has to make it, with image ids that are unique over the ranks (``update(image_ids=...)``).
        ``gather=False``: this rank's images only, no communication (a caller that scores on one rank)."""
        from .coco_eval import evaluate_detection
        from ..parallel import gather_evaluation
        if gather:
            dets, gts, _ = gather_evaluation(self.detections, self.ground_truth, self.image_ids, group=group)
        else:
            order = sorted(range(len(self.image_ids)), key=lambda i: int(self.image_ids[i]))
            dets, gts = [self.detections[i] for i in order], [self.ground_truth[i] for i in order]
        out = evaluate_detection(gts, dets, height=self.height, width=self.width, classes=self.classes)
        self.detections, self.ground_truth, self.image_ids = [], [], []
        return {k.replace("AP", "mAP"): v for k, v in out.items()}
