"""Detection metrics with the interface of ``src/dagr/utils/coco_eval.py`` (``evaluate_detection`` :64-94): AP, AP_50,
AP_75, AP_S, AP_M, AP_L of per-image detections against per-image ground truth.

The reference converts both lists to COCO dictionaries (:175-233) and hands them to ``pycocotools.COCO`` +
``detectron2``'s ``COCOeval_opt`` -- third-party, absent here, **parity unpinned**.  The evaluation protocol those
packages implement is public and is restated below in numpy (``_evaluate_image`` / ``_accumulate`` follow COCOeval's
``evaluateImg`` / ``accumulate`` / ``summarize``: greedy score-ordered matching per IoU threshold in {0.50 .. 0.95}, 100
detections per image, area ranges all / small < 32^2 / medium / large > 96^2 with out-of-range ground truth ignored,
101-point interpolated precision, mean over classes with at least one ground-truth box).

What is the reference's own and kept: images without ground truth are not evaluated (``_match_times`` walks the
ground-truth timestamps, :110-144 -- and ``_to_prophesee`` leaves every timestamp at 0, so one image = one window holding
all of its boxes and detections), class ids shift by one, boxes are (x1, y1, x2, y2) on input and (x, y, w, h) in the
evaluation, an evaluation without any detection returns zeros (:45-49)."""
import numpy as np

IOU_THRS = np.linspace(.5, 0.95, int(np.round((0.95 - .5) / .05)) + 1, endpoint=True)
REC_THRS = np.linspace(.0, 1.00, int(np.round((1.00 - .0) / .01)) + 1, endpoint=True)
AREA_RNG = [[0 ** 2, 1e5 ** 2], [0 ** 2, 32 ** 2], [32 ** 2, 96 ** 2], [96 ** 2, 1e5 ** 2]]
MAX_DETS = 100
OUT_KEYS = ("AP", "AP_50", "AP_75", "AP_S", "AP_M", "AP_L")


def _xywh(d):
    b = np.asarray(d["boxes"].cpu() if hasattr(d["boxes"], "cpu") else d["boxes"], dtype=np.float64).reshape(-1, 4)
    # through float32 like the reference's structured array (BBOX_DTYPE: '<f4')
    b = b.astype(np.float32)
    return np.stack([b[:, 0], b[:, 1], b[:, 2] - b[:, 0], b[:, 3] - b[:, 1]], 1).astype(np.float64)


def _arr(v, dtype):
    return np.asarray(v.cpu() if hasattr(v, "cpu") else v).astype(dtype).reshape(-1)


def _iou_xywh(d, g):
    """IoU matrix [len(d), len(g)] of (x, y, w, h) boxes (maskApi bbIou, iscrowd = 0)."""
    if len(d) == 0 or len(g) == 0:
        return np.zeros((len(d), len(g)))
    w = np.minimum(d[:, None, 0] + d[:, None, 2], g[None, :, 0] + g[None, :, 2]) - np.maximum(d[:, None, 0], g[None, :, 0])
    h = np.minimum(d[:, None, 1] + d[:, None, 3], g[None, :, 1] + g[None, :, 3]) - np.maximum(d[:, None, 1], g[None, :, 1])
    inter = np.where((w > 0) & (h > 0), w * h, 0.0)
    union = (d[:, 2] * d[:, 3])[:, None] + (g[:, 2] * g[:, 3])[None, :] - inter
    return inter / union


def _evaluate_image(gt, dt, scores, area_rng):
    """One (image, class, area range): matches of the score-sorted detections per IoU threshold.
    Returns (scores sorted, matched [T, D] bool, det ignored [T, D] bool, gt ignored [G] bool) or None when empty."""
    if len(gt) == 0 and len(dt) == 0:
        return None
    g_area = gt[:, 2] * gt[:, 3]
    g_ign = (g_area < area_rng[0]) | (g_area > area_rng[1])
    g_order = np.argsort(g_ign, kind="mergesort")                    # evaluated ground truth first
    gt, g_ign = gt[g_order], g_ign[g_order]
    d_order = np.argsort(-scores, kind="mergesort")[:MAX_DETS]
    dt, scores = dt[d_order], scores[d_order]
    ious = _iou_xywh(dt, gt)
    T, D, G = len(IOU_THRS), len(dt), len(gt)
    gtm = -np.ones((T, G), dtype=np.int64)
    dtm = np.zeros((T, D), dtype=bool)
    dt_ign = np.zeros((T, D), dtype=bool)
    for ti, thr in enumerate(IOU_THRS):
        for di in range(D):
            best, m = min(thr, 1 - 1e-10), -1
            for gi in range(G):
                if gtm[ti, gi] >= 0:
                    continue
                if m > -1 and not g_ign[m] and g_ign[gi]:
                    break                                            # only ignored ground truth left: keep the match
                if ious[di, gi] < best:
                    continue
                best, m = ious[di, gi], gi
            if m == -1:
                continue
            dt_ign[ti, di] = g_ign[m]
            dtm[ti, di] = True
            gtm[ti, m] = di
    d_area = dt[:, 2] * dt[:, 3]
    out_of_range = (d_area < area_rng[0]) | (d_area > area_rng[1])
    dt_ign = dt_ign | (~dtm & out_of_range[None, :])
    return scores, dtm, dt_ign, g_ign


def _accumulate(per_image):
    """Precision at the 101 recall points per IoU threshold for one (class, area range); None without ground truth."""
    per_image = [e for e in per_image if e is not None]
    if not per_image:
        return None
    scores = np.concatenate([e[0] for e in per_image])
    order = np.argsort(-scores, kind="mergesort")
    dtm = np.concatenate([e[1] for e in per_image], axis=1)[:, order]
    dt_ign = np.concatenate([e[2] for e in per_image], axis=1)[:, order]
    n_gt = int(sum((~e[3]).sum() for e in per_image))
    if n_gt == 0:
        return None
    tps = np.cumsum(dtm & ~dt_ign, axis=1).astype(np.float64)
    fps = np.cumsum(~dtm & ~dt_ign, axis=1).astype(np.float64)
    precision = np.zeros((len(IOU_THRS), len(REC_THRS)))
    for ti in range(len(IOU_THRS)):
        tp, fp = tps[ti], fps[ti]
        rc = tp / n_gt
        pr = tp / (fp + tp + np.spacing(1))
        for i in range(len(pr) - 1, 0, -1):                          # precision envelope
            if pr[i] > pr[i - 1]:
                pr[i - 1] = pr[i]
        idx = np.searchsorted(rc, REC_THRS, side="left")
        ok = idx < len(pr)
        precision[ti, ok] = pr[idx[ok]]
    return precision


def evaluated_images(gt_boxes_list, dt_boxes_list):
    """What the reference's ``_convert_to_coco_format`` (:15-60) hands to COCO: one entry per image WITH ground truth, in
    input order -- (gt boxes xywh, gt classes, detection boxes xywh, detection classes, scores)."""
    images = []
    for gt, dt in zip(gt_boxes_list, dt_boxes_list):
        g_box = _xywh(gt)
        if len(g_box) == 0:
            continue                                                 # KPIs only where there is at least one box (:29-30)
        d_box = _xywh(dt)
        d_score = _arr(dt["scores"], np.float32).astype(np.float64) if "scores" in dt else np.ones(len(d_box))
        images.append((g_box, _arr(gt["labels"], np.int64), d_box, _arr(dt["labels"], np.int64), d_score))
    return images


def evaluate_detection(gt_boxes_list, dt_boxes_list, classes=("car", "pedestrian"), height=240, width=304,
                       time_tol=50000):
    """gt / dt: one dict per image, ``boxes`` [n, 4] (x1, y1, x2, y2), ``labels`` [n], detections also ``scores`` [n]."""
    images = evaluated_images(gt_boxes_list, dt_boxes_list)
    if sum(len(im[2]) for im in images) == 0:
        return {k: 0 for k in OUT_KEYS}
    prec = -np.ones((len(IOU_THRS), len(REC_THRS), len(classes), len(AREA_RNG)))
    for c in range(len(classes)):
        for ai, rng in enumerate(AREA_RNG):
            per_image = [_evaluate_image(g[gl == c], d[dl == c], s[dl == c], rng) for g, gl, d, dl, s in images]
            p = _accumulate(per_image)
            if p is not None:
                prec[:, :, c, ai] = p

    def mean(sel):
        v = sel[sel > -1]
        return float(v.mean()) if v.size else -1.0
    return {"AP": mean(prec[:, :, :, 0]), "AP_50": mean(prec[0, :, :, 0]), "AP_75": mean(prec[5, :, :, 0]),
            "AP_S": mean(prec[:, :, :, 1]), "AP_M": mean(prec[:, :, :, 2]), "AP_L": mean(prec[:, :, :, 3])}
