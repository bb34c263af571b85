"""CPU suite: the COCO-protocol detection metrics (dagr_amd/utils/coco_eval.py; pycocotools / detectron2 in the reference:
third-party, absent, parity unpinned) on cases whose AP follows by hand from the protocol's definition."""
import numpy as np
import pytest
import torch

from dagr_amd.utils.coco_eval import evaluate_detection


def _img(gt, dt):
    g = np.asarray(gt, dtype=np.float32).reshape(-1, 5)
    d = np.asarray(dt, dtype=np.float32).reshape(-1, 6)
    return (dict(boxes=torch.from_numpy(g[:, :4]), labels=torch.from_numpy(g[:, 4]).long()),
            dict(boxes=torch.from_numpy(d[:, :4]), labels=torch.from_numpy(d[:, 4]).long(), scores=torch.from_numpy(d[:, 5])))


def _run(pairs, **kw):
    gts, dts = zip(*pairs)
    return evaluate_detection(list(gts), list(dts), **kw)


def test_perfect_detections_score_one_and_only_populated_area_ranges_count():
    # two medium boxes (50x50 = 2500 px^2, between 32^2 and 96^2), detected exactly
    out = _run([_img([[10, 10, 60, 60, 0]], [[10, 10, 60, 60, 0, 0.9]]),
                _img([[100, 80, 150, 130, 1]], [[100, 80, 150, 130, 1, 0.8]])])
    assert out["AP"] == pytest.approx(1.0) and out["AP_50"] == pytest.approx(1.0) and out["AP_75"] == pytest.approx(1.0)
    assert out["AP_M"] == pytest.approx(1.0)
    assert out["AP_S"] == -1.0 and out["AP_L"] == -1.0            # no ground truth in those ranges: undefined, as COCO reports


def test_iou_threshold_sweep():
    # one box, detected with IoU 0.62: a true positive at thresholds 0.50 .. 0.60 (3 of 10), a false positive above
    g = [0, 0, 100, 100, 0]
    w = 100 * 0.62                                                 # same height, narrower: IoU = w / 100
    out = _run([_img([g], [[0, 0, w, 100, 0, 0.9]])])
    assert out["AP_50"] == pytest.approx(1.0) and out["AP_75"] == pytest.approx(0.0)
    assert out["AP"] == pytest.approx(0.3)


def test_precision_recall_curve_with_a_false_positive_in_between():
    # class 0: two ground-truth boxes in two images; detections by score: TP (0.9), FP (0.8), TP (0.7)
    # precision envelope: recall <= 0.5 -> 1.0, recall in (0.5, 1.0] -> 2/3; 101 points: 51 x 1.0 + 50 x 2/3
    big = [0, 0, 100, 100, 0]
    out = _run([_img([big], [[0, 0, 100, 100, 0, 0.9], [300, 300, 400, 400, 0, 0.8]]),
                _img([big], [[0, 0, 100, 100, 0, 0.7]])])
    want = (51 * 1.0 + 50 * (2 / 3)) / 101
    assert out["AP_50"] == pytest.approx(want, rel=1e-9) and out["AP"] == pytest.approx(want, rel=1e-9)
    assert out["AP_L"] == pytest.approx(want, rel=1e-9)            # 100 x 100 > 96^2


def test_classes_average_and_images_without_ground_truth_are_skipped():
    a = _img([[0, 0, 50, 50, 0]], [[0, 0, 50, 50, 0, 0.9]])                      # class 0: AP 1
    b = _img([[0, 0, 50, 50, 1]], [[200, 200, 250, 250, 1, 0.9]])                # class 1: missed, AP 0
    empty = _img(np.zeros((0, 5)), [[0, 0, 50, 50, 0, 0.99], [0, 0, 50, 50, 1, 0.99]])   # no boxes: not evaluated
    out = _run([a, b, empty])
    assert out["AP"] == pytest.approx(0.5) and out["AP_50"] == pytest.approx(0.5)


def test_greedy_matching_prefers_the_best_overlap_and_duplicates_are_false_positives():
    # two detections on one box: the higher-scoring one takes it, the second is a false positive after it
    out = _run([_img([[0, 0, 100, 100, 0]], [[0, 0, 100, 100, 0, 0.6], [2, 0, 100, 100, 0, 0.9]])])
    # score order: (IoU 0.98, 0.9) matches, then the exact box finds the ground truth taken -> FP; recall 1 at precision 1
    assert out["AP"] == pytest.approx(1.0)
    # reversed scores: the exact box matches first -> still AP 1 (FP comes after full recall)
    out = _run([_img([[0, 0, 100, 100, 0]], [[0, 0, 100, 100, 0, 0.9], [2, 0, 100, 100, 0, 0.6]])])
    assert out["AP"] == pytest.approx(1.0)
    # a false positive BEFORE the match halves the precision everywhere
    out = _run([_img([[0, 0, 100, 100, 0]], [[300, 300, 400, 400, 0, 0.95], [0, 0, 100, 100, 0, 0.9]])])
    assert out["AP"] == pytest.approx(0.5)


def test_no_detections_returns_zeros_like_the_reference():
    g, d = _img([[0, 0, 50, 50, 0]], np.zeros((0, 6)))
    assert evaluate_detection([g], [d]) == {k: 0 for k in ("AP", "AP_50", "AP_75", "AP_S", "AP_M", "AP_L")}


def test_detection_buffer_compute_renames_to_map():
    from dagr_amd.utils.buffers import DetectionBuffer
    buf = DetectionBuffer(height=215, width=320, classes=("car", "pedestrian"))
    g, d = _img([[10, 10, 60, 60, 1]], [[10, 10, 60, 60, 1, 0.9]])
    buf.update([d], [g])
    out = buf.compute()
    assert set(out) == {"mAP", "mAP_50", "mAP_75", "mAP_S", "mAP_M", "mAP_L"} and out["mAP"] == pytest.approx(1.0)
    assert buf.detections == [] and buf.ground_truth == []
