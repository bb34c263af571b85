"""Test driver with the reference's entry point ``run_test_with_visualization`` (``src/dagr/utils/testing.py:16-55``):
same signature and return convention, so ``scripts/run_test.py:61`` / ``run_test_interframe.py:83`` call it unchanged.

Per batch: ``data.cuda()`` -> ``format_data`` -> ``model(data)`` -> (optionally) collect the detections with their
sequence / timestamp, feed the mAP buffer.  Differences that follow from this stack: the per-window work is one
device pipeline with a single synchronisation (the survivor counts of the NMS), visualisation (wandb image logging) is
not part of the hot path and is skipped, and the COCO evaluation is the numpy restatement in ``utils/coco_eval.py``
(pycocotools / detectron2 are absent); ``no_eval=True`` skips it (datasets without boxes).  Under a process group the
window batches are sharded over the ranks; the metrics are those of the WHOLE run on every rank (detections and ground
truth gathered once, ``parallel.gather_evaluation``)."""
import torch

from .buffers import DetectionBuffer, format_data


def to_npy(detections):
    return [{k: v.cpu().numpy() for k, v in d.items()} for d in detections]


def format_detections(sequences, t, detections):
    """Detections of one batch as numpy dicts tagged with their sequence name and timestamp (testing.py:9-14)."""
    out = to_npy(detections)
    t = t.tolist() if torch.is_tensor(t) else list(t)
    for det, seq, ts in zip(out, sequences, t):
        det["sequence"] = seq
        det["t"] = ts
    return out


def run_test_with_visualization(loader, model, dataset: str, log_every_n_batch=-1, name="", compile_detections=False,
                                no_eval=False):
    model.eval()
    scorer = None
    if not no_eval:
        ds = loader.dataset
        scorer = DetectionBuffer(height=ds.height, width=ds.width, classes=ds.classes)
    collected = [] if compile_detections else None
    # global index of an image in the run (``DataLoader.image_ids``: right for both ways of sharding a loader); a foreign
    # loader numbers its images in order of arrival
    for step, data in enumerate(loader):
        if torch.cuda.is_available():
            data = data.cuda(non_blocking=True)
        data = format_data(data)
        out = model(data, return_targets=scorer is not None)
        detections = out[0]
        if collected is not None:
            stamps = data.t1 if hasattr(data, "t1") else [0] * len(detections)
            collected.extend(format_detections(data.sequence, stamps, detections))
        if scorer is not None:
            if len(out) < 2:
                raise RuntimeError("evaluation needs ground-truth boxes (data.bbox); pass no_eval=True without them")
            ids = loader.image_ids(step) if hasattr(loader, "image_ids") else None
            scorer.update(detections, out[1], dataset, data.height[0], data.width[0], image_ids=ids)
    metrics = scorer.compute() if scorer is not None else None
    return (metrics, collected) if compile_detections else metrics
