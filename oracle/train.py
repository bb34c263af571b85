"""CPU oracle of one training forward (BASELINE config 5; ``scripts/train_ncaltech101.py:49-58``): ``model.train()`` ->
``DAGR.forward`` training branch (dagr.py:78-88) -> ``YOLOX.forward`` -> ``GNNHead.forward`` losses (dagr.py:238-282).

TEST INFRASTRUCTURE ONLY -- see oracle/__init__.py.  The forward is ``oracle.model.forward_events`` on the non-LUT
message path (``cache_luts`` is an evaluation-time step, run_test.py:59; training evaluates the spline basis on the edge
attributes, spline_conv.py:64-78) with batch-statistics BatchNorm (``oracle.ops.batch_statistics``); torch autograd through
those restatements gives the gradients the GPU path is compared with.  Pinned to the reference's own code by
``tests/golden/ref_py_model.npz`` (training cases of tests/make_golden_refpy_model.py); the loss itself is third-party
(``oracle/yolox_loss.py``: parity unpinned)."""
import torch

from . import model as om
from . import ops
from .yolox_loss import LossHead


def sequential_counter(counts):
    """``_sequential_counter`` (model/utils.py:136-156): [2, 3] -> [0, 1, 0, 1, 2]."""
    return torch.cat([torch.arange(int(c)) for c in counts]) if len(counts) else torch.zeros(0, dtype=torch.long)


def convert_to_training_format(bbox, batch, batch_size):
    """``model/utils.py:47-60``: rows (x, y, w, h, class, ...) -> [B, 100, 5] of (class, cx, cy, w, h), zero padded."""
    targets = torch.zeros((batch_size, 100, 5), dtype=torch.float32)
    _, counts = torch.unique(batch, return_counts=True)
    counter = sequential_counter(counts)
    bbox = bbox.clone()
    bbox[:, :2] += bbox[:, 2:4] * .5
    bbox = torch.roll(bbox[:, :5], dims=1, shifts=1)
    targets[batch, counter] = bbox
    return targets


def training_losses(sd, args, height, width, x, y, t, p, b, batch_size, bbox, bbox_batch, image_feat=None, cnn_out=None,
                    bbox0=None, bbox0_batch=None, exact_pos_mean=False):
    """The 6-tuple of ``get_losses`` (total, 5*iou, obj, cls, l1 = 0, matched anchors / ground truths) for one batch of
    windows; ``sd`` may hold leaf tensors that require grad.

    ``--use_image`` (dagr.py:197-222,241-268): ``image_feat`` = the image branch's feature maps (sampled into the graph
    DETACHED, net.py:118,129,...), ``cnn_out`` = the CNN head's raw maps on the resized image outputs (dict of lists
    ``cls_output`` / ``reg_output`` / ``obj_output``; they enter the hybrid sum detached, and train on their own through a
    second ``get_losses`` against the boxes of the EARLIER frame, ``bbox0``); the two loss tuples are added element-wise
    for the first five entries, the sixth is the image branch's.

    ``exact_pos_mean``: pooled positions from the correctly rounded mean (``oracle.ops.pooling``): the form the GPU
    comparisons use -- a cluster mean that lands within 1e-4 px of a pixel boundary floors differently under fp32
    sequential summation, which moves ONE pooled node by a pixel and with it ~1 % of a mid-level weight gradient."""
    nc = om.NetConstants(args, height, width)
    detached = None
    if cnn_out is not None:
        detached = {k: [m.detach() for m in v] for k, v in cnn_out.items()}
        image_feat = [f.detach() for f in image_feat]
    with ops.batch_statistics():
        _, raw = om.forward_events(sd, args, height, width, x, y, t, p, b, batch_size, use_lut=False,
                                   image_feat=image_feat, cnn_out=detached, exact_pos_mean=exact_pos_mean)
    maps = [torch.cat([reg_o, obj_o, cls_o], 1) for (cls_o, reg_o, obj_o) in raw]      # collect_outputs, dagr.py:293-294
    labels = convert_to_training_format(bbox, bbox_batch, batch_size)
    n_cls = maps[0].shape[1] - 5
    events = LossHead(n_cls, len(maps)).losses_from_maps(maps, nc.strides, labels)
    if cnn_out is None:
        return events
    image_maps = [torch.cat([cnn_out["reg_output"][k], cnn_out["obj_output"][k], cnn_out["cls_output"][k]], 1)
                  for k in range(len(cnn_out["cls_output"]))]
    labels0 = convert_to_training_format(bbox0, bbox0_batch, batch_size)
    image = list(LossHead(n_cls, len(image_maps)).losses_from_maps(image_maps, nc.strides, labels0))
    for i in range(5):
        image[i] = image[i] + events[i]
    return tuple(image)
