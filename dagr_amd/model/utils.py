"""Hot-path helpers with the names of ``src/dagr/model/utils.py``: ``voxel_size_to_params`` :112-116,
``postprocess_network_output`` :61-110 (batched on the device: csrc/nms.hip replaces the per-image
``batched_nms_coordinate_trick`` / torchvision NMS loop), ``convert_to_evaluation_format`` :35-44,
``init_subnetwork`` :9-23."""
import numpy as np
import torch


def voxel_size_to_params(pooling_layer, height, width):
    rx = int(np.ceil(2 * pooling_layer.voxel_size[0].cpu().numpy() * width))
    ry = int(np.ceil(2 * pooling_layer.voxel_size[1].cpu().numpy() * height))
    M = pooling_layer.transform.max
    return rx, ry, M


def init_subnetwork(net, state_dict, name="backbone.net.", freeze=False):
    assert name.endswith(".")
    for attr in name.split(".")[:-1]:
        net = getattr(net, attr)
    sub_state_dict = {k.replace(name, ""): v for k, v in state_dict.items() if name in k}
    net.load_state_dict(sub_state_dict)
    if freeze:
        for param in net.parameters():
            param.requires_grad = False


def postprocess_network_output(prediction, num_classes, conf_thre=0.01, nms_thre=0.65, height=640, width=640,
                               filtering=True):
    """Batched, device-side version of ``model/utils.py:61-110``: cxcywh -> xyxy, class max, the reference's
    confidence mask (obj * cls * cls >= thr) and class-offset greedy NMS for all images in ONE kernel launch
    (``dagr_postprocess``, csrc/nms.hip).  ``prediction``: [B, A, 5 + C] on the GPU."""
    if not prediction.is_cuda:
        raise RuntimeError("postprocess_network_output expects the decoded head outputs on the GPU")
    if not filtering:
        # :87-88,101-102: neither the confidence mask nor the NMS indices are applied -- every anchor comes back,
        # in anchor order
        B = prediction.shape[0]
        xy, wh = prediction[..., :2], prediction[..., 2:4]
        x1y1 = xy - wh / 2
        boxes = torch.cat((x1y1, wh + x1y1), dim=-1)                                   # same op order as :62-63
        class_conf, class_pred = torch.max(prediction[..., 5:5 + num_classes], dim=-1)
        scores = prediction[..., 4] * class_conf                                       # image_pred[:, 4:5] *= class_conf
        return [{"boxes": boxes[b], "scores": scores[b], "labels": class_pred[b].long()} for b in range(B)]
    return detections_from_device(*postprocess_device(prediction, num_classes, conf_thre, nms_thre, height, width))


def detections_from_device(det, n_keep):
    """``(det[B, A, 6], n_keep[B])`` -> the reference's per-image dicts (model/utils.py:104-108).  ONE D2H copy (the B
    survivor counts) cuts them: the return type is variable-length, so one synchronisation is inherent.  The rows are
    copied out of `det` (a captured window hands out its static buffer, rewritten by the next one)."""
    counts = n_keep.tolist()
    det = det.clone()
    return [{"boxes": det[b, :n, :4], "scores": det[b, :n, 4], "labels": det[b, :n, 5].long()}
            for b, n in enumerate(counts)]


def postprocess_device(prediction, num_classes, conf_thre=0.01, nms_thre=0.65, height=640, width=640):
    """One launch, no synchronisation (``dagr_postprocess``): returns ``det[B, A, 6]`` whose first ``n_keep[b]`` rows
    per image are the surviving detections (x1, y1, x2, y2, score, label) by descending score, and ``n_keep[B]``."""
    from .. import _lib
    B, A, C = prediction.shape
    pred = prediction.contiguous()
    det = torch.empty((B, A, 6), dtype=torch.float32, device=pred.device)
    n_keep = torch.empty((B,), dtype=torch.int32, device=pred.device)
    _lib.check(_lib.lib().dagr_postprocess(_lib.ptr(pred), B, A, int(num_classes), float(conf_thre), float(nms_thre),
                                           float(max(width, height) + 1), _lib.ptr(det), _lib.ptr(n_keep),
                                           _lib.cur_stream(pred.device)), "postprocess")
    return det, n_keep


def convert_to_evaluation_format(data):
    targets = []
    bb = data.bbox
    bidx = getattr(data, "bbox_batch", torch.zeros(len(bb), dtype=torch.long, device=bb.device))
    for i in range(data.num_graphs):
        bbox = bb[bidx == i].clone()
        bbox[:, 2:4] += bbox[:, :2]
        targets.append({"boxes": bbox[:, :4], "labels": bbox[:, 4].long()})
    return targets


def convert_to_training_format(bbox, batch, batch_size):
    """``model/utils.py:47-60``: per-sample rows (class, cx, cy, w, h) in pixels, zero-padded to 100 boxes."""
    max_detections = 100
    targets = torch.zeros((batch_size, max_detections, 5), dtype=torch.float32, device=bbox.device)
    if bbox.shape[0] == 0:
        return targets
    # running index of every box inside its sample (boxes of a sample are contiguous, samples ascending)
    first = torch.ones_like(batch, dtype=torch.bool)
    first[1:] = batch[1:] != batch[:-1]
    idx = torch.arange(len(batch), device=batch.device)
    counter = idx - torch.cummax(torch.where(first, idx, torch.zeros_like(idx)), 0).values     # (no mask indexing: no host sync)
    rows = bbox[:, :5].clone().float()
    rows[:, :2] += rows[:, 2:4] * .5                       # corner -> centre
    targets[batch, counter] = torch.roll(rows, shifts=1, dims=1)
    return targets


def shallow_copy(data):
    """``model/utils.py:158-166``: a new ``Data`` sharing graph tensors with ``data`` but owning a copy of ``x`` (the
    cached CSR of the graph is shared, the reference's ``adj_t`` is not carried over -- it is recomputed there)."""
    out = data.__class__()
    keep = ("edge_index", "edge_attr", "pos", "batch", "pooling", "num_image_channels", "skipped", "pooled", "width",
            "height", "time_window", "edge_attr_max", "_dagr_csr", "_dagr_exact", "_dagr_pixel_codes", "_lazy")
    for k in keep:
        if k in data.__dict__:
            out.__dict__[k] = data.__dict__[k]
    out.x = data.x.clone()
    return out


def init_grid_and_stride(hw, strides, like):
    """``model/utils.py:119-134``: anchor grid and stride per output cell, concatenated over the scales."""
    grids, all_strides = [], []
    for (hsize, wsize), stride in zip(hw, strides):
        yv, xv = torch.meshgrid(torch.arange(hsize), torch.arange(wsize), indexing="ij")
        grid = torch.stack((xv, yv), 2).view(1, -1, 2)
        grids.append(grid)
        all_strides.append(torch.full((1, grid.shape[1], 1), stride))
    return (torch.cat(grids, dim=1).to(like.dtype).to(like.device),
            torch.cat(all_strides, dim=1).to(like.dtype).to(like.device))
