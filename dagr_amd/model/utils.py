"""Mirror of the hot-path helpers in ``src/dagr/model/utils.py``: ``voxel_size_to_params`` :112-116,
``postprocess_network_output`` :61-110 (+ ``batched_nms_coordinate_trick`` :25-33 with a torch NMS,
torchvision being absent), ``convert_to_evaluation_format`` :35-44, ``init_subnetwork`` :9-23."""
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


def nms(boxes, scores, iou_threshold):
    """Greedy NMS with torchvision.ops.nms semantics (keep indices sorted by decreasing score;
    suppress IoU > threshold).  Runs on whatever device the boxes are on; O(n^2) on <= 175 boxes/sample."""
    n = boxes.shape[0]
    if n == 0:
        return torch.empty((0,), dtype=torch.int64, device=boxes.device)
    order = torch.argsort(scores, descending=True, stable=True)
    b = boxes[order]
    area = (b[:, 2] - b[:, 0]) * (b[:, 3] - b[:, 1])
    lt = torch.max(b[:, None, :2], b[None, :, :2])
    rb = torch.min(b[:, None, 2:], b[None, :, 2:])
    wh = (rb - lt).clamp(min=0)
    inter = wh[..., 0] * wh[..., 1]
    iou = inter / (area[:, None] + area[None, :] - inter)
    over = (iou > iou_threshold).cpu()
    keep = []
    suppressed = torch.zeros(n, dtype=torch.bool)
    for i in range(n):
        if suppressed[i]:
            continue
        keep.append(i)
        suppressed |= over[i]
    return order[torch.as_tensor(keep, dtype=torch.int64, device=boxes.device)]


def batched_nms_coordinate_trick(boxes, scores, idxs, iou_threshold, width, height):
    if boxes.numel() == 0:
        return torch.empty((0,), dtype=torch.int64, device=boxes.device)
    max_dim = max([width, height])
    offsets = idxs * float(max_dim + 1)
    return nms(boxes + offsets[:, None], scores, iou_threshold)


def _empty(device):
    return {"boxes": torch.zeros(0, 4, dtype=torch.float32, device=device),
            "scores": torch.zeros(0, dtype=torch.float, device=device),
            "labels": torch.zeros(0, dtype=torch.long, device=device)}


def postprocess_network_output(prediction, num_classes, conf_thre=0.01, nms_thre=0.65, height=640, width=640,
                               filtering=True):
    prediction[..., :2] -= prediction[..., 2:4] / 2  # cxcywh -> xywh
    prediction[..., 2:4] += prediction[..., :2]
    output = []
    for image_pred in prediction:
        if len(image_pred) == 0:
            output.append(_empty(prediction.device))
            continue
        class_conf, class_pred = torch.max(image_pred[:, 5:5 + num_classes], 1, keepdim=True)
        image_pred[:, 4:5] *= class_conf
        conf_mask = (image_pred[:, 4] * class_conf.squeeze() >= conf_thre).squeeze()
        detections = torch.cat((image_pred[:, :5], class_pred), 1)
        if filtering:
            detections = detections[conf_mask]
        if len(detections) == 0:
            output.append(_empty(prediction.device))
            continue
        keep = batched_nms_coordinate_trick(detections[:, :4], detections[:, 4], detections[:, 5], nms_thre,
                                            width=width, height=height)
        if filtering:
            detections = detections[keep]
        output.append({"boxes": detections[:, :4], "scores": detections[:, 4], "labels": detections[:, -1].long()})
    return output


def convert_to_evaluation_format(data):
    targets = []
    bb = data.bbox
    bidx = getattr(data, "bbox_batch", torch.zeros(len(bb), dtype=torch.long, device=bb.device))
    for i in range(data.num_graphs):
        bbox = bb[bidx == i].clone()
        bbox[:, 2:4] += bbox[:, :2]
        targets.append({"boxes": bbox[:, :4], "labels": bbox[:, 4].long()})
    return targets
