"""CPU oracle for the detection post-processing (TEST INFRASTRUCTURE ONLY -- see oracle/__init__.py).

Restates ``postprocess_network_output`` / ``batched_nms_coordinate_trick`` (src/dagr/model/utils.py:25-33,
61-110) with torchvision.ops.nms's published greedy algorithm written out (torchvision is absent here):
sort by descending score, a box is dropped when its IoU with an already kept box exceeds the threshold."""
import torch


def nms(boxes, scores, iou_threshold):
    n = boxes.shape[0]
    if n == 0:
        return torch.empty((0,), dtype=torch.int64)
    order = torch.argsort(scores, descending=True, stable=True)
    b = boxes[order]
    area = (b[:, 2] - b[:, 0]) * (b[:, 3] - b[:, 1])
    keep, suppressed = [], torch.zeros(n, dtype=torch.bool)
    for i in range(n):
        if suppressed[i]:
            continue
        keep.append(i)
        lt = torch.max(b[i, :2], b[:, :2])
        rb = torch.min(b[i, 2:], b[:, 2:])
        wh = (rb - lt).clamp(min=0)
        inter = wh[:, 0] * wh[:, 1]
        iou = inter / (area[i] + area - inter)
        suppressed |= iou > iou_threshold
    return order[torch.as_tensor(keep, dtype=torch.int64)]


def postprocess_network_output(prediction, num_classes, conf_thre=0.01, nms_thre=0.65, height=640, width=640,
                               filtering=True):
    """model/utils.py:61-110, CPU tensors."""
    prediction = prediction.clone()
    prediction[..., :2] -= prediction[..., 2:4] / 2
    prediction[..., 2:4] += prediction[..., :2]
    output = []
    for image_pred in prediction:
        class_conf, class_pred = torch.max(image_pred[:, 5:5 + num_classes], 1, keepdim=True)
        image_pred[:, 4:5] *= class_conf
        conf_mask = (image_pred[:, 4] * class_conf.squeeze(1) >= conf_thre)
        det = torch.cat((image_pred[:, :5], class_pred.float()), 1)
        if filtering:
            det = det[conf_mask]
        if len(det) == 0:
            output.append({"boxes": torch.zeros(0, 4), "scores": torch.zeros(0), "labels": torch.zeros(0, dtype=torch.long)})
            continue
        offsets = det[:, 5] * float(max(width, height) + 1)
        keep = nms(det[:, :4] + offsets[:, None], det[:, 4], nms_thre)
        if filtering:
            det = det[keep]
        output.append({"boxes": det[:, :4], "scores": det[:, 4], "labels": det[:, 5].long()})
    return output
