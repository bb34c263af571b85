"""Training losses of the detection head: what ``GNNHead`` inherits from ``yolox.models.YOLOXHead`` (third-party,
pinned by the reference at yolox@618fd8c0, ``download_and_install_dependencies.sh:13-15``; not vendored, absent here --
**parity unpinned**: the functions below restate the published YOLOX algorithm and are anchored on the reference's call
sites, ``dagr.py:238-282`` (``get_losses`` on the hybrid / image outputs), ``:292-297`` (``get_output_and_grid``) and
``:168-171`` (``use_l1 = False``, ``IOUloss(reduction="none")``, ``BCEWithLogitsLoss(reduction="none")``).

Pieces (all plain torch on <= 175 anchors per image -- host-side glue next to the HIP path, differentiable w.r.t. the maps):
  * ``output_and_grid``   : [B, 5+C, h, w] raw map -> [B, h*w, 5+C] with xy decoded ((xy + grid) * stride) and
                            wh = exp(wh) * stride, plus the cell grid
  * ``simota_assign``     : SimOTA label assignment of one image, ``simota_assign_batch`` the same for all images at
                            once over the padded label rows (candidate anchors = centre inside a box OR inside the
                            2.5-stride square around its centre; cost = BCE(sqrt(cls*obj), one-hot) + 3 * -log(IoU) +
                            1e5 * [not in box AND centre]; dynamic k = clamp(sum of the 10 best IoUs, 1); conflicts go
                            to the cheapest ground truth)
  * ``detection_losses``  : 5 * (1 - IoU^2) on the matched anchors + BCE objectness on all anchors + BCE class (IoU-
                            weighted one-hot) on the matched anchors, each summed and divided by the number of matched
                            anchors; returns the reference's 6-tuple (total, iou, obj, cls, l1 = 0, matched / ground truths)
"""
import torch
import torch.nn.functional as F

CENTER_RADIUS = 2.5
N_CANDIDATE_K = 10
REG_WEIGHT = 5.0


def output_and_grid(output, stride):
    """YOLOXHead.get_output_and_grid for one scale (n_anchors = 1)."""
    B, n_ch, h, w = output.shape
    yv, xv = torch.meshgrid(torch.arange(h, device=output.device), torch.arange(w, device=output.device), indexing="ij")
    grid = torch.stack((xv, yv), 2).view(1, h * w, 2).to(output.dtype)
    out = output.flatten(start_dim=2).permute(0, 2, 1)                    # [B, h*w, 5+C]
    xy = (out[..., :2] + grid) * stride
    wh = torch.exp(out[..., 2:4]) * stride
    return torch.cat([xy, wh, out[..., 4:]], dim=-1), grid


def pairwise_iou_cxcywh(a, b):
    """IoU matrix [len(a), len(b)] of boxes given as (cx, cy, w, h)."""
    a_lo, a_hi = a[:, None, :2] - a[:, None, 2:] / 2, a[:, None, :2] + a[:, None, 2:] / 2
    b_lo, b_hi = b[None, :, :2] - b[None, :, 2:] / 2, b[None, :, :2] + b[None, :, 2:] / 2
    lo, hi = torch.max(a_lo, b_lo), torch.min(a_hi, b_hi)
    overlap = (lo < hi).all(dim=2).to(a.dtype)
    inter = (hi - lo).prod(dim=2) * overlap
    return inter / (a[:, 2:].prod(1)[:, None] + b[:, 2:].prod(1)[None, :] - inter)


def iou_loss(pred, target):
    """yolox IOUloss(reduction='none', loss_type='iou'): 1 - IoU^2 per row, boxes as (cx, cy, w, h)."""
    lo = torch.max(pred[:, :2] - pred[:, 2:] / 2, target[:, :2] - target[:, 2:] / 2)
    hi = torch.min(pred[:, :2] + pred[:, 2:] / 2, target[:, :2] + target[:, 2:] / 2)
    # (two-factor products written out: ``prod(dim)``'s backward counts zeros with a host read-back)
    overlap = ((lo[:, 0] < hi[:, 0]) & (lo[:, 1] < hi[:, 1])).to(pred.dtype)
    d = hi - lo
    inter = d[:, 0] * d[:, 1] * overlap
    union = pred[:, 2] * pred[:, 3] + target[:, 2] * target[:, 3] - inter
    iou = inter / (union + 1e-16)
    return 1 - iou ** 2


def candidate_anchors(gt, centers, strides):
    """(candidate [A] bool, in_box_and_center [G, A'] bool over the candidates): YOLOXHead.get_in_boxes_info."""
    cx, cy = centers[:, 0][None, :], centers[:, 1][None, :]
    gx, gy, gw, gh = (gt[:, k][:, None] for k in range(4))
    in_box = torch.stack([cx - (gx - gw / 2), cy - (gy - gh / 2), (gx + gw / 2) - cx, (gy + gh / 2) - cy], 2).min(2).values > 0
    rad = CENTER_RADIUS * strides[None, :]
    in_ctr = torch.stack([cx - (gx - rad), cy - (gy - rad), (gx + rad) - cx, (gy + rad) - cy], 2).min(2).values > 0
    cand = in_box.any(0) | in_ctr.any(0)
    return cand, (in_box & in_ctr)[:, cand]


@torch.no_grad()
def simota_assign(gt_boxes, gt_classes, boxes, cls_logits, obj_logits, centers, strides, num_classes):
    """SimOTA for one image.  gt_boxes [G, 4] (cx, cy, w, h, pixels), boxes [A, 4] decoded predictions,
    cls_logits [A, C], obj_logits [A, 1], centers [A, 2] anchor-cell centres in pixels, strides [A].
    Returns (fg [A] bool, matched_gt [F] long, matched_iou [F]) with F = fg.sum()."""
    A = boxes.shape[0]
    cand, both = candidate_anchors(gt_boxes, centers, strides)
    idx = cand.nonzero(as_tuple=True)[0]
    fg = torch.zeros(A, dtype=torch.bool, device=boxes.device)
    if idx.numel() == 0:
        return fg, torch.zeros(0, dtype=torch.long, device=boxes.device), boxes.new_zeros(0)
    ious = pairwise_iou_cxcywh(gt_boxes, boxes[idx])                                   # [G, A']
    onehot = F.one_hot(gt_classes.long(), num_classes).to(boxes.dtype)[:, None, :].expand(-1, idx.numel(), -1)
    joint = (cls_logits[idx].float().sigmoid() * obj_logits[idx].float().sigmoid()).sqrt()
    cls_cost = F.binary_cross_entropy(joint[None].expand(gt_boxes.shape[0], -1, -1), onehot, reduction="none").sum(-1)
    cost = cls_cost + 3.0 * -torch.log(ious + 1e-8) + 100000.0 * (~both).to(boxes.dtype)
    # dynamic k per ground truth, then the k cheapest candidates of each (rank < k)
    topk = torch.topk(ious, min(N_CANDIDATE_K, ious.shape[1]), dim=1).values
    dyn_k = topk.sum(1).int().clamp(min=1)
    rank = cost.argsort(dim=1, stable=True).argsort(dim=1, stable=True)
    match = rank < dyn_k[:, None]                                                      # [G, A']
    multi = match.sum(0) > 1
    if multi.any():                                                                    # contested anchors: cheapest gt
        best = cost[:, multi].argmin(dim=0)
        match[:, multi] = False
        match[best, multi.nonzero(as_tuple=True)[0]] = True
    taken = match.any(0)
    fg[idx[taken]] = True
    matched_gt = match[:, taken].to(torch.uint8).argmax(0)
    matched_iou = (match.to(ious.dtype) * ious).sum(0)[taken]
    return fg, matched_gt, matched_iou


@torch.no_grad()
def simota_assign_batch(labels, boxes, cls_logits, obj_logits, centers, strides, num_classes):
    """``simota_assign`` for the whole batch at once over the padded label rows (labels [B, R, 5], a row is a ground truth
    iff it sums to > 0): the same candidate sets, costs, ranks and conflict rule per image -- rows that are not ground
    truths and anchors that are not candidates of their image are masked instead of cut out, so that nothing depends on
    a per-image count and nothing synchronises with the host.  Returns (fg [B, A] bool, matched_gt [B, A] long row index
    of the anchor's ground truth, matched_iou [B, A]); the last two are meaningful where fg."""
    B, R, _ = labels.shape
    A = boxes.shape[1]
    dt = boxes.dtype
    valid = labels.sum(dim=2) > 0                                                      # [B, R]
    gt = labels[..., 1:5]
    cx, cy = centers[:, 0].view(1, 1, A), centers[:, 1].view(1, 1, A)
    gx, gy, gw, gh = (gt[..., k].unsqueeze(2) for k in range(4))
    in_box = torch.stack([cx - (gx - gw / 2), cy - (gy - gh / 2), (gx + gw / 2) - cx, (gy + gh / 2) - cy], 3).min(3).values > 0
    rad = CENTER_RADIUS * strides.view(1, 1, A)
    in_ctr = torch.stack([cx - (gx - rad), cy - (gy - rad), (gx + rad) - cx, (gy + rad) - cy], 3).min(3).values > 0
    v3 = valid.unsqueeze(2)
    cand = ((in_box | in_ctr) & v3).any(1)                                             # [B, A]: candidates of the image
    both = in_box & in_ctr
    # IoU of every (row, anchor) pair, as pairwise_iou_cxcywh
    a_lo, a_hi = (gt[..., :2] - gt[..., 2:] / 2).unsqueeze(2), (gt[..., :2] + gt[..., 2:] / 2).unsqueeze(2)
    b_lo, b_hi = (boxes[..., :2] - boxes[..., 2:] / 2).unsqueeze(1), (boxes[..., :2] + boxes[..., 2:] / 2).unsqueeze(1)
    lo, hi = torch.max(a_lo, b_lo), torch.min(a_hi, b_hi)
    inter = (hi - lo).prod(dim=3) * (lo < hi).all(dim=3).to(dt)
    ious = inter / (gt[..., 2:].prod(2).unsqueeze(2) + boxes[..., 2:].prod(2).unsqueeze(1) - inter)
    joint = (cls_logits.float().sigmoid() * obj_logits.float().sigmoid()).sqrt()      # [B, A, C]
    onehot = F.one_hot(labels[..., 0].long().clamp(0, num_classes - 1), num_classes).to(dt)
    cls_cost = F.binary_cross_entropy(joint.unsqueeze(1).expand(-1, R, -1, -1), onehot.unsqueeze(2).expand(-1, -1, A, -1),
                                      reduction="none").sum(-1)
    cost = cls_cost + 3.0 * -torch.log(ious + 1e-8) + 100000.0 * (~both).to(dt)
    live = v3 & cand.unsqueeze(1)                                                      # pairs that exist in the per-image form
    cost = torch.where(live, cost, torch.full_like(cost, float("inf")))
    ious = torch.where(live, ious, torch.zeros_like(ious))
    dyn_k = torch.topk(ious, min(N_CANDIDATE_K, A), dim=2).values.sum(2).int().clamp(min=1)
    rank = cost.argsort(dim=2, stable=True).argsort(dim=2, stable=True)
    match = (rank < dyn_k.unsqueeze(2)) & live                                         # [B, R, A]
    multi = match.sum(1) > 1                                                           # contested anchors: cheapest gt
    best = F.one_hot(cost.argmin(dim=1), R).permute(0, 2, 1).bool()
    match = torch.where(multi.unsqueeze(1), best, match)
    fg = match.any(1)
    matched_gt = match.to(torch.uint8).argmax(1)
    matched_iou = (match.to(dt) * ious).sum(1)
    return fg, matched_gt, matched_iou


def detection_losses(labels, outputs, grids, strides, num_classes):
    """YOLOXHead.get_losses with use_l1 = False (dagr.py:168).  labels [B, 100, 5] = (class, cx, cy, w, h) rows, zero
    padded (``convert_to_training_format``); outputs [B, A, 5+C] from ``output_and_grid`` concatenated over the scales;
    grids: list of [1, A_k, 2]; strides: per-scale stride.  One batched assignment, no host synchronisation (the counts
    in the result are tensors)."""
    B, A, _ = outputs.shape
    dev, dt = outputs.device, outputs.dtype
    grid = torch.cat(grids, 1)[0]
    stride = torch.cat([torch.full((g.shape[1],), float(s), device=dev, dtype=dt) for g, s in zip(grids, strides)])
    centers = (grid + 0.5) * stride[:, None]
    box, obj, cls = outputs[..., :4], outputs[..., 4:5], outputs[..., 5:]
    num_gts = (labels.sum(dim=2) > 0).sum().to(dt)
    fg, m_gt, m_iou = simota_assign_batch(labels.to(dt), box.detach(), cls.detach(), obj.detach(), centers, stride,
                                          num_classes)
    fgf = fg.to(dt)
    # targets of every anchor (those of non-foreground anchors are multiplied away below); per-image order of the
    # foreground anchors = row-major order of fg, as the per-image form concatenates them
    rows = torch.gather(labels.to(dt), 1, m_gt.unsqueeze(2).expand(-1, -1, 5))         # [B, A, 5]
    reg_t = rows[..., 1:5]
    cls_t = F.one_hot(rows[..., 0].long().clamp(0, num_classes - 1), num_classes).to(dt) * m_iou.unsqueeze(2)
    num_fg = fgf.sum().clamp(min=1.0)
    # sums over the matched anchors written as masked sums over all anchors (boolean-mask indexing would synchronise with
    # the host, here and again in its backward): an unmatched anchor's boxes are replaced by a unit box on both sides
    # (IoU 1 -> loss exactly 0, and `where` routes its gradient to the constant), its class term is multiplied away
    sel = fg.view(-1, 1)
    unit = torch.ones((1, 4), device=dev, dtype=dt)
    loss_iou = (iou_loss(torch.where(sel, box.reshape(-1, 4), unit), torch.where(sel, reg_t.reshape(-1, 4), unit))
                * fgf.view(-1)).sum() / num_fg
    loss_obj = F.binary_cross_entropy_with_logits(obj.reshape(-1, 1), fgf.view(-1, 1), reduction="none").sum() / num_fg
    loss_cls = (F.binary_cross_entropy_with_logits(cls.reshape(-1, num_classes), cls_t.reshape(-1, num_classes),
                                                   reduction="none") * fgf.view(-1, 1)).sum() / num_fg
    loss_l1 = 0.0
    total = REG_WEIGHT * loss_iou + loss_obj + loss_cls + loss_l1
    return total, REG_WEIGHT * loss_iou, loss_obj, loss_cls, loss_l1, num_fg / num_gts.clamp(min=1.0)
