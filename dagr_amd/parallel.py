"""Window sharding across GPUs (one process per GPU, torch.distributed; backend "nccl" = RCCL on ROCm).

The hot path shards by independent window batches (`reset=True` windows share no state:
``src/dagr/model/networks/dagr.py:74,90``, ``src/dagr/model/layers/ev_tgn.py:47-49``), so there is no
collective on the data path.  The only exchange is the gather of detections at the end of a run
(what ``scripts/run_test*.py`` feed into the mAP buffer on one process): variable-length per rank,
so counts are gathered first, then the padded payload.
"""
import torch
import torch.distributed as dist


def shard_indices(num_items, rank, world_size):
    """Window w -> rank w mod G; returns this rank's window indices in order."""
    return list(range(rank, num_items, world_size))


def gather_detections(rows, group=None):
    """rows: float tensor [n_i, C] of this rank's detections (any n_i >= 0, same C everywhere),
    e.g. columns (window_id, x1, y1, x2, y2, score, label).  Returns the concatenation over ranks in
    rank order on every rank.  Two collectives: counts (all_gather) + padded payload (all_gather)."""
    if not dist.is_available() or not dist.is_initialized():
        return rows
    world = dist.get_world_size(group)
    n = torch.tensor([rows.shape[0]], dtype=torch.int64, device=rows.device)
    counts = [torch.zeros_like(n) for _ in range(world)]
    dist.all_gather(counts, n, group=group)
    counts = [int(c.item()) for c in counts]
    nmax = max(counts + [1])
    padded = torch.zeros((nmax, rows.shape[1]), dtype=rows.dtype, device=rows.device)
    padded[:rows.shape[0]] = rows
    parts = [torch.zeros_like(padded) for _ in range(world)]
    dist.all_gather(parts, padded, group=group)
    return torch.cat([p[:c] for p, c in zip(parts, counts)], dim=0)


def restore_window_order(rows, window_col=0):
    """Detections gathered rank by rank -> sorted by window id (stable), the order
    ``run_test_interframe.py:34-45`` writes them in."""
    order = torch.argsort(rows[:, window_col], stable=True)
    return rows[order]


def data_parallel(model, device, bucket_cap_mb=16):
    """Data-parallel training replica (BASELINE config 5; the reference trains on one GPU, train_ncaltech101.py:130):
    ``DistributedDataParallel`` over the default group -- gradients averaged by bucketed all-reduce (RCCL over xGMI on
    GPUs) that overlaps the backward pass.  Choices that follow from this model and this fabric:
      * the dense ``YOLOXHead`` module lists inside ``GNNHead`` (``stems``, ``cls_convs`` ... ``obj_preds``) exist only for
        checkpoint compatibility and never run (dagr.py:137): they are frozen here so that the reducer does not wait for
        gradients that never arrive (no ``find_unused_parameters`` graph walk per step);
      * 16-MB buckets: dagr-l + head is O(10^7) parameters (~40 MB fp32), so three or four buckets let the first
        all-reduce start while layer 2's backward still runs; a ring all-reduce over xGMI is bound by one ~153 GB/s link,
        i.e. ~0.2 ms per bucket at 8 GPUs -- far below a step, so finer buckets would only add launch latency;
      * ``broadcast_buffers=False``: BatchNorm running statistics stay per replica, as in the reference (no SyncBN)."""
    head = getattr(model, "head", None)
    if head is not None:
        for name in ("stems", "cls_convs", "reg_convs", "cls_preds", "reg_preds", "obj_preds"):
            sub = getattr(head, name, None)
            if sub is not None:
                sub.requires_grad_(False)
    kw = dict(device_ids=[device.index], output_device=device.index) if device.type == "cuda" else {}
    return torch.nn.parallel.DistributedDataParallel(model, broadcast_buffers=False, bucket_cap_mb=bucket_cap_mb,
                                                     gradient_as_bucket_view=True, **kw)
