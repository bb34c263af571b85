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
