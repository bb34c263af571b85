"""Window sharding across GPUs (one process per GPU, torch.distributed; backend "nccl" = RCCL on ROCm).

The hot path shards by independent window batches (`reset=True` windows share no state:
``src/dagr/model/networks/dagr.py:74,90``, ``src/dagr/model/layers/ev_tgn.py:47-49``), so there is no
collective on the data path.  The only exchange is the gather of detections at the end of a run
(what ``scripts/run_test*.py`` feed into the mAP buffer on one process): variable-length per rank,
so counts are gathered first, then the padded payload.
"""
import torch
import torch.distributed as dist


def rank_environment(local_rank=None, local_world=None, apply=True):
    """What one process of an N-rank job on one node sets up BEFORE its first kernel, so that eight ranks starting together
    do not step on each other (each call is a no-op at N = 1):
      * ``MIOPEN_USER_DB_PATH`` / ``MIOPEN_CUSTOM_CACHE_DIR`` per rank: with ``cudnn.benchmark`` every rank runs MIOpen's
        find at start-up and WRITES its user perf-db; eight processes would share one file otherwise;
      * CPU affinity: rank r gets the r-th contiguous slice of the host cores this process may run on (launch-bound host
        threads of eight ranks do not migrate over each other), and torch's intra-op pool is sized to the slice.
    Reads LOCAL_RANK / LOCAL_WORLD_SIZE (``torch.distributed.run`` sets both).  Returns what it chose."""
    import os
    if local_rank is None:
        local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    if local_world is None:
        local_world = int(os.environ.get("LOCAL_WORLD_SIZE", os.environ.get("WORLD_SIZE", "1")))
    out = dict(local_rank=local_rank, local_world=local_world, miopen_db=None, cores=None)
    if local_world <= 1:
        return out
    base = os.environ.get("DAGR_RANK_CACHE_DIR", os.path.join(os.environ.get("TMPDIR", "/tmp"), "dagr_rank_cache"))
    db = os.path.join(base, f"miopen_rank{local_rank}")
    try:
        avail = sorted(os.sched_getaffinity(0))
    except AttributeError:                       # (not Linux)
        avail = list(range(os.cpu_count() or 1))
    per = max(1, len(avail) // local_world)
    cores = avail[(local_rank * per) % len(avail):][:per] or avail
    out.update(miopen_db=db, cores=cores)
    if apply:
        os.makedirs(db, exist_ok=True)
        os.environ.setdefault("MIOPEN_USER_DB_PATH", db)
        os.environ.setdefault("MIOPEN_CUSTOM_CACHE_DIR", db)
        try:
            os.sched_setaffinity(0, cores)
        except (AttributeError, OSError):
            pass
        torch.set_num_threads(max(1, len(cores)))
        out["miopen_db"] = os.environ["MIOPEN_USER_DB_PATH"]
    return out


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


def gather_evaluation(detections, ground_truth, image_ids, group=None):
    """The run's per-image detections and ground truth from ALL ranks, ordered by global image id, on every rank: what
    the reference's single process hands to its evaluation (``scripts/run_test.py:61-65`` ->
    ``utils/coco_eval.py:64-94``), so that a sharded run prints THE mAP of the run and not one per shard.
    ``detections[i]`` = {boxes [n,4], scores [n], labels [n]}, ``ground_truth[i]`` = {boxes, labels} of image
    ``image_ids[i]``.  One gather (counts + padded payload, ``gather_detections``): per image a header row
    (id, 2, n_gt, n_det), its ground-truth rows (id, 0, box, 0, label) and its detection rows (id, 1, box, score, label),
    float64 (float32 boxes / scores and integer labels travel exactly).  Without a process group: the inputs, sorted."""
    import numpy as np

    def host(v):
        return np.asarray(v.detach().cpu() if torch.is_tensor(v) else v)
    order = sorted(range(len(image_ids)), key=lambda i: int(image_ids[i]))
    if not dist.is_available() or not dist.is_initialized():
        return [detections[i] for i in order], [ground_truth[i] for i in order], [int(image_ids[i]) for i in order]
    # (a one-rank group goes through the collective too: the same code path whatever the world size)
    rows = []
    for i in order:
        d, g, iid = detections[i], ground_truth[i], float(image_ids[i])
        gb, db = host(g["boxes"]).reshape(-1, 4).astype(np.float64), host(d["boxes"]).reshape(-1, 4).astype(np.float64)
        rows.append(np.array([[iid, 2.0, len(gb), len(db), 0, 0, 0, 0]], dtype=np.float64))
        if len(gb):
            rows.append(np.concatenate([np.full((len(gb), 1), iid), np.zeros((len(gb), 1)), gb, np.zeros((len(gb), 1)),
                                        host(g["labels"]).reshape(-1, 1).astype(np.float64)], 1))
        if len(db):
            rows.append(np.concatenate([np.full((len(db), 1), iid), np.ones((len(db), 1)), db,
                                        host(d["scores"]).reshape(-1, 1).astype(np.float64),
                                        host(d["labels"]).reshape(-1, 1).astype(np.float64)], 1))
    mine = torch.from_numpy(np.concatenate(rows, 0) if rows else np.zeros((0, 8)))
    dev = torch.device("cuda", torch.cuda.current_device()) if dist.get_backend(group) == "nccl" else torch.device("cpu")
    allrows = gather_detections(mine.to(dev), group=group).cpu().numpy()
    # one stable sort by (image id, row kind) groups every image's rows (ground truth, detections, header; each kind in
    # the order its rank sent it) -- a mask per image was quadratic in the images of a full test split
    allrows = allrows[np.lexsort((allrows[:, 1], allrows[:, 0]))]
    heads = allrows[allrows[:, 1] == 2.0]
    ids = [int(v) for v in heads[:, 0]]
    if len(set(ids)) != len(ids):
        raise RuntimeError("gather_evaluation: an image id arrived from two ranks")
    dets, gts = [], []
    pos = 0
    for h in heads:                                   # rows of one image: n_gt x kind 0, n_det x kind 1, the header
        n_gt, n_det = int(h[2]), int(h[3])
        g, d = allrows[pos:pos + n_gt], allrows[pos + n_gt:pos + n_gt + n_det]
        pos += n_gt + n_det + 1
        gts.append(dict(boxes=torch.from_numpy(g[:, 2:6].astype(np.float32)), labels=torch.from_numpy(g[:, 7].astype(np.int64))))
        dets.append(dict(boxes=torch.from_numpy(d[:, 2:6].astype(np.float32)),
                         scores=torch.from_numpy(d[:, 6].astype(np.float32)),
                         labels=torch.from_numpy(d[:, 7].astype(np.int64))))
    if pos != len(allrows):
        raise RuntimeError("gather_evaluation: rows without a header")
    return dets, gts, ids


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
