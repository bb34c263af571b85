#!/usr/bin/env python
"""Twin of the reference's ``scripts/train_ncaltech101.py`` (:41-182) on this stack, same flow through the same import
paths (``dagr.*`` resolves to ``dagr_amd``): datasets with the training / testing augmentations -> loaders ->
``DAGR(args, height, width).cuda()`` -> ``ModelEMA`` -> AdamW at ``l_r * sqrt(batch / 64)`` -> ``LambdaLR(LRSchedule)`` ->
``Checkpointer`` -> per iteration: ``format_data`` -> ``model(data)`` (loss dict) -> backward -> ``clip_grad_value_`` ->
NaN-gradient fix -> optimizer / scheduler step -> ``ema.update`` -> every third epoch a validation pass on ``ema.ema``.

BASELINE config 5 runs it data-parallel: under ``torch.distributed.run`` (one process per GPU) every rank holds a
replica, takes its slice of each global batch (``DataLoader(shard=(rank, world))``: same seeded permutation everywhere),
and the gradients are averaged by ``DistributedDataParallel`` over RCCL -- bucketed all-reduce overlapped with the
backward pass; BatchNorm statistics stay per replica (the reference has no SyncBN).  ``--batch_size`` is the GLOBAL batch
(the learning-rate rule above refers to it).

The N-Caltech101 reader needs h5py (absent here): with ``--dataset_directory`` it is used, without it the run is on
``SyntheticObjects`` (labelled synthetic rectangles, 240 x 180).  The validation pass computes the COCO-protocol mAP
(``dagr/utils/coco_eval.py``) of ``ema.ema``'s detections and keeps the best checkpoint, as the reference does.

The reference's command line is taken as it is (readme.md:168-171); ``--config`` also accepts a short name:

  python scripts/train_ncaltech101.py --config config/dagr-l-ncaltech.yaml --exp_name ncaltech_l \
         --dataset_directory $DAGR_DIR/data/ --output_directory $DAGR_DIR/logs/
  python scripts/train_ncaltech101.py --epochs 3 --samples 256 --batch_size 16
  python -m torch.distributed.run --nproc-per-node 8 --master-addr 127.0.0.1 scripts/train_ncaltech101.py --batch_size 64
"""
import argparse
import os
import random
import sys
import time
from pathlib import Path

import numpy as np
import torch

import _common as C
from dagr import parallel
from dagr.data import DataLoader
from dagr.data.augment import Augmentations
from dagr.data.synthetic_data import SyntheticObjects
from dagr.model.networks.dagr import DAGR
from dagr.model.networks.ema import ModelEMA
from dagr.utils.args import MODEL_CONFIGS
from dagr.utils.buffers import format_data
from dagr.utils.learning_rate_scheduler import LRSchedule
from dagr.utils.logging import Checkpointer, log_hparams, set_up_logging_directory


def flags(argv=None, preset="ncaltech101"):
    """The reference's ``FLAGS()`` (readme.md:168-171 / :180-184 parse as they are) plus the options of the synthetic
    stand-in data and of short runs.  A short ``--config`` name (``dagr-s``) under the N-Caltech101 preset means
    ``config/dagr-l-ncaltech.yaml`` at that model's widths."""
    def more(p):
        g = p.add_argument_group("short runs / synthetic stand-in data")
        g.add_argument("--epochs", type=int, default=None, help="alias of --tot_num_epochs")
        g.add_argument("--samples", type=int, default=512, help="synthetic training samples (no dataset to read)")
        g.add_argument("--val_samples", type=int, default=64)
        g.add_argument("--max_iters", type=int, default=-1, help="stop after this many iterations (smoke runs)")
        g.add_argument("--resume_checkpoint", type=Path, default=None)
    argv = list(sys.argv[1:] if argv is None else argv)
    widths = {}
    if preset == "ncaltech101" and "--config" in argv:
        name = argv[argv.index("--config") + 1]
        if name in MODEL_CONFIGS:
            widths = MODEL_CONFIGS[name]
            argv[argv.index("--config") + 1] = "dagr-l-ncaltech.yaml"
    a = C.flags(__doc__, argv, extra=more,
                default_config="dagr-l-ncaltech.yaml" if preset == "ncaltech101" else "dagr-s-dsec.yaml")
    for k, v in widths.items():
        if f"--{k}" not in argv:
            setattr(a, k, v)
    if a.epochs is not None:
        a.tot_num_epochs = a.epochs
    if "exp_name" not in a:
        a.exp_name = "train"
    return a


def gradients_broken(model):
    return any(p.grad is not None and bool(torch.isnan(p.grad).any()) for p in model.parameters())


def fix_gradients(model):
    """train_ncaltech101.py:36-39: NaN gradient entries become zeros before the step."""
    for p in model.parameters():
        if p.grad is not None:
            torch.nan_to_num_(p.grad, nan=0.0)


def train_epoch(loader, net, module, ema, scheduler, optimizer, clip, dev, log, max_iters=-1):
    net.train()
    for data in loader:
        data = format_data(data.to(dev, non_blocking=True))
        optimizer.zero_grad(set_to_none=True)
        out = net(data)
        loss = out["total_loss"]
        loss.backward()                       # DDP: bucketed gradient all-reduce overlaps this call
        torch.nn.utils.clip_grad_value_(module.parameters(), clip)
        fix_gradients(module)
        optimizer.step()
        scheduler.step()
        ema.update(module)
        log.append({"loss": float(loss.detach()), "lr": float(scheduler.get_last_lr()[-1]),
                    **{k: float(v) for k, v in out.items() if k != "total_loss"}})
        if 0 < max_iters <= len(log):
            return True
    return False


@torch.no_grad()
def validate(loader, net, dev, dry_run_steps=-1, detections=True):
    """``run_test`` of the training script (:76-99): ``ema.ema`` in eval mode through the window engine, detections and
    targets into the mAP buffer under their global image ids; every rank runs its slice of the validation batches and
    ``compute`` gathers the slices (one collective, ``parallel.gather_evaluation``): the same mAP on every rank.  ``detections=False`` (stand-in models without an eval branch): the mean training-mode
    loss of the validation batches instead, reported as a negative "mAP" so that the best-checkpoint logic still works."""
    from dagr.utils.buffers import DetectionBuffer
    if detections:
        net.eval()
        buf = DetectionBuffer(height=loader.dataset.height, width=loader.dataset.width, classes=loader.dataset.classes)
        for i, data in enumerate(loader):
            data = format_data(data.to(dev))
            dets, targets = net(data)
            buf.update(dets, targets, "ncaltech101", data.height[0], data.width[0],
                       image_ids=loader.image_ids(i) if hasattr(loader, "image_ids") else None)
            if 0 < dry_run_steps == i:
                break
        return buf.compute()
    was = net.training
    net.train()
    saved = {k: v.clone() for k, v in net.state_dict().items() if "running_" in k or "num_batches" in k}
    tot, n = 0.0, 0
    for i, data in enumerate(loader):
        tot += float(net(format_data(data.to(dev)))["total_loss"])
        n += 1
        if 0 < dry_run_steps == i:
            break
    net.load_state_dict(saved, strict=False)       # the pass must not move the running statistics
    net.train(was)
    tot_t = torch.tensor([tot, float(n)], dtype=torch.float64, device=dev)
    if torch.distributed.is_available() and torch.distributed.is_initialized():
        torch.distributed.all_reduce(tot_t)
    val = float(tot_t[0] / max(1.0, float(tot_t[1])))
    return {"mAP": -val, "val_loss": val}


def build(a, world, rank, dev, model_factory=None, preset="ncaltech101"):
    per_rank = a.batch_size // world
    if per_rank * world != a.batch_size or per_rank < 1:
        raise ValueError(f"--batch_size {a.batch_size} must be a positive multiple of the {world} ranks")
    # the namespace the model and the augmentations read (every key of the YAML under the command line); its
    # batch_size is this replica's share of the global batch
    args = argparse.Namespace(**dict(vars(a), batch_size=per_rank))
    aug = Augmentations(args)
    real = C.real_data_available(a, "DSEC" if preset == "dsec" else "N-Caltech101")
    if real and preset == "dsec":
        from dagr.data.dsec_data import DSEC                    # train_dsec.py:122,125-128
        root = a.dataset_directory / args.dataset
        train_ds = DSEC(root=root, split="train", transform=aug.transform_training, min_bbox_diag=15, min_bbox_height=10)
        val_ds = DSEC(root=root, split="val", transform=aug.transform_testing, min_bbox_diag=15, min_bbox_height=10)
    elif real:
        from dagr.data.ncaltech101_data import NCaltech101      # train_ncaltech101.py:122-125
        root = a.dataset_directory / args.dataset
        train_ds = NCaltech101(root, "training", aug.transform_training, num_events=args.n_nodes)
        val_ds = NCaltech101(root, "validation", aug.transform_testing, num_events=args.n_nodes)
    else:
        size = dict(width=320, height=215, use_image=a.use_image) if preset == "dsec" else {}
        train_ds = SyntheticObjects(a.samples, min(args.n_nodes, 20000), seed=7, transform=aug.transform_training, **size)
        val_ds = SyntheticObjects(a.val_samples, min(args.n_nodes, 20000), seed=100007, transform=aug.transform_testing,
                                  **size)
    follow = ["bbox", "bbox0"]
    train_loader = DataLoader(train_ds, follow_batch=follow, batch_size=a.batch_size, shuffle=True, drop_last=True,
                              shard=(rank, world), seed=42)
    order = np.random.default_rng(42).permutation(len(val_ds)).tolist()     # :124 (a fixed random order)
    val_loader = DataLoader(val_ds, sampler=order, follow_batch=follow, batch_size=a.batch_size, drop_last=True,
                            shard=(rank, world))
    if model_factory is not None:
        model = model_factory(args, train_ds).to(dev)
    else:
        model = DAGR(args, height=train_ds.height, width=train_ds.width).to(dev)
        model.cache_luts(width=train_ds.width, height=train_ds.height, radius=args.radius)
    return args, train_loader, val_loader, model


def main(argv=None, model_factory=None, preset="ncaltech101"):
    a = flags(argv, preset)
    world, rank, dev = C.distributed()
    seed = 42
    torch.manual_seed(seed)
    np.random.seed(seed)
    random.seed(seed)
    args, train_loader, val_loader, model = build(a, world, rank, dev, model_factory, preset)
    if rank == 0:
        log_hparams(args)
        print(f"Training with {sum(p.numel() for p in model.parameters())} number of parameters.")
    ema = ModelEMA(model)
    # (DAGR_FORCE_DDP=1 wraps a single process too: exercises the reducer on one GPU)
    ddp = world > 1 or (os.environ.get("DAGR_FORCE_DDP") == "1" and torch.distributed.is_initialized())
    net = parallel.data_parallel(model, dev) if ddp else model
    lr = float(args.l_r * np.sqrt(a.batch_size) / np.sqrt(64))                    # :132-133, nominal batch 64
    # (fused: one multi-tensor launch per step instead of a dozen -- the step is bound by the host issuing launches)
    optimizer = torch.optim.AdamW([p for p in model.parameters() if p.requires_grad], lr=lr,
                                  weight_decay=args.weight_decay, fused=(dev.type == "cuda"))
    schedule = LRSchedule(warmup_epochs=.3, num_iters_per_epoch=len(train_loader), tot_num_epochs=args.tot_num_epochs)
    scheduler = torch.optim.lr_scheduler.LambdaLR(optimizer=optimizer, lr_lambda=schedule)
    out_dir = set_up_logging_directory(args.dataset, args.task, a.output_directory, exp_name=a.exp_name)   # :115
    ckpt = Checkpointer(output_directory=out_dir, model=model, optimizer=optimizer, scheduler=scheduler, ema=ema, args=args)
    if model_factory is not None:
        ckpt.mAP_max = float("-inf")        # stand-in models report a negative validation loss in place of mAP
    start_epoch = 0
    if a.resume_checkpoint is not None:
        start_epoch = ckpt.restore_checkpoint(a.resume_checkpoint, best=False)
        if rank == 0:
            print(f"Resume from checkpoint at epoch {start_epoch}")
    log = []
    t0 = time.perf_counter()
    for epoch in range(start_epoch, args.tot_num_epochs):
        stop = train_epoch(train_loader, net, model, ema, scheduler, optimizer, args.clip, dev, log, a.max_iters)
        if rank == 0:
            ckpt.checkpoint(epoch, name="last_model")
            tail = log[-len(train_loader):] or log
            print(f"epoch {epoch}: loss {np.mean([r['loss'] for r in tail]):.4f}  lr {log[-1]['lr']:.3e}  "
                  f"{time.perf_counter() - t0:.1f} s")
        if stop:
            break
        if epoch % 3 > 0:
            continue
        metrics = validate(val_loader, ema.ema if model_factory is None else model, dev, detections=model_factory is None)
        if rank == 0:
            print(f"epoch {epoch}: validation {metrics}")
            ckpt.process(metrics, epoch)
    C.finish(world)
    return out_dir, log


if __name__ == "__main__":
    main()
