#!/usr/bin/env python
"""Twin of the reference's ``scripts/run_test.py`` (:31-66) on this stack, same flow through the same import paths
(``dagr.*`` resolves to ``dagr_amd``): dataset -> loader -> ``DAGR(args, height, width).cuda()`` -> ``ModelEMA`` ->
checkpoint into ``ema.ema`` (strict) -> ``cache_luts`` -> ``run_test_with_visualization(loader, ema.ema, dataset=...)``.
Under ``torch.distributed.run`` (one process per GPU) the window batches are sharded over the ranks and the detections
gathered once at the end (RCCL).  The DSEC reader's dependencies are absent here, so the dataset is the benchmark's
synthetic event stream with the DSEC sample contract (``dagr/data/synthetic_data.py``); without ``--checkpoint`` the model
keeps seeded random weights.  The synthetic windows carry no boxes, so that run is ``no_eval`` and writes detection records;
with ``--dataset_directory`` (DSEC) or ``--labelled`` (synthetic objects with boxes) the detections are also scored
(COCO-protocol mAP, ``dagr/utils/coco_eval.py``): ONE mAP for the run -- a sharded run gathers detections and ground truth
of all ranks before the evaluation, as the reference's single process sees them (run_test.py:61-65).

The reference's command line is taken as it is (readme.md:107-113); ``--config`` also accepts a short name:

  python scripts/run_test.py --config config/dagr-s-dsec.yaml --use_image --img_net resnet50 \
         --checkpoint data/dagr_s_50.pth --batch_size 8 --dataset_directory $DSEC_ROOT --output_directory $LOG_DIR
  python scripts/run_test.py --config dagr-s --windows 64 --batch_size 8 --output_directory /tmp/out
  python -m torch.distributed.run --nproc-per-node 8 --master-addr 127.0.0.1 scripts/run_test.py ...
"""
import time

import numpy as np
import torch

import _common as C
from dagr.utils.logging import log_hparams, set_up_logging_directory
from dagr.utils.testing import run_test_with_visualization


def main(argv=None, model_factory=None):
    a = C.flags(__doc__, argv)
    world, rank, dev = C.distributed()
    torch.manual_seed(42)
    np.random.seed(42)
    ds, loader = C.dataset_and_loader(a, world, rank)
    args, net = (model_factory or C.build_model)(a, ds, dev)
    out_dir = set_up_logging_directory(C.logging_dataset(a), a.task, a.output_directory,
                                       exp_name=getattr(a, "exp_name", "run_test"))
    if rank == 0:
        log_hparams(args)
    t0 = time.perf_counter()
    labelled = C.is_labelled(a)
    with torch.no_grad():
        metrics, detections = run_test_with_visualization(loader, net, dataset="dsec" if labelled else "synthetic",
                                                          compile_detections=True, no_eval=not labelled)
    if labelled and rank == 0:
        # run_test.py:61-65: ONE set of metrics for the run -- under a process group detections and ground truth of all
        # ranks were gathered before the evaluation (utils/buffers.py:DetectionBuffer.compute)
        print(f"metrics of the run ({world} rank(s)):", metrics)
    C.save_metrics(metrics, out_dir, rank)
    names = C.sequence_names(ds)
    files = C.gather_and_save(C.detection_rows(detections, dev, names), out_dir, rank, names)
    if rank == 0:
        print(f"{len(ds) // a.batch_size * a.batch_size} windows on {world} GPU(s) in {time.perf_counter() - t0:.2f} s "
              f"(incl. synthetic data generation on the host) -> {out_dir}: {files}")
    C.finish(world)
    return out_dir


if __name__ == "__main__":
    main()
