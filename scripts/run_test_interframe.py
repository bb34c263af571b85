#!/usr/bin/env python
"""Twin of the reference's ``scripts/run_test_interframe.py`` (:47-90): for ``--num_interframe_steps`` offsets
``n_us in linspace(0, 50000, steps)`` the dataset is truncated to the first ``n_us`` microseconds after each frame
(``dataset.set_num_us``, dsec_data.py:114-115,159-161), every truncated window goes through
``run_test_with_visualization(..., compile_detections=True)``, and all detections are written per sequence sorted by
timestamp (``save_detections``, :34-45).  The (window x offset) grid is embarrassingly parallel: under
``torch.distributed.run`` the window batches of every offset are sharded over the ranks, one gather at the end."""
import time

import numpy as np
import torch

import _common as C
from dagr.utils.logging import log_hparams, set_up_logging_directory
from dagr.utils.testing import run_test_with_visualization


def main(argv=None, model_factory=None):
    a = C.flags(__doc__, argv)            # --num_interframe_steps / --no_eval are FLAGS' own (args.py:64,68)
    a.interframe = True
    world, rank, dev = C.distributed()
    torch.manual_seed(42)
    np.random.seed(42)
    ds, loader = C.dataset_and_loader(a, world, rank)
    args, net = (model_factory or C.build_model)(a, ds, dev)
    out_dir = set_up_logging_directory(C.logging_dataset(a), a.task, a.output_directory,
                                       exp_name=getattr(a, "exp_name", "run_test_interframe"))
    if rank == 0:
        log_hparams(args)
    detections = []
    t0 = time.perf_counter()
    with torch.no_grad():
        for n_us in np.linspace(0, 50000, a.num_interframe_steps):
            loader.dataset.set_num_us(int(n_us))
            labelled = C.is_labelled(a)
            metrics, one_offset = run_test_with_visualization(loader, net, dataset="dsec" if labelled else "synthetic",
                                                              compile_detections=True, no_eval=not labelled)
            if metrics is not None and rank == 0:     # the run's mAP at this offset (all ranks' windows, gathered)
                print(f"Time Window: {int(n_us)} us \t mAP: {metrics.get('mAP')}")
            C.save_metrics(metrics, out_dir, rank, name=f"metrics_{int(n_us):06d}us.json")
            detections.extend(one_offset)
    names = C.sequence_names(ds)
    files = C.gather_and_save(C.detection_rows(detections, dev, names), out_dir, rank, names)
    if rank == 0:
        print(f"{a.num_interframe_steps} offsets x {len(ds) // a.batch_size * a.batch_size} windows on {world} GPU(s) in "
              f"{time.perf_counter() - t0:.2f} s -> {out_dir}: {files}")
    C.finish(world)
    return out_dir


if __name__ == "__main__":
    main()
