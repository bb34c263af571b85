"""Shared pieces of the test scripts: flags, the (synthetic) dataset / sharded loader, model + EMA set-up exactly as
``scripts/run_test.py:52-59`` does it, and the record writer of ``scripts/run_test_interframe.py:21-45``."""
import argparse
import os
import sys
from pathlib import Path

import numpy as np
import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from dagr import parallel                                        # noqa: E402
from dagr.data import DataLoader                                 # noqa: E402
from dagr.data.augment import Augmentations                      # noqa: E402
from dagr.data.synthetic_data import SyntheticObjects, SyntheticWindows   # noqa: E402
from dagr.model.networks.dagr import DAGR                        # noqa: E402
from dagr.model.networks.ema import ModelEMA                     # noqa: E402
from dagr.utils.args import MODEL_CONFIGS, model_args            # noqa: E402
from dagr.utils.buffers import detections_to_records             # noqa: E402
from dagr.utils.testing_weights import randomize_                # noqa: E402


def flags(description, extra=None):
    p = argparse.ArgumentParser(description=description, formatter_class=argparse.RawDescriptionHelpFormatter)
    p.add_argument("--config", default="dagr-s", choices=sorted(MODEL_CONFIGS))
    p.add_argument("--checkpoint", type=Path, default=None, help="torch.load(path)['ema'] -> ema.ema (strict)")
    p.add_argument("--output_directory", type=Path, default=Path("run_test_out"))
    p.add_argument("--batch_size", type=int, default=8)
    p.add_argument("--windows", type=int, default=32)
    p.add_argument("--events_per_window", type=int, default=50000)
    p.add_argument("--width", type=int, default=640)
    p.add_argument("--height", type=int, default=480)
    p.add_argument("--stream", default="uniform", choices=["uniform", "edges"])
    p.add_argument("--dataset_directory", type=Path, default=None,
                   help="DSEC root (run_test.py:45: DSEC(root, 'test', transform_testing, ...)); needs dsec-det + h5py + "
                        "hdf5plugin.  Default: the synthetic event stream with the DSEC sample contract")
    p.add_argument("--split", default="test")
    p.add_argument("--no_eval", action="store_true", help="utils/args.py:62: no ground truth is loaded / scored")
    p.add_argument("--labelled", action="store_true",
                   help="synthetic windows WITH boxes (dagr/data/synthetic_data.py:SyntheticObjects): the run is scored "
                        "(COCO-protocol mAP of the whole run, also when it is sharded over several GPUs)")
    p.add_argument("--use_image", action="store_true")
    p.add_argument("--img_net", default="resnet50")
    if extra:
        extra(p)
    return p


def distributed():
    """(world, rank, device) of a ``torch.distributed.run`` launch (one process per GPU); backend "nccl" = RCCL."""
    world, rank = int(os.environ.get("WORLD_SIZE", "1")), int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    if torch.cuda.is_available():
        torch.cuda.set_device(local_rank)
        dev = torch.device("cuda", local_rank)
    else:
        dev = torch.device("cpu")
    # a launcher's rendezvous is honoured at every world size (one rank under ``torch.distributed.run`` is a one-rank RCCL
    # group: the gathers at the end of the run are then the collectives an 8-rank run makes)
    launched = all(k in os.environ for k in ("RANK", "WORLD_SIZE", "MASTER_ADDR", "MASTER_PORT"))
    if (world > 1 or launched) and not torch.distributed.is_initialized():
        torch.distributed.init_process_group("nccl" if dev.type == "cuda" else "gloo",
                                             **({"device_id": dev} if dev.type == "cuda" else {}))
        global _OWNS_GROUP
        _OWNS_GROUP = True
    return world, rank, dev


_OWNS_GROUP = False


def finish(world):
    """End of a script: ranks leave together; the process group goes if this script made it (or the run is sharded)."""
    global _OWNS_GROUP
    if torch.distributed.is_available() and torch.distributed.is_initialized() and (world > 1 or _OWNS_GROUP):
        torch.distributed.barrier()
        torch.distributed.destroy_process_group()
        _OWNS_GROUP = False


def dataset_and_loader(a, world, rank):
    """The dataset and THIS rank's loader: batch k holds windows [k*B, (k+1)*B) (drop_last=True, run_test.py:48) and
    goes to rank k mod G -- independent windows, no collective on the data path."""
    if getattr(a, "dataset_directory", None) is not None:
        # run_test.py:43 / run_test_interframe.py:66-68: the DSEC test split, boxes of at least 15 px diagonal and 10 px
        # height; run_test.py scores every frame pair (no_eval left at its default False), the interframe script restricts
        # itself to perfect tracks and passes --no_eval through
        from dagr.data.dsec_data import DSEC
        interframe = hasattr(a, "num_interframe_steps")
        ds = DSEC(a.dataset_directory, a.split, Augmentations.transform_testing, debug=False, min_bbox_diag=15,
                  min_bbox_height=10, only_perfect_tracks=interframe, no_eval=bool(a.no_eval) if interframe else False)
    elif getattr(a, "labelled", False):
        ds = SyntheticObjects(a.windows, a.events_per_window, a.width, a.height, transform=Augmentations.transform_testing,
                              use_image=a.use_image)
    else:
        ds = SyntheticWindows(a.windows, a.events_per_window, a.width, a.height, a.stream, a.use_image,
                              transform=Augmentations.transform_testing)
    n_batches = len(ds) // a.batch_size
    loader = DataLoader(ds, follow_batch=["bbox", "bbox0"], batch_size=a.batch_size, shuffle=False, drop_last=True,
                        batches=parallel.shard_indices(n_batches, rank, world))
    return ds, loader


def build_model(a, ds, dev):
    """run_test.py:52-59: DAGR(args, height, width).cuda() -> ModelEMA -> checkpoint['ema'] (strict) -> cache_luts."""
    args = model_args(a.config, batch_size=a.batch_size, use_image=a.use_image, img_net=a.img_net)
    model = DAGR(args, height=ds.height, width=ds.width)
    if a.checkpoint is None:
        model = randomize_(model, seed=0)
    model = model.to(dev)
    ema = ModelEMA(model)
    if a.checkpoint is not None:
        ema.ema.load_state_dict(torch.load(a.checkpoint, map_location=dev)["ema"])
    else:
        ema.ema.load_state_dict(model.state_dict())
    ema.ema.cache_luts(radius=args.radius, height=ds.height, width=ds.width)
    return args, ema.ema


def is_labelled(a):
    """Whether the run is scored: DSEC with ground truth (run_test.py:43: no_eval stays False), or labelled synthetic data."""
    return (a.dataset_directory is not None or bool(getattr(a, "labelled", False))) and not a.no_eval


def save_metrics(metrics, output_directory, rank, name="metrics.json"):
    """The run's metrics (ONE set for the whole run: ``DetectionBuffer.compute`` gathers the shards) -> rank 0's file."""
    if metrics is None or rank != 0:
        return
    import json
    output_directory.mkdir(parents=True, exist_ok=True)
    with open(output_directory / name, "w") as f:
        json.dump(metrics, f)


def sequence_names(ds):
    """Every sequence name of the dataset, sorted: the table all ranks share, so that a sequence travels through the
    gather as its index and comes out as its own name (``save_detections`` writes one ``detections_{sequence}.npy`` per
    sequence string, run_test_interframe.py:34-45)."""
    if hasattr(ds, "sequence_names"):
        return sorted(ds.sequence_names())
    return sorted(folder.name for folder in ds.dataset.subsequence_directories)      # DSEC (dsec_data.py:144-150)


def detection_rows(detections, device, names):
    """Per-window detection dicts ({boxes, scores, labels, sequence, t, [window]}) -> float rows
    (index of the sequence in `names`, t, x1, y1, x2, y2, score, label) for the gather."""
    index = {n: i for i, n in enumerate(names)}
    rows = []
    for d in detections:
        n = len(d["boxes"])
        if n:
            seq = float(index[str(d["sequence"])])
            rows.append(np.concatenate([np.full((n, 1), seq), np.full((n, 1), float(d["t"])), d["boxes"],
                                        d["scores"].reshape(-1, 1), d["labels"].reshape(-1, 1).astype(np.float64)], 1))
    arr = np.concatenate(rows, 0) if rows else np.zeros((0, 8))
    return torch.from_numpy(arr).to(torch.float64).to(device)


def gather_and_save(rows, output_directory, rank, names):
    """One gather of the run's detections (variable length), then rank 0 writes ``detections_{sequence}.npy`` per sequence
    name with the record layout of run_test_interframe.py:21-45, sorted by time."""
    allrows = parallel.gather_detections(rows).cpu().numpy()
    if rank != 0:
        return None
    output_directory.mkdir(parents=True, exist_ok=True)
    files = {}
    for seq in np.unique(allrows[:, 0]) if len(allrows) else []:
        r = allrows[allrows[:, 0] == seq]
        r = r[np.argsort(r[:, 1], kind="stable")]
        rec = detections_to_records(dict(boxes=r[:, 2:6], scores=r[:, 6], labels=r[:, 7]), r[:, 1].astype(np.uint64))
        path = output_directory / f"detections_{names[int(seq)]}.npy"
        np.save(path, rec)
        files[path.name] = len(rec)
    return files
