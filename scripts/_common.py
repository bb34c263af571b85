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
from dagr.utils.args import SCRIPT_FLAGS                         # noqa: E402
from dagr.utils.buffers import detections_to_records             # noqa: E402
from dagr.utils.testing_weights import randomize_                # noqa: E402


def _synthetic_options(p):
    """Options of the synthetic stand-in data (no counterpart in the reference): used when the run has no
    ``--dataset_directory`` or the dataset's readers (dsec-det, h5py, hdf5plugin) are not installed."""
    g = p.add_argument_group("synthetic stand-in data")
    g.add_argument("--windows", type=int, default=None, help="number of 50 ms windows (default: max(32, 4 x batch_size))")
    g.add_argument("--events_per_window", type=int, default=50000)
    g.add_argument("--width", type=int, default=640)
    g.add_argument("--height", type=int, default=480)
    g.add_argument("--stream", default="uniform", choices=["uniform", "edges"])
    g.add_argument("--split", default="test")
    g.add_argument("--labelled", action="store_true",
                   help="synthetic windows WITH boxes (dagr/data/synthetic_data.py:SyntheticObjects): the run is scored "
                        "(COCO-protocol mAP of the whole run, also when it is sharded over several GPUs)")


def flags(description, argv=None, extra=None, default_config="dagr-s-dsec.yaml"):
    """The reference's ``FLAGS()`` (``dagr.utils.args``: same parser, ``--config <yaml>`` merged under the command line --
    run_test.py:31, readme.md:107-113) plus the synthetic-data options.  Returns the namespace the model is built from."""
    def more(p):
        _synthetic_options(p)
        if extra:
            extra(p)
    a = SCRIPT_FLAGS(argv, description=description, default_config=default_config, extra=more)
    if a.windows is None:
        a.windows = max(32, 4 * a.batch_size)
    return a


def real_data_available(a, what="DSEC"):
    """Whether ``--dataset_directory`` can be read here; a printed notice when it cannot (the run then goes over the synthetic
    stand-in stream: same sample contract, same engine path)."""
    if a.dataset_directory is None:
        return False
    missing = []
    for name in (("dsec_det", "h5py", "hdf5plugin") if what == "DSEC" else ("h5py",)):
        try:
            __import__(name)
        except ImportError:
            missing.append(name)
    reason = None
    if missing:
        reason = f"the {what} reader needs {', '.join(missing)} (not installed)"
    elif not Path(a.dataset_directory).exists():
        reason = f"{a.dataset_directory} does not exist"
    if reason is None:
        return True
    if int(os.environ.get("RANK", "0")) == 0:
        print(f"NOTICE: --dataset_directory {a.dataset_directory}: {reason}; running on the SYNTHETIC stand-in data "
              f"instead (no ground truth, no mAP).", flush=True)
    return False


def distributed():
    """(world, rank, device) of a ``torch.distributed.run`` launch (one process per GPU); backend "nccl" = RCCL."""
    world, rank = int(os.environ.get("WORLD_SIZE", "1")), int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    parallel.rank_environment()      # N ranks on a node: a MIOpen perf-db and a slice of the host cores per rank
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
    a.real_data = real_data_available(a)
    if a.real_data:
        # run_test.py:43 / run_test_interframe.py:66-68: the DSEC test split, boxes of at least 15 px diagonal and 10 px
        # height; run_test.py scores every frame pair (no_eval left at its default False), the interframe script restricts
        # itself to perfect tracks and passes --no_eval through
        from dagr.data.dsec_data import DSEC
        interframe = bool(getattr(a, "interframe", False))
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
    """run_test.py:52-59: DAGR(args, height, width).cuda() -> ModelEMA -> checkpoint['ema'] (strict) -> cache_luts, with
    ``args`` = the parsed namespace (every key of the YAML under the command line).  The reference asserts a checkpoint;
    here a run without one keeps seeded random weights, with a printed notice."""
    model = DAGR(a, height=ds.height, width=ds.width)
    checkpoint = a.checkpoint if "checkpoint" in a else None
    if checkpoint is None:
        if int(os.environ.get("RANK", "0")) == 0:
            print("NOTICE: no --checkpoint: the model keeps seeded random weights.", flush=True)
        model = randomize_(model, seed=0)
    model = model.to(dev)
    ema = ModelEMA(model)
    if checkpoint is not None:
        ema.ema.load_state_dict(torch.load(checkpoint, map_location=dev)["ema"])
    else:
        ema.ema.load_state_dict(model.state_dict())
    ema.ema.cache_luts(radius=a.radius, height=ds.height, width=ds.width)
    return a, ema.ema


def logging_dataset(a):
    """First component of the run directory (logging.py:101-110: <output>/<dataset>/<task>/<exp_name>): the YAML's dataset
    when its data is read, "synthetic" when the stand-in stream is."""
    return a.dataset if getattr(a, "real_data", False) else "synthetic"


def is_labelled(a):
    """Whether the run is scored: DSEC with ground truth (run_test.py:43: no_eval stays False), or labelled synthetic data."""
    return (bool(getattr(a, "real_data", False)) or bool(getattr(a, "labelled", False))) and not a.no_eval


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
