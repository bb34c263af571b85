"""Mirror of ``src/dagr/utils/args.py``: same flags (every option ``default=SUPPRESS`` so that the YAML
only fills keys absent from the command line, args.py:104-110) and the shipped model configs as data."""
import argparse
import types
from pathlib import Path

import yaml

# config/dagr-{n,s,m,l}-dsec.yaml differ only in net_stem_width / yolo_stem_width (lines 23-24)
MODEL_CONFIGS = {
    "dagr-n": dict(net_stem_width=0.25, yolo_stem_width=0.25),
    "dagr-s": dict(net_stem_width=0.5, yolo_stem_width=0.5),
    "dagr-m": dict(net_stem_width=0.75, yolo_stem_width=0.75),
    "dagr-l": dict(net_stem_width=1.0, yolo_stem_width=1.0),
}
BASE_CONFIG = dict(task="detection", dataset="dsec", radius=0.01, time_window_us=1000000, max_neighbors=16,
                   n_nodes=50000, batch_size=64, activation="relu", edge_attr_dim=2, aggr="sum", kernel_size=5,
                   pooling_aggr="max", base_width=0.5, after_pool_width=1, num_scales=2, weight_decay=0.00001,
                   clip=0.1, pooling_dim_at_output="5x7", aug_trans=0.1, aug_zoom=1.5, aug_p_flip=0.5,
                   img_net="resnet18", l_r=0.0002, tot_num_epochs=801)


def model_args(name="dagr-s", **over):
    """Namespace equivalent to ``FLAGS()`` with ``--config config/<name>-dsec.yaml``."""
    cfg = dict(BASE_CONFIG, use_image=False, no_events=False, pretrain_cnn=False, keep_temporal_ordering=False)
    cfg.update(MODEL_CONFIGS[name])
    cfg.update(over)
    return types.SimpleNamespace(**cfg)


def BASE_FLAGS():
    S = argparse.SUPPRESS
    p = argparse.ArgumentParser("")
    p.add_argument("--dataset_directory", type=Path, default=S)
    p.add_argument("--output_directory", type=Path, default=S)
    p.add_argument("--checkpoint", type=Path, default=S)
    p.add_argument("--img_net", default=S, type=str)
    p.add_argument("--img_net_checkpoint", type=Path, default=S)
    p.add_argument("--config", type=Path, default="../config/detection.yaml")
    for flag in ("--use_image", "--no_events", "--pretrain_cnn", "--keep_temporal_ordering"):
        p.add_argument(flag, action="store_true")
    for name, typ in (("task", str), ("dataset", str), ("radius", float), ("time_window_us", int),
                      ("max_neighbors", int), ("n_nodes", int), ("batch_size", int), ("activation", str),
                      ("edge_attr_dim", int), ("aggr", str), ("kernel_size", int), ("pooling_aggr", str),
                      ("base_width", float), ("after_pool_width", float), ("net_stem_width", float),
                      ("yolo_stem_width", float), ("num_scales", int), ("weight_decay", float), ("clip", float),
                      ("aug_p_flip", float)):
        p.add_argument("--" + name, default=S, type=typ)
    p.add_argument("--pooling_dim_at_output", default=S)
    return p


def parse_config(args, config):
    with Path(config).open() as f:
        cfg = yaml.load(f, Loader=yaml.SafeLoader)
    for k, v in cfg.items():
        if k not in args:
            setattr(args, k, v)
    return args


def FLAGS(argv=None):
    p = BASE_FLAGS()
    S = argparse.SUPPRESS
    for name, typ in (("aug_trans", float), ("aug_zoom", float), ("exp_name", str), ("l_r", float),
                      ("tot_num_epochs", int)):
        p.add_argument("--" + name, default=S, type=typ)
    p.add_argument("--no_eval", action="store_true")
    p.add_argument("--run_test", action="store_true")
    p.add_argument("--num_interframe_steps", type=int, default=10)
    args = p.parse_args(argv)
    if args.config != "":
        args = parse_config(args, args.config)
    args.dataset_directory = Path(args.dataset_directory)
    args.output_directory = Path(args.output_directory)
    if "checkpoint" in args:
        args.checkpoint = Path(args.checkpoint)
    return args
