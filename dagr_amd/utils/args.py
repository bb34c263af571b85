"""Mirror of ``src/dagr/utils/args.py``: same flags (every option ``default=SUPPRESS`` so that the YAML
only fills keys absent from the command line, args.py:104-110).  The model configurations are the YAML files under
``config/`` at the repository root (data: ``dagr-{n,s,m,l}-dsec.yaml``, ``dagr-l-ncaltech.yaml``); ``--config`` takes a
path to one of them exactly as the reference's command lines do (readme.md:68-75,107-113,131-138,168-171,180-184), or a
short name (``dagr-s``) as an alias."""
import argparse
import sys
import types
from pathlib import Path

import yaml

CONFIG_DIR = Path(__file__).resolve().parents[2] / "config"


def _load_yaml(path):
    with Path(path).open() as f:
        return yaml.load(f, Loader=yaml.SafeLoader)


def _model_configs():
    out = {}
    for path in sorted(CONFIG_DIR.glob("dagr-*-dsec.yaml")):
        cfg = _load_yaml(path)
        out[path.name[:-len("-dsec.yaml")]] = dict(net_stem_width=float(cfg["net_stem_width"]),
                                                    yolo_stem_width=float(cfg["yolo_stem_width"]))
    return out


# short name -> the two keys in which config/dagr-{n,s,m,l}-dsec.yaml differ (read from the files)
MODEL_CONFIGS = _model_configs()


def resolve_config(config):
    """``--config`` as the reference's command lines give it (a YAML path, relative to the working directory or to the
    repository root), or a short name (``dagr-s`` -> ``config/dagr-s-dsec.yaml``).  The reference's readme also names
    ``config/eagr-s-dsec.yaml`` (readme.md:122,134), a file its tree does not hold: it resolves to ``dagr-s-dsec.yaml``
    with a printed notice."""
    config = Path(config)
    tried = [config, CONFIG_DIR.parent / config, CONFIG_DIR / config.name]
    name = config.name
    if not name.endswith((".yaml", ".yml")):
        tried += [CONFIG_DIR / f"{name}-dsec.yaml", CONFIG_DIR / f"{name}.yaml"]
    if name.startswith("eagr-"):
        tried.append(CONFIG_DIR / ("dagr-" + name[len("eagr-"):]))
    for path in tried:
        if path.is_file():
            if path.name != name:
                print(f"[dagr] --config {config}: using {path}", file=sys.stderr)
            return path
    raise FileNotFoundError(f"--config {config}: no such file (looked in {', '.join(str(t) for t in tried)})")


def model_args(name="dagr-s", **over):
    """Namespace equivalent to ``FLAGS()`` with ``--config config/<name>-dsec.yaml``: every key of that file."""
    path = name if str(name).endswith((".yaml", ".yml")) else CONFIG_DIR / f"{name}-dsec.yaml"
    cfg = dict(_load_yaml(resolve_config(path)), use_image=False, no_events=False, pretrain_cnn=False,
               keep_temporal_ordering=False)
    cfg.pop("dataset_directory", None)
    cfg.pop("output_directory", None)
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


def FLAGS_PARSER():
    """The parser of ``FLAGS`` (args.py:54-70); the scripts add their synthetic-data options to it."""
    p = BASE_FLAGS()
    S = argparse.SUPPRESS
    for name, typ in (("aug_trans", float), ("aug_zoom", float), ("exp_name", str), ("l_r", float),
                      ("tot_num_epochs", int)):
        p.add_argument("--" + name, default=S, type=typ)
    p.add_argument("--no_eval", action="store_true")
    p.add_argument("--run_test", action="store_true")
    p.add_argument("--num_interframe_steps", type=int, default=10)
    return p


def FLAGS(argv=None):
    args = FLAGS_PARSER().parse_args(argv)
    if args.config != "":
        args = parse_config(args, args.config)
    args.dataset_directory = Path(args.dataset_directory)
    args.output_directory = Path(args.output_directory)
    if "checkpoint" in args:
        args.checkpoint = Path(args.checkpoint)
    return args


def SCRIPT_FLAGS(argv=None, description=None, default_config="dagr-s-dsec.yaml", extra=None):
    """``FLAGS`` for the shipped scripts: the reference's command lines parse to the reference's namespace (same parser, same
    YAML merge), plus (i) ``--config`` also takes a short name / is looked up under the repository's ``config/``, with
    `default_config` when absent (the reference's default ``../config/detection.yaml`` exists nowhere), (ii) the options of
    the synthetic stand-in data (`extra(parser)`), used when the dataset's readers cannot run, (iii) ``--dataset_directory``
    may be left out (-> None; the YAML's ``dataset_directory`` is a placeholder and is not used)."""
    p = FLAGS_PARSER()
    p.description = description
    p.formatter_class = argparse.RawDescriptionHelpFormatter
    p.set_defaults(config=None)
    if extra is not None:
        extra(p)
    args = p.parse_args(argv)
    given = set(vars(args))                          # SUPPRESS defaults: present <=> given on the command line
    args.config = resolve_config(args.config if args.config is not None else default_config)
    args = parse_config(args, args.config)
    args.dataset_directory = Path(args.dataset_directory) if "dataset_directory" in given else None
    args.output_directory = Path(args.output_directory)
    if "checkpoint" in given:                        # absent stays absent ("checkpoint" in args, run_test.py:56)
        args.checkpoint = Path(args.checkpoint)
    return args
