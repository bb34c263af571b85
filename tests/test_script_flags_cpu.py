"""The shipped scripts take the reference's documented command lines verbatim (north_star: "scripts/run_test*.py drop in").
Each of the five command lines of the reference's readme (readme.md:68-75, 107-113, 131-138, 168-171, 180-184) goes through
the parser of the script it belongs to; the namespace must equal what the reference's own ``FLAGS()`` made of it
(tests/golden/ref_py_flags.json <- tests/make_golden_refpy_flags.py, which imports /root/reference/src/dagr/utils/args.py).
Then the namespace IS what the model is built from: every YAML key reaches ``DAGR(args, ...)``."""
import json
import os
import sys

import pytest
import yaml

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "scripts"))
GOLD = json.load(open(os.path.join(ROOT, "tests", "golden", "ref_py_flags.json")))
sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
from make_golden_refpy_flags import README_LINES  # noqa: E402  (the argv lists; the generator's main() is not run)

# keys the scripts add on top of the reference's namespace (synthetic stand-in data / short runs) -- nothing else may differ
EXTENSIONS = {"windows", "events_per_window", "width", "height", "stream", "split", "labelled", "epochs", "samples",
              "val_samples", "max_iters", "resume_checkpoint", "interframe", "real_data"}


def _parse(key, argv):
    script = key.split("@")[0]
    if script in ("run_test.py", "run_test_interframe.py"):
        import _common as C
        return vars(C.flags("", argv))
    import train_ncaltech101 as T
    return vars(T.flags(argv, preset="dsec" if script == "train_dsec.py" else "ncaltech101"))


@pytest.mark.parametrize("key", sorted(README_LINES))
def test_reference_readme_command_line_parses_to_the_reference_namespace(key, capsys):
    mine = _parse(key, list(README_LINES[key]))
    want = GOLD[key]
    for k, v in want.items():
        if k == "config":       # same file name under this repo's config/ (eagr-s-dsec.yaml -> dagr-s-dsec.yaml, see resolve_config)
            assert os.path.basename(str(mine[k])) == os.path.basename(v)
            assert os.path.dirname(str(mine[k])) == os.path.join(ROOT, "config") or str(mine[k]) == v
            continue
        if k == "path":         # dagr-l-ncaltech.yaml's unused `path` key: a placeholder directory in both files
            assert k in mine
            continue
        got = mine[k]
        got = str(got) if not isinstance(got, (int, float, bool, str)) else got
        assert got == v and type(got) is type(v), (key, k, got, v)
    extra = set(mine) - set(want)
    assert extra <= EXTENSIONS, extra
    if "eagr-" in " ".join(README_LINES[key]):
        assert "using" in capsys.readouterr().err          # the alias is announced


def test_shipped_configs_hold_the_reference_schema():
    """config/*.yaml: the five files of the reference's config/, same keys; same values except the two directory placeholders
    (the goldens above pin the values through FLAGS for dagr-s / dagr-l-ncaltech; flags_json of ref_py_functions.npz pins
    n / s / m / l through model_args, tests/test_oracle_refpy.py)."""
    names = sorted(os.listdir(os.path.join(ROOT, "config")))
    assert names == ["dagr-l-dsec.yaml", "dagr-l-ncaltech.yaml", "dagr-m-dsec.yaml", "dagr-n-dsec.yaml", "dagr-s-dsec.yaml"]
    s = yaml.safe_load(open(os.path.join(ROOT, "config", "dagr-s-dsec.yaml")))
    want = GOLD["run_test.py@readme:107-113"]
    cli = {"config", "batch_size", "checkpoint", "dataset_directory", "output_directory", "img_net", "use_image"}
    flag_defaults = {"no_eval", "no_events", "run_test", "pretrain_cnn", "keep_temporal_ordering", "num_interframe_steps"}
    assert set(s) == (set(want) - cli - flag_defaults) | {"batch_size", "dataset_directory", "output_directory", "img_net"}
    for n, w in (("n", 0.25), ("s", 0.5), ("m", 0.75), ("l", 1.0)):
        c = yaml.safe_load(open(os.path.join(ROOT, "config", f"dagr-{n}-dsec.yaml")))
        assert c["net_stem_width"] == w and c["yolo_stem_width"] == w
        assert {k: v for k, v in c.items() if "stem_width" not in k} == {k: v for k, v in s.items() if "stem_width" not in k}


def test_short_names_and_defaults_still_work():
    import _common as C
    a = C.flags("", ["--config", "dagr-l", "--batch_size", "4"])
    assert a.net_stem_width == 1 and a.batch_size == 4 and a.dataset_directory is None and "checkpoint" not in a
    assert os.path.basename(str(a.config)) == "dagr-l-dsec.yaml" and a.windows == 32
    a = C.flags("", [])
    assert os.path.basename(str(a.config)) == "dagr-s-dsec.yaml" and a.batch_size == 64 and a.windows == 256
    import train_ncaltech101 as T
    a = T.flags(["--config", "dagr-s", "--epochs", "2"], preset="ncaltech101")
    assert (a.dataset, a.num_scales, a.net_stem_width, a.tot_num_epochs, a.aug_zoom) == ("ncaltech101", 1, 0.5, 2, 1)
    a = T.flags([], preset="ncaltech101")
    assert os.path.basename(str(a.config)) == "dagr-l-ncaltech.yaml" and a.net_stem_width == 1 and a.exp_name == "train"
    with pytest.raises(FileNotFoundError):
        C.flags("", ["--config", "config/no-such.yaml"])


def test_yaml_keys_reach_the_model(tmp_path):
    """``DAGR(args, ...)`` is built from the parsed namespace: a YAML with different radius / max_neighbors / num_scales /
    pooling_dim_at_output / widths changes the model accordingly (constructor only: no GPU work)."""
    import _common as C
    from dagr_amd.model.networks.dagr import DAGR
    cfg = yaml.safe_load(open(os.path.join(ROOT, "config", "dagr-s-dsec.yaml")))
    cfg.update(radius=0.02, max_neighbors=8, num_scales=1, net_stem_width=0.75, yolo_stem_width=0.75)
    path = tmp_path / "custom.yaml"
    yaml.safe_dump(cfg, open(path, "w"))
    a = C.flags("", ["--config", str(path), "--batch_size", "2"])
    m = DAGR(a, height=215, width=320)
    assert m.backbone.num_scales == 1 and not hasattr(m.head, "stem2")
    assert (m.backbone.events_to_graph.max_neighbors, m.backbone.events_to_graph.radius) == (8, 0.02)
    assert m.backbone.layer5.conv_block1.conv.weight.shape[-1] == int(128 * 0.75)
    b = C.flags("", ["--config", "config/dagr-s-dsec.yaml", "--batch_size", "2"])
    m2 = DAGR(b, height=215, width=320)
    assert m2.backbone.num_scales == 2 and m2.backbone.layer5.conv_block1.conv.weight.shape[-1] == 64
