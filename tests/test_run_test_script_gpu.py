"""Script-level drop-in (scripts/run_test.py, scripts/run_test_interframe.py: twins of the reference's scripts of the same
names): tiny synthetic runs write per-sequence detection record files with the layout of run_test_interframe.py:21-45."""
import os
import subprocess
import sys

import numpy as np
import pytest

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
NAMES = ("t", "x", "y", "w", "h", "class_id", "class_confidence")


def _run(script, out, *extra):
    cmd = [sys.executable, os.path.join(ROOT, "scripts", script), "--windows", "6", "--batch_size", "2",
           "--events_per_window", "3000", "--width", "320", "--height", "215", "--stream", "edges",
           "--output_directory", str(out), *extra]
    r = subprocess.run(cmd, capture_output=True, text=True, timeout=600)
    assert r.returncode == 0, r.stderr[-2000:]
    return r.stdout


def test_run_test_script_writes_detection_records(tmp_path):
    out = tmp_path / "out"
    stdout = _run("run_test.py", out)
    rec = np.load(out / "synthetic" / "detection" / "run_test" / "detections_synthetic000.npy")
    assert rec.dtype.names == NAMES
    assert len(rec) > 0 and (np.diff(rec["t"].astype(np.int64)) >= 0).all()                 # sorted by timestamp
    assert set(np.unique(rec["t"])) <= {50000 * (w + 1) for w in range(6)}                   # t1 of the six windows
    assert (rec["w"] > 0).all() and (rec["h"] > 0).all() and (rec["class_confidence"] >= 0.001).all()
    assert "6 windows on 1 GPU(s)" in stdout


def test_run_test_interframe_script(tmp_path):
    out = tmp_path / "out"
    stdout = _run("run_test_interframe.py", out, "--num_interframe_steps", "3")
    rec = np.load(out / "synthetic" / "detection" / "run_test_interframe" / "detections_synthetic000.npy")
    assert rec.dtype.names == NAMES and len(rec) > 0
    # three offsets (0, 25 ms, 50 ms after each frame) x six windows: the timestamps are t0 + n_us
    want = {50000 * w + off for w in range(6) for off in (0, 25000, 50000)}
    assert set(np.unique(rec["t"])) <= want and (np.diff(rec["t"].astype(np.int64)) >= 0).all()
    assert "3 offsets x 6 windows" in stdout


def test_run_test_script_scores_a_labelled_run(tmp_path):
    """``--labelled``: synthetic windows with boxes through the real model -> ONE set of COCO-protocol metrics for the run
    (scripts/run_test.py:61-65; under a process group detections and ground truth are gathered first:
    tests/test_sharding_gloo.py holds world-2 == world-1)."""
    import json
    out = tmp_path / "out"
    cmd = [sys.executable, os.path.join(ROOT, "scripts", "run_test.py"), "--labelled", "--windows", "6", "--batch_size", "2",
           "--events_per_window", "4000", "--width", "240", "--height", "180", "--output_directory", str(out)]
    r = subprocess.run(cmd, capture_output=True, text=True, timeout=600)
    assert r.returncode == 0, r.stderr[-2000:]
    assert "metrics of the run (1 rank(s))" in r.stdout
    m = json.load(open(out / "synthetic" / "detection" / "run_test" / "metrics.json"))
    assert set(m) >= {"mAP", "mAP_50", "mAP_75"} and all(np.isfinite(v) for v in m.values())


def test_run_test_takes_the_reference_command_line(tmp_path):
    """readme.md:107-113 as it stands (minus the checkpoint file, which is not in this image): ``--config <yaml path>
    --use_image --img_net resnet50 --batch_size 8 --dataset_directory $DSEC_ROOT --output_directory $LOG_DIR``.  The DSEC
    readers are absent, so the run announces the synthetic stand-in stream and goes through the whole engine path
    (ResNet-50 branch + event graph, B = 8) built from the YAML's keys."""
    out = tmp_path / "log"
    cmd = [sys.executable, os.path.join(ROOT, "scripts", "run_test.py"), "--config", "config/dagr-s-dsec.yaml", "--use_image",
           "--img_net", "resnet50", "--batch_size", "8", "--dataset_directory", str(tmp_path / "DSEC_ROOT"),
           "--output_directory", str(out), "--windows", "16", "--events_per_window", "20000"]
    r = subprocess.run(cmd, capture_output=True, text=True, timeout=900, cwd=str(tmp_path))   # config/ is found from any cwd
    assert r.returncode == 0, r.stderr[-2000:]
    assert "NOTICE: --dataset_directory" in r.stdout and "SYNTHETIC" in r.stdout
    assert "NOTICE: no --checkpoint" in r.stdout
    assert "'img_net': 'resnet50'" in r.stdout and "'use_image': True" in r.stdout and "'max_neighbors': 16" in r.stdout
    assert "16 windows on 1 GPU(s)" in r.stdout
    rec = np.load(out / "synthetic" / "detection" / "run_test" / "detections_synthetic000.npy")
    assert rec.dtype.names == NAMES and len(rec) > 0


def test_run_test_interframe_takes_the_reference_command_line(tmp_path):
    """readme.md:131-138 (``config/eagr-s-dsec.yaml`` -- a name the reference's own tree lacks -- resolves to
    ``dagr-s-dsec.yaml`` with a notice; ``--num_interframe_steps``, ``--no_eval``)."""
    out = tmp_path / "log"
    cmd = [sys.executable, os.path.join(ROOT, "scripts", "run_test_interframe.py"), "--config", "config/eagr-s-dsec.yaml",
           "--use_image", "--img_net", "resnet18", "--batch_size", "2", "--dataset_directory", str(tmp_path / "DSEC_ROOT"),
           "--no_eval", "--output_directory", str(out), "--num_interframe_steps", "2", "--windows", "4",
           "--events_per_window", "3000", "--width", "320", "--height", "215"]
    r = subprocess.run(cmd, capture_output=True, text=True, timeout=900)
    assert r.returncode == 0, r.stderr[-2000:]
    assert "dagr-s-dsec.yaml" in r.stderr and "2 offsets x 4 windows" in r.stdout
