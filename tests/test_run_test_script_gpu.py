"""Script-level drop-in (scripts/run_test.py, twin of the reference's scripts/run_test.py:31-66): a tiny synthetic run
writes the detection record file of utils/buffers.py:46-66."""
import os
import subprocess
import sys

import numpy as np
import pytest

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def test_run_test_script_writes_detection_records(tmp_path):
    out = tmp_path / "out"
    cmd = [sys.executable, os.path.join(ROOT, "scripts", "run_test.py"), "--windows", "6", "--batch_size", "2",
           "--events_per_window", "3000", "--width", "320", "--height", "215", "--stream", "edges",
           "--output_directory", str(out)]
    r = subprocess.run(cmd, capture_output=True, text=True, timeout=300)
    assert r.returncode == 0, r.stderr[-2000:]
    rec = np.load(out / "detections.npy")
    assert rec.dtype.names == ("window", "t", "x", "y", "w", "h", "class_id", "class_confidence")
    assert len(rec) > 0 and set(np.unique(rec["window"])) <= set(range(6))
    assert (np.diff(rec["window"].astype(np.int64)) >= 0).all()          # restored window order
    assert (rec["w"] > 0).all() and (rec["h"] > 0).all() and (rec["class_confidence"] >= 0.001).all()
    assert "6 windows, 18000 events" in r.stdout
