"""Device event downsampler (dagr.data.downsample.downsample_events, csrc/downsample.hip) against the restated reference
loop (oracle/downsample.py <- scripts/downsample_events.py:91-124): same surviving events, same integrator state, over
two consecutive chunks that share the change map (the way the reference streams a recording)."""
import numpy as np
import pytest
import torch

from oracle import downsample as od
from dagr_amd.utils import synthetic as syn

pytestmark = pytest.mark.gpu


@pytest.mark.parametrize("factor,stream", [(2, "edges"), (2, "uniform"), (4, "edges")])
def test_downsample_matches_reference_loop(factor, stream):
    from dagr_amd.data.downsample import downsample_events
    W, H = 640, 480
    Wo, Ho = W // factor, H // factor
    gen = syn.edges_window if stream == "edges" else syn.uniform_window
    cm_o = cm_d = None
    for chunk in range(2):
        x, y, t, p = gen(60000, W, H, seed=90 + chunk)
        ev = dict(x=x.astype(np.int64), y=y.astype(np.int64), t=t.astype(np.int64), p=p.astype(np.int8))
        want, cm_o = od.downsample_events({k: v.copy() for k, v in ev.items()}, H, W, Ho, Wo, change_map=cm_o)
        dev = {k: torch.from_numpy(v).cuda() for k, v in ev.items()}
        got, cm_d = downsample_events(dev, H, W, Ho, Wo, change_map=cm_d)
        assert len(want["t"]) > 5
        for k in ("x", "y", "t", "p"):
            assert np.array_equal(got[k].cpu().numpy().astype(np.int64), want[k].astype(np.int64)), (chunk, k)
        assert np.array_equal(cm_d.cpu().numpy(), cm_o)


def test_downsample_script_on_the_device_matches_the_reference_main_loop(tmp_path):
    """scripts/downsample_events.py end to end with the device kernel: 230 123 events in three chunks (the last one partial,
    with the reference's {0, 1} polarities), the change map resident on the GPU in between -> the digests of the reference
    script's own main loop (tests/golden/ref_py_data.npz)."""
    import os
    import sys
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    sys.path.insert(0, os.path.join(root, "scripts"))
    import downsample_events as D
    from tests.test_data_refpy import G, _check_stream_output, _stream_case
    dst = tmp_path / "events_2x.npz"
    counts = D.main(["--input_path", str(_stream_case(tmp_path)), "--output_path", str(dst), "--input_height", "48",
                     "--input_width", "64", "--output_height", "24", "--output_width", "32"])
    assert counts["t"] == int(G["stream_count"])
    _check_stream_output(dst)
