"""Generate tests/golden/graph_ref_small.npz with the reference's own kernels (run on the GPU box:
``gpurun -- python tests/make_golden_graph.py``; output lands in gpurun_out/golden/ and is then copied
into tests/golden/).  Stores inputs + the reference edge_index of every small case."""
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)

from oracle import ref_harness  # noqa: E402
from oracle import graph as og  # noqa: E402
from tests.graph_cases import small_cases, medium_cases  # noqa: E402


def main():
    assert ref_harness.available(), "needs oracle/_ref/libev_graph_ref.so and a GPU"
    out = {}
    ok = True
    for c in small_cases() + medium_cases():
        ref = ref_harness.reference_window_graph(c["x"], c["y"], c["t"], c["b"], c["W"], c["H"], c["B"], c["r"],
                                                 c["dt"], K=c["K"], Q=c["Q"])
        orc = og.build_window_graph(c["x"], c["y"], c["t"], c["b"], c["W"], c["H"], c["B"], c["r"], c["dt"],
                                    K=c["K"], Q=c["Q"])
        same = ref.shape == orc.shape and (ref == orc).all()
        ok &= bool(same)
        print(f"{c['name']:24s} N={len(c['x']):6d} E_ref={ref.shape[1]:8d} oracle==reference: {same}")
        if len(c["x"]) <= 6000:  # keep the committed fixture small
            n = c["name"]
            for k in ("x", "y", "t", "b"):
                out[f"{n}/{k}"] = c[k]
            out[f"{n}/params"] = np.array([c["W"], c["H"], c["B"], c["r"], c["dt"], c["K"], c["Q"]], np.int64)
            out[f"{n}/edge_index"] = ref.astype(np.int32)
    dst = os.path.join(ROOT, "gpurun_out", "golden")
    os.makedirs(dst, exist_ok=True)
    np.savez_compressed(os.path.join(dst, "graph_ref_small.npz"), **out)
    print("oracle pinned to reference kernels:", ok)
    return 0 if ok else 1


if __name__ == "__main__":
    sys.exit(main())
