#!/usr/bin/env python
"""profiles/<round>_traffic.json from the PMC passes of tools/pmc.sh (per kernel, per bench config):
python tools/make_traffic_json.py <image tag> <events-only tag> [round]"""
import json
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def load(path):
    d = {}
    for i, line in enumerate(open(path)):
        if i == 0:
            continue
        name, _, val = line.rstrip("\n").rsplit(",", 2)
        d[name] = float(val)
    return d


sys.path.insert(0, ROOT)
import bench  # noqa: E402  (source_stamp: the line's roofline.traffic is refused when the kernels have changed since)

out = {"source_stamp": bench.source_stamp(),
       "_note": "rocprofv3 --pmc FETCH_SIZE / WRITE_SIZE in separate passes (plus --kernel-trace only) over `bench.py --steps 3 "
                "--warmup 1 --engines 1` (tools/pmc.sh); KB per dispatch averaged over the dispatches of the run; traffic_bytes = "
                "(2*FETCH_SIZE + WRITE_SIZE)*1024 as MI355X_MICROARCH.md prescribes for gfx950 (FETCH_SIZE reports 1/2 of coalesced reads). "
                "Calibration on this library's patterns (profiles/r1_pmc_calibration.csv, round 1): coalesced 4 B and 16 B/lane streams 0.50x, 64-byte "
                "row gathers 1.00x, writes 1.00x -- for the gather-dominated conv kernels the doubled figure is an upper bound.",
       "configs": {}}
for cfg, tag in (("use_image", sys.argv[1]), ("events_only", sys.argv[2])):
    f = load(os.path.join(ROOT, "gpurun_out", tag, "pmc_fetch.csv"))
    w = load(os.path.join(ROOT, "gpurun_out", tag, "pmc_write.csv"))
    ks = {}
    for k in f:
        if k.startswith("dagr::"):
            ks[k.replace("dagr::", "")] = {"fetch_kb": f[k], "write_kb": w.get(k, 0.0),
                                           "traffic_bytes": int((2 * f[k] + w.get(k, 0.0)) * 1024)}
    out["configs"][cfg] = ks
rnd = sys.argv[3] if len(sys.argv) > 3 else "r2"
json.dump(out, open(os.path.join(ROOT, "profiles", rnd + "_traffic.json"), "w"), indent=1)
print({c: len(v) for c, v in out["configs"].items()})
