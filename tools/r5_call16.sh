#!/bin/bash
set -u
ROOT=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$ROOT/gpurun_out/r5c16
mkdir -p "$OUT"
cd "$ROOT"
PROBE_IMAGE=1 bash tools/prof_trace.sh r5c16_tl tools/latency_trace.py 1 25000 4 > "$OUT/trace.log" 2>&1
python - gpurun_out/r5c16_tl/kernel_trace.csv > "$OUT/timeline_img_b1.txt" <<'PY'
import csv, sys
rows = list(csv.DictReader(open(sys.argv[1])))
rows.sort(key=lambda r: int(r["Start_Timestamp"]))
marks = [i for i, r in enumerate(rows) if "single_scan_kernel" in r["Kernel_Name"]]
a = marks[-1]
win = rows[a + 1:]
t0 = int(win[0]["Start_Timestamp"])
prev = None
for r in win:
    s, e = int(r["Start_Timestamp"]), int(r["End_Timestamp"])
    name = r["Kernel_Name"].replace("dagr::(anonymous namespace)::", "").replace("void ", "").split("(")[0][:70]
    print("%8.1f %7.1f gap %6.1f q%s  %s" % ((s - t0) / 1e3, (e - s) / 1e3, (s - prev) / 1e3 if prev else 0, r.get("Queue_Id", "?"), name))
    prev = max(prev or 0, e)
print("span us", (max(int(r["End_Timestamp"]) for r in win) - t0) / 1e3, "kernels", len(win), "busy", sum(int(r["End_Timestamp"]) - int(r["Start_Timestamp"]) for r in win) / 1e3)
PY
rm -f gpurun_out/r5c16_tl/kernel_trace.csv
tail -5 "$OUT/timeline_img_b1.txt"
