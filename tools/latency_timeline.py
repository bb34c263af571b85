#!/usr/bin/env python
"""tools/latency_timeline.py <kernel_trace.csv>: per-kernel timeline of the LAST window of a tools/latency_trace.py run
(windows are separated by a rocprim single_scan marker kernel): start, duration, gap to the previous kernel."""
import csv
import sys

rows = list(csv.DictReader(open(sys.argv[1])))
rows.sort(key=lambda r: int(r["Start_Timestamp"]))
marks = [i for i, r in enumerate(rows) if "single_scan_kernel" in r["Kernel_Name"]]
spans = []
for k, a in enumerate(marks):
    b = marks[k + 1] if k + 1 < len(marks) else len(rows)
    win = rows[a + 1:b]
    if k + 1 < len(marks):
        win = win[:-1]                      # the zeros() fill that precedes the next marker
    t0 = int(win[0]["Start_Timestamp"])
    busy = sum(int(r["End_Timestamp"]) - int(r["Start_Timestamp"]) for r in win)
    spans.append(((int(win[-1]["End_Timestamp"]) - t0) / 1e3, busy / 1e3, len(win)))
    if k == len(marks) - 1:
        prev = None
        for r in win:
            s, e = int(r["Start_Timestamp"]), int(r["End_Timestamp"])
            name = r["Kernel_Name"].replace("dagr::(anonymous namespace)::", "").replace("void ", "").split("(")[0][:52]
            print("%8.1f %7.1f gap %6.1f  %s" % ((s - t0) / 1e3, (e - s) / 1e3, (s - prev) / 1e3 if prev else 0, name))
            prev = e
print("windows (span us, busy us, kernels):", spans)
