#!/usr/bin/env python
"""Aggregate a rocprofv3 counter_collection.csv per kernel: dispatches and mean counter value per dispatch."""
import csv
import re
import sys
from collections import defaultdict

rows = list(csv.DictReader(open(sys.argv[1])))
acc = defaultdict(lambda: defaultdict(float))
disp = defaultdict(set)
for r in rows:
    name = re.sub(r"\(anonymous namespace\)::", "", r["Kernel_Name"]).split("(")[0].replace("void ", "")[:60]
    acc[name][r["Counter_Name"]] += float(r["Counter_Value"])
    disp[name].add(r["Dispatch_Id"])
counters = sorted({c for k in acc for c in acc[k]})
print("kernel,dispatches," + ",".join(counters))
for k in sorted(acc, key=lambda k: -sum(acc[k].values())):
    n = len(disp[k])
    print(f"{k},{n}," + ",".join(f"{acc[k].get(c, 0) / n:.1f}" for c in counters))
