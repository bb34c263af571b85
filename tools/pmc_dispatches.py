#!/usr/bin/env python
"""Per-dispatch values of one counter for the kernels whose name contains a pattern, in dispatch order:
python tools/pmc_dispatches.py <counter_collection.csv> <pattern>"""
import csv
import sys
from collections import OrderedDict

acc = OrderedDict()
for r in csv.DictReader(open(sys.argv[1])):
    if sys.argv[2] in r["Kernel_Name"]:
        key = int(r["Dispatch_Id"])
        acc.setdefault(key, {}).setdefault(r["Counter_Name"], 0.0)
        acc[key][r["Counter_Name"]] += float(r["Counter_Value"])
for k in sorted(acc):
    print(k, " ".join(f"{c}={v:.0f}" for c, v in acc[k].items()))
