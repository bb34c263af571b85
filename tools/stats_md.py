#!/usr/bin/env python
"""kernel_stats.csv (rocprofv3 --stats) -> a short markdown table for profiles/.
usage: stats_md.py <kernel_stats.csv> <out.md> "<command line>" ["<note>"]"""
import csv
import re
import sys

src, dst, cmd = sys.argv[1:4]
note = sys.argv[4] if len(sys.argv) > 4 else ""
rows = list(csv.reader(open(src)))[1:]
setup = [r for r in rows if r[0].startswith("naive_conv")]   # MIOpen benchmark-mode search at start-up
rows = [r for r in rows if not r[0].startswith("naive_conv")]
total = sum(int(r[2]) for r in rows) or 1


def short(name):
    name = name.replace("(anonymous namespace)::", "").replace("void ", "")
    name = re.sub(r"\(.*$", "", name)
    return name[:90]


with open(dst, "w") as f:
    f.write(f"# rocprofv3 --kernel-trace --stats -- {cmd}\n")
    if note:
        f.write(f"# {note}\n")
    if setup:
        ms = sum(int(r[2]) for r in setup) / 1e6
        f.write(f"# excluded from the table and the percentages: MIOpen's benchmark-mode search at start-up "
                f"(naive_conv_*, {sum(int(r[1]) for r in setup)} calls, {ms:.0f} ms, untimed)\n")
    f.write("\n| kernel | calls | total ms | avg us | % of kernel time |\n|---|---|---|---|---|\n")
    for r in rows[:45]:
        f.write(f"| {short(r[0])} | {r[1]} | {int(r[2]) / 1e6:.2f} | {float(r[3]) / 1e3:.1f} | {100 * int(r[2]) / total:.2f} |\n")
