#!/usr/bin/env python
"""tools/kernel_resources.py <out.md>: registers / scratch / occupancy / static LDS of every kernel of libdagr_hip.so
(hipcc -Rpass-analysis=kernel-resource-usage over dagr_amd/csrc/*.hip with the Makefile's flags; no GPU needed)."""
import glob
import os
import re
import subprocess
import sys
import tempfile

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
out = sys.argv[1] if len(sys.argv) > 1 else os.path.join(ROOT, "profiles", "kernel_resources.md")
rows = []
for f in sorted(glob.glob(os.path.join(ROOT, "dagr_amd", "csrc", "*.hip"))):
    with tempfile.NamedTemporaryFile(suffix=".o") as tmp:
        r = subprocess.run(["hipcc", "--offload-arch=gfx950", "-O3", "-std=c++17", "-fPIC", "-ffp-contract=off",
                            "-I" + os.path.join(ROOT, "include"), "-c", f, "-o", tmp.name,
                            "-Rpass-analysis=kernel-resource-usage"], capture_output=True, text=True)
    cur = {}
    for line in r.stderr.splitlines():
        m = re.search(r"remark:\s+(.*?)\s+\[-Rpass", line)
        if not m:
            continue
        t = m.group(1)
        if t.startswith("Function Name:"):
            cur = {"name": t.split(":", 1)[1].strip()}
        for key, pat in (("v", r"^VGPRs: (\d+)"), ("a", r"^AGPRs: (\d+)"), ("s", r"^SGPRs: (\d+)"),
                         ("sc", r"^ScratchSize \[bytes/lane\]: (\d+)"), ("o", r"^Occupancy \[waves/SIMD\]: (\d+)"),
                         ("l", r"^LDS Size \[bytes/block\]: (\d+)")):
            mm = re.match(pat, t)
            if mm:
                cur[key] = mm.group(1)
                if key == "l":
                    rows.append((os.path.basename(f), dict(cur)))
names = sorted({c["name"] for _, c in rows})
dem = subprocess.run(["c++filt"] + names, capture_output=True, text=True).stdout.splitlines()
nice = {}
for n, d in zip(names, dem):
    d = d.replace("dagr::(anonymous namespace)::", "").replace("void ", "")
    nice[n] = re.sub(r"\(.*$", "", d)
with open(out, "w") as fo:
    fo.write("# Register / scratch / LDS budget of every kernel of libdagr_hip.so (hipcc -Rpass-analysis=kernel-resource-usage,\n"
             "# the Makefile's flags; dynamic LDS is chosen at launch and not listed)\n\n")
    fo.write("| file | kernel | VGPRs | AGPRs | scratch B/lane | waves/SIMD | static LDS B |\n|---|---|---|---|---|---|---|\n")
    for f, c in rows:
        fo.write(f"| {f} | `{nice[c['name']]}` | {c.get('v', '')} | {c.get('a', '')} | {c.get('sc', '')} | "
                 f"{c.get('o', '')} | {c.get('l', '')} |\n")
print(len(rows), "kernels ->", out)
