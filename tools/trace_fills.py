#!/usr/bin/env python
"""Where do the memset (fillBufferAligned) launches of one training step come from?  Last complete step of a kernel trace: every fill
with the kernels before / after it (tools)."""
import csv, glob, os, re, sys, collections
f = glob.glob(os.path.join(sys.argv[1], "**", "*kernel_trace.csv"), recursive=True)[0]
rows = sorted(csv.DictReader(open(f)), key=lambda r: int(r["Start_Timestamp"]))
idx = [i for i, r in enumerate(rows) if "im2col_patch" in r["Kernel_Name"]]
sel = rows[idx[-2]:idx[-1]]
sh = lambda n: re.sub(r"\(.*", "", n.replace("void lseg::(anonymous namespace)::", "").replace("lseg::(anonymous namespace)::", ""))[:48]
c = collections.Counter()
for i, r in enumerate(sel):
    if "fillBuffer" in r["Kernel_Name"]:
        nxt = next((sh(q["Kernel_Name"]) for q in sel[i + 1:] if "fillBuffer" not in q["Kernel_Name"]), "end")
        c[(nxt, r["Grid_Size_X"])] += 1
print(len(sel), "kernels in the step;", sum(c.values()), "fills")
for (k, g), n in c.most_common(25):
    print(f"  {n:4d} x fill(grid {g}) before {k}")
