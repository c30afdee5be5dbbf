#!/usr/bin/env python
"""Per-(kernel, grid) time of ONE training step: the last complete step of a rocprofv3 kernel trace of tools/train_bench.py
(steps are cut at the patch-embedding im2col).  Usage: trace_train.py <rocprofv3 output dir> [rows]"""
import csv, glob, os, re, sys, collections
f = glob.glob(os.path.join(sys.argv[1], "**", "*kernel_trace.csv"), recursive=True)[0]
top = int(sys.argv[2]) if len(sys.argv) > 2 else 60
rows = sorted(csv.DictReader(open(f)), key=lambda r: int(r["Start_Timestamp"]))
idx = [i for i, r in enumerate(rows) if "im2col_patch" in r["Kernel_Name"]]
sel = rows[idx[-2]:idx[-1]]
def sh(n):
    n = n.replace("void lseg::(anonymous namespace)::", "").replace("lseg::(anonymous namespace)::", "").replace("lseg::", "")
    n = re.sub(r"\(anonymous namespace\)::", "", n)
    return re.sub(r"\((?:[^()]|\([^()]*\))*\)$", "", n)[:86]
agg = collections.defaultdict(lambda: [0, 0])
for r in sel:
    a = agg[(sh(r["Kernel_Name"]), r["Grid_Size_X"], r["LDS_Block_Size"] if "LDS_Block_Size" in r else "")]
    a[0] += 1; a[1] += int(r["End_Timestamp"]) - int(r["Start_Timestamp"])
tot = sum(v[1] for v in agg.values())
span = int(sel[-1]["End_Timestamp"]) - int(sel[0]["Start_Timestamp"])
print(f"{len(sel)} kernels in the step; kernel time {tot / 1e6:.2f} ms, span {span / 1e6:.2f} ms")
print(f"{'calls':>6} {'total_ms':>9} {'avg_us':>9} {'%':>6} {'grid':>9}  name")
for (nm, g, _), (c, ns) in sorted(agg.items(), key=lambda kv: -kv[1][1])[:top]:
    print(f"{c:6d} {ns / 1e6:9.3f} {ns / c / 1e3:9.1f} {100 * ns / tot:6.2f} {g:>9}  {nm}")
