#!/usr/bin/env python
"""One forward of a rocprofv3 --kernel-trace run, kernel by kernel: queue, duration, gap to the previous kernel's end on the same queue,
grid size (tools).  The last complete forward = the kernels between the last two im2col_patch launches."""
import csv, glob, os, re, sys
path = sys.argv[1]
f = glob.glob(os.path.join(path, "**", "*kernel_trace.csv"), recursive=True)[0]
rows = sorted(csv.DictReader(open(f)), key=lambda r: int(r["Start_Timestamp"]))
idx = [i for i, r in enumerate(rows) if "im2col_patch" in r["Kernel_Name"]]
a, b = idx[-2], idx[-1]
sel = rows[a:b]
def short(n):
    n = n.replace("void lseg::(anonymous namespace)::", "").replace("lseg::(anonymous namespace)::", "").replace("lseg::", "")
    n = re.sub(r"TileCfg<(\d+), (\d+), \d+, \d+, \d+(, \d+)?>", r"T\1x\2", n)
    n = re.sub(r"\(.*", "", n)
    return n[:64]
last_end = {}
t0 = int(sel[0]["Start_Timestamp"])
tot = {}
for r in sel:
    q = r["Queue_Id"]; s, e = int(r["Start_Timestamp"]), int(r["End_Timestamp"])
    gap = (s - last_end[q]) / 1e3 if q in last_end else 0.0
    last_end[q] = e
    nm = short(r["Kernel_Name"])
    tot.setdefault((q, nm), [0, 0.0])
    tot[(q, nm)][0] += 1; tot[(q, nm)][1] += (e - s) / 1e3
    if len(sys.argv) > 2:
        print(f"q{q} t={(s - t0) / 1e3:8.1f} us dur {(e - s) / 1e3:7.1f} gap {gap:6.1f}  grid {int(r['Grid_Size_X']) // int(r['Workgroup_Size_X']):5d}x{r['Grid_Size_Y']:>4s}  {nm}")
print(f"forward wall {(max(int(r['End_Timestamp']) for r in sel) - t0) / 1e3:.1f} us, {len(sel)} kernels")
for (q, nm), (n, us) in sorted(tot.items(), key=lambda kv: -kv[1][1])[:24]:
    print(f"  q{q} {us:8.1f} us  n={n:3d} avg {us / n:6.1f}  {nm}")
