#!/usr/bin/env python
"""Condense rocprofv3 output (kernel stats + per-kernel PMC averages) into small text/JSON files
that are committed under profiles/."""
import collections, csv, glob, json, os, sys
out = sys.argv[1]
res = {}
st = glob.glob(os.path.join(out, "stats", "**", "*kernel_stats.csv"), recursive=True)
if st:
    rows = list(csv.DictReader(open(st[0])))
    tot = sum(int(r["TotalDurationNs"]) for r in rows)
    print("== rocprofv3 --kernel-trace --stats (python bench.py --steps 5 --warmup 2): top kernels")
    print(f"{'calls':>7} {'total_ms':>10} {'avg_us':>10} {'%':>6}  name")
    for r in rows[:22]:
        nm = r["Name"].replace("void lseg::(anonymous namespace)::", "").replace("lseg::(anonymous namespace)::", "")
        print(f"{int(r['Calls']):7d} {int(r['TotalDurationNs'])/1e6:10.3f} {float(r['AverageNs'])/1e3:10.2f} {100*int(r['TotalDurationNs'])/tot:6.2f}  {nm[:120]}")
    res["kernel_stats"] = [{"name": r["Name"], "calls": int(r["Calls"]), "avg_ns": float(r["AverageNs"]),
                            "total_ns": int(r["TotalDurationNs"])} for r in rows[:40]]
pm = {}
for f in glob.glob(os.path.join(out, "pmc_*", "**", "*counter_collection.csv"), recursive=True):
    agg = collections.defaultdict(lambda: collections.defaultdict(list))
    for r in csv.DictReader(open(f)):
        agg[r["Kernel_Name"]][r["Counter_Name"]].append(float(r["Counter_Value"]))
    for k, d in agg.items():
        for c, v in d.items():
            pm.setdefault(k, {})[c] = sum(v) / len(v)
print("\n== PMC per-launch averages for the GEMM / attention kernels")
for k, d in pm.items():
    if "gemm_kernel" not in k and "attention" not in k:
        continue
    nm = k.replace("void lseg::(anonymous namespace)::", "")[:110]
    print(nm)
    for c, v in sorted(d.items()):
        print(f"    {c:34s} {v:16.1f}")
res["pmc"] = pm
json.dump(res, open(os.path.join(out, "summary.json"), "w"), indent=1)
