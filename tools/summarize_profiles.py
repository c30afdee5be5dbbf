#!/usr/bin/env python
"""Condense rocprofv3 output (kernel trace + per-kernel PMC averages) into small text/JSON files that are committed under profiles/.
Kernels are grouped by (name, grid size): the bench's B = 1 self-check, the text tower and the B = 36 image tower launch the same
symbols with different grids, and a name-only average mixes them."""
import collections, csv, glob, json, os, sys
out = sys.argv[1]
res = {}


def short(n):
    return n.replace("void lseg::(anonymous namespace)::", "").replace("lseg::(anonymous namespace)::", "")


tr = glob.glob(os.path.join(out, "stats", "**", "*kernel_trace.csv"), recursive=True)
if tr:
    agg = collections.defaultdict(lambda: [0, 0])
    for r in csv.DictReader(open(tr[0])):
        grid = f"{r['Grid_Size_X']}x{r['Grid_Size_Y']}x{r['Grid_Size_Z']}"
        a = agg[(r["Kernel_Name"], grid)]
        a[0] += 1; a[1] += int(r["End_Timestamp"]) - int(r["Start_Timestamp"])
    rows = sorted(agg.items(), key=lambda kv: -kv[1][1])
    tot = sum(v[1] for _, v in rows)
    print("== rocprofv3 --kernel-trace (python bench.py --steps 5 --warmup 2 ...): top kernels by (name, grid)")
    print(f"{'calls':>7} {'total_ms':>10} {'avg_us':>10} {'%':>6}  {'grid(threads)':>16}  name")
    for (nm, grid), (calls, ns) in rows[:32]:
        print(f"{calls:7d} {ns / 1e6:10.3f} {ns / calls / 1e3:10.2f} {100 * ns / tot:6.2f}  {grid:>16}  {short(nm)[:120]}")
    res["kernel_stats"] = [{"name": nm, "grid": grid, "calls": c, "avg_ns": ns / c, "total_ns": ns} for (nm, grid), (c, ns) in rows[:80]]
pm = {}
for f in glob.glob(os.path.join(out, "pmc_*", "**", "*counter_collection.csv"), recursive=True):
    agg = collections.defaultdict(lambda: collections.defaultdict(list))
    for r in csv.DictReader(open(f)):
        grid = r.get("Grid_Size", "")
        agg[(r["Kernel_Name"], grid)][r["Counter_Name"]].append(float(r["Counter_Value"]))
    for k, d in agg.items():
        for c, v in d.items():
            pm.setdefault("|".join(k), {})[c] = sum(v) / len(v)
            pm["|".join(k)]["_launches"] = len(v)
print("\n== PMC per-launch averages for the largest GEMM / attention instances")
nbig = int(sys.argv[2]) if len(sys.argv) > 2 else 9
big = sorted((k for k in pm if "gemm_kernel" in k or "attention" in k or "attn_" in k or "corr_planes" in k), key=lambda k: -pm[k].get("GRBM_GUI_ACTIVE", pm[k].get("FETCH_SIZE", 0)))[:nbig]
for k in big:
    print(short(k)[:150])
    for c, v in sorted(pm[k].items()):
        print(f"    {c:34s} {v:16.1f}")
res["pmc"] = pm
json.dump(res, open(os.path.join(out, "summary.json"), "w"), indent=1)
