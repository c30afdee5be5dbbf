#!/usr/bin/env python
"""Per-kernel totals from a rocprofv3 --kernel-trace run (rocpd sqlite `*_results.db` or `*_kernel_trace.csv`)."""
import csv, glob, os, re, sqlite3, sys
from collections import defaultdict

def rows_from(path):
    dbs = glob.glob(os.path.join(path, "**", "*_results.db"), recursive=True) if os.path.isdir(path) else [path]
    out = []
    for db in dbs:
        if db.endswith(".db"):
            cur = sqlite3.connect(db).cursor()
            out += cur.execute("select s.kernel_name, d.end - d.start from rocpd_kernel_dispatch d join rocpd_info_kernel_symbol s on d.kernel_id = s.id").fetchall()
    for f in glob.glob(os.path.join(path, "**", "*kernel_trace.csv"), recursive=True) if os.path.isdir(path) else []:
        for r in csv.DictReader(open(f)):
            out.append((r["Kernel_Name"], int(r["End_Timestamp"]) - int(r["Start_Timestamp"])))
    return out

def short(n):
    n = re.sub(r"^_ZN4lseg12_GLOBAL__N_1\d+", "", n)
    n = n.replace("NS_4BF16E", "BF16").replace("NS_3F16E", "F16")
    m = re.search(r"TileCfgILi(\d+)ELi(\d+)", n)
    tile = f" tile{m.group(1)}x{m.group(2)}" if m else ""
    e = re.search(r"EEELb([01])ELb([01])ELi(\d)ELi(\d)", n)
    epi = f" conv{e.group(1)} relu{e.group(2)} epi{e.group(3)} tag{e.group(4)}" if e else ""
    return re.sub(r"I.*", "", n)[:40] + ("<BF16>" if "BF16" in n else "<F16>" if "F16" in n else "") + tile + epi

if __name__ == "__main__":
    rows = rows_from(sys.argv[1])
    div = float(sys.argv[2]) if len(sys.argv) > 2 else 1.0
    agg = defaultdict(lambda: [0, 0])
    for n, d in rows:
        a = agg[short(n)]; a[0] += 1; a[1] += d
    tot = sum(v[1] for v in agg.values())
    print(f"total kernel time {tot / 1e6 / div:.3f} ms per step (divisor {div})")
    for k, (c, t) in sorted(agg.items(), key=lambda kv: -kv[1][1])[:45]:
        print(f"{t / 1e6 / div:9.3f} ms {100 * t / tot:5.1f}%  n={c / div:7.1f}  avg {t / c / 1e3:9.1f} us  {k}")
