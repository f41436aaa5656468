#!/usr/bin/env python3
"""Condense a tools/profile_r01.sh output directory into one small text/JSON summary for profiles/.

usage: tools/summarize_prof.py gpurun_out/prof_<tag> profiles/<name>   (writes <name>.txt and <name>.json)
"""
import csv
import glob
import json
import os
import sys
from collections import defaultdict

src, dst = sys.argv[1], sys.argv[2]
lines, js = [], {}
for f in glob.glob(os.path.join(src, "trace", "*", "*kernel_stats.csv")):
    lines.append("== rocprofv3 --kernel-trace --stats (kernel_stats.csv) ==")
    rows = list(csv.DictReader(open(f)))
    js["kernel_stats"] = rows
    for r in rows:
        lines.append(f"{r['Name'][:110]:110s} calls={r['Calls']:>4s} avg_ns={float(r['AverageNs']):>12.1f} min_ns={r['MinNs']:>9s} max_ns={r['MaxNs']:>9s} pct={r['Percentage']}")
pmc = defaultdict(lambda: defaultdict(list))
meta = {}
for f in glob.glob(os.path.join(src, "pmc_*", "*", "*counter_collection.csv")):
    for r in csv.DictReader(open(f)):
        k = r["Kernel_Name"]
        if "pynqs" not in k:
            continue
        pmc[k][r["Counter_Name"]].append(float(r["Counter_Value"]))
        meta[k] = {x: r[x] for x in ("Grid_Size", "Workgroup_Size", "LDS_Block_Size", "Scratch_Size", "VGPR_Count", "Accum_VGPR_Count", "SGPR_Count")}
lines.append("")
lines.append("== rocprofv3 --pmc (separate passes), mean per dispatch ==")
js["pmc"] = {}
for k, d in pmc.items():
    lines.append(k[:150])
    lines.append("   " + " ".join(f"{a}={b}" for a, b in meta[k].items()))
    js["pmc"][k] = {"meta": meta[k]}
    for c, v in sorted(d.items()):
        m = sum(v) / len(v)
        js["pmc"][k][c] = m
        lines.append(f"   {c:40s} {m:18.1f}   (n={len(v)})")
open(dst + ".txt", "w").write("\n".join(lines) + "\n")
json.dump(js, open(dst + ".json", "w"), indent=1)
print("\n".join(lines))
