#!/usr/bin/env python3
"""Per-kernel totals of a `rocprofv3 --kernel-trace --stats --output-format csv -d DIR` run: tools/kernel_stats_summary.py DIR OUT.txt "title" """
import csv, glob, os, sys
src, dst, title = sys.argv[1], sys.argv[2], sys.argv[3] if len(sys.argv) > 3 else ""
rows = []
for f in glob.glob(os.path.join(src, "**", "*kernel_stats.csv"), recursive=True):
    rows += list(csv.DictReader(open(f)))
rows.sort(key=lambda r: -float(r["TotalDurationNs"]))
with open(dst, "w") as o:
    o.write(f"== {title} ==\n")
    for r in rows[:50]:
        o.write("%-130s calls=%6s avg_us=%10.2f total_ms=%10.2f pct=%s\n" % (r["Name"][:130], r["Calls"], float(r["AverageNs"]) / 1e3, float(r["TotalDurationNs"]) / 1e6, r["Percentage"]))
print(open(dst).read()[:2500])
