#!/usr/bin/env python3
"""Condense a rocprofv3 --kernel-trace --stats CSV directory into a small per-kernel table (markdown) for profiles/."""
import csv, glob, os, sys
d = sys.argv[1]
out = sys.argv[2]
files = glob.glob(os.path.join(d, "**", "*kernel_stats.csv"), recursive=True)
rows = []
for f in files:
    with open(f) as fh:
        for r in csv.DictReader(fh):
            rows.append(r)
rows.sort(key=lambda r: -float(r.get("TotalDurationNs", 0) or 0))
tot = sum(float(r["TotalDurationNs"]) for r in rows) or 1.0
with open(out, "w") as o:
    o.write(f"# rocprofv3 --kernel-trace --stats summary ({os.path.basename(d)})\n\n")
    o.write("| kernel | calls | total ms | avg us | min us | max us | % |\n|---|---|---|---|---|---|---|\n")
    for r in rows[:60]:
        name = r["Name"]
        name = name.replace("(anonymous namespace)::", "")
        if len(name) > 110: name = name[:107] + "..."
        o.write(f"| `{name}` | {r['Calls']} | {float(r['TotalDurationNs'])/1e6:.3f} | {float(r['AverageNs'])/1e3:.2f} | "
                f"{float(r['MinNs'])/1e3:.2f} | {float(r['MaxNs'])/1e3:.2f} | {100*float(r['TotalDurationNs'])/tot:.1f} |\n")
print("wrote", out, len(rows), "kernels")
