#!/usr/bin/env python3
"""Which call sites launch a kernel?  From a `rocprofv3 --kernel-trace --output-format csv -d <dir>` collection of an EAGER run: every launch whose
name contains <pattern>, grouped by (grid size, the kernel launched before it, the kernel launched after it), most frequent first.

    python3 tools/kernel_trace_sites.py <dir> <pattern> [<exclude pattern>]

(round 6: found the generic `col_partial_kernel` launches of a training step to be the TCN's / stem's BatchNorms on the side streams, and counted the
copy / fill / elementwise glue launches of a 16-clip step per site)"""
import collections
import csv
import glob
import sys

f = glob.glob(sys.argv[1] + "/**/*kernel_trace.csv", recursive=True)[0]
rows = sorted(csv.DictReader(open(f)), key=lambda r: int(r["Start_Timestamp"]))
pat, excl = sys.argv[2], (sys.argv[3] if len(sys.argv) > 3 else None)


def short(n):
    return n.replace("(anonymous namespace)::", "").replace("void ", "")[:48]


c = collections.Counter()
for i, r in enumerate(rows):
    n = r["Kernel_Name"]
    if pat in n and not (excl and excl in n):
        prev = short(rows[i - 1]["Kernel_Name"]) if i else ""
        nxt = short(rows[i + 1]["Kernel_Name"]) if i + 1 < len(rows) else ""
        c[(r["Grid_Size_X"], r["Grid_Size_Y"], prev, nxt)] += 1
print(f"{sum(c.values())} launches of *{pat}* in {len(rows)}")
for k, v in c.most_common(40):
    print(f"{v:5d}  grid {k[0]} x {k[1]}   after {k[2]}   before {k[3]}")
