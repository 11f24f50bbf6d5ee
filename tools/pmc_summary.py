#!/usr/bin/env python3
"""Average PMC counters per dispatch, grouped by kernel (rocprofv3 --pmc ... --output-format csv)."""
import csv, glob, sys, collections
f = glob.glob(sys.argv[1] + "/**/*counter_collection.csv", recursive=True)[0]
agg = collections.defaultdict(lambda: collections.defaultdict(float)); cnt = collections.defaultdict(set)
for r in csv.DictReader(open(f)):
    k = (r["Kernel_Name"].replace("(anonymous namespace)::", "")[:70], r["Grid_Size"])
    agg[k][r["Counter_Name"]] += float(r["Counter_Value"]); cnt[k].add(r["Dispatch_Id"])
for k, v in agg.items():
    n = len(cnt[k])
    if len(sys.argv) > 2 and sys.argv[2] not in k[0]: continue
    print(k[0], "grid", k[1], "x", n)
    print("   ", {c: round(x / n) for c, x in sorted(v.items())})
