#!/usr/bin/env python3
"""Per-step kernel table from a rocprofv3 kernel_trace CSV: groups launches by (kernel, grid) over the last N steps."""
import csv, glob, os, sys, collections
d, steps = sys.argv[1], int(sys.argv[2])
f = glob.glob(os.path.join(d, "**", "*kernel_trace.csv"), recursive=True)[0]
rows = list(csv.DictReader(open(f)))
rows.sort(key=lambda r: int(r["Start_Timestamp"]))
rows = [r for r in rows if "copyBuffer" not in r["Kernel_Name"] and "fillBuffer" not in r["Kernel_Name"]]
# a step starts at each mel_power_kernel launch
starts = [i for i, r in enumerate(rows) if "mel_power_kernel" in r["Kernel_Name"]]
starts = starts[-steps - 1:-1] if len(starts) > steps else starts[:-1]
end = [i for i, r in enumerate(rows) if "mel_power_kernel" in r["Kernel_Name"]][-1]
sel = rows[starts[0]:end]
n = len(starts)
agg = collections.OrderedDict()
for r in sel:
    name = r["Kernel_Name"].replace("(anonymous namespace)::", "").split("(")[0][:60]
    key = (name, r["Grid_Size_X"], r["Grid_Size_Y"], r["Grid_Size_Z"])
    a = agg.setdefault(key, [0, 0.0])
    a[0] += 1
    a[1] += (int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) / 1e3
span = (int(sel[-1]["End_Timestamp"]) - int(sel[0]["Start_Timestamp"])) / 1e3 / n
busy = sum(a[1] for a in agg.values()) / n
print(f"steps={n}  span/step={span:.1f} us  kernel-busy/step={busy:.1f} us")
for (name, gx, gy, gz), (cnt, us) in sorted(agg.items(), key=lambda kv: -kv[1][1]):
    print(f"{us / n:8.1f} us/step  {cnt / n:5.1f} x {us / cnt:7.1f} us  {name} grid=({gx},{gy},{gz})")
