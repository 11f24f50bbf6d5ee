import csv, glob, sys, collections
f = glob.glob(sys.argv[1] + "/**/*kernel_trace.csv", recursive=True)[0]
rows = list(csv.DictReader(open(f)))
rows.sort(key=lambda r: int(r["Start_Timestamp"]))
c = collections.Counter()
for i, r in enumerate(rows):
    if r["Kernel_Name"].startswith(sys.argv[2]):
        prev = rows[i-1]["Kernel_Name"][:50]; nxt = rows[i+1]["Kernel_Name"][:50] if i+1 < len(rows) else ""
        c[(r["Grid_Size_X"], r["Grid_Size_Y"], r["Workgroup_Size_X"], prev, nxt)] += 1
for k, v in c.most_common(30): print(v, k)
