#!/usr/bin/env python3
"""Summarise hipcc -Rpass-analysis=kernel-resource-usage output: one line per kernel."""
import re, subprocess, sys
src = sys.argv[1]
out = subprocess.run(["hipcc", "--offload-arch=gfx950", "-O3", "-std=c++17", "-fPIC", "-c", "-I../../include", src, "-o", "/tmp/_ru.o",
                      "-Rpass-analysis=kernel-resource-usage"], capture_output=True, text=True).stderr
cur = None; rows = {}
for line in out.splitlines():
    m = re.search(r"remark: (.*?): (.*?) \[-Rpass", line) or re.search(r"remark:\s+(.*?): (.*?) \[-Rpass", line)
    if not m: continue
    k, v = m.group(1).strip(), m.group(2).strip()
    if k == "Function Name":
        cur = subprocess.run(["c++filt", v], capture_output=True, text=True).stdout.strip()[:90]; rows[cur] = {}
    elif cur: rows[cur][k] = v
for k, r in rows.items():
    print(f"{k:92s} V={r.get('VGPRs')} A={r.get('AGPRs')} spill={r.get('VGPRs Spill')}/{r.get('ScratchSize [bytes/lane]')} LDS={r.get('LDS Size [bytes/block]')} occ={r.get('Occupancy [waves/SIMD]')}")
