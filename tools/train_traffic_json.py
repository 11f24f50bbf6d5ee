#!/usr/bin/env python3
"""Memory-side bytes of ONE training step, per kernel and in total, from two rocprofv3 PMC passes (FETCH_SIZE, WRITE_SIZE; one counter per pass)
of `bench.py --train --train-batch B --no-train-graph --no-extra-legs --steps S --warmup W` (eager issue: one dispatch per kernel).

usage: train_traffic_json.py <dir of the FETCH_SIZE pass> <dir of the WRITE_SIZE pass> <steps traced = S + W + 2> <batch> <out.json> [out.md]

Units and corrections as tools/traffic_json.py (MI355X_MICROARCH.md, HBM section): read bytes = 2 x FETCH_SIZE x 1024 (gfx950 tallies the 128-B
requests of wide coalesced reads at 64 B), write bytes = WRITE_SIZE x 1024; Infinity-Cache hits are included in both.  The traced process runs
S + W timed / warm-up steps plus the two eager steps bench.py uses to count launches, hence the divisor.  bench.py reads `gb_per_step` from the
newest profiles/*_train_traffic_b{B}.json."""
import collections
import csv
import glob
import json
import re
import sys


def load(d, counter):
    f = glob.glob(d + "/**/*counter_collection.csv", recursive=True)[0]
    tot = collections.defaultdict(float)
    n = collections.defaultdict(set)
    for r in csv.DictReader(open(f)):
        if r["Counter_Name"] != counter:
            continue
        name = re.sub(r"\(anonymous namespace\)::", "", r["Kernel_Name"])
        name = re.sub(r"^void ", "", name)
        name = re.sub(r"\(.*$", "", name)
        tot[name] += float(r["Counter_Value"])
        n[name].add(r["Dispatch_Id"])
    return tot, {k: len(v) for k, v in n.items()}


def main():
    fetch, nf = load(sys.argv[1], "FETCH_SIZE")
    write, nw = load(sys.argv[2], "WRITE_SIZE")
    steps, batch, out = int(sys.argv[3]), int(sys.argv[4]), sys.argv[5]
    rows = {}
    for k in sorted(set(fetch) | set(write)):
        rd = 2.0 * 1024.0 * fetch.get(k, 0.0) / steps
        wr = 1024.0 * write.get(k, 0.0) / steps
        rows[k] = {"launches_per_step": round(max(nf.get(k, 0), nw.get(k, 0)) / steps, 2), "read_gb_per_step": round(rd / 1e9, 4),
                   "write_gb_per_step": round(wr / 1e9, 4)}
    total = sum(v["read_gb_per_step"] + v["write_gb_per_step"] for v in rows.values())
    doc = {"source": "rocprofv3 --kernel-trace --pmc FETCH_SIZE / WRITE_SIZE (separate passes) on bench.py --train --no-train-graph; read = 2 x FETCH_SIZE x 1024",
           "clips_per_step": batch, "steps_traced": steps, "gb_per_step": round(total, 2),
           "read_gb_per_step": round(sum(v["read_gb_per_step"] for v in rows.values()), 2),
           "write_gb_per_step": round(sum(v["write_gb_per_step"] for v in rows.values()), 2), "kernels": rows}
    json.dump(doc, open(out, "w"), indent=1)
    top = sorted(rows.items(), key=lambda kv: -(kv[1]["read_gb_per_step"] + kv[1]["write_gb_per_step"]))
    lines = [f"# Memory-side bytes of one {batch}-clip training step: {doc['gb_per_step']} GB ({doc['read_gb_per_step']} read + {doc['write_gb_per_step']} written)", "",
             "| kernel | launches / step | read GB / step | write GB / step |", "|---|---|---|---|"]
    for k, v in top[:40]:
        lines.append(f"| `{k[:100]}` | {v['launches_per_step']} | {v['read_gb_per_step']} | {v['write_gb_per_step']} |")
    text = "\n".join(lines) + "\n"
    if len(sys.argv) > 6:
        open(sys.argv[6], "w").write(text)
    print(text[:3000])


if __name__ == "__main__":
    main()
