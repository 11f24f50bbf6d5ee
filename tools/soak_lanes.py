#!/usr/bin/env python3
"""Soak test of the multi-lane path at the bench workload (B=64): every step's outputs are compared bit for bit with the
single-lane reference for the same resident batch.  usage: soak_lanes.py [steps=300] [lanes=4]"""
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch

import bench
from emotiongestures_amd.pipeline import ClipPipeline

steps = int(sys.argv[1]) if len(sys.argv) > 1 else 300
lanes = int(sys.argv[2]) if len(sys.argv) > 2 else 4
dev = torch.device("cuda:0")
gen, vae, mel, _, _ = bench.build_models("bf16x3", dev)
inp = bench.make_inputs(64, seed=1000)
g = {k: torch.from_numpy(v).to(dev) for k, v in inp.items()}
one = ClipPipeline((gen, vae, mel), g, dev, lanes=1)
i0 = one.launch_next()
one.wait(i0)
ref = [t.clone() for t in one.outputs(i0)]
torch.cuda.synchronize()
pipe = ClipPipeline((gen, vae, mel), g, dev, lanes=lanes)
bad = 0
worst = 0.0
pending = []
for s in range(steps):
    if len(pending) == lanes:
        i = pending.pop(0)
        pipe.wait(i)
        eq = all(torch.equal(a, b) for a, b in zip(pipe.outputs(i), ref))
        if not eq:
            bad += 1
            worst = max(worst, max(float((a - b).abs().max()) for a, b in zip(pipe.outputs(i), ref)))
    pending.append(pipe.launch_next())
for i in pending:
    pipe.wait(i)
    bad += int(not all(torch.equal(a, b) for a, b in zip(pipe.outputs(i), ref)))
print(f"{steps} steps, {lanes} lanes in flight, B=64: steps whose outputs differ from the single-lane reference: {bad} (max |diff| {worst:.3g})")
sys.exit(1 if bad else 0)
