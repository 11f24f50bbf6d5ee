#!/usr/bin/env python3
"""Per-parameter gradient error of one generator training step (HIP path vs the CPU oracle's autograd), in network order.
usage: python tools/train_grad_report.py [batch=2]"""
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import torch

from emotiongestures_amd.builders import build_mirror
from emotiongestures_amd.synth import hash_unit, synth_inputs
from emotiongestures_amd.train import functional as F
from oracle import emogest_oracle as O

B = int(sys.argv[1]) if len(sys.argv) > 1 else 2
dev = "cuda:0"
model = build_mirror("spatial", 34, 126, 4, 4, seed=0, precision="f32")
sd = {k: v.detach().clone() for k, v in model.state_dict().items()}
for v in sd.values():
    if v.is_floating_point():
        v.requires_grad_(True)
inp = synth_inputs(B, 34, 126, 4, seed=0)
target = torch.from_numpy((hash_unit("train.target_pose", B * 34 * 126, 0) - 0.5).astype(np.float32).reshape(B, 34, 126))
label = torch.from_numpy(inp["label"]).argmax(1)
loss_ref, _, _ = O.generator_train_loss(sd, O.GenCfg(), torch.from_numpy(inp["spec"]), torch.from_numpy(inp["text"]),
                                        torch.from_numpy(inp["pre_pose"]), target, label)
loss_ref.backward()
# a second oracle pass in float64 shows how much of the difference is fp32 round-off of the ORACLE itself
sd64 = {k: (v.detach().double().requires_grad_(True) if v.is_floating_point() else v) for k, v in sd.items()}
l64, _, _ = O.generator_train_loss(sd64, O.GenCfg(), torch.from_numpy(inp["spec"]).double(), torch.from_numpy(inp["text"]),
                                   torch.from_numpy(inp["pre_pose"]).double(), target.double(), label)
l64.backward()
model.to(dev).train()
pose, emo, sem, pred, txt = model(torch.from_numpy(inp["spec"]).to(dev), torch.from_numpy(inp["text"]).to(dev),
                                  torch.from_numpy(inp["pre_pose"]).to(dev), None)
loss = F.add(F.smooth_l1_loss(pose, target.to(dev), 1.0, 100.0), F.cross_entropy(pred, label.to(dev)))
loss.backward()
print(f"loss hip {float(loss):.6f}  oracle f32 {float(loss_ref):.6f}  oracle f64 {float(l64):.6f}")
rel = lambda a, b: float((a.double().cpu() - b.double()).norm() / b.double().norm().clamp_min(1e-30))
print(f"{'parameter':70s} {'hip vs f64':>11s} {'oracle32 vs f64':>16s}")
for k, p in model.named_parameters():
    if sd[k].grad is None or p.grad is None:
        continue
    print(f"{k:70s} {rel(p.grad, sd64[k].grad):11.2e} {rel(sd[k].grad, sd64[k].grad):16.2e}")
