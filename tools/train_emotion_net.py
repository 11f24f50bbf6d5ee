#!/usr/bin/env python3
"""The reference's one training loop (train_audio_classifier_K_fold.py:109-200) on the HIP training path, synthetic data:
EmotionNet in train() mode, 100 x FocalLoss(alpha from class counts, gamma 2), Adam(lr, betas=(0.5, 0.999), weight_decay=1e-5),
periodic validation accuracy (compute_acc, :57-62) in eval() mode on the inference kernels.

    python tools/train_emotion_net.py [--steps 30] [--batch 8] [--lr 1e-4]

Synthetic task: the label is encoded in the spectrogram (a per-class band offset), so the loss must fall and accuracy rise."""
import argparse
import os
import sys
import time

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import torch

from emotiongestures_amd.model.audio_emotion_classifer import EmotionNet
from emotiongestures_amd.synth import hash_unit, load_synth_weights
from emotiongestures_amd.train import functional as F
from emotiongestures_amd.train.optim import FlatAdam, flatten_parameters

ap = argparse.ArgumentParser()
ap.add_argument("--steps", type=int, default=30)
ap.add_argument("--batch", type=int, default=8)          # the script's FocalLoss broadcasts its 8 class weights over the batch axis: batch 8
ap.add_argument("--lr", type=float, default=1e-4)
args = ap.parse_args()
dev = torch.device("cuda:0")


def batch(step, n):
    lab = torch.from_numpy((hash_unit("lab", n, step) * 8).astype(np.int64) % 8)
    x = (-80.0 * hash_unit("x", n * 128 * 128, step)).astype(np.float16).astype(np.float32).reshape(n, 128, 128)
    x = torch.from_numpy(x)
    for i, l in enumerate(lab.tolist()):
        x[i, 16 * l:16 * l + 16, :] += 30.0
    return x.clamp_(-80, 0), lab


net = load_synth_weights(EmotionNet(precision="f32"), 31).to(dev).train()
fp = flatten_parameters(net)
opt = FlatAdam(fp, lr=args.lr, betas=(0.5, 0.999), weight_decay=1e-5)            # :128
class_count = np.ones(8)
train_s = 0.0
for step in range(args.steps):
    t0 = time.perf_counter()
    x, lab = batch(step, args.batch)
    for l in lab.tolist():
        class_count[l] += 1
    class_weights = class_count.sum() / (len(class_count) * class_count)       # :146-148
    alpha = torch.tensor(class_weights[:args.batch] if args.batch == 8 else [1.0] * args.batch, dtype=torch.float32)
    net.train()
    opt.zero_grad()
    loss = F.focal_loss(net(x.to(dev)), lab.to(dev), alpha, 2.0, 100.0)         # criterion(output, label) * 100  (:168)
    loss.backward()
    opt.step()
    torch.cuda.synchronize()
    train_s += time.perf_counter() - t0
    if step % 5 == 4 or step == args.steps - 1:
        tv = time.perf_counter()
        net.eval()              # the inference engine repacks the updated weights on the host (the 65536 x 4096 first Linear dominates)
        with torch.no_grad():
            vx, vl = batch(10_000 + step, 16)
            acc = float((net(vx.to(dev)).argmax(1).cpu() == vl).float().mean())
        print(f"step {step + 1:3d}  loss {float(loss.detach()):9.3f}  val acc {100 * acc:5.1f} %  ({train_s / (step + 1) * 1e3:.0f} ms per training step; validation incl. repack {time.perf_counter() - tv:.1f} s)")
