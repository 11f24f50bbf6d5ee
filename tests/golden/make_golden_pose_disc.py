#!/usr/bin/env python3
"""Generate tests/golden/pose_disc.npz from the REFERENCE's Full_model.Models_spatial_memory.Pose_Discriminator (:671-704), on CPU
(build container only: needs /root/reference, which never ships).

    python tests/golden/make_golden_pose_disc.py

The class's head is `Linear(282, 64)` on the encoder output and its encoder adds a d_word_vec-wide positional table to the raw poses, so
its forward runs only with d_word_vec = d_model = 282 (the upstream defaults, 128, fail at the first add -- SURVEY §0 lists the same defect
for Motion_Discriminator); that one consistent choice is what is captured: 3 encoder layers, 8 heads of 64, d_inner 1024, 60 positions,
poses [2, 60, 282].
  eval/out                 eval()-mode probabilities [2, 60, 1]
  train/out, train/loss    train() mode with every nn.Dropout p = 0 (gradient-parity configuration); loss = smooth_l1(out, 1)
  train/dx/*, train/g/*    input gradient and per-parameter gradient fingerprints (norm, sum, strided sample) as make_golden_grad.py
"""
import json
import os
import sys

import numpy as np
import torch
import torch.nn as nn
import torch.nn.functional as F

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests", "golden"))
REF = "/root/reference"

from emotiongestures_amd.synth import hash_unit, load_synth_weights  # noqa: E402
from make_golden_grad import fingerprint, NS  # noqa: E402
from make_golden_training_types import _stubs  # noqa: E402

CFG = dict(d_word_vec=282, d_model=282, d_inner=1024, n_layers=3, n_head=8, d_k=64, d_v=64, n_position=60)
SEED = 23


def poses():
    return ((hash_unit("pose_disc.x", 2 * 60 * 282, SEED) * 2 - 1) * 0.5).astype(np.float32).reshape(2, 60, 282)


def main():
    torch.manual_seed(0)
    torch.set_num_threads(8)
    _stubs()
    sys.path.insert(0, REF)
    os.chdir(REF)
    from Full_model.Models_spatial_memory import Pose_Discriminator
    out = {}
    pd = Pose_Discriminator(**CFG)
    load_synth_weights(pd, SEED)
    json.dump([[k, list(v.shape)] for k, v in pd.state_dict().items()], open(os.path.join(ROOT, "tests", "golden", "pose_disc_schema.json"), "w"))
    x = torch.from_numpy(poses())
    pd.eval()
    with torch.no_grad():
        out["eval/out"] = pd(x).numpy()
    pd.train()
    for mod in pd.modules():
        if isinstance(mod, nn.Dropout):
            mod.p = 0.0
    xr = x.clone().requires_grad_(True)
    prob = pd(xr)
    loss = F.smooth_l1_loss(prob, torch.ones_like(prob))
    loss.backward()
    out["train/out"], out["train/loss"] = prob.detach().numpy(), np.float64(loss.item())
    g = xr.grad.reshape(-1).double().numpy()
    out["train/dx/norm"], out["train/dx/sample"] = np.float64(np.linalg.norm(g)), g[:: max(1, g.size // NS)][:NS].astype(np.float32)
    fingerprint(out, "train", pd)
    print("eval out", out["eval/out"].ravel()[:4], "loss", loss.item(), "without grad", list(out["train/nograd"]))
    path = os.path.join(ROOT, "tests", "golden", "pose_disc.npz")
    np.savez_compressed(path, **out)
    print("wrote", path, os.path.getsize(path) // 1024, "KiB")


if __name__ == "__main__":
    main()
