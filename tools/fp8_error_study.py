#!/usr/bin/env python3
"""What fp8 GEMM operands would cost in accuracy on BASELINE configs[4]'s per-draw part (fusion_proj -> encoder -> decoder -> post_projector),
measured on the CPU oracle by quantising both operands of every nn.Linear of that part -- the numbers behind DESIGN.md's decision not to ship
an fp8 mode.  Three operand formats:
  bf16    plain bfloat16 (the library's fast mode, for scale)
  e4m3    OCP e4m3 with one scale per TENSOR (round 2's study)
  mxfp8   the gfx950-native block-scaled form (v_mfma_scale_f32_16x16x128_f8f6f4): OCP MX, e4m3 elements with one shared power-of-two (e8m0)
          scale per 32 consecutive K elements of each row, scale = 2^(floor(log2(max |v|)) - 8), elements saturated to +-448
CPU only.

    python tools/fp8_error_study.py [clips=16]"""
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import torch
import torch.nn.functional as TF

from emotiongestures_amd.builders import build_mirror, clip_rel_l2
from emotiongestures_amd.harness import calculate_frechet_distance
from emotiongestures_amd.synth import load_synth_weights, synth_inputs
from oracle import emogest_oracle as O

n = int(sys.argv[1]) if len(sys.argv) > 1 else 16
model = build_mirror("spatial", 34, 126, 4, 4, seed=11)
sd = {k: v.detach() for k, v in model.state_dict().items()}
inp = synth_inputs(n, 34, 126, 4, seed=11)
t = {k: torch.from_numpy(v) for k, v in inp.items()}
MODE = {"q": None}


def quant(x):
    if MODE["q"] == "e4m3":
        s = x.abs().max().clamp_min(1e-30) / 448.0
        return (x / s).to(torch.float8_e4m3fn).float() * s
    if MODE["q"] == "mxfp8":
        shp = x.shape
        k = shp[-1]
        pad = (-k) % 32
        v = torch.nn.functional.pad(x, (0, pad)).reshape(-1, (k + pad) // 32, 32)
        amax = v.abs().amax(-1, keepdim=True).clamp_min(2.0 ** -126)
        scale = torch.exp2(torch.floor(torch.log2(amax)) - 8.0)                  # e8m0: a power of two per 32-element block (OCP MX v1.0 6.3)
        q = (v / scale).clamp(-448.0, 448.0).to(torch.float8_e4m3fn).float() * scale
        return q.reshape(*shp[:-1], k + pad)[..., :k]
    if MODE["q"] == "bf16":
        return x.bfloat16().float()
    return x


orig_lin = O._lin


def lin_q(sd_, p, x):
    if MODE["q"] and p.split(".")[0] in ("fusion_proj", "encoder", "decoder", "post_projector"):
        return TF.linear(quant(x), quant(sd_[p + ".weight"]), sd_.get(p + ".bias"))
    return orig_lin(sd_, p, x)


O._lin = lin_q
res = {}
with torch.no_grad():
    for mode in (None, "bf16", "e4m3", "mxfp8"):
        MODE["q"] = mode
        res[mode] = O.generator_forward(sd, O.GenCfg(), t["spec"], t["text"], t["pre_pose"], t["sampled"])[0].numpy()
O._lin = orig_lin
# FGD auto-encoder features on the CPU oracle (model/FGD.py:26-82)
from emotiongestures_amd.harness import MLP_Reconstruct
ae_sd = {k: v.detach() for k, v in load_synth_weights(MLP_Reconstruct(pose_dim=126), 5).state_dict().items()}
feat = lambda p: O.fgd_autoencoder(ae_sd, torch.from_numpy(p))[1].reshape(-1, 512).numpy().astype(np.float64)
f0 = feat(res[None])
for mode in ("bf16", "e4m3", "mxfp8"):
    f1 = feat(res[mode])
    fgd = float(np.real(calculate_frechet_distance(f0.mean(0), np.cov(f0, rowvar=False), f1.mean(0), np.cov(f1, rowvar=False))))
    e = np.linalg.norm((res[mode] - res[None]).reshape(n, -1), axis=1) / np.linalg.norm(res[None].reshape(n, -1), axis=1)
    print(f"{mode:5s} operands in the per-draw transformer part: pose rel-L2 per clip median {np.median(e):.2e} max {e.max():.2e}  "
          f"(bar 1e-3)   Frechet distance of FGD features vs fp32 poses {fgd:.3e}")
