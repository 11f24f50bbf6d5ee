#!/usr/bin/env python3
"""Would a TWO-MFMA product (fp16 operands on v_mfma_f32_16x16x32_f16) keep the pose inside the 1e-3 bar?  (round-5 verdict, item 2)

The shipped arithmetic is `bf16x3`: x = x_hi + x_lo, w = w_hi + w_lo in bfloat16, product = x_hi w_hi + x_hi w_lo + x_lo w_hi with fp32 accumulation
(three MFMAs per tile step).  fp16 carries 11 significand bits against bfloat16's 8, so two-term schemes are conceivable:

  f16_xw2   activations ONE fp16 value, weights (hi + lo) fp16          product = x_h w_hi + x_h w_lo      (error: the rounding of x, 2^-12 rms-ish)
  f16_x2w   activations (hi + lo) fp16, weights ONE fp16 value          product = x_hi w_h + x_lo w_h      (error: the rounding of w)
  f16       both single fp16 (one MFMA), for scale
  bf16x3    the shipped arithmetic, emulated the same way (calibrates the emulation against the measured 4.9e-5)
  bf16      single bfloat16 (one MFMA), for scale (measured on the GPU: 3.4e-2)

Everything runs on the CPU oracle (oracle/emogest_oracle.py) with the operands of every MFMA-side contraction quantised as the scheme says and the
products accumulated in fp32 by torch: the 3x3 / 1x1 convolutions of the audio tower from 32 input channels up, every Linear, the TCN's conv1d.  The
stem (1 -> 32, fp32 VALU in the library) and the attention products (two activation operands, 0.4 % of the FLOPs) stay exact.  `where` restricts a
scheme to a part of the network, the rest runs bf16x3:
  all        every contraction above
  no_stage1  everything but the six 32 -> 32 convolutions of tower stage 1 (HBM / latency bound: fewer MFMAs buy nothing there)
  tower23    only tower stages 2-3 (64 / 128 channels)
  xformer    only the Linears (transformer, projections, heads)

Reported per configuration: per-clip pose rel-L2 (median / max over the clips) against the exact-fp32 oracle, max |d| of the emotion logits, and the
Frechet distance between FGD-autoencoder features of the two pose sets.  Go = max pose error <= 2e-4 everywhere (5 x inside the 1e-3 bar).

    python tools/f16x2_study.py [clips=16] [ted|beat_long]"""
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import torch
import torch.nn.functional as TF

from emotiongestures_amd.builders import build_mirror
from emotiongestures_amd.harness import MLP_Reconstruct, calculate_frechet_distance
from emotiongestures_amd.synth import load_synth_weights, synth_inputs
from oracle import emogest_oracle as O

n = int(sys.argv[1]) if len(sys.argv) > 1 else 16
shape = sys.argv[2] if len(sys.argv) > 2 else "ted"
if shape == "ted":
    frames, pose_dim, prior, chunk, spec_len = 34, 126, 4, 4, 124
else:               # BEAT-long: 10 s of audio, 120 frames (BASELINE configs[3])
    frames, pose_dim, prior, chunk, spec_len = 120, 282, 10, 10, 312
if shape == "ted":
    model = build_mirror("spatial", frames, pose_dim, prior, chunk, seed=11, spec_len=spec_len)
else:
    from emotiongestures_amd.builders import make_args, make_lang
    from emotiongestures_amd.Full_model.Models_spatial_memory import Transformer
    model = load_synth_weights(Transformer(make_args(chunk), make_lang(200), frames=frames, pose_dim=pose_dim, prior_frames=prior, d_word_vec=512, d_model=512,
                                           d_inner=2048, n_layers=3, n_head=8, d_k=64, d_v=64, n_position=frames, spec_len=spec_len), 11).eval()
sd = {k: v.detach() for k, v in model.state_dict().items()}
inp = synth_inputs(n, frames, pose_dim, prior, spec_len=spec_len, seed=11)
t = {k: torch.from_numpy(v) for k, v in inp.items()}
cfg = O.GenCfg() if shape == "ted" else O.GenCfg(frames=frames, pose_dim=pose_dim, prior_frames=prior, chunk=chunk)


def split(x, dt):
    hi = x.to(dt).float()
    return hi, (x - hi).to(dt).float()


def contract(op, x, w, scheme):
    """op(x, w) -> the product, linear in both arguments (bias added by the caller)."""
    if scheme is None:
        return op(x, w)
    if scheme == "bf16":
        return op(x.bfloat16().float(), w.bfloat16().float())
    if scheme == "f16":
        return op(x.half().float(), w.half().float())
    if scheme == "bf16x3":
        xh, xl = split(x, torch.bfloat16)
        wh, wl = split(w, torch.bfloat16)
        return op(xh, wh) + (op(xh, wl) + op(xl, wh))
    if scheme == "f16_xw2":
        xh = x.half().float()
        wh, wl = split(w, torch.float16)
        return op(xh, wh) + op(xh, wl)
    if scheme == "f16_x2w":
        xh, xl = split(x, torch.float16)
        wh = w.half().float()
        return op(xh, wh) + op(xl, wh)
    raise ValueError(scheme)


STATE = {"scheme": None, "where": "all"}


def scheme_for(kind, cin=0, cout=0):
    """kind: 'conv2d' | 'linear' | 'conv1d'.  The part of the network a scheme does not cover runs the shipped bf16x3."""
    s, where = STATE["scheme"], STATE["where"]
    if s is None:
        return None
    stage1 = kind == "conv2d" and cin == 32 and cout == 32
    inside = {"all": True, "no_stage1": not stage1, "tower23": kind == "conv2d" and not stage1 and cin >= 32,
              "xformer": kind == "linear"}[where]
    return s if inside else "bf16x3"


class FProxy:
    """torch.nn.functional as the oracle sees it, with the MFMA-side contractions quantised."""

    def __getattr__(self, name):
        return getattr(TF, name)

    @staticmethod
    def conv2d(x, w, b=None, stride=1, padding=0):
        if w.shape[1] < 32:            # the stem: fp32 VALU in the library
            return TF.conv2d(x, w, b, stride=stride, padding=padding)
        y = contract(lambda a, c: TF.conv2d(a, c, None, stride=stride, padding=padding), x, w, scheme_for("conv2d", w.shape[1], w.shape[0]))
        return y if b is None else y + b.view(1, -1, 1, 1)

    @staticmethod
    def conv1d(x, w, b=None, stride=1, padding=0, dilation=1):
        y = contract(lambda a, c: TF.conv1d(a, c, None, stride=stride, padding=padding, dilation=dilation), x, w, scheme_for("conv1d"))
        return y if b is None else y + b.view(1, -1, 1)

    @staticmethod
    def linear(x, w, b=None):
        y = contract(lambda a, c: TF.linear(a, c), x, w, scheme_for("linear"))
        return y if b is None else y + b


O.F = FProxy()
ae_sd = {k: v.detach() for k, v in load_synth_weights(MLP_Reconstruct(pose_dim=pose_dim), 5).state_dict().items()}


def run():
    with torch.no_grad():
        out = O.generator_forward(sd, cfg, t["spec"], t["text"], t["pre_pose"], t["sampled"])
    return out[0].numpy(), out[3].numpy()


def feats(p):
    keep = O.F
    O.F = TF
    try:
        with torch.no_grad():
            return O.fgd_autoencoder(ae_sd, torch.from_numpy(p))[1].reshape(-1, 512).numpy().astype(np.float64)
    finally:
        O.F = keep


STATE["scheme"] = None
pose0, logit0 = run()
f0 = feats(pose0)
print(f"# {shape}: {n} clips, {frames} frames x {pose_dim}; reference = exact fp32 oracle; bar 1e-3, go threshold 2e-4")
print(f"{'scheme':9s} {'where':10s} {'MFMAs':>5s}  {'pose rel-L2 median':>18s} {'max':>9s}  {'logits max|d|':>13s}  {'FGD shift':>10s}")
rows = [("bf16x3", "all", 3), ("bf16", "all", 1), ("f16", "all", 1), ("f16_xw2", "all", 2), ("f16_x2w", "all", 2),
        ("f16_xw2", "no_stage1", 2), ("f16_x2w", "no_stage1", 2), ("f16_xw2", "tower23", 2), ("f16_x2w", "tower23", 2),
        ("f16_xw2", "xformer", 2), ("f16_x2w", "xformer", 2)]
for scheme, where, mf in rows:
    STATE["scheme"], STATE["where"] = scheme, where
    pose, logit = run()
    e = np.linalg.norm((pose - pose0).reshape(n, -1), axis=1) / np.linalg.norm(pose0.reshape(n, -1), axis=1)
    try:
        f1 = feats(pose)
        fgd = float(np.real(calculate_frechet_distance(f0.mean(0), np.cov(f0, rowvar=False), f1.mean(0), np.cov(f1, rowvar=False))))
    except Exception as ex:          # the FGD auto-encoder is built for one pose layout; report what failed instead of a number
        fgd = float("nan")
    verdict = "go" if e.max() <= 2e-4 else ("inside bar" if e.max() <= 1e-3 else "OVER BAR")
    print(f"{scheme:9s} {where:10s} {mf:5d}  {np.median(e):18.2e} {e.max():9.2e}  {np.abs(logit - logit0).max():13.2e}  {fgd:10.3e}  {verdict}", flush=True)
