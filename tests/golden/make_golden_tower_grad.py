#!/usr/bin/env python3
"""Generate tests/golden/tower_grads.npz: block-level gradient goldens for EVERY site of the audio tower -- the stem, the 13
SEBasicBlocks and final_conv1 -- from the REFERENCE's own modules (build container only: needs /root/reference, which never ships).

    python tests/golden/make_golden_tower_grad.py

Why block level: the end-to-end gradient test can only bound the ReLU / BatchNorm tower statistically (one flipped ReLU mask element
between two fp32 forwards changes everything upstream of it).  Fed IDENTICAL inputs, a block has no such excuse.  So, per site:

  1. one training step of the reference's Transformer (Models_spatial_memory, TED shapes, B = 2, .train(), dropout p = 0, loss =
     100 smooth_l1 + CE: the step of make_golden_grad.py) is run with forward / backward hooks that capture the site's REAL input
     activation and the REAL upstream gradient at its output;
  2. a crop of both (the site's geometry keeps ragged tile edges: widths 44 / 38 / 31) is rounded to fp16 -- the gradient after scaling
     by a power of two -- and stored: both sides then start from bit-identical fp32 values;
  3. the reference's module for that site (ResNetBlocks.SEBasicBlock with its SELayer and downsample; conv1 -> ReLU -> bn1 for the stem,
     final_conv1 -> bn1) is run in float64 on those stored tensors, train mode; the file holds fingerprints (L2 norm, sum, 64-value
     strided sample) of the output, the input gradient and every parameter gradient, plus the bit-packed ReLU masks (conv1's ReLU and
     the block's final ReLU), so that the test can tell a genuine error from a mask element the fp32 path decides differently.

Weights come from emotiongestures_amd.synth (integer hash), exactly as in make_golden_grad.py; stand-ins only for imports unused on this
path (torch_dct, torchvision*, umap, fasttext).
"""
import os
import sys
from types import SimpleNamespace

import numpy as np
import torch
import torch.nn as nn
import torch.nn.functional as F

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests", "golden"))

from make_golden_grad import stub, train_targets  # noqa: E402
from emotiongestures_amd.synth import load_synth_weights, synth_inputs  # noqa: E402

NS = 64
# (rows, cols) of the input crop per stage; the stride-2 entry blocks take their crop from the previous stage's map
CROP = {1: (12, 44), 2: (10, 38), 3: (8, 31)}
CROP_S2 = {2: (20, 44), 3: (16, 38)}          # -> outputs 10 x 22 and 8 x 19
ORIGIN = {1: (40, 30), 2: (20, 12), 3: (12, 0)}


def fp(out, key, t):
    v = t.detach().reshape(-1).double().numpy()
    stride = max(1, v.size // NS)
    out[key + "/norm"] = np.float64(np.linalg.norm(v))
    out[key + "/sum"] = np.float64(v.sum())
    out[key + "/sample"] = v[::stride][:NS].astype(np.float64)


def pow2_scale(t):
    m = float(t.abs().max())
    return 1.0 if m == 0 else 2.0 ** np.floor(np.log2(1024.0 / m))        # largest value lands in [1024, 2048): far from fp16's limits


def site_list(enc):
    sites = [("stem", None)]
    for li, layer in enumerate((enc.layer1, enc.layer2, enc.layer3)):
        for bi, blk in enumerate(layer):
            sites.append((f"layer{li + 1}.{bi}", blk))
    sites.append(("final", None))
    return sites


def main():
    torch.manual_seed(0)
    torch.set_num_threads(8)
    stub()
    from Full_model.Models_spatial_memory import Transformer
    seed, batch = 0, 2
    args = SimpleNamespace(chunk=4, hidden_size=300, n_layers=3, freeze_wordembed=False, wordembed_dim=300, dropout_prob=0.1)
    lang = SimpleNamespace(n_words=200, word_embedding_weights=None)
    m = Transformer(args, lang, frames=34, pose_dim=126, prior_frames=4, d_word_vec=512, d_model=512, d_inner=2048, n_layers=3,
                    n_head=8, d_k=64, d_v=64)
    load_synth_weights(m, seed)
    m.train()
    for mod in m.modules():
        if isinstance(mod, nn.Dropout):
            mod.p = 0.0
    ae = m.audio_encoder
    enc = ae.feat_extractor

    # ---- 1. the real step, with the sites' inputs / upstream gradients captured --------------------------------------------------
    cap = {}

    def hook_io(name):
        def fwd(_mod, inp, outp):
            cap[name + "/x"] = inp[0].detach().clone()
            outp.register_hook(lambda g, n=name: cap.__setitem__(n + "/g", g.detach().clone()))
        return fwd

    handles = []
    for name, blk in site_list(enc):
        if blk is not None:
            handles.append(blk.register_forward_hook(hook_io(name)))
    # stem: input = the spectrogram (conv1's input), output = bn1's output; final: input of final_conv1, output of ae.bn1
    handles.append(enc.conv1.register_forward_hook(lambda _m, i, o: cap.__setitem__("stem/x", i[0].detach().clone())))
    def grad_tap(key):
        def fwd(_mod, _inp, outp):
            outp.register_hook(lambda g: cap.__setitem__(key, g.detach().clone()))
        return fwd                  # returns None: a forward hook's return value would replace the module's output

    handles.append(enc.bn1.register_forward_hook(grad_tap("stem/g")))
    handles.append(ae.final_conv1.register_forward_hook(lambda _m, i, o: cap.__setitem__("final/x", i[0].detach().clone())))
    handles.append(ae.bn1.register_forward_hook(grad_tap("final/g")))

    inp = synth_inputs(batch, 34, 126, 4, seed=seed)
    target = torch.from_numpy(train_targets(batch, 34, 126, seed))
    label = torch.from_numpy(inp["label"]).argmax(1)
    pose, emo, sem, pred, txt = m(torch.from_numpy(inp["spec"]), torch.from_numpy(inp["text"]), torch.from_numpy(inp["pre_pose"]), None)
    loss = 100.0 * F.smooth_l1_loss(pose, target) + F.cross_entropy(pred, label)
    loss.backward()
    for h in handles:
        h.remove()
    print("step loss", loss.item())

    # ---- 2. + 3. per site: crop, round to fp16, run the reference's module in float64 -----------------------------------------------
    out = {"meta": np.asarray([batch, seed], np.int64), "sites": np.array([n for n, _ in site_list(enc)])}
    m.zero_grad(set_to_none=True)
    m.double()
    for name, blk in site_list(enc):
        x_full, g_full = cap[name + "/x"], cap[name + "/g"]
        if name == "stem":
            stage, stride = 1, 1
            (h, w), (r0, c0) = (32, 60), (8, 10)
        elif name == "final":
            stage, stride = 3, 1
            (h, w), (r0, c0) = CROP[3], ORIGIN[3]
        else:
            stage = int(name[5])
            stride = blk.conv1.stride[0]
            (h, w) = CROP_S2[stage] if stride == 2 else CROP[stage]
            (r0, c0) = ORIGIN[stage - 1] if stride == 2 else ORIGIN[stage]
            if stride == 2:
                r0, c0 = r0 // 2 * 2, c0 // 2 * 2
        x = x_full[:, :, r0:r0 + h, c0:c0 + w].contiguous()
        ho, wo = (h - 1) // stride + 1, (w - 1) // stride + 1
        g = g_full[:, :, r0 // stride:r0 // stride + ho, c0 // stride:c0 // stride + wo].contiguous()
        assert g.shape[2:] == (ho, wo), (name, g.shape, ho, wo)
        gs = pow2_scale(g)
        x16 = x.half()
        g16 = (g * gs).half()
        out[f"{name}/x"] = x16.numpy()                        # NCHW fp16
        out[f"{name}/g"] = g16.numpy()
        out[f"{name}/g_scale"] = np.float64(gs)
        out[f"{name}/stride"] = np.int64(stride)
        xd = x16.double().requires_grad_(True)
        gd = g16.double() / gs
        masks = {}
        if name == "stem":
            r1 = F.relu(enc.conv1(xd))
            y = enc.bn1(r1)
            params = {"conv1.weight": enc.conv1.weight, "conv1.bias": enc.conv1.bias, "bn1.weight": enc.bn1.weight, "bn1.bias": enc.bn1.bias}
            masks["r1"] = r1.detach() > 0
        elif name == "final":
            y = ae.bn1(ae.final_conv1(xd))
            params = {"final_conv1.weight": ae.final_conv1.weight, "final_conv1.bias": ae.final_conv1.bias, "bn1.weight": ae.bn1.weight, "bn1.bias": ae.bn1.bias}
        else:
            r1h = []
            hh = blk.relu.register_forward_hook(lambda _m, i, o: r1h.append(o.detach() > 0))      # first call = conv1's ReLU, second = the block's last
            y = blk(xd)
            hh.remove()
            masks["r1"], masks["out"] = r1h[0], r1h[1]
            params = {k: p for k, p in blk.named_parameters()}
        for p in params.values():
            p.grad = None
        y.backward(gd)
        fp(out, f"{name}/out", y)
        fp(out, f"{name}/dx", xd.grad)
        for k, p in params.items():
            fp(out, f"{name}/p/{k}", p.grad)
        for k, mk in masks.items():
            out[f"{name}/mask/{k}"] = np.packbits(mk.numpy().reshape(-1))
        print(f"{name:10s} x {tuple(x.shape)} g {tuple(g.shape)} scale 2^{int(np.log2(gs))}  |out| {float(y.norm()):.4e}  |dx| {float(xd.grad.norm()):.4e}  "
              f"{len(params)} parameter gradients")
    path = os.path.join(ROOT, "tests", "golden", "tower_grads.npz")
    np.savez_compressed(path, **out)
    print("wrote", path, os.path.getsize(path) // 1024, "KiB")


if __name__ == "__main__":
    main()
