#!/usr/bin/env python3
"""Generate tests/golden/memory_grads.npz: forward values and gradients of the reference's SP_Memory_Net_v1 and TM_Memory_Net modules
(Full_model/Models_memory.py:215-293) run STANDALONE under torch autograd (build container only: needs /root/reference).

    python tests/golden/make_golden_memory_grad.py

Inside the synthetically initialised generator TM_Memory_Net's softmax saturates (its gradients are ~1e-7: tests/golden/grads.npz,
case genmem), so the modules' backward passes are pinned here on inputs that keep the sigmoid gate and the softmax in their
non-saturated range: hash-synthesised prior poses / predicted frames in +-0.5, module weights from emotiongestures_amd.synth scaled by
0.2, a hash-synthesised upstream gradient.  TM_Memory_Net couples the clips of the batch (score = mem (mem^T enc), :288-289): the batch is
fixed at 4.  Stored: outputs and gradients with respect to both inputs (full arrays), fingerprints of every parameter gradient.
Stand-ins only for imports unused on this path (torch_dct, torchvision*, umap, fasttext); dropout p = 0.
"""
import os
import sys
from types import SimpleNamespace

import numpy as np
import torch
import torch.nn as nn

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests", "golden"))

from make_golden_grad import stub  # noqa: E402
from emotiongestures_amd.synth import hash_unit, load_synth_weights  # noqa: E402

B, P, PRED, D, CHUNK, SEED = 4, 4, 30, 126, 4, 5


def inputs():
    prior = (hash_unit("mem.prior", B * P * D, SEED) - 0.5).astype(np.float32).reshape(B, P, D)
    pred = (hash_unit("mem.pred", B * PRED * D, SEED) - 0.5).astype(np.float32).reshape(B, PRED, D)
    g = (hash_unit("mem.g", B * PRED * D, SEED) - 0.5).astype(np.float32).reshape(B, PRED, D)
    return prior, pred, g


def main():
    torch.manual_seed(0)
    stub()
    from Full_model.Models_memory import SP_Memory_Net_v1, TM_Memory_Net
    args = SimpleNamespace(chunk=CHUNK)
    prior_np, pred_np, g_np = inputs()
    out = {"meta": np.asarray([B, P, PRED, D, CHUNK, SEED], np.int64)}
    for name, cls in (("sp", SP_Memory_Net_v1), ("tm", TM_Memory_Net)):
        m = cls(args, P, PRED, D, 512)
        load_synth_weights(m, SEED)
        with torch.no_grad():
            for p in m.parameters():
                p.mul_(0.2)
        m.train()
        for mod in m.modules():
            if isinstance(mod, nn.Dropout):
                mod.p = 0.0
        prior = torch.from_numpy(prior_np).clone().requires_grad_(True)
        pred0 = torch.from_numpy(pred_np).clone().requires_grad_(True)
        pred = pred0 * 1.0                      # the modules write into their `pred_feature` argument in place: hand them a non-leaf
        y = m(prior, pred)
        y.backward(torch.from_numpy(g_np))
        out[f"{name}/out"] = y.detach().numpy()
        out[f"{name}/dprior"] = prior.grad.numpy()
        out[f"{name}/dpred"] = pred0.grad.numpy()
        for k, p in m.named_parameters():         # fingerprints as in make_golden_grad.py: L2 norm, sum, 64-value strided sample
            gv = p.grad.detach().reshape(-1).double().numpy()
            stride = max(1, gv.size // 64)
            out[f"{name}/p/{k}/norm"] = np.float64(np.linalg.norm(gv))
            out[f"{name}/p/{k}/sum"] = np.float64(gv.sum())
            out[f"{name}/p/{k}/sample"] = gv[::stride][:64]
        d = np.abs(y.detach().numpy()[:, :CHUNK] - pred_np[:, :CHUNK]).mean()
        print(name, "mean |out - pred| on the first chunk", float(d), " |dprior|", float(prior.grad.norm()), " parameter gradient norms",
              [round(float(p.grad.norm()), 4) for p in m.parameters()])
    path = os.path.join(ROOT, "tests", "golden", "memory_grads.npz")
    np.savez_compressed(path, **out)
    print("wrote", path, os.path.getsize(path) // 1024, "KiB")


if __name__ == "__main__":
    main()
