#!/usr/bin/env python3
"""Generate tests/golden/attention_mask.npz from the REFERENCE's Full_model.SubLayers.MultiHeadAttention with its mask argument (build container
only: needs /root/reference).  The gesture path never passes a mask (Models_spatial_memory.py:574,611); the classes accept one
(Modules.py:18-19, SubLayers.py:44-45), so the mirror honours it and this pins the semantics: a padding mask [B, 1, Lk] (broadcast over the
queries), a full [B, Lq, Lk] mask (causal), and a row with every key masked (-1e9 everywhere: uniform over ALL keys).

    python tests/golden/make_golden_mask.py
"""
import os
import sys

import numpy as np
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests", "golden"))
REF = "/root/reference"

from emotiongestures_amd.synth import hash_unit, load_synth_weights  # noqa: E402
from make_golden_grad import stub  # noqa: E402

B, LQ, LK, D, H = 3, 34, 40, 512, 8
SEED = 31


def inputs():
    q = ((hash_unit("mask.q", B * LQ * D, SEED) * 2 - 1)).astype(np.float32).reshape(B, LQ, D)
    kv = ((hash_unit("mask.kv", B * LK * D, SEED) * 2 - 1)).astype(np.float32).reshape(B, LK, D)
    pad = np.ones((B, 1, LK), np.int64)
    pad[0, 0, 30:] = 0                      # clip 0: the last 10 keys are padding
    pad[1, 0, ::3] = 0                      # clip 1: every third key masked
    pad[2, 0, :] = 0                        # clip 2: EVERY key masked
    full = np.ones((B, LQ, LK), np.int64)
    for i in range(LQ):
        full[:, i, i + 1:] = 0              # causal: query i sees keys 0..i
    full[1, 5, :] = 0                       # one fully masked row
    return q, kv, pad, full


def main():
    stub()
    sys.path.insert(0, REF)
    os.chdir(REF)
    from Full_model.SubLayers import MultiHeadAttention
    mha = MultiHeadAttention(H, D, 64, 64, dropout=0.2).eval()
    load_synth_weights(mha, SEED)
    q, kv, pad, full = inputs()
    out = {}
    with torch.no_grad():
        for name, m in (("pad", pad), ("full", full)):
            y, attn = mha(torch.from_numpy(q), torch.from_numpy(kv), torch.from_numpy(kv), mask=torch.from_numpy(m))
            out[f"{name}/out"], out[f"{name}/attn"] = y.numpy()[:, :, ::4].copy(), attn.numpy()[:, ::4].copy()        # every 4th feature; heads 0 and 4
            print(name, float(y.abs().mean()), attn.shape, float(attn[2, 0, 0].max()) if name == "pad" else "")
    path = os.path.join(ROOT, "tests", "golden", "attention_mask.npz")
    np.savez_compressed(path, **out)
    print("wrote", path, os.path.getsize(path) // 1024, "KiB")


if __name__ == "__main__":
    main()
