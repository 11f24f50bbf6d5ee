#!/usr/bin/env python3
"""Generate tests/golden/dropout_grads.npz: one training step of the REFERENCE's generator with its nn.Dropout layers ACTIVE, under masks that are
not torch's but the HIP library's own counter-based stream (round-5 verdict item 4) -- build container only (needs /root/reference).

    python tests/golden/make_golden_dropout_grad.py

How: `oracle.dropout_site_plan` lists the Dropout modules on the path to the losses by the name they have in the reference's `named_modules()`, in
the order the library's train-mode forward visits them; `oracle.dropout_plan_masks` computes the library's keep decisions for that order from the
integer restatement of its hash (no GPU).  Here every nn.Dropout of the reference model is told to multiply by the mask filed under ITS OWN module
name (shape and p are checked against the live module and tensor); Dropout modules outside the plan (text branch, SP_Memory_Net_v2: they feed no
loss) are the identity.  Stored: loss, pose, emotion logits, per-parameter gradient fingerprints (as tests/golden/make_golden_grad.py), the plan's
(offset, numel) list and the seed.  tests/test_training_oracle.py pins the ORACLE's Dropout placements against this file on the CPU;
tests/test_gpu_training.py pins the HIP step against the oracle (every gradient element-wise) and against this file's loss / outputs."""
import os
import sys
from types import SimpleNamespace

import numpy as np
import torch
import torch.nn as nn
import torch.nn.functional as F

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(os.path.dirname(HERE))
sys.path.insert(0, ROOT)
sys.path.insert(0, HERE)

from make_golden_grad import fingerprint, stub, train_targets  # noqa: E402
from emotiongestures_amd.synth import load_synth_weights, synth_inputs  # noqa: E402
from oracle import emogest_oracle as O  # noqa: E402

SEED_W, SEED_MASK, BATCH = 0, 4321, 2


def main():
    stub()
    from Full_model.Models_spatial_memory import Transformer
    args = SimpleNamespace(chunk=4, hidden_size=300, n_layers=3, freeze_wordembed=False, wordembed_dim=300, dropout_prob=0.1)
    lang = SimpleNamespace(n_words=200, word_embedding_weights=None)
    m = Transformer(args, lang, frames=34, pose_dim=126, prior_frames=4, d_word_vec=512, d_model=512, d_inner=2048, n_layers=3,
                    n_head=8, d_k=64, d_v=64)
    load_synth_weights(m, SEED_W)
    m.train()
    plan = O.dropout_site_plan(O.GenCfg(), BATCH)
    masks, where = O.dropout_plan_masks(plan, SEED_MASK)
    p_of = {site: p for site, _shape, p in plan}
    names = {mod: name for name, mod in m.named_modules() if isinstance(mod, nn.Dropout)}
    assert set(p_of) <= set(names.values()), sorted(set(p_of) - set(names.values()))
    used = []

    def forward(self, x):
        name = names[self]
        if name not in masks:
            return x                              # off the loss path (text branch, SP_Memory_Net_v2's discarded result)
        assert self.training and abs(self.p - p_of[name]) < 1e-12, (name, self.p, p_of[name])
        assert tuple(x.shape) == tuple(masks[name].shape), (name, tuple(x.shape), tuple(masks[name].shape))
        used.append(name)
        return x * masks[name]

    nn.Dropout.forward = forward
    inp = synth_inputs(BATCH, 34, 126, 4, seed=SEED_W)
    target = torch.from_numpy(train_targets(BATCH, 34, 126, SEED_W))
    label = torch.from_numpy(inp["label"]).argmax(1)
    pose, emo, sem, pred, txt = m(torch.from_numpy(inp["spec"]), torch.from_numpy(inp["text"]), torch.from_numpy(inp["pre_pose"]), None)
    loss = 100.0 * F.smooth_l1_loss(pose, target) + F.cross_entropy(pred, label)
    loss.backward()
    assert sorted(used) == sorted(p_of), (sorted(set(p_of) - set(used)), len(used))          # every planned site ran exactly once
    out = {"gen/loss": np.float64(loss.item()), "gen/pose": pose.detach().numpy(), "gen/emotion_prediction": pred.detach().numpy(),
           "gen/emotion_feature": emo.detach().numpy()[:, ::4, ::16].copy(),
           "gen/meta": np.asarray([BATCH, SEED_W, SEED_MASK], np.int64), "gen/sites": np.asarray(where, np.int64),
           "gen/site_names": np.array([s for s, _sh, _p in plan]), "gen/reference_call_order": np.array(used)}
    fingerprint(out, "gen", m)
    path = os.path.join(HERE, "dropout_grads.npz")
    np.savez_compressed(path, **out)
    print("loss", loss.item(), "sites", len(used), "file", os.path.getsize(path), "bytes")
    print("reference call order == library call order:", used == [s for s, _sh, _p in plan])


if __name__ == "__main__":
    main()
