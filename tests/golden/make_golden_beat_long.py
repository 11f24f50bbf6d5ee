#!/usr/bin/env python3
"""Generate tests/golden/beat_long_b2.npz: BASELINE configs[3] (10 s audio -> spec 128x312, 120 frames, pose_dim 282, prior 10)
from the REFERENCE's own classes, with the sizes the reference hard-codes replaced after construction:

  * Audio_ResNetEncoder.fc1 = nn.Linear(32*31, d_model)   (Full_model/Models_spatial_memory.py:105)  -> nn.Linear(32*78, 512)
  * MLP_Reconstruct_v3: Conv1d(60, 32) / Conv1d(32, 60) / BatchNorm1d(60) / Conv1d(60, 60)  (CAVE/BEAT_CVAE.py:320,365-368) -> 120
  * n_position is a constructor argument (:477): 120

Every other line of the reference's forward runs unmodified.  Build container only:  python tests/golden/make_golden_beat_long.py"""
import os
import sys
from types import SimpleNamespace

import numpy as np
import torch
import torch.nn as nn

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests", "golden"))
from make_golden import REF, _flat, _stub_unused_imports  # noqa: E402

from emotiongestures_amd.synth import digest, load_synth_weights, synth_inputs  # noqa: E402

F_, D_, P_, T_, SEED, B_ = 120, 282, 10, 312, 21, 2


def main():
    _stub_unused_imports()
    if REF not in sys.path:
        sys.path.insert(0, REF)
    from CAVE.BEAT_CVAE import MLP_Reconstruct_v3
    from Full_model.Models_spatial_memory import Transformer
    torch.manual_seed(0)
    args = SimpleNamespace(chunk=10, hidden_size=300, n_layers=3, freeze_wordembed=False, wordembed_dim=300, dropout_prob=0.1)
    lang = SimpleNamespace(n_words=200, word_embedding_weights=None)
    m = Transformer(args, lang, frames=F_, pose_dim=D_, prior_frames=P_, d_word_vec=512, d_model=512, d_inner=2048, n_layers=3, n_head=8,
                    d_k=64, d_v=64, n_position=F_)
    m.audio_encoder.fc1 = nn.Linear(32 * 78, 512)                      # 128x312 spectrogram -> 32x78 map
    load_synth_weights(m, SEED)
    m.eval()
    vae = MLP_Reconstruct_v3()
    vae.Encoder[0] = nn.Conv1d(F_, 32, 3, padding=1)
    vae.Decoder[9] = nn.Conv1d(32, F_, 3, padding=1)
    vae.Decoder[11] = nn.BatchNorm1d(F_)
    vae.Decoder[12] = nn.Conv1d(F_, F_, 3, padding=1)
    load_synth_weights(vae, SEED)
    vae.eval()
    inp = synth_inputs(B_, F_, D_, P_, spec_len=T_, seed=SEED)
    y, z = torch.from_numpy(inp["label"]), torch.from_numpy(inp["z"])
    real_randn = torch.randn
    torch.randn = lambda *a, **k: z.clone()          # sample() draws torch.randn(n, 32) (BEAT_CVAE.py:441)
    try:
        with torch.no_grad():
            sampled = vae.sample(y)
    finally:
        torch.randn = real_randn
    assert tuple(sampled.shape) == (B_, F_, 512)
    with torch.no_grad():
        pose, emo, sem, pred, text = m(torch.from_numpy(inp["spec"]), torch.from_numpy(inp["text"]), torch.from_numpy(inp["pre_pose"]), sampled)
    out = {"pose": pose.numpy(), "emotion_prediction": pred.numpy(), "meta": np.asarray([B_, F_, D_, P_, 10, T_, 200, SEED, 1], np.int64)}
    out.update(_flat("cvae_sample", digest(sampled.numpy(), 16384)))
    out.update(_flat("emotion_feature", digest(emo.numpy(), 8192)))
    out.update(_flat("semantic_feature", digest(sem.numpy(), 8192)))
    path = os.path.join(ROOT, "tests", "golden", "beat_long_b2.npz")
    np.savez_compressed(path, **out)
    print("beat_long_b2 pose", pose.shape, "L2/clip", np.linalg.norm(pose.numpy().reshape(B_, -1), axis=1), os.path.getsize(path) // 1024, "KiB")


if __name__ == "__main__":
    main()
