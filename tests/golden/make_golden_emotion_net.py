#!/usr/bin/env python3
"""Generate tests/golden/emotion_net.npz from the reference's model.audio_emotion_classifer.EmotionNet (CPU, eval mode).

Build container only:   python tests/golden/make_golden_emotion_net.py

Weights and the [B,128,128] input come from emotiongestures_amd.synth (integer hash); the golden holds logits, the
[B,256,16,16] feature statistics and a corner of it.  torchvision (imported at model/audio_emotion_classifer.py:7, unused) gets
an empty stand-in.
"""
import json
import os
import sys
import types

import numpy as np
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
REF = "/root/reference"

from emotiongestures_amd.synth import hash_unit, load_synth_weights  # noqa: E402


def emotion_input(batch, seed):
    """fp16-rounded dB values in [-80, 0] like a stored spectrogram (utils/train_utils_BEAT.py:189), [B,128,128]."""
    v = (-80.0 * hash_unit("emotion.mfcc", batch * 128 * 128, seed)).astype(np.float16).astype(np.float32)
    return v.reshape(batch, 128, 128)


def main():
    for name in ("torchvision", "torchvision.models", "torchvision.utils", "torchvision.transforms", "fasttext", "torch_dct", "umap"):
        if name not in sys.modules:
            sys.modules[name] = types.ModuleType(name)
    sys.modules["torchvision"].models = sys.modules["torchvision.models"]
    sys.path.insert(0, REF)
    os.chdir(REF)
    from model.audio_emotion_classifer import EmotionNet
    torch.manual_seed(0)
    net = EmotionNet().eval()
    load_synth_weights(net, 31)
    x = torch.from_numpy(emotion_input(2, 31))
    with torch.no_grad():
        feat = net.emotion_encoder(x.unsqueeze(1))
        logits = net(x)
    out = {"logits": logits.numpy(), "feat_mean": feat.mean(dim=(0, 2, 3)).numpy(), "feat_std": feat.std(dim=(0, 2, 3)).numpy(),
           "feat_corner": feat[:, :, :4, :4].numpy()}
    json.dump([[k, list(v.shape)] for k, v in net.state_dict().items()],
              open(os.path.join(ROOT, "tests", "golden", "emotion_net_schema.json"), "w"))
    path = os.path.join(ROOT, "tests", "golden", "emotion_net.npz")
    np.savez_compressed(path, **out)
    print("wrote", path, os.path.getsize(path) // 1024, "KiB", logits.numpy())


if __name__ == "__main__":
    main()
