#!/usr/bin/env python3
"""Driver of the product loop `emotiongestures_amd.train.loops.train_k_fold` (the reference's train_audio_classifier_K_fold.py:109-200) on a
synthetic BEAT-shaped data set built through the reference's own data path: clips -> datapath.DataPreprocessor -> sample store ->
datapath.SpeechMotionDataset -> audio_classifier_collate_fn -> EmotionNet on the HIP training operators.

    python tools/train_emotion_net.py [--clips 16] [--folds 2] [--epochs 2] [--batch 8] [--lr 1e-4] [--precision f32|bf16x3] [--save DIR]

Synthetic task: the label is encoded in the spectrogram (a per-class band offset), so the loss must fall and accuracy rise."""
import argparse
import os
import sys
import time

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import torch

from emotiongestures_amd import datapath as D
from emotiongestures_amd.synth import synth_clip
from emotiongestures_amd.train import loops

ap = argparse.ArgumentParser()
ap.add_argument("--clips", type=int, default=16)
ap.add_argument("--folds", type=int, default=2)
ap.add_argument("--epochs", type=int, default=2)
ap.add_argument("--batch", type=int, default=8)
ap.add_argument("--lr", type=float, default=1e-4)
ap.add_argument("--val-every", type=int, default=5)
ap.add_argument("--precision", default="f32", choices=("f32", "bf16x3"))
ap.add_argument("--save", default=None)
args = ap.parse_args()
dev = torch.device("cuda:0")

eids = [1, 66, 75, 82, 90, 100, 105, 115]                   # one BEAT recording id per emotion class (lmdb_loader_BEAT_full.py:78-118)
videos = []
for i in range(args.clips):
    clip = synth_clip(seed=i, duration=9.0)
    k = i % 8
    spec = clip["audio_feat"].astype(np.float32)
    spec[16 * k:16 * k + 16, :] = np.minimum(spec[16 * k:16 * k + 16, :] + 40.0, 0.0)
    clip["audio_feat"] = spec.astype(np.float16)
    videos.append({"eid": "1_spk_0_%d_%d" % (eids[k], eids[k]), "clips": [clip]})
store = D.DictStore()
D.DataPreprocessor(videos, store, 62, 20, 15).run()          # 62 poses at 15 fps = a 128-frame spectrogram per sample
ds = D.SpeechMotionDataset(store, 62, 20, 15)
print(f"{len(ds)} samples, spectrogram [128, {ds.expected_spectrogram_length}]")
t0 = time.perf_counter()
hist = loops.train_k_fold(ds, device=dev, n_splits=args.folds, total_epoch=args.epochs, batch_size=args.batch, lr=args.lr, val_every=args.val_every,
                          save_dir=args.save, test_dataset=ds, precision=args.precision)
torch.cuda.synchronize()
for h in hist:
    print(f"fold {h['fold']}: {h['iterations']} iterations, loss {h['loss'][0]:.3f} -> {h['loss'][-1]:.3f}, val {h['val_acc'][-1:]} test {h['test_acc'][-1:]}")
print(f"{time.perf_counter() - t0:.1f} s")
