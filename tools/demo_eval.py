#!/usr/bin/env python3
"""End-to-end walk through the reference's evaluation flow on synthetic data, every stage on the HIP path:

  raw 16 kHz clip audio --extract_melspectrogram (GPU)--> whole-clip mel --DataPreprocessor--> sample records (fp16 spec, audio,
  poses, words) --SpeechMotionDataset / collate--> batches --harness.evaluate (CVAE sample -> generator -> FGD auto-encoder,
  skeleton emotion classifier, Frechet distance, diversity, MPJRE, L2)--> the summary metrics of
  test_emotion_gesture_diversity_iterative.py:191-261 (beat score excluded).

Weights are synthetic (integer hash), so the metric values carry no meaning; the script shows the call sequence a user of the
reference would keep and prints the throughput of the loop.  usage: demo_eval.py [n_clips=6] [seconds=12]"""
import os
import sys
import time

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import torch

from emotiongestures_amd.builders import build_mirror
from emotiongestures_amd import datapath as D
from emotiongestures_amd import harness as H
from emotiongestures_amd.CAVE.BEAT_CVAE import MLP_Reconstruct_v3
from emotiongestures_amd.synth import hash_unit, load_synth_weights, synth_audio

n_clips = int(sys.argv[1]) if len(sys.argv) > 1 else 6
seconds = float(sys.argv[2]) if len(sys.argv) > 2 else 12.0
dev = torch.device("cuda:0")
FRAMES, POSE_DIM, PRIOR, FPS = 60, 282, 10, 15          # BEAT timing: 60 poses @ 15 fps = 4 s -> spec [128,124]

t0 = time.perf_counter()
videos = []
for c in range(n_clips):
    audio = synth_audio(1, int(seconds * 16000), seed=50 + c)[0]
    n_skel = int(seconds * 30)
    skel = (hash_unit("demo.skel", n_skel * 94 * 3, c) * 2 - 1).astype(np.float32).reshape(n_skel, 94, 3)     # 94 x 3 = 282
    words = [["w%d" % i, 0.35 * i, 0.35 * i + 0.25] for i in range(int(seconds / 0.35))]
    videos.append(D.clips_from_raw_audio("1_demo_0_%d_%d" % (60 + 8 * c, 60 + 8 * c), audio, skel, words, 30, device=dev))
store = D.DictStore()
D.DataPreprocessor(videos, store, FRAMES, 15, FPS).run()
ds = D.SpeechMotionDataset(store, FRAMES, 15, FPS)
t_data = time.perf_counter() - t0

B = 8
batches = []
for s in range(0, len(ds) - B + 1, B):
    audio, spec, pose, label, aux = D.audio_classifier_collate_fn([ds[i] for i in range(s, s + B)])
    batches.append({"spec": spec, "text": torch.zeros(B, 60, dtype=torch.int64), "pose_seq": pose, "label": label})
print(f"data path: {n_clips} clips x {seconds:.0f} s -> {len(ds)} samples -> {len(batches)} batches of {B}  ({t_data:.2f} s incl. GPU mel)")

gen = build_mirror("spatial", FRAMES, POSE_DIM, PRIOR, PRIOR, seed=7, precision="bf16x3").to(dev)
vae = load_synth_weights(MLP_Reconstruct_v3(frames=FRAMES), 7).eval().to(dev)
fgd = load_synth_weights(H.MLP_Reconstruct(pose_dim=POSE_DIM), 7).eval().to(dev)
cls = load_synth_weights(H.SkeletonTransformer(class_dim=8, pose_dim=POSE_DIM, d_word_vec=512, d_model=512, d_inner=1024, n_layers=3, n_head=8,
                                               d_k=64, d_v=64, n_position=FRAMES), 7).eval().to(dev)
H.evaluate(gen, vae, fgd, cls, batches[:1], PRIOR, device=dev)                    # warm-up (packs weights, allocates workspaces)
torch.cuda.synchronize()
t0 = time.perf_counter()
metrics = H.evaluate(gen, vae, fgd, cls, batches, PRIOR, device=dev)
torch.cuda.synchronize()
dt = time.perf_counter() - t0
print("metrics:", {k: round(v, 4) for k, v in metrics.items()})
print(f"evaluation loop: {len(batches) * B} samples in {dt:.2f} s ({len(batches) * B / dt:.0f} samples/s incl. host-side Frechet / diversity)")
