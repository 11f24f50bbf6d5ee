#!/usr/bin/env python3
"""Throughput of the other BASELINE.json configurations on one GPU (informational; the contract line is bench.py):
  cfg 4  BEAT-long: 10 s audio -> mel [128,312] -> CVAE (120 ch) -> generator (120 frames x 282) at B=16 per step
  cfg 5  diversity: B=64 TED clips, 32 CVAE draws per clip; audio tower once per clip, fusion/enc/dec/post per draw
  h2d    the headline workload through ClipPipeline.run with inputs starting in pinned host memory
usage: bench_configs.py [long|draws|h2d] [precision]"""
import os
import sys
import time

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch

from emotiongestures_amd.builders import build_mirror, make_args, make_lang
from emotiongestures_amd.CAVE.BEAT_CVAE import MLP_Reconstruct_v3
from emotiongestures_amd.engine import MelFrontEnd
from emotiongestures_amd.synth import load_synth_weights, synth_audio, synth_inputs

dev = torch.device("cuda:0")


def timeit(fn, iters=10, warm=3):
    for _ in range(warm):
        fn()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(iters):
        fn()
    torch.cuda.synchronize()
    return (time.perf_counter() - t0) / iters


def long_clips(prec):
    from emotiongestures_amd.Full_model.Models_spatial_memory import Transformer
    from emotiongestures_amd.pipeline import ClipPipeline
    F, D, P, T, B = 120, 282, 10, 312, 16
    model = Transformer(make_args(10), make_lang(200), frames=F, pose_dim=D, prior_frames=P, d_word_vec=512, d_model=512, d_inner=2048,
                        n_layers=3, n_head=8, d_k=64, d_v=64, n_position=F, spec_len=T, precision=prec)
    load_synth_weights(model, 21).eval().to(dev)
    vae = load_synth_weights(MLP_Reconstruct_v3(frames=F), 21).eval().to(dev)
    inp = synth_inputs(B, F, D, P, spec_len=T, seed=21)
    inp["audio"] = synth_audio(B, 160000, seed=21)
    g = {k: torch.from_numpy(inp[k]).to(dev) for k in ("audio", "text", "pre_pose", "label", "z")}
    for lanes in (1, 4):
        pipe = ClipPipeline((model, vae, MelFrontEnd(dev)), g, dev, lanes=lanes)
        dt = timeit(lambda: pipe.launch_next(), iters=24, warm=8)
        print(f"cfg4 BEAT-long {prec}: B={B} per step, {lanes} step(s) in flight: {dt * 1e3:.3f} ms/step, {B / dt:.0f} clips/s "
              f"({B / dt * 10:.0f} s of audio per second)")


def draws(prec):
    B, R = 64, 32
    model = build_mirror("spatial", 34, 126, 4, 4, seed=9, precision=prec).to(dev)
    vae = load_synth_weights(MLP_Reconstruct_v3(frames=34), 9).eval().to(dev)
    inp = synth_inputs(B, seed=9)
    g = {k: torch.from_numpy(v).to(dev) for k, v in inp.items()}
    lab = g["label"].repeat_interleave(R, 0)
    z = torch.randn(B * R, 32, device=dev)

    def step():
        with torch.no_grad():
            s = vae.sample(lab, z=z).view(B, R, 34, 512)
            return model.forward_draws(g["spec"], g["pre_pose"], s)
    dt = timeit(step)
    print(f"cfg5 diversity {prec}: B={B} clips x {R} draws: {dt * 1e3:.2f} ms/step, {B / dt:.0f} clips/s, {B * R / dt:.0f} pose sequences/s")


def emotion_net(prec):
    """EmotionNet (audio emotion classifier) inference throughput, B=64 spectrograms [128,128]."""
    from emotiongestures_amd.model.audio_emotion_classifer import EmotionNet
    net = EmotionNet(precision=prec).eval()
    load_synth_weights(net, 31)
    net.to(dev)
    x = (torch.rand(64, 128, 128, device=dev) * -80.0)
    with torch.no_grad():
        dt = timeit(lambda: net(x), iters=10, warm=3)
    print(f"EmotionNet {prec}: B=64, {dt * 1e3:.2f} ms per batch, {64 / dt:.0f} clips/s")


def host_inputs(prec):
    """PCIe-inclusive rate of the headline workload: every batch starts in pinned host memory (audio 16 MB + small tensors),
    ClipPipeline.run copies it into a lane's buffers on the lane's stream and returns the poses to the caller's stream."""
    import bench
    from emotiongestures_amd.pipeline import ClipPipeline
    gen, vae, mel, _, _ = bench.build_models(prec, dev)
    inp = bench.make_inputs(64, seed=1000)
    keys = ("audio", "text", "pre_pose", "label", "z")
    host = {k: torch.from_numpy(inp[k]).pin_memory() for k in keys}
    g = {k: v.to(dev) for k, v in host.items()}
    pipe = ClipPipeline((gen, vae, mel), g, dev, lanes=4)
    for src, label in ((g, "inputs resident in HBM"), (host, "inputs in pinned host memory (H2D inside the timed region)")):
        n = 60
        for out in pipe.run(src for _ in range(8)):
            pass
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        for out in pipe.run(src for _ in range(n)):
            pass
        torch.cuda.synchronize()
        dt = (time.perf_counter() - t0) / n
        print(f"ClipPipeline.run, {label}: {dt * 1e3:.3f} ms/step, {64 / dt:.0f} clips/s")


if __name__ == "__main__":
    what = sys.argv[1] if len(sys.argv) > 1 else "both"
    prec = sys.argv[2] if len(sys.argv) > 2 else "bf16x3"
    if what in ("long", "both"):
        long_clips(prec)
    if what in ("draws", "both"):
        draws(prec)
    if what == "h2d":
        host_inputs(prec)
    if what == "emotion":
        emotion_net(prec)
