#!/usr/bin/env python3
"""bench.py -- gesture clips/sec of the EmotionGesture hot path on MI355X.

One "step" = one pass of the hot path over one batch of 64 synthetic TED-shaped clips per GPU, inputs already
resident in HBM:   16 kHz audio [64, 64000] -> HIP mel front-end -> spec [64,128,124]
                   one-hot label + latent z -> HIP CVAE sample -> emotion map [64,34,512]
                   (spec, text, prior poses, emotion map) -> HIP generator -> pose [64,34,126] (+ the 4 auxiliary returns)
This is BASELINE.json configs[1] ("Batch=64 synthetic 16 kHz audio -> gesture inference, 1xMI355X").

    python bench.py [--gpus N] [--steps K] [--warmup W] [--precision f32|bf16x3|bf16]
    python -m torch.distributed.run --nnodes=1 --nproc-per-node N ... bench.py --gpus N ...

N > 1: one process per GPU, clips sharded across ranks (weak scaling: 64 clips per GPU), no collective on the data
path (clips are independent); torch.distributed (RCCL) is used only for the barrier and the max-over-ranks
time.  Rank 0 prints ONE JSON line.  Started WITHOUT an outer launcher, `python bench.py --gpus N` launches itself: before
any HIP call the parent spawns N fresh worker processes (RANK / LOCAL_RANK / WORLD_SIZE / MASTER_* set, 127.0.0.1
rendezvous), waits for them and exits with the worst status; it refuses N greater than the visible GPU count.  This replaces
the reference's in-process `nn.DataParallel` (test_emotion_gesture_diversity_iterative.py:137-138).
"""
import argparse
import ctypes as C
import json
import os
import socket
import subprocess
import sys
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

import numpy as np
import torch

FLOP_PER_CLIP = 9.235e9        # 2 x MACs of the TED generator forward (reference FlopCounterMode probe)
MEL_FLOP_PER_CLIP = 0.023e9
CVAE_FLOP_PER_CLIP = 0.0189e9 * 34 / 60
PEAK_TFLOPS = {"f32": 157.3, "bf16x3": 2500.0, "bf16": 2500.0}     # MI355X_MICROARCH.md: dense MFMA peak of the MFMA dtype


def build_models(precision, dev, seed=0):
    from emotiongestures_amd.builders import build_mirror
    from emotiongestures_amd.CAVE.BEAT_CVAE import MLP_Reconstruct_v3
    from emotiongestures_amd.engine import MelFrontEnd
    from emotiongestures_amd.synth import load_synth_weights
    gen = build_mirror("spatial", 34, 126, 4, 4, seed=seed, precision=precision)
    vae = load_synth_weights(MLP_Reconstruct_v3(frames=34), seed).eval()
    sd_g = {k: v.detach().clone() for k, v in gen.state_dict().items()}
    sd_v = {k: v.detach().clone() for k, v in vae.state_dict().items()}
    gen.to(dev)
    vae.to(dev)
    return gen, vae, MelFrontEnd(dev), sd_g, sd_v


def make_inputs(batch, seed):
    from emotiongestures_amd.synth import synth_audio, synth_inputs
    inp = synth_inputs(batch, 34, 126, 4, seed=seed)
    inp["audio"] = synth_audio(batch, 64000, seed=seed)
    return inp


def cpu_baseline(sd_g, sd_v, inp, budget_s=20.0):
    """The CPU oracle (port of the reference's CPU path: same torch fp32 CPU kernels) on the SAME workload -- whole B=64
    batches of mel + CVAE sample + generator -- on the host cores of this box, bounded to ~budget_s."""
    from oracle import emogest_oracle as O
    avail = len(os.sched_getaffinity(0)) if hasattr(os, "sched_getaffinity") else (os.cpu_count() or 1)
    n = inp["audio"].shape[0]
    t = {k: torch.from_numpy(v[:n]) for k, v in inp.items() if k != "audio"}
    audio = inp["audio"][:n]

    def step(m=n):
        with torch.no_grad():
            spec = torch.from_numpy(O.melspectrogram(audio[:m], out_frames=124))
            sampled = O.cvae_sample(sd_v, t["label"][:m], t["z"][:m])
            return O.generator_forward(sd_g, O.GenCfg(), spec, t["text"][:m], t["pre_pose"][:m], sampled)[0]

    # pick the thread count that is fastest for this workload on a 16-clip probe (more threads than the small convs can use only
    # adds synchronisation cost on many-core hosts); `cores` reports the threads actually used
    best, cores = None, 1
    for th in sorted({c for c in (8, 16, 32, 64, avail) if c <= avail}):
        torch.set_num_threads(th)
        step(16)
        t0 = time.perf_counter()
        step(16)
        el = time.perf_counter() - t0
        if best is None or el < best:
            best, cores = el, th
        if el > 8.0:
            break
    torch.set_num_threads(cores)
    t0 = time.perf_counter()
    iters = 0
    while True:
        step()
        iters += 1
        el = time.perf_counter() - t0
        if el > budget_s or iters >= 40:
            break
    torch.set_num_threads(cores)
    t1 = time.perf_counter()
    step(1)                                   # BASELINE configs[0]: single-clip latency of the CPU path
    lat = time.perf_counter() - t1
    return {"value": round(n * iters / el, 2), "unit": "clips/s", "cores": cores, "kind": "port",
            "b1_latency_ms": round(lat * 1e3, 1),
            "sample": f"{iters} x B={n} passes of oracle mel+cvae_sample+generator_forward (torch fp32 CPU, {cores} threads)"}


def measured_traffic(tag, precision):
    """HBM-side bytes per launch of the dominant kernel, from the committed PMC passes (profiles/*_traffic.json, written by
    tools/traffic_json.py from `rocprofv3 --pmc FETCH_SIZE` / `--pmc WRITE_SIZE` runs of this same bench command).  Counters
    cannot be read inside the timed process, so this is a lookup, None when the profile does not cover the kernel / precision."""
    pdir = os.path.join(ROOT, "profiles")
    cands = sorted(f for f in os.listdir(pdir) if f.endswith("_traffic.json")) if os.path.isdir(pdir) else []
    if precision != "bf16x3" or not cands:
        return None
    kern = json.load(open(os.path.join(pdir, cands[-1])))["kernels"]      # newest round's file
    if tag in (2, 3, 4):
        prefix = {2: "gemm_glds_kernel<3", 3: "gemm_presplit", 4: "gemm_bf16_kernel<3"}[tag]
    elif tag >= 1000:
        cin, cout, s = tag // 1000000, (tag // 1000) % 1000, (tag // 100) % 10
        prefix = f"conv3x3_bf16_kernel<{cin}, {(cout + 15) // 16}, {s},"
    else:
        return None
    hits = [v for k, v in kern.items() if k.startswith(prefix)]
    if not hits and tag >= 1000 and tag // 1000000 == 32 and (tag // 1000) % 1000 == 32:      # the 32 -> 32 stage runs on the persistent kernel
        hits = [v for k, v in kern.items() if k.startswith("conv3x3_c32_persistent_kernel<3")]       # <TERMS, TH, STAMP>: any tile height, production build
    return max(hits, key=lambda v: v.get("launches_per_step", 0))["bytes"] if hits else None      # several instantiations: the one the step launches most


def train_hbm_gb_per_step(batch):
    """Memory-side GB moved by one training step at `batch` clips, from the newest committed PMC summary profiles/*_train_traffic_b{batch}.json
    (tools/train_traffic_json.py over `rocprofv3 --pmc FETCH_SIZE` / `WRITE_SIZE` passes of `bench.py --train --train-batch {batch}`); None when
    no such profile is committed.  A lookup: counters cannot be read inside the timed process."""
    pdir = os.path.join(ROOT, "profiles")
    cands = sorted(f for f in os.listdir(pdir) if f.endswith(f"_train_traffic_b{batch}.json")) if os.path.isdir(pdir) else []
    if not cands:
        return None
    d = json.load(open(os.path.join(pdir, cands[-1])))
    return {"gb_per_step": d["gb_per_step"], "source": "profiles/" + cands[-1]}


def parse_args():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=20)
    ap.add_argument("--warmup", type=int, default=5)
    ap.add_argument("--batch", type=int, default=64, help="clips per GPU per step")
    ap.add_argument("--precision", default=os.environ.get("EG_PRECISION", "bf16x3"), choices=["f32", "bf16x3", "bf16"])
    ap.add_argument("--no-graph", action="store_true", help="launch eagerly instead of replaying a captured hipGraph")
    ap.add_argument("--no-concurrent", action="store_true", help="keep the independent branches of a step on one stream")
    ap.add_argument("--concurrent", action="store_true",
                    help="fork the text / prior / CVAE branches of a step onto side streams (default only with --in-flight 1: with "
                         "several steps in flight the overlap across steps already fills the GPU and the forks cost 5-7 %%)")
    ap.add_argument("--in-flight", type=int, default=int(os.environ.get("EG_IN_FLIGHT", "4")),
                    help="independent steps (batches) in flight per GPU: each lane has its own workspaces, I/O buffers, hipGraph and stream")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--no-roofline", action="store_true")
    ap.add_argument("--no-extra-legs", action="store_true", help="skip the f32 parity-mode leg and the sustained (>= 2 s) leg")
    ap.add_argument("--sustain-seconds", type=float, default=2.0)
    ap.add_argument("--no-train-legs", action="store_true", help="skip the two training legs (16 and 128 clips per step) of the default line")
    ap.add_argument("--fold-affine", action="store_true",
                    help="fold the Dropout-only Linear chains at pack time (fewer launches and FLOPs than the reference graph; reported in config)")
    ap.add_argument("--no-fuse-se", action="store_true", help="A/B: keep the SE tail of identity blocks as a separate pass (round-1 data flow)")
    ap.add_argument("--train", action="store_true",
                    help="BASELINE configs[2]: one step = generator forward + 100*smooth_l1 + CE + backward + bucketed gradient "
                         "all-reduce (RCCL) + fused Adam on --train-batch clips per GPU (fp32 operators)")
    ap.add_argument("--train-precision", choices=("f32", "bf16x3"), default="bf16x3",
                    help="--train: arithmetic of the convolutions and Linear products; f32 is the gradient-parity configuration (always timed as the f32_eager leg)")
    ap.add_argument("--no-train-graph", action="store_true",
                    help="--train: issue every kernel through autograd instead of replaying the step from one captured hipGraph")
    ap.add_argument("--train-graph", action="store_true", help="(default since round 2; kept for old command lines)")
    ap.add_argument("--no-train-dropout", action="store_true",
                    help="--train: run the step with every Dropout at p = 0 (the gradient-parity configuration) instead of the reference's train() graph")
    ap.add_argument("--dp-train-batch", type=int, default=16,
                    help="clips per GPU per step of the data-parallel training leg that the default line carries at --gpus N > 1 (16 x 8 GPUs = global 128)")
    ap.add_argument("--train-batch", type=int, default=16, help="clips per GPU per training step (16 = global 128 on 8 GPUs, SURVEY.md §8d cfg 3)")
    return ap.parse_args()


# ---- self-launch: `python bench.py --gpus N` without an outer torchrun ------------------------------------------------------
def self_launch(n):
    """Parent side of `--gpus N`.  Runs BEFORE anything touches the GPU (torch.cuda.device_count() does not initialise it on
    this image; is_available() would), never re-execs: N fresh children, one per GPU, this process only waits for them."""
    dry = os.environ.get("EG_BENCH_DRY") == "1"          # CPU test of this launcher (tests/test_bench_launcher.py)
    if not dry:
        ndev = torch.cuda.device_count()
        if ndev < n and os.environ.get("EG_BENCH_BACKEND", "nccl") == "nccl":       # gloo: the one-GPU test of the data-parallel path shares the device
            raise SystemExit(f"bench.py: --gpus {n} but only {ndev} GPU(s) are visible; refusing to oversubscribe")
    # rendezvous through a file the ranks open themselves (emotiongestures_amd/dist.py init_process_group): no probed-then-released TCP port that
    # another process could take before rank 0 binds it
    from emotiongestures_amd.dist import new_store_path
    store = new_store_path("eg_bench")
    procs = []
    for r in range(n):
        env = dict(os.environ, RANK=str(r), LOCAL_RANK=str(r), WORLD_SIZE=str(n), LOCAL_WORLD_SIZE=str(n), EG_DIST_STORE=store, EG_BENCH_CHILD="1")
        env.pop("MASTER_PORT", None)
        env.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
        procs.append(subprocess.Popen([sys.executable, os.path.abspath(__file__)] + sys.argv[1:], env=env))
    # poll: the first worker that fails takes its siblings down (they would otherwise sit in the rendezvous / a collective until a timeout);
    # an overall limit bounds the whole run.  Only the children this function started are signalled.
    deadline = time.monotonic() + float(os.environ.get("EG_BENCH_TIMEOUT_S", "3600"))
    rcs = [None] * n
    failed = None
    while any(rc is None for rc in rcs):
        for r, p in enumerate(procs):
            if rcs[r] is None:
                rcs[r] = p.poll()
                if rcs[r] not in (None, 0) and failed is None:
                    failed = (r, rcs[r])
        if failed is not None or time.monotonic() > deadline:
            for r, p in enumerate(procs):
                if rcs[r] is None:
                    p.terminate()
            for r, p in enumerate(procs):
                if rcs[r] is None:
                    try:
                        rcs[r] = p.wait(timeout=10)
                    except subprocess.TimeoutExpired:
                        p.kill()
                        rcs[r] = p.wait()
            if failed is None:
                raise SystemExit(f"bench.py: workers exceeded EG_BENCH_TIMEOUT_S; terminated (exit codes {rcs})")
            raise SystemExit(f"bench.py: worker rank {failed[0]} failed with exit code {failed[1]}; siblings terminated (exit codes {rcs})")
        time.sleep(0.05)
    try:
        os.unlink(store)
    except OSError:
        pass
    return 0


def dry_worker(args, rank, world):
    """Launcher plumbing only (no GPU): rendezvous over gloo, barrier, max-over-ranks reduce, one JSON line from rank 0."""
    from emotiongestures_amd.dist import init_process_group
    dist = init_process_group("gloo", rank, world)
    dist.barrier()
    tt = torch.tensor([1.0 + rank], dtype=torch.float64)
    dist.all_reduce(tt, op=dist.ReduceOp.MAX)
    dist.barrier()
    dist.destroy_process_group()
    if rank == 0:
        print(json.dumps({"metric": "gesture clips/sec (34 frames, 43 joints)", "dry_run": True, "n_gpus": world,
                          "steps": args.steps, "warmup": args.warmup, "max_elapsed": float(tt.item())}))
    return 0


# ---- informational legs of the default line: the other BASELINE.json configurations on one GPU -------------------------------------------
def gpu_b1_latency(gen, vae, mel, inp, sd_g, sd_v, dev, reps=50):
    """BASELINE configs[0] on the GPU: ONE clip (4 s of audio -> mel -> CVAE sample -> generator -> pose), one lane, the step replayed from its
    hipGraph and synchronised every time; median of `reps` wall-clock latencies, pose checked against the CPU oracle for that clip."""
    from emotiongestures_amd.builders import clip_rel_l2
    from emotiongestures_amd.pipeline import ClipPipeline
    from oracle import emogest_oracle as O
    g1 = {k: torch.from_numpy(v[:1]).to(dev) for k, v in inp.items()}
    lat_by_mode = {}
    for branch in (False, True):            # one lane alone: forking the text / prior / CVAE branches onto side streams may shorten the critical path
        pipe = ClipPipeline((gen, vae, mel), g1, dev, lanes=1, branch_streams=branch)
        for _ in range(5):
            pipe.wait(pipe.launch_next())
        ts = []
        for _ in range(reps):
            torch.cuda.synchronize(dev)
            t0 = time.perf_counter()
            pipe.wait(pipe.launch_next())
            ts.append((time.perf_counter() - t0) * 1e3)
        lat_by_mode[branch] = (float(np.median(ts)), float(np.min(ts)), pipe.outputs(0)[0].cpu().numpy())
        del pipe
    best = min(lat_by_mode, key=lambda k: lat_by_mode[k][0])
    pose = lat_by_mode[best][2]
    with torch.no_grad():
        t1 = {k: torch.from_numpy(v[:1]) for k, v in inp.items() if k != "audio"}
        spec = torch.from_numpy(O.melspectrogram(inp["audio"][:1], out_frames=124))
        ref = O.generator_forward(sd_g, O.GenCfg(), spec, t1["text"], t1["pre_pose"], O.cvae_sample(sd_v, t1["label"], t1["z"]))[0]
    return {"latency_ms_median": round(lat_by_mode[best][0], 4), "latency_ms_min": round(lat_by_mode[best][1], 4), "reps": reps,
            "clips_per_step": 1, "launch": "hipGraph replay, 1 lane, synchronised per clip, " + ("branch streams" if best else "one stream"),
            "latency_ms_median_one_stream": round(lat_by_mode[False][0], 4), "latency_ms_median_branch_streams": round(lat_by_mode[True][0], 4),
            "pose_rel_l2_vs_cpu_oracle": clip_rel_l2(pose, ref.numpy())}


def beat_long_leg(precision, dev, steps, B=16, lanes=4):
    """BASELINE configs[3] on one GPU: BEAT-dataset-shaped long clips -- 10 s of 16 kHz audio -> mel [128, 312] -> CVAE sample (120 channels) ->
    generator (120 frames x 282, 10 prior frames), B clips per step, `lanes` steps in flight; pose of every clip against the CPU oracle."""
    from emotiongestures_amd.builders import clip_rel_l2, make_args, make_lang
    from emotiongestures_amd.CAVE.BEAT_CVAE import MLP_Reconstruct_v3
    from emotiongestures_amd.engine import MelFrontEnd
    from emotiongestures_amd.Full_model.Models_spatial_memory import Transformer
    from emotiongestures_amd.pipeline import ClipPipeline
    from emotiongestures_amd.synth import load_synth_weights, synth_audio, synth_inputs
    from oracle import emogest_oracle as O
    Fr, D, P, T = 120, 282, 10, 312
    model = Transformer(make_args(10), make_lang(200), frames=Fr, pose_dim=D, prior_frames=P, d_word_vec=512, d_model=512, d_inner=2048,
                        n_layers=3, n_head=8, d_k=64, d_v=64, n_position=Fr, spec_len=T, precision=precision)
    load_synth_weights(model, 21).eval()
    vae = load_synth_weights(MLP_Reconstruct_v3(frames=Fr), 21).eval()
    sd = {k: v.detach().clone() for k, v in model.state_dict().items()}
    sdv = {k: v.detach().clone() for k, v in vae.state_dict().items()}
    model.to(dev); vae.to(dev)
    inp = synth_inputs(B, Fr, D, P, spec_len=T, seed=21)
    inp["audio"] = synth_audio(B, 160000, seed=21)
    g = {k: torch.from_numpy(inp[k]).to(dev) for k in ("audio", "text", "pre_pose", "label", "z")}
    pipe = ClipPipeline((model, vae, MelFrontEnd(dev)), g, dev, lanes=lanes)
    for _ in range(2 * lanes):
        pipe.launch_next()
    torch.cuda.synchronize(dev)
    t0 = time.perf_counter()
    for _ in range(steps):
        pipe.launch_next()
    torch.cuda.synchronize(dev)
    el = time.perf_counter() - t0
    pose = pipe.outputs(0)[0].cpu().numpy()
    n_chk = B                                   # every clip of the step (oracle: ~0.1-0.2 s per long clip on the host)
    with torch.no_grad():
        t = {k: torch.from_numpy(inp[k][:n_chk]) for k in ("text", "pre_pose", "label", "z")}
        spec = torch.from_numpy(O.melspectrogram(inp["audio"][:n_chk], out_frames=T))
        ref = O.generator_forward(sd, O.GenCfg(frames=Fr, pose_dim=D, prior_frames=P, chunk=10), spec, t["text"], t["pre_pose"],
                                  O.cvae_sample(sdv, t["label"], t["z"]))[0]
    return {"value": round(B * steps / el, 2), "unit": "clips/s", "ms_per_step": round(el / steps * 1e3, 4), "dtype": precision,
            "config": f"BEAT-long: 10 s audio -> mel(128x{T}) -> CVAE({Fr} ch) -> generator -> {Fr}x{D} pose; {B} clips per step, {lanes} steps in flight",
            "audio_seconds_per_second": round(B * steps / el * 10.0, 1),
            "pose_rel_l2_vs_cpu_oracle": clip_rel_l2(pose[:n_chk], ref.numpy()), "parity_clips_checked": n_chk}


def diversity_leg(precision, dev, steps, B=64, R=32):
    """BASELINE configs[4] on one GPU, in the headline's arithmetic (bf16x3; fp8 is reported separately when built): 32 CVAE latent draws per clip,
    audio tower once per clip, fusion -> encoder -> decoder -> post_projector per draw (M = B*R*34 rows).  A subset of (clip, draw) pairs against
    the CPU oracle."""
    from emotiongestures_amd.builders import build_mirror, clip_rel_l2
    from emotiongestures_amd.CAVE.BEAT_CVAE import MLP_Reconstruct_v3
    from emotiongestures_amd.synth import hash_unit, load_synth_weights, synth_inputs
    from oracle import emogest_oracle as O
    model = build_mirror("spatial", 34, 126, 4, 4, seed=9, precision=precision)
    vae = load_synth_weights(MLP_Reconstruct_v3(frames=34), 9).eval()
    sd = {k: v.detach().clone() for k, v in model.state_dict().items()}
    sdv = {k: v.detach().clone() for k, v in vae.state_dict().items()}
    model.to(dev); vae.to(dev)
    inp = synth_inputs(B, seed=9)
    g = {k: torch.from_numpy(v).to(dev) for k, v in inp.items()}
    z = (hash_unit("bench.draws.z", B * R * 32, 9).reshape(B, R, 32) * 2 - 1).astype(np.float32) * 1.7
    lab = g["label"][:, None, :].expand(B, R, 8).reshape(B * R, 8).contiguous()
    zd = torch.from_numpy(z).reshape(B * R, 32).to(dev)

    # hipGraph replay with `lanes` steps in flight, like every other leg: one captured step (CVAE draws -> forward_draws) per lane, each lane its own
    # stream, workspaces (slot) and outputs; the lanes share the models and the weight arena
    lanes = 2

    def step(slot=0):
        with torch.no_grad():
            return model.forward_draws(g["spec"], g["pre_pose"], vae.sample(lab, z=zd, slot=slot).view(B, R, 34, 512), slot=slot)
    graphs = []
    for i in range(lanes):
        st = torch.cuda.Stream(dev)
        st.wait_stream(torch.cuda.current_stream(dev))
        with torch.cuda.stream(st):
            for _ in range(2):
                step(i)
        torch.cuda.synchronize(dev)
        gr = torch.cuda.CUDAGraph()
        from emotiongestures_amd.pipeline import CAPTURE_MODE
        with torch.cuda.graph(gr, stream=st, capture_error_mode=CAPTURE_MODE):
            out_i = step(i)
        graphs.append((gr, st, out_i))

    def launch(k):
        gr, st, _o = graphs[k % lanes]
        with torch.cuda.stream(st):
            gr.replay()
    for k in range(2 * lanes):
        launch(k)
    torch.cuda.synchronize(dev)
    t0 = time.perf_counter()
    for k in range(steps):
        launch(k)
    torch.cuda.synchronize(dev)
    el = time.perf_counter() - t0
    poses = graphs[0][2]
    if lanes > 1 and not all(torch.equal(graphs[i][2], poses) for i in range(1, lanes)):
        raise SystemExit("bench.py: diversity lanes disagree on the same batch (invalid run)")
    clips, draws = [0, 31, 63], [0, 15, 31]
    err = 0.0
    t = {k: torch.from_numpy(v[clips]) for k, v in inp.items()}
    for r in draws:
        with torch.no_grad():
            ref = O.generator_forward(sd, O.GenCfg(), t["spec"], t["text"], t["pre_pose"], O.cvae_sample(sdv, t["label"], torch.from_numpy(z[clips, r])))[0]
        err = max(err, clip_rel_l2(poses[clips, r].cpu().numpy(), ref.numpy()))
    return {"value": round(B * R * steps / el, 1), "unit": "pose sequences/s", "clips_per_s": round(B * steps / el, 1), "ms_per_step": round(el / steps * 1e3, 3),
            "dtype": precision, "config": f"{B} clips x {R} CVAE draws per step (audio tower once per clip, transformer per draw), hipGraph replay, {lanes} steps in flight",
            "pose_rel_l2_vs_cpu_oracle": err, "parity_pairs_checked": len(clips) * len(draws)}


TRAIN_FLOP_PER_CLIP = 3.0 * FLOP_PER_CLIP      # forward + input gradients + weight gradients of the generator; CVAE and losses not counted


def train_leg(dev, B, precision, graph, steps, warmup, rank=0, world=1, dist=None, backend="nccl", segments=0, dropout=False, payload=None):
    """One timed configuration of the training step (generator + emotion CVAE: forward + 100*smooth_l1 + CE + backward + bucketed gradient
    all-reduce + fused Adam on B clips per GPU).  precision: arithmetic of the convolutions / Linear products ("f32" = the gradient-parity
    configuration); graph: replay the step from captured hipGraph(s) instead of issuing every kernel through autograd.
    dropout=True: the graph the REFERENCE trains -- every nn.Dropout of its train() mode active (encoder input, MHA / FFN outputs, the 0.2 layers
    of the projection MLPs and of the CVAE, the attention probabilities; SubLayers.py:54,79, Modules.py:21) on the library's counter-based mask
    stream (`train_dropout`; under a graph the mask epoch lives on the device: GraphedStep / SegmentedStep(stochastic=True)).  dropout=False is the
    gradient-parity configuration (p = 0).  payload: what the gradient buckets travel as ("f32" | "bf16"; default EG_GRAD_PAYLOAD or f32)."""
    from emotiongestures_amd import _lib
    from emotiongestures_amd.builders import build_mirror
    from emotiongestures_amd.CAVE.BEAT_CVAE import MLP_Reconstruct_v3
    from emotiongestures_amd.synth import hash_unit, load_synth_weights, synth_inputs
    from emotiongestures_amd.train import functional as F
    from emotiongestures_amd.train.optim import FlatAdam, GradBuckets, flatten_parameters
    lib = _lib.load()
    inp = synth_inputs(B, 34, 126, 4, seed=2000 + rank)
    g = {k: torch.from_numpy(v).to(dev) for k, v in inp.items()}
    target = torch.from_numpy((hash_unit("train.target_pose", B * 34 * 126, 2000 + rank) - 0.5).astype(np.float32).reshape(B, 34, 126)).to(dev)
    label = g["label"].argmax(1)
    eps = torch.from_numpy(synth_inputs(B, seed=3000 + rank)["z"]).to(dev)

    def barrier():
        torch.cuda.synchronize(dev)
        if dist is not None:
            dist.barrier()
        torch.cuda.synchronize(dev)

    F.set_precision(precision)
    try:
        return _train_leg_body(dev, B, precision, graph, steps, warmup, rank, world, dist, backend, dropout, lib, g, target, label, eps, barrier, payload)
    finally:                    # also when a leg raises: the process-wide precision / image registry / mask epoch never leak into the next leg
        F.reset_state()
        torch.cuda.empty_cache()


def _train_leg_body(dev, B, precision, graph, steps, warmup, rank, world, dist, backend, dropout, lib, g, target, label, eps, barrier, payload=None):
    from emotiongestures_amd.builders import build_mirror
    from emotiongestures_amd.CAVE.BEAT_CVAE import MLP_Reconstruct_v3
    from emotiongestures_amd.synth import load_synth_weights
    from emotiongestures_amd.train import functional as F
    from emotiongestures_amd.train.optim import FlatAdam, GradBuckets, flatten_parameters, stage_splits
    model = build_mirror("spatial", 34, 126, 4, 4, seed=0, precision="f32").to(dev).train()
    vae = load_synth_weights(MLP_Reconstruct_v3(frames=34), 0).to(dev).train()
    model.train_dropout = vae.train_dropout = bool(dropout)
    F.manual_seed(1234 + rank)
    both = torch.nn.ModuleList([model, vae])         # one flat parameter / gradient buffer, one optimiser, one set of buckets
    fp = flatten_parameters(both)
    from emotiongestures_amd.train import nets as _nets
    fp.enable_weight_images(*_nets.weight_image_plan(both))     # every Linear / conv weight image of a step from one launch (after the optimiser); Q|K|V and K|V as one image each
    opt = FlatAdam(fp, lr=2e-4, betas=(0.5, 0.999), weight_decay=1e-5)           # test_emotion_gesture_diversity_iterative.py:355-366 (lr 2e-4)
    collective = world > 1 or (dist is not None and os.environ.get("EG_FORCE_COLLECTIVES") == "1")       # world 1 + EG_FORCE_COLLECTIVES: the data-parallel step over a 1-rank RCCL group
    gb = GradBuckets(fp, bucket_mb=25.0, split_at=stage_splits(model, fp) if collective else ()).attach()     # buckets end at the segmented backward's phase boundaries
    ar_ms = []
    seg = [None]

    def forward_loss():
        pose, emo, _s, pred, _t = model(g["spec"], g["text"], g["pre_pose"], None)
        # the emotion CVAE learns to reconstruct the generator's emotion feature map under the clip's label (its eval-time role:
        # sample(label) replaces that map, test_emotion_gesture_diversity_iterative.py:203-205)
        rec, mu, logvar = vae(emo.detach(), g["label"], eps)
        return F.add(F.add(F.smooth_l1_loss(pose, target, 1.0, 100.0), F.cross_entropy(pred, label)),
                     F.add(F.smooth_l1_loss(rec, emo.detach(), 1.0, 1.0), F.kld_loss(mu, logvar, 1.0)))

    # One rank: the emotion CVAE's forward + backward (its input is the DETACHED emotion map, so nothing of it feeds the generator's gradient) runs
    # on a side stream beside the generator's losses and backward -- one fork / join pair per step, captured into the step's hipGraph with it.
    # Same kernels on the same data: the parameters after a step are bitwise those of the one-stream step (tests/test_gpu_training.py).
    side = torch.cuda.Stream(dev) if (not collective and os.environ.get("EG_TRAIN_SIDE_CVAE", "1") != "0") else None
    if side is not None:
        model.aux_stream = torch.cuda.Stream(dev)            # the generator's gradient-free text branch beside its audio tower (train/nets.py)

    def forward_backward():
        if side is None:
            loss = forward_loss()
            loss.backward()
            return loss
        cur = torch.cuda.current_stream(dev)
        pose, emo, _s, pred, _t = model(g["spec"], g["text"], g["pre_pose"], None)
        emo_d = emo.detach()
        side.wait_stream(cur)
        with torch.cuda.stream(side):
            rec, mu, logvar = vae(emo_d, g["label"], eps)
            loss_v = F.add(F.smooth_l1_loss(rec, emo_d, 1.0, 1.0), F.kld_loss(mu, logvar, 1.0))
            loss_v.backward()
        loss_g = F.add(F.smooth_l1_loss(pose, target, 1.0, 100.0), F.cross_entropy(pred, label))
        loss_g.backward()
        cur.wait_stream(side)
        return F.add(loss_g.detach(), loss_v.detach())

    def step(timed=True):
        opt.zero_grad()
        gb.begin()
        loss = forward_backward()
        if not timed:                   # inside a stream capture: no timing events
            gb.finish()
            opt.step(collected=True)
            return loss
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        gb.finish()                     # waits for the bucket all-reduces still running behind backward: the EXPOSED part
        e1.record()
        opt.step(collected=True)
        ar_ms.append((e0, e1))
        return loss

    # launches of the library per step, counted on one eager step (a captured step records the same launches once)
    first_loss = float(step().detach())          # step 1 from the synthetic initial weights: comparable across precisions / launch modes
    torch.cuda.synchronize(dev)
    n0 = int(lib.eg_launch_count())
    step()
    torch.cuda.synchronize(dev)
    launches = int(lib.eg_launch_count()) - n0
    ar_ms.clear()

    run = step
    mode = "eager (autograd issues every kernel)"
    if graph:
        from emotiongestures_amd.train.graph import GraphedStep, SegmentedStep
        if not collective:
            gs = GraphedStep(lambda _inputs: step(timed=False), g, opt, warmup=max(1, warmup), stochastic=bool(dropout))
            run = gs.run
            mode = "one captured hipGraph per step"
        else:
            # data parallel: the step is cut into per-bucket graph segments (forward + the backward up to bucket 0 complete, then one segment per
            # further bucket); bucket k's all-reduce runs on the side stream while segment k+1 replays, Adam follows the last reduction
            gb.payload = payload or os.environ.get("EG_GRAD_PAYLOAD", "f32")          # "bf16": buckets travel as bfloat16 (half the xGMI bytes)
            # the emotion CVAE's forward + backward beside the generator's losses and backward here too (as in the one-graph step above): forked inside
            # segment 0's loss function, joined at the end of segment 0 (after_backward), all inside that captured segment
            side_dp = torch.cuda.Stream(dev) if os.environ.get("EG_TRAIN_SIDE_CVAE", "1") != "0" else None
            if side_dp is not None and os.environ.get("EG_TRAIN_DP_AUX", "1") != "0":
                model.aux_stream = torch.cuda.Stream(dev)        # the generator's text branch and prior-pose branch beside its audio tower, as in the one-graph step
            hold = {}

            def loss_dp():
                cur = torch.cuda.current_stream(dev)
                pose, emo, _s, pred, _t = model(g["spec"], g["text"], g["pre_pose"], None)
                emo_d = emo.detach()
                side_dp.wait_stream(cur)
                with torch.cuda.stream(side_dp):
                    rec, mu, logvar = vae(emo_d, g["label"], eps)
                    loss_v = F.add(F.smooth_l1_loss(rec, emo_d, 1.0, 1.0), F.kld_loss(mu, logvar, 1.0))
                    loss_v.backward()
                    hold["v"] = loss_v.detach()
                return F.add(F.smooth_l1_loss(pose, target, 1.0, 100.0), F.cross_entropy(pred, label))

            def join_dp(step_):
                torch.cuda.current_stream(dev).wait_stream(side_dp)
                step_.loss = F.add(step_.loss.detach(), hold["v"])            # the reported loss: generator + CVAE, as the one-stream step's

            ss = SegmentedStep(forward_loss if side_dp is None else loss_dp, gb, opt, device=dev, warmup=max(1, warmup), stochastic=bool(dropout),
                               after_backward=None if side_dp is None else join_dp)
            run = lambda: ss.run(exposed=ar_ms)
            seg[0] = ss
            mode = f"{ss.n_segments} hipGraph segments per step (backward cut at the tower output and per tower stage), bucket all-reduces between them on a side stream ({gb.payload} payload)"

    for _ in range(max(1, warmup)):
        loss = run()
    barrier()
    ar_ms.clear()
    t0 = time.perf_counter()
    for _ in range(steps):
        loss = run()
    barrier()
    el = time.perf_counter() - t0
    exposed = float(np.mean([a.elapsed_time(b) for a, b in ar_ms])) if ar_ms else None
    by_rank = None
    if dist is not None:
        # every rank's own clock over the same K steps (both ends behind a barrier, so they differ only by barrier skew) and its exposed wait
        on = dev if backend == "nccl" else "cpu"
        mine = torch.tensor([el, -1.0 if exposed is None else exposed], device=on, dtype=torch.float64)
        seen = [torch.zeros_like(mine) for _ in range(world)]
        dist.all_gather(seen, mine)
        els = [float(t[0]) for t in seen]
        exs = [float(t[1]) for t in seen if float(t[1]) >= 0]
        el = max(els)
        by_rank = {"ms_per_step_max": round(max(els) / steps * 1e3, 3), "ms_per_step_min": round(min(els) / steps * 1e3, 3),
                   "allreduce_exposed_ms_per_step_max": round(max(exs), 3) if exs else None,
                   "allreduce_exposed_ms_per_step_min": round(min(exs), 3) if exs else None}
    value = B * world * steps / el
    tf = value * TRAIN_FLOP_PER_CLIP / 1e12 / world
    out = {"value": round(value, 2), "ms_per_step": round(el / steps * 1e3, 3), "dtype": precision, "clips_per_gpu_per_step": B,
           "dropout": "reference placements active (train_dropout)" if dropout else "p = 0 (gradient-parity configuration)",
           "launch": mode, "library_launches_per_step": launches, "first_loss": first_loss, "final_loss": float(loss.detach()),
           "algorithmic_tflops_per_gpu": round(tf, 1), "frac_of_mfma_peak": round(tf / PEAK_TFLOPS["bf16x3" if precision != "f32" else "f32"], 4),
           "allreduce_exposed_ms_per_step": None if exposed is None else round(exposed, 3),
           "allreduce_exposed_bytes_per_step": None if seg[0] is None else seg[0].exposed_bytes(),
           "trainable_parameters": int(sum(p.numel() for p in fp.params)), "buckets": len(gb.buckets)}
    if by_rank is not None:
        out["ranks"] = by_rank
        if by_rank["allreduce_exposed_ms_per_step_max"] is not None:          # the step waits for the slowest rank's exposed reduction
            out["allreduce_exposed_ms_per_step"] = by_rank["allreduce_exposed_ms_per_step_max"]
        out["gradient_payload"] = gb.payload if graph and collective else "f32"
    if os.environ.get("EG_TRAIN_DIGEST") == "1":     # bitwise identity of the parameters after warmup + steps (tests: segmented step over RCCL vs the one-graph step)
        import hashlib
        out["param_digest"] = hashlib.sha256(fp.flat.cpu().numpy().tobytes()).hexdigest()
    del model, vae, both, fp, opt, gb
    return out


def emit(line):
    """The ONE JSON line, as the LAST line of stdout: RCCL writes its version banner through libc's stdout buffer, which would otherwise be flushed
    behind Python's at exit."""
    import ctypes
    try:
        ctypes.CDLL(None).fflush(None)
    except Exception:       # noqa: BLE001
        pass
    sys.stdout.flush()
    print(json.dumps(line), flush=True)


def collective_facts(dist, backend, dev):
    """What the process group really is: backend, RCCL version, and the ranks an all_reduce / all_gather actually reached (every rank calls)."""
    if dist is None:
        return None
    on = dev if backend == "nccl" else "cpu"
    one = torch.ones(1, device=on, dtype=torch.float64)
    dist.all_reduce(one, op=dist.ReduceOp.SUM)
    mine = torch.tensor([dist.get_rank()], device=on, dtype=torch.int64)
    seen = [torch.zeros_like(mine) for _ in range(dist.get_world_size())]
    dist.all_gather(seen, mine)
    ver = None
    if backend == "nccl":
        try:
            ver = ".".join(str(v) for v in torch.cuda.nccl.version())
        except Exception as e:      # noqa: BLE001
            ver = f"unavailable ({type(e).__name__})"
    return {"backend": "rccl (torch.distributed 'nccl')" if backend == "nccl" else backend, "rccl_version": ver, "world_size": dist.get_world_size(),
            "ranks_seen_by_all_reduce": int(one.item()), "ranks_seen_by_all_gather": sorted(int(t.item()) for t in seen),
            "forced_at_world_1": os.environ.get("EG_FORCE_COLLECTIVES") == "1" and dist.get_world_size() == 1}


def train_worker(args, rank, world, dev, dist, backend):
    """--train: data-parallel training step of the generator + emotion CVAE (SURVEY.md §8d cfg 3, §8e): every rank its own synthetic
    clips, gradients averaged with bucketed all-reduces (emotiongestures_amd/train/optim.py).  The reported leg runs the arithmetic and
    launch mode the flags name (default: split-bf16 MFMA, the step replayed from captured hipGraphs); `f32_eager` is the
    gradient-parity configuration (fp32 operators issued through autograd) timed in the same run."""
    B = args.train_batch
    main_leg = train_leg(dev, B, args.train_precision, not args.no_train_graph, args.steps, args.warmup, rank, world, dist, backend,
                         dropout=not args.no_train_dropout)
    parity = None
    if not args.no_extra_legs and (args.train_precision != "f32" or not args.no_train_graph):
        parity = train_leg(dev, B, "f32", False, args.steps, args.warmup, rank, world, dist, backend)
    facts = collective_facts(dist, backend, dev)
    if dist is not None:
        dist.barrier()
        dist.destroy_process_group()
    if rank == 0:
        nparam = main_leg.pop("trainable_parameters")
        nb = main_leg.pop("buckets")
        line = {
            "metric": "training clips/sec (generator + emotion CVAE: forward + backward + all-reduce + Adam)", "value": main_leg["value"],
            "unit": "clips/s", "n_gpus": world, "steps": args.steps, "warmup": args.warmup, "ms_per_step": main_leg["ms_per_step"],
            "higher_is_better": True, "scaling": "weak", "vs_baseline": None, "dtype": main_leg["dtype"], "data": "synthetic",
            "config": {"workload": "TED clips: spec(128x124) + prior poses -> generator (train mode) -> 100*smooth_l1(pose) + CE(emotion); emotion map -> CVAE (train mode) -> smooth_l1(recon) + KLD; backward -> Adam",
                       "clips_per_gpu_per_step": B, "global_batch": B * world, "parallelism": f"data parallel x{world}, bucketed gradient all-reduce (25 MB buckets, backward order, side stream)",
                       "trainable_parameters": nparam, "gradient_bytes_per_step": 4 * nparam, "buckets": nb},
            "dropout": main_leg["dropout"], "final_loss": main_leg["final_loss"], "allreduce_exposed_ms_per_step": main_leg["allreduce_exposed_ms_per_step"],
            "allreduce_exposed_bytes_per_step": main_leg["allreduce_exposed_bytes_per_step"], "launch": main_leg["launch"],
            "library_launches_per_step": main_leg["library_launches_per_step"], "algorithmic_tflops_per_gpu": main_leg["algorithmic_tflops_per_gpu"],
            "frac_of_mfma_peak": main_leg["frac_of_mfma_peak"]}
        if "param_digest" in main_leg:
            line["param_digest"] = main_leg["param_digest"]
        if "ranks" in main_leg:
            line["ranks"] = main_leg["ranks"]
        if facts is not None:
            line["collectives"] = facts
        if parity is not None:
            parity.pop("trainable_parameters"); parity.pop("buckets")
            line["f32_eager"] = parity
        emit(line)
    return 0


def main():
    args = parse_args()
    launched = "WORLD_SIZE" in os.environ and "RANK" in os.environ
    if not launched and args.gpus > 1:
        return self_launch(args.gpus)
    rank = int(os.environ.get("RANK", "0"))
    world = int(os.environ.get("WORLD_SIZE", "1"))
    local = int(os.environ.get("LOCAL_RANK", "0"))
    if world != max(args.gpus, 1):
        raise SystemExit(f"bench.py: --gpus {args.gpus} but the launcher started WORLD_SIZE={world} ranks")
    if os.environ.get("EG_BENCH_DRY") == "1":
        return dry_worker(args, rank, world)
    if not torch.cuda.is_available():
        raise SystemExit("bench.py needs a GPU: the HIP path has no CPU fallback")
    ndev = torch.cuda.device_count()
    backend = os.environ.get("EG_BENCH_BACKEND", "nccl")       # "nccl" is RCCL on ROCm
    if local >= ndev:
        if backend == "nccl":
            raise SystemExit(f"bench.py: local rank {local} has no GPU ({ndev} visible): one process per GPU")
        local = local % max(ndev, 1)        # several gloo ranks on one GPU: only the single-GPU test of this launcher path
    torch.cuda.set_device(local)
    dev = torch.device("cuda", local)
    dist = None
    if world > 1 or os.environ.get("EG_FORCE_COLLECTIVES") == "1":      # the latter: a 1-rank RCCL group, so that every collective site really issues its call
        from emotiongestures_amd.dist import init_process_group, new_store_path
        if "MASTER_PORT" not in os.environ and "EG_DIST_STORE" not in os.environ:       # a lone rank that was asked for a group (EG_FORCE_COLLECTIVES)
            os.environ["EG_DIST_STORE"] = new_store_path("eg_bench1")
        dist = init_process_group(backend, rank, world, device=dev)

    from emotiongestures_amd import _lib
    lib = _lib.load()
    if args.train:
        return train_worker(args, rank, world, dev, dist, backend)
    gen, vae, mel, sd_g, sd_v = build_models(args.precision, dev)
    lanes = 1 if args.no_graph else max(1, args.in_flight)
    gen.concurrent = lanes == 1 and (args.concurrent or not args.no_concurrent)
    gen.fold_affine = bool(args.fold_affine)
    gen.fuse_se = not args.no_fuse_se
    B = args.batch
    inp = make_inputs(B, seed=1000 + rank)          # every rank generates its own shard of clips
    g = {k: torch.from_numpy(v).to(dev) for k, v in inp.items()}

    def make_step(gen, vae, mel, side):
        def step():
            with torch.no_grad():
                cur = torch.cuda.current_stream(dev)
                if side is not None:        # the CVAE draw does not depend on the audio: fork it beside the mel front-end
                    side.wait_stream(cur)
                    with torch.cuda.stream(side):
                        sampled = vae.sample(g["label"], z=g["z"])
                    spec = mel(g["audio"], out_frames=124)
                    cur.wait_stream(side)
                else:
                    spec = mel(g["audio"], out_frames=124)
                    sampled = vae.sample(g["label"], z=g["z"])
                return gen(spec, g["text"], g["pre_pose"], sampled)
        return step

    def barrier():
        torch.cuda.synchronize(dev)
        if dist is not None:
            dist.barrier()
        torch.cuda.synchronize(dev)

    def make_runner(gen_, lanes_):
        """(step callable, pipeline or None) for a generator: hipGraph lanes unless --no-graph."""
        if args.no_graph:
            # eager (profiling passes): EG_BENCH_SHARED_CHIP=1 keeps the tile policy of the timed 4-lane configuration, so that the per-kernel PMC
            # numbers describe the kernels the headline runs
            gen_.shared_chip = os.environ.get("EG_BENCH_SHARED_CHIP") == "1"
            side = torch.cuda.Stream(dev) if gen_.concurrent else None
            return make_step(gen_, vae, mel, side), None
        # One lane = one captured step (all launches + the fork/join of its side streams) replayed on its own stream.
        # Steps are independent batches, so `--in-flight` lanes (own workspaces, I/O buffers and outputs each; the models and
        # their weights arena are shared) are replayed round-robin: the low-occupancy GEMM / attention phase of one batch
        # overlaps the convolution phase of the next.  The timed region still covers exactly K complete steps.
        from emotiongestures_amd.pipeline import ClipPipeline
        pipe_ = ClipPipeline((gen_, vae, mel), g, dev, lanes=lanes_, branch_streams=gen_.concurrent)
        return (lambda: pipe_.outputs(pipe_.launch_next())), pipe_

    def timed(step_, steps, warmup):
        for _ in range(warmup):
            o = step_()
        barrier()
        t0 = time.perf_counter()
        for _ in range(steps):
            o = step_()
        barrier()
        el = time.perf_counter() - t0
        if dist is not None:
            tt = torch.tensor([el], device=dev if backend == "nccl" else "cpu", dtype=torch.float64)
            dist.all_reduce(tt, op=dist.ReduceOp.MAX)
            el = float(tt.item())
        return el, o

    step, pipe = make_runner(gen, lanes)
    elapsed, out = timed(step, args.steps, args.warmup)
    total_clips = B * world * args.steps
    value = total_clips / elapsed

    timed_concurrent = bool(gen.concurrent)
    launch_desc = "eager" if pipe is None else f"hipGraph replay, {lanes} step(s) in flight"
    # every lane ran the same resident batch: their outputs must agree bit for bit (catches any cross-lane interference)
    lanes_equal = None
    if pipe is not None and lanes > 1:
        base = pipe.outputs(0)
        lanes_equal = all(torch.equal(a, b) for i in range(1, lanes) for a, b in zip(pipe.outputs(i), base))
        if not lanes_equal:
            raise SystemExit("bench.py: lanes disagree on the same batch -- results depend on concurrent work (invalid run)")

    # ---- sustained leg: the same step loop for >= 2 s of wall clock, so that clocks / thermals under load are represented ----
    sustained = None
    if not args.no_extra_legs:
        chunk = max(args.steps, 20)
        done, el_tot = 0, 0.0
        while el_tot < args.sustain_seconds and done < 200 * chunk:
            el, _ = timed(step, chunk, 0)
            el_tot += el
            done += chunk
        sustained = {"steps": done, "seconds": round(el_tot, 3), "value": round(B * world * done / el_tot, 2),
                     "ms_per_step": round(el_tot / done * 1e3, 4)}

    # ---- parity of the WHOLE timed batch against the CPU oracle (rank 0), and the "FGD vs ref" half of the metric ----
    parity = None
    fgd = None
    roof = None
    cpu = None
    f32 = None
    pose_ref = None
    if rank == 0:
        from emotiongestures_amd.builders import clip_rel_l2
        from oracle import emogest_oracle as O
        with torch.no_grad():
            tn = {k: torch.from_numpy(v) for k, v in inp.items() if k != "audio"}
            spec_ref = torch.from_numpy(O.melspectrogram(inp["audio"], out_frames=124))
            s_ref = O.cvae_sample(sd_v, tn["label"], tn["z"])
            pose_ref = O.generator_forward(sd_g, O.GenCfg(), spec_ref, tn["text"], tn["pre_pose"], s_ref)[0]
        parity = clip_rel_l2(out[0].cpu().numpy(), pose_ref.numpy())          # max over all B clips of the timed batch
    if rank == 0 and B >= 32:
        # Frechet distance between the FGD auto-encoder latents (model/FGD.py:26-82, HIP path, synthetic weights) of the GPU
        # poses and of the CPU oracle's poses for the same B clips (B x 34 feature rows)
        from emotiongestures_amd.harness import MLP_Reconstruct, calculate_frechet_distance
        from emotiongestures_amd.synth import load_synth_weights
        with torch.no_grad():
            ae = load_synth_weights(MLP_Reconstruct(pose_dim=126), 5).eval().to(dev)
            fa = ae(out[0].contiguous())[1].reshape(-1, 512).cpu().numpy().astype(np.float64)
            fb = ae(pose_ref.to(dev))[1].reshape(-1, 512).cpu().numpy().astype(np.float64)
        fgd = float(np.real(calculate_frechet_distance(fa.mean(0), np.cov(fa, rowvar=False), fb.mean(0), np.cov(fb, rowvar=False))))

    # ---- extra legs in the same run: f32 parity mode (SURVEY.md §8d cfg 2), the flagged affine folds, a 256-clip batch ----
    extra = {}
    if not args.no_extra_legs:
        def leg(precision, fold, batch, lanes_):
            nonlocal g
            g_main = g
            gen_x = build_models(precision, dev)[0]
            gen_x.concurrent = False
            gen_x.fold_affine = fold
            if batch != B:
                inp_x = make_inputs(batch, seed=1000 + rank)
                g = {k: torch.from_numpy(v).to(dev) for k, v in inp_x.items()}
            try:
                step_x, pipe_x = make_runner(gen_x, lanes_)
                el_x, out_x = timed(step_x, args.steps, max(2, args.warmup // 2))
            finally:
                g = g_main
            rec = None
            if rank == 0:
                rec = {"value": round(batch * world * args.steps / el_x, 2), "ms_per_step": round(el_x / args.steps * 1e3, 4), "dtype": precision,
                       "clips_per_gpu_per_step": batch, "fold_affine": fold, "steps_in_flight": lanes_}
                if batch == B:
                    rec["pose_rel_l2_vs_cpu_oracle"] = clip_rel_l2(out_x[0].cpu().numpy(), pose_ref.numpy())
            del step_x, pipe_x, gen_x
            torch.cuda.empty_cache()
            return rec
        if args.precision != "f32":
            f32 = leg("f32", False, B, lanes)
            if f32 is not None:
                f32["mfma_peak_tflops"] = PEAK_TFLOPS["f32"]
                f32["frac_of_f32_mfma_peak"] = round(f32["value"] * FLOP_PER_CLIP / 1e12 / PEAK_TFLOPS["f32"] / world, 4)
        if not args.fold_affine:
            extra["fold_affine"] = leg(args.precision, True, B, lanes)          # same arithmetic mode, fewer products than the reference graph
        if B == 64:
            extra["b256"] = leg(args.precision, False, 256, 2)                  # informational: the same path at 4x the batch (NOT the headline config)
        if rank == 0 and world == 1:
            for name, fn in (("gpu_b1", lambda: gpu_b1_latency(gen, vae, mel, inp, sd_g, sd_v, dev)),
                             ("beat_long", lambda: beat_long_leg(args.precision, dev, args.steps)),
                             ("diversity_32", lambda: diversity_leg(args.precision, dev, max(4, args.steps // 4)))):
                try:
                    extra[name] = fn()
                except Exception as e:          # informational legs: reported, never fatal to the headline
                    print(f"bench.py: extra leg {name} FAILED: {e!r}", file=sys.stderr)
                    extra[name] = {"error": repr(e)[:400]}
                torch.cuda.empty_cache()

    # ---- N > 1: BASELINE configs[2] -- the data-parallel training step (16 clips per GPU: global batch 128 on 8 GPUs), every rank, gradients
    # averaged over RCCL in per-stage buckets between the backward's graph segments; fp32 and bf16 bucket payloads.  Then the process group is torn
    # down: what follows (roofline, per-launch events) is rank 0's alone, and no other rank sits in a collective waiting for it. ----
    train_dp = None
    facts = None
    if dist is not None:
        if world > 1 and not args.no_train_legs:
            step = pipe = None
            torch.cuda.empty_cache()
            train_dp = {}
            for pay in ("f32", "bf16"):
                rec = train_leg(dev, args.dp_train_batch, "bf16x3", True, max(5, args.steps // 2), 3, rank, world, dist, backend, dropout=True, payload=pay)
                nparam, nb = rec.pop("trainable_parameters"), rec.pop("buckets")
                rec["gradient_bytes_per_step"] = (4 if pay == "f32" else 2) * nparam
                rec["buckets"] = nb
                train_dp[f"payload_{pay}"] = rec
        facts = collective_facts(dist, backend, dev)
        dist.barrier()
        dist.destroy_process_group()
        dist = None
        if rank != 0:
            return 0

    # ---- roofline leg: per-launch HIP-event timing of the contraction kernels over K more steps (same stream) ----
    if rank == 0 and not args.no_roofline:
        cap = 400 * max(args.steps, 1)
        gen.concurrent = False                  # per-launch durations are only meaningful without overlapping side streams
        gen.shared_chip = lanes > 1             # the tile policy of the TIMED configuration, for this eager step (pipelines carry their own hint)
        eager_step = make_step(gen, vae, mel, None)
        for _ in range(3):                      # warm: clocks up, kernels / weights resident, before per-launch events are taken
            eager_step()
        torch.cuda.synchronize(dev)
        _lib.check(lib.eg_profile_enable(cap), "eg_profile_enable")
        for _ in range(args.steps):
            eager_step()                    # per-launch event timing runs eagerly (events are not part of the graph)
        torch.cuda.synchronize(dev)
        lib.eg_profile_disable()
        tags = np.zeros(cap, np.int64); fl = np.zeros(cap, np.float64); ms = np.zeros(cap, np.float32); wgs = np.zeros(cap, np.int32)
        lib.eg_profile_read_workgroups(wgs.ctypes.data_as(C.c_void_p), cap)         # before eg_profile_read, which resets the record list
        n = lib.eg_profile_read(tags.ctypes.data_as(C.c_void_p), fl.ctypes.data_as(C.c_void_p), ms.ctypes.data_as(C.c_void_p), cap)
        if n < 0:
            _lib.check(n, "eg_profile_read")
        tags, fl, ms, wgs = tags[:n], fl[:n], ms[:n], wgs[:n]
        # share of the chip a launch occupies: a product on 68 workgroups (128 x 128 tiles at N = 512) holds a quarter of the 256 CUs for its
        # duration and the other lanes' convolutions run on the rest; 0 = not recorded (convolutions: grids of >= 512 workgroups)
        share = np.where(wgs > 0, np.minimum(1.0, wgs / 256.0), 1.0)
        groups = {}
        for tg in np.unique(tags):
            sel = tags == tg
            groups[int(tg)] = (float(ms[sel].sum()), float(ms[sel].mean()), float(fl[sel].mean()), int(sel.sum()))
        contraction = {k: v for k, v in groups.items() if v[2] > 0}             # tags with FLOPs: GEMM / conv / attention
        dom = max(contraction, key=lambda k: contraction[k][0])
        tot_ms, avg_ms, flop, cnt = groups[dom]
        achieved = flop / (avg_ms * 1e-3) / 1e12
        peak = PEAK_TFLOPS[args.precision]
        traffic = measured_traffic(dom, args.precision)
        seld = tags == dom
        cu_ms = float((ms[seld] * share[seld]).sum())
        roof = {"bound": "mfma", "kernel": kernel_name(dom), "achieved": round(achieved, 2), "peak": peak, "unit": "TFLOP/s",
                "frac": round(achieved / peak, 4), "traffic": traffic, "avg_launch_ms": round(avg_ms, 4),
                "launches_per_step": cnt // max(args.steps, 1), "kernel_ms_per_step_isolated": round(tot_ms / args.steps, 4),
                # the same launches weighted with the share of the 256 CUs each occupies (workgroups / 256, capped at 1): what the family costs
                # the step as it is run -- four batches in flight, the rest of the chip busy with other lanes' kernels -- and its rate per occupied CU share
                "kernel_cu_ms_per_step": round(cu_ms / args.steps, 4), "mean_workgroups_per_launch": round(float(wgs[seld].mean()), 1),
                "achieved_per_occupied_share": round(float(fl[seld].sum()) / (cu_ms * 1e-3) / 1e12, 2) if cu_ms > 0 else None,
                "tile_policy": os.environ.get("EG_GEMM_TILE", "auto: the caller's shared-chip hint (several batches in flight) -> 128 x 128 from 64 workgroups up; stand-alone 64 x 64"),
                "flop_per_launch": flop,
                "by_kernel_ms_per_step": {kernel_name(k): round(v[0] / args.steps, 4) for k, v in sorted(groups.items())},
                "by_kernel_tflops": {kernel_name(k): round(v[2] / (v[1] * 1e-3) / 1e12, 1) for k, v in sorted(contraction.items())}}
        # the family that bounds the step: all 3x3 convolutions of the audio tower together (isolated per-launch times, summed per step)
        conv_tags = [k for k in contraction if k >= 1000]
        if conv_tags:
            c_ms = sum(groups[k][0] for k in conv_tags) / args.steps
            c_fl = sum(groups[k][2] * groups[k][3] for k in conv_tags) / args.steps
            roof["conv_family"] = {"kernels": len(conv_tags), "launches_per_step": sum(groups[k][3] for k in conv_tags) // max(args.steps, 1),
                                   "ms_per_step_isolated": round(c_ms, 4), "gflop_per_step": round(c_fl / 1e9, 1),
                                   "achieved": round(c_fl / (c_ms * 1e-3) / 1e12, 1), "peak": peak, "unit": "TFLOP/s",
                                   "frac": round(c_fl / (c_ms * 1e-3) / 1e12 / peak, 4), "share_of_step_flop": round(c_fl / (B * FLOP_PER_CLIP), 3)}
        roof["step_frac_of_peak"] = round(value / world * FLOP_PER_CLIP / 1e12 / peak, 4)       # whole step: clips/s x 9.235 GFLOP over the dense MFMA peak
        # memory side of the one stage whose convolutions are traffic- rather than MFMA-limited (32 -> 32 channels at 128 x 124): algorithmic
        # bytes per launch (input + output map, + the residual for the conv2 launches: average of the block's two) over the measured duration
        c32 = 32 * 1000000 + 32 * 1000 + 100 + 1
        if c32 in groups and B > 0:
            amap = B * 128 * 124 * 32 * 4
            alg = 2.5 * amap
            hb = alg / (groups[c32][1] * 1e-3) / 1e9
            roof["hbm_view_conv32"] = {"bound": "hbm", "kernel": kernel_name(c32), "achieved": round(hb, 1), "peak": 8000.0, "unit": "GB/s",
                                       "frac": round(hb / 8000.0, 4), "algorithmic_bytes": int(alg), "traffic": measured_traffic(c32, args.precision),
                                       "avg_launch_ms": round(groups[c32][1], 4)}
    # ---- training legs in the default line (BASELINE configs[2] on one GPU): 16 clips per step (the 8-GPU global-batch-128 share) and 128 ----
    train = None
    if rank == 0 and world == 1 and not args.no_extra_legs and not args.no_train_legs:
        step = pipe = None                # release the lanes' graphs and workspaces before the training legs allocate theirs
        torch.cuda.empty_cache()
        train = {}
        for tb in (16, 128):
            try:
                # headline of the leg: the graph the reference trains (every Dropout active); `parity_p0` = the same step with p = 0 (the
                # configuration the gradient tests pin against the reference's autograd); `f32_eager` = fp32 operators issued through autograd
                rec = train_leg(dev, tb, "bf16x3", True, max(5, args.steps // 2), 3, dropout=True)
                p0 = train_leg(dev, tb, "bf16x3", True, max(5, args.steps // 2), 3, dropout=False)
                par = train_leg(dev, tb, "f32", False, 3, 1)
                for k in ("trainable_parameters", "buckets", "allreduce_exposed_ms_per_step"):
                    rec.pop(k, None)
                rec["parity"] = ("element-wise with Dropout ON and at p = 0: the library's masks are a pure function of (seed, stream offset + index, p); "
                                 "tests/test_gpu_training.py injects them at the reference's 26 Dropout sites (oracle autograd, and the reference's own modules "
                                 "for tests/golden/dropout_grads.npz) and compares every parameter gradient of this step (TED, B = 2, f32); `parity_p0` times "
                                 "the same step with every Dropout at p = 0")
                rec["parity_p0"] = {k: p0[k] for k in ("value", "ms_per_step", "first_loss", "final_loss", "library_launches_per_step", "frac_of_mfma_peak")}
                rec["f32_eager"] = {k: par[k] for k in ("value", "ms_per_step", "first_loss", "library_launches_per_step", "frac_of_mfma_peak")}
                rec["hbm"] = train_hbm_gb_per_step(tb)
                train[f"b{tb}"] = rec
            except Exception as e:      # a failed leg must not take the (already measured) headline with it: it is reported in the line, loudly
                print(f"bench.py: training leg at {tb} clips FAILED: {e!r}", file=sys.stderr)
                train[f"b{tb}"] = {"error": repr(e)[:400]}
                torch.cuda.empty_cache()
    if rank == 0 and world == 1 and not args.no_cpu_baseline:
        cpu = cpu_baseline(sd_g, sd_v, inp)

    if rank == 0:
        if train_dp is not None:
            train = {f"b{args.dp_train_batch}_data_parallel": dict(train_dp, global_batch=args.dp_train_batch * world,
                                               note="BASELINE configs[2]: generator + emotion CVAE, reference Dropout placements active, bf16x3 arithmetic, one "
                                                    "process per GPU, SegmentedStep (per-stage hipGraph segments, bucket all-reduces between them on a side stream)")}
        line = {
            "metric": "gesture clips/sec (34 frames, 43 joints)", "value": round(value, 2), "unit": "clips/s",
            "n_gpus": world, "steps": args.steps, "warmup": args.warmup, "ms_per_step": round(elapsed / args.steps * 1e3, 4),
            "higher_is_better": True, "scaling": "weak", "vs_baseline": None, "dtype": args.precision, "data": "synthetic",
            "config": {"workload": "TED clips: 4 s 16 kHz audio -> mel(128x124) -> CVAE sample -> generator -> 34x126 pose",
                       "clips_per_gpu_per_step": B, "global_batch": B * world, "variant": "Models_spatial_memory",
                       "parallelism": f"clip-sharded x{world}, no data-path collective",
                       "launch": launch_desc,
                       "branch_streams": timed_concurrent, "fold_affine": bool(getattr(gen, "fold_affine", False)),
                       "algorithmic_gflop_per_clip": round((FLOP_PER_CLIP + MEL_FLOP_PER_CLIP + CVAE_FLOP_PER_CLIP) / 1e9, 3)},
            "pose_rel_l2_vs_cpu_oracle": parity, "parity_clips_checked": B, "fgd_vs_cpu_oracle": fgd,
            "lanes_bitwise_equal": lanes_equal, "sustained": sustained, "f32": f32, "extra_legs": extra, "train": train, "roofline": roof, "cpu_baseline": cpu,
        }
        if facts is not None:       # N > 1 (or EG_FORCE_COLLECTIVES): what the barrier / max-over-ranks really ran over
            line["collectives"] = facts
        emit(line)
    return 0


GEMM_NAMES = {2: "gemm_glds_kernel (fp32 X via LDS-DMA)", 3: "gemm_presplit_kernel (pre-split X)", 4: "gemm_bf16_kernel (causal shift)",
              5: "gemm_kernel (f32 MFMA)", 6: "attention_mfma_kernel", 7: "ffn_slab_kernel (fused w_1 -> ReLU -> w_2 + residual, hidden in LDS)"}


def kernel_name(tag):
    if tag in GEMM_NAMES:
        return GEMM_NAMES[tag]
    if tag >= 1000:
        return f"conv3x3<cin={tag // 1000000},cout={(tag // 1000) % 1000},stride={(tag // 100) % 10}>"
    return f"tag{tag}"


if __name__ == "__main__":
    sys.exit(main())
