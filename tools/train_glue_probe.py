#!/usr/bin/env python3
"""Which torch-side (aten) launches does one eager training step issue, and from which line of the package?  (GPU box.)

    python tools/train_glue_probe.py [clips=16] [precision=bf16x3]

The library's own launches are counted by eg_launch_count(); everything else in a step (fills, copies, element-wise glue issued through torch)
shows up here grouped by (aten op, innermost emotiongestures_amd / bench frame)."""
import collections
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import torch


def trace_python_sites(step):
    """Patch the torch entry points that fill / copy and record the innermost package frame of every call during one step (the autograd
    thread included: the patches are process-wide)."""
    import traceback
    sites = collections.Counter()

    def site():
        for fr in reversed(traceback.extract_stack(limit=14)[:-2]):
            if "emotiongestures_amd" in fr.filename or fr.filename.endswith("train_glue_probe.py"):
                return f"{fr.filename.split('emotiongestures_amd/')[-1]}:{fr.lineno} {fr.line[:90] if fr.line else ''}"
        return "?"

    def wrap(owner, name, label, pred=None):
        orig = getattr(owner, name)

        def f(*a, **k):
            if pred is None or pred(*a, **k):
                sites[(label, site())] += 1
            return orig(*a, **k)
        setattr(owner, name, f)
        return owner, name, orig

    T = torch.Tensor
    patched = [wrap(T, "copy_", "copy_"), wrap(T, "zero_", "zero_"), wrap(T, "fill_", "fill_"), wrap(torch, "zeros", "zeros"),
               wrap(torch, "zeros_like", "zeros_like"), wrap(torch, "ones", "ones"), wrap(torch, "full", "full"), wrap(T, "clone", "clone"),
               wrap(T, "contiguous", "contiguous(copy)", lambda t, *a, **k: not t.is_contiguous()),
               wrap(T, "reshape", "reshape(copy)", lambda t, *a, **k: not t.is_contiguous()),
               wrap(T, "to", "to"), wrap(torch, "cat", "cat"), wrap(T, "float", "float"), wrap(T, "sum", "sum"), wrap(T, "mean", "mean"),
               wrap(T, "mul_", "mul_"), wrap(T, "add_", "add_"), wrap(T, "__getitem__", "index(tensor)", lambda t, i: isinstance(i, torch.Tensor))]
    try:
        step()
        torch.cuda.synchronize()
    finally:
        for owner, name, orig in patched:
            setattr(owner, name, orig)
    print("python-level fill / copy calls in one step:", sum(sites.values()))
    for (label, where), n in sites.most_common(70):
        print(f"  {n:4d}  {label:18s} {where}")


def main():
    B = int(sys.argv[1]) if len(sys.argv) > 1 else 16
    precision = sys.argv[2] if len(sys.argv) > 2 else "bf16x3"
    dev = torch.device("cuda:0")
    from emotiongestures_amd import _lib
    from emotiongestures_amd.builders import build_mirror
    from emotiongestures_amd.CAVE.BEAT_CVAE import MLP_Reconstruct_v3
    from emotiongestures_amd.synth import hash_unit, load_synth_weights, synth_inputs
    from emotiongestures_amd.train import functional as F
    from emotiongestures_amd.train.optim import FlatAdam, GradBuckets, flatten_parameters
    lib = _lib.load()
    inp = synth_inputs(B, 34, 126, 4, seed=2000)
    g = {k: torch.from_numpy(v).to(dev) for k, v in inp.items()}
    target = torch.from_numpy((hash_unit("train.target_pose", B * 34 * 126, 2000) - 0.5).astype(np.float32).reshape(B, 34, 126)).to(dev)
    label = g["label"].argmax(1)
    eps = torch.from_numpy(synth_inputs(B, seed=3000)["z"]).to(dev)
    F.set_precision(precision)
    model = build_mirror("spatial", 34, 126, 4, 4, seed=0, precision="f32").to(dev).train()
    vae = load_synth_weights(MLP_Reconstruct_v3(frames=34), 0).to(dev).train()
    fp = flatten_parameters(torch.nn.ModuleList([model, vae]))
    fp.enable_weight_images()
    opt = FlatAdam(fp, lr=2e-4, betas=(0.5, 0.999), weight_decay=1e-5)
    gb = GradBuckets(fp, bucket_mb=25.0).attach()

    def step():
        opt.zero_grad()
        gb.begin()
        pose, emo, _s, pred, _t = model(g["spec"], g["text"], g["pre_pose"], None)
        rec, mu, logvar = vae(emo.detach(), g["label"], eps)
        loss = F.add(F.add(F.smooth_l1_loss(pose, target, 1.0, 100.0), F.cross_entropy(pred, label)),
                     F.add(F.smooth_l1_loss(rec, emo.detach(), 1.0, 1.0), F.kld_loss(mu, logvar, 1.0)))
        loss.backward()
        gb.finish()
        opt.step(collected=True)
        return loss

    for _ in range(3):
        step()
    torch.cuda.synchronize()
    trace_python_sites(step)
    # parameters whose gradient arrives in fresh memory (copied into the flat buffer) instead of being written into its slice
    names = {id(p): k for k, p in torch.nn.ModuleList([model, vae]).named_parameters()}
    orig_collect = type(fp).collect_one
    copied = []

    def collect_one(self, p):
        o = self.offsets[self.index[id(p)]]
        if p.grad is not None and p.grad.data_ptr() != self.grad[o:o + p.numel()].data_ptr():
            copied.append(f"{names.get(id(p), '?')} {tuple(p.shape)}")
        return orig_collect(self, p)
    type(fp).collect_one = collect_one
    step()
    torch.cuda.synchronize()
    type(fp).collect_one = orig_collect
    print("gradients copied into the flat buffer:", len(copied))
    for c in copied:
        print("   ", c)
    n0 = int(lib.eg_launch_count())
    from torch.profiler import ProfilerActivity, profile
    with profile(activities=[ProfilerActivity.CPU, ProfilerActivity.CUDA], with_stack=True) as prof:
        step()
        torch.cuda.synchronize()
    print("library launches in the profiled step:", int(lib.eg_launch_count()) - n0)
    ev = prof.events()
    kernels = collections.Counter()
    for e in ev:
        if e.device_type == torch.autograd.DeviceType.CUDA:
            kernels[e.name[:70]] += 1
    print("device activities:", sum(kernels.values()))
    for k, n in kernels.most_common(12):
        print(f"  {n:5d}  {k}")
    # aten ops that launched something, by innermost package frame
    by_site = collections.Counter()
    t_site = collections.Counter()
    for e in ev:
        if e.device_type != torch.autograd.DeviceType.CPU or not e.name.startswith("aten::"):
            continue
        if not e.kernels:
            continue
        # skip ops nested in another aten op that already owns the kernels (only leaves own `kernels`)
        site = "?"
        for fr in (e.stack or []):
            if "emotiongestures_amd" in fr or "train_glue_probe" in fr:
                site = fr.split("emotiongestures_amd/")[-1] if "emotiongestures_amd/" in fr else fr.split("/")[-1]
                break
        by_site[(e.name, site)] += len(e.kernels)
        t_site[(e.name, site)] += sum(k.duration for k in e.kernels)
    print("torch-side launches by (aten op, package frame):", sum(by_site.values()))
    for (name, site), n in by_site.most_common(60):
        print(f"  {n:5d}  {t_site[(name, site)]:8.0f} us  {name:28s} {site[:150]}")


if __name__ == "__main__":
    main()
