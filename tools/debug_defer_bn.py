import sys, torch, numpy as np
sys.path.insert(0, "/root/repo")
from emotiongestures_amd.builders import build_mirror
from emotiongestures_amd.synth import synth_inputs, hash_unit
from emotiongestures_amd.train import functional as F
from emotiongestures_amd.train.optim import flatten_parameters
DEV = "cuda:0"
B = int(sys.argv[1]) if len(sys.argv) > 1 else 4
inp = synth_inputs(B, 34, 126, 4, seed=31)
g = {k: torch.from_numpy(v).to(DEV) for k, v in inp.items()}
target = torch.from_numpy((hash_unit("t", B * 34 * 126, 1) - 0.5).astype(np.float32).reshape(B, 34, 126)).to(DEV)
label = g["label"].argmax(1)
res = []
for defer in (False, True):
    F.DEFER_BN_APPLY = defer
    F.set_precision("bf16x3")
    model = build_mirror("spatial", 34, 126, 4, 4, seed=2, precision="f32").to(DEV).train()
    fp = flatten_parameters(model)
    F.manual_seed(5)
    pose, emo, _s, pred, _t = model(g["spec"], g["text"], g["pre_pose"], None)
    loss = F.add(F.smooth_l1_loss(pose, target, 1.0, 100.0), F.cross_entropy(pred, label))
    loss.backward()
    torch.cuda.synchronize()
    res.append((float(loss.detach()), {n: p.grad.detach().clone() for n, p in model.named_parameters() if p.grad is not None},
                {n: b.detach().clone() for n, b in model.named_buffers() if "running" in n}))
(l0, g0, b0), (l1, g1, b1) = res
print("loss", l0, l1)
rel = lambda a, b: float((a.double() - b.double()).norm() / max(float(b.double().norm()), 1e-12))
errs = sorted(((rel(g1[k], g0[k]), k) for k in g0), reverse=True)
print(errs[:8])
berr = sorted(((rel(b1[k], b0[k]), k) for k in b0), reverse=True)
print(berr[:4])

# ---- per-block gradient trace: where do the two modes part?
from emotiongestures_amd.train import nets
orig = nets.se_basic_block
trace = {}
def wrapped(blk, x):
    out = orig(blk, x)
    key = id(blk)
    def hook(gr, key=key):
        trace.setdefault(key, []).append(gr.detach().clone())
    if out.requires_grad:
        out.register_hook(hook)
    return out
nets.se_basic_block = wrapped
res2 = []
for defer in (False, True):
    F.DEFER_BN_APPLY = defer
    F.set_precision("bf16x3")
    model = build_mirror("spatial", 34, 126, 4, 4, seed=2, precision="f32").to(DEV).train()
    fp = flatten_parameters(model)
    F.manual_seed(5)
    trace.clear()
    pose, emo, _s, pred, _t = model(g["spec"], g["text"], g["pre_pose"], None)
    loss = F.add(F.smooth_l1_loss(pose, target, 1.0, 100.0), F.cross_entropy(pred, label))
    loss.backward()
    torch.cuda.synchronize()
    fe = model.audio_encoder.feat_extractor
    blocks = [b for layer in (fe.layer1, fe.layer2, fe.layer3) for b in layer]
    res2.append([trace[id(b)][0] for b in blocks])
for i, (a, b) in enumerate(zip(*res2)):
    print("block", i, "d(out) rel diff", rel(b, a))

# ---- forward agreement at the tower output / pose, and the A/A noise floor (same mode twice must be bitwise)
outs = []
for defer in (False, True, True):
    F.DEFER_BN_APPLY = defer
    F.set_precision("bf16x3")
    model = build_mirror("spatial", 34, 126, 4, 4, seed=2, precision="f32").to(DEV).train()
    F.manual_seed(5)
    with torch.no_grad():
        feat = nets.resnetse_forward(model.audio_encoder.feat_extractor, g["spec"])
    pose = model(g["spec"], g["text"], g["pre_pose"], None)[0]
    outs.append((feat.clone(), pose.detach().clone()))
print("tower output rel diff (materialised vs deferred):", rel(outs[1][0], outs[0][0]), " pose:", rel(outs[1][1], outs[0][1]))
print("deferred twice bitwise:", torch.equal(outs[1][0], outs[2][0]), torch.equal(outs[1][1], outs[2][1]))
