"""Print gradient norms / a checksum of a few parameters for the fused (chain) and the operator-by-operator training step (the A/B of
tests/test_gpu_training.py::test_fused_blocks_step_equals_the_operator_by_operator_step) -- for comparing two builds of the library."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "tests"))
import numpy as np, torch
from conftest import build_mirror
from emotiongestures_amd.CAVE.BEAT_CVAE import MLP_Reconstruct_v3
from emotiongestures_amd.synth import hash_unit, load_synth_weights, synth_inputs
from emotiongestures_amd.train import functional as F, nets
from emotiongestures_amd.train.optim import flatten_parameters
DEV = "cuda:0"
def T(key, shape, lo=-1.0, hi=1.0, seed=0):
    n = int(np.prod(shape)); return torch.from_numpy((lo + (hi - lo) * hash_unit(key, n, seed)).astype(np.float32).reshape(shape))
B = 3
F.PRESPLIT_ROWS = int(os.environ.get("ROWS", "64"))
inp = synth_inputs(B, 34, 126, 4, seed=31)
g = {k: torch.from_numpy(v).to(DEV) for k, v in inp.items()}
target = T("tgt", (B, 34, 126), -0.5, 0.5).to(DEV); label = g["label"].argmax(1); eps = g["z"]
keys = ("0.prior_seq_encoder.post_header.0.weight", "0.prior_seq_encoder.pred_conv.5.weight", "0.decoder.layer_stack.0.enc_attn.w_qs.weight", "0.decoder.layer_stack.0.enc_attn.w_ks.weight",
        "0.decoder.layer_stack.0.pos_ffn.w_1.weight", "0.decoder.layer_stack.2.pos_ffn.w_2.weight", "0.post_projector.0.weight", "0.encoder.layer_stack.2.pos_ffn.w_1.weight", "0.audio_encoder.fc2.weight")
MASKS = {}
_lin, _relu, _lrelu = F._linear_ex, F.relu, F.leaky_relu
def lin_rec(x, lda, w, ldw, bias, res, y, M, N, K, relu, prec, *a, **k):
    out = _lin(x, lda, w, ldw, bias, res, y, M, N, K, relu, prec, *a, **k)
    if relu: MASKS[CUR].append(("lin %dx%dx%d" % (M, N, K), (y > 0).clone(), y.clone()))
    return out
def relu_rec(x):
    y = _relu(x); MASKS[CUR].append(("relu %s" % (tuple(x.shape),), (y.detach() > 0).clone(), x.detach().clone())); return y
def lrelu_rec(x, slope=0.2):
    y = _lrelu(x, slope); MASKS[CUR].append(("lrelu %s" % (tuple(x.shape),), (y.detach() > 0).clone(), x.detach().clone())); return y
F._linear_ex, F.relu, F.leaky_relu = lin_rec, relu_rec, lrelu_rec
for fused in (False, True):
    CUR = fused; MASKS[CUR] = []
    F.FUSE_BLOCKS = fused; F.set_precision("bf16x3")
    model = build_mirror("spatial", 34, 126, 4, 4, seed=2, precision="f32").to(DEV).train()
    vae = load_synth_weights(MLP_Reconstruct_v3(frames=34), 2).to(DEV).train()
    model.train_dropout = vae.train_dropout = True
    both = torch.nn.ModuleList([model, vae]); fp = flatten_parameters(both); fp.enable_weight_images(*nets.weight_image_plan(both))
    F.manual_seed(99)
    pose, emo, _s, pred, _t = model(g["spec"], g["text"], g["pre_pose"], None)
    rec, mu, logvar = vae(emo.detach(), g["label"], eps)
    loss = F.add(F.add(F.smooth_l1_loss(pose, target, 1.0, 100.0), F.cross_entropy(pred, label)), F.add(F.smooth_l1_loss(rec, emo.detach(), 1.0, 1.0), F.kld_loss(mu, logvar, 1.0)))
    for nm, tt in (("pose", pose), ("emo", emo), ("pred", pred), ("rec", rec), ("mu", mu), ("logvar", logvar)):
        d = tt.detach().double(); print("      %-7s sum %.10e  norm %.10e" % (nm, float(d.sum()), float(d.norm())))
    for nm, tt in (("l_pose", F.smooth_l1_loss(pose, target, 1.0, 100.0)), ("l_ce", F.cross_entropy(pred, label)), ("l_rec", F.smooth_l1_loss(rec, emo.detach(), 1.0, 1.0)), ("l_kld", F.kld_loss(mu, logvar, 1.0))):
        print("      %-7s %.10e" % (nm, float(tt.detach())))
    loss.backward(); torch.cuda.synchronize()
    gr = {n: p.grad.detach().double().cpu() for n, p in both.named_parameters() if p.grad is not None}
    print("fused" if fused else "unfused", "loss %.9f" % float(loss.detach()), "pose sum %.9f" % float(pose.detach().double().sum()))
    for k in keys: print("   %-55s norm %.12e  sum %.12e" % (k, float(gr[k].norm()), float(gr[k].sum())))
    F.unregister_weight_images(fp.images); F.reset_state(); F.PRESPLIT_ROWS = int(os.environ.get("ROWS", "64"))

a, b = MASKS[False], MASKS[True]
print("relu sites", len(a), len(b))
for i, ((na, ma, ya), (nb, mb, yb)) in enumerate(zip(a, b)):
    if na != nb or ma.shape != mb.shape: print("  site", i, "MISMATCH", na, nb); continue
    d = (ma != mb)
    if int(d.sum()):
        idx = d.nonzero()[:4].tolist()
        print("  site", i, na, "flips", int(d.sum()), "values", [(float(ya[tuple(j)]), float(yb[tuple(j)])) for j in idx])
