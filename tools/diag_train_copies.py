"""Diagnostic: which torch-side copies (hipMemcpy D2D / elementwise copy kernels) one eager training step of bench.py's configuration issues, by call site,
and which parameters' gradients still travel through FlatParams.collect_one's copy.  Usage: python tools/diag_train_copies.py [batch]"""
import collections, os, sys, traceback
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
from emotiongestures_amd.builders import build_mirror
from emotiongestures_amd.CAVE.BEAT_CVAE import MLP_Reconstruct_v3
from emotiongestures_amd.synth import hash_unit, load_synth_weights, synth_inputs
from emotiongestures_amd.train import functional as F, nets, optim

B = int(sys.argv[1]) if len(sys.argv) > 1 else 16
dev = torch.device("cuda:0")
F.set_precision("bf16x3")
inp = synth_inputs(B, 34, 126, 4, seed=2000)
g = {k: torch.from_numpy(v).to(dev) for k, v in inp.items()}
target = torch.from_numpy((hash_unit("train.target_pose", B * 34 * 126, 2000) - 0.5).astype(np.float32).reshape(B, 34, 126)).to(dev)
label = g["label"].argmax(1)
eps = torch.from_numpy(synth_inputs(B, seed=3000)["z"]).to(dev)
model = build_mirror("spatial", 34, 126, 4, 4, seed=0, precision="f32").to(dev).train()
vae = load_synth_weights(MLP_Reconstruct_v3(frames=34), 0).to(dev).train()
model.train_dropout = vae.train_dropout = True
F.manual_seed(1234)
both = torch.nn.ModuleList([model, vae])
fp = optim.flatten_parameters(both)
fp.enable_weight_images(*nets.weight_image_plan(both))
opt = optim.FlatAdam(fp, lr=2e-4, betas=(0.5, 0.999), weight_decay=1e-5)
gb = optim.GradBuckets(fp, bucket_mb=25.0).attach()
names = {id(p): n for n, p in both.named_parameters()}

def step():
    opt.zero_grad(); gb.begin()
    pose, emo, _s, pred, _t = model(g["spec"], g["text"], g["pre_pose"], None)
    rec, mu, logvar = vae(emo.detach(), g["label"], eps)
    loss = F.add(F.add(F.smooth_l1_loss(pose, target, 1.0, 100.0), F.cross_entropy(pred, label)),
                 F.add(F.smooth_l1_loss(rec, emo.detach(), 1.0, 1.0), F.kld_loss(mu, logvar, 1.0)))
    loss.backward(); gb.finish(); opt.step(collected=True)

step(); step()
sites, copied = collections.Counter(), []
orig = {n: getattr(torch.Tensor, n) for n in ("copy_", "clone", "contiguous", "zero_", "fill_")}
def wrap(n):
    def f(self, *a, **k):
        if self.is_cuda and not (n == "contiguous" and self.is_contiguous()):
            st = [fr for fr in traceback.extract_stack()[:-1] if "emotiongestures_amd" in fr.filename or "bench" in fr.filename]
            key = " <- ".join(f"{os.path.basename(fr.filename)}:{fr.lineno}" for fr in st[-2:][::-1])
            sites[(n, key, tuple(self.shape) if self.dim() < 3 else self.numel())] += 1
        return orig[n](self, *a, **k)
    return f
for n in orig:
    setattr(torch.Tensor, n, wrap(n))
oc = optim.FlatParams.collect_one
def collect_one(self, p):
    o = self.offsets[self.index[id(p)]]
    if p.grad is not None and p.grad.data_ptr() != self.grad[o:o + 1].data_ptr():
        copied.append((names.get(id(p), "?"), tuple(p.shape)))
    return oc(self, p)
optim.FlatParams.collect_one = collect_one
step()
torch.cuda.synchronize()
for n in orig:
    setattr(torch.Tensor, n, orig[n])
print("== torch-side tensor ops of one step, by site")
for (n, key, shp), c in sorted(sites.items(), key=lambda kv: -kv[1]):
    print(f"{c:4d}  {n:10s} {shp}  {key}")
print("== gradients copied into the flat buffer:", len(copied))
for n, s in copied:
    print("   ", n, s)

# launches by launch-site label of one more step (needs EG_LAUNCH_HIST=1 in the environment)
import ctypes
from emotiongestures_amd import _lib
lib = _lib.load()
lib.eg_launch_histogram(None, 0, 1)
step()
torch.cuda.synchronize()
buf = ctypes.create_string_buffer(1 << 16)
lib.eg_launch_histogram(buf, len(buf), 1)
rows = [l.rsplit(" ", 1) for l in buf.value.decode().splitlines()]
print("== library launches of one step by site:", sum(int(c) for _, c in rows))
for n, c in sorted(rows, key=lambda r: -int(r[1])):
    print(f"{int(c):4d}  {n}")
