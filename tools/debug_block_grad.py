#!/usr/bin/env python3
"""Debug aid: backward of ONE SEBasicBlock (train mode) on realistic activations, HIP vs torch CPU, tensor by tensor."""
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import torch
import torch.nn.functional as TF

from emotiongestures_amd.builders import build_mirror
from emotiongestures_amd.synth import hash_unit, synth_inputs
from emotiongestures_amd.train import functional as F
from oracle import emogest_oracle as O

dev = "cuda:0"
li, bi = (int(sys.argv[1]), int(sys.argv[2])) if len(sys.argv) > 2 else (3, 5)
model = build_mirror("spatial", 34, 126, 4, 4, seed=0, precision="f32")
sd = {k: v.detach().clone() for k, v in model.state_dict().items()}
inp = synth_inputs(2, 34, 126, 4, seed=0)
p0 = "audio_encoder.feat_extractor"
with torch.no_grad(), O.bn_training():
    x = torch.from_numpy(inp["spec"]).unsqueeze(1)
    x = O._bn(sd, p0 + ".bn1", TF.relu(TF.conv2d(x, sd[p0 + ".conv1.weight"], sd[p0 + ".conv1.bias"], padding=1)))
    done = False
    for l, n in enumerate((3, 4, 6)):
        for b in range(n):
            if (l + 1, b) == (li, bi):
                done = True
                break
            x = O.se_basic_block(sd, f"{p0}.layer{l + 1}.{b}", x, 2 if (l > 0 and b == 0) else 1)
        if done:
            break
p = f"{p0}.layer{li}.{bi}"
stride = 2 if (li > 1 and bi == 0) else 1
print("block", p, "input", tuple(x.shape), "stride", stride)
g = torch.from_numpy((hash_unit("dbg.g", x.numel(), 1) - 0.5).astype(np.float32).reshape(x.shape))
if len(sys.argv) > 3 and sys.argv[3] == "real" and (li, bi) == (3, 5):
    # the REAL upstream gradient of this block in a training step (float64 oracle), rounded to fp32
    sd64 = {k: (v.detach().double().requires_grad_(True) if v.is_floating_point() else v) for k, v in sd.items()}
    taps = {}
    target = torch.from_numpy((hash_unit("train.target_pose", 2 * 34 * 126, 0) - 0.5).reshape(2, 34, 126)).double()
    label = torch.from_numpy(inp["label"]).argmax(1)
    with O.bn_training():
        pose, _e, _s, pred, _t = O.generator_forward(sd64, O.GenCfg(), torch.from_numpy(inp["spec"]).double(), torch.from_numpy(inp["text"]),
                                                   torch.from_numpy(inp["pre_pose"]).double(), None, taps=taps)
    taps["layer3"].retain_grad()
    (100.0 * TF.smooth_l1_loss(pose, target) + TF.cross_entropy(pred, label)).backward()
    g = taps["layer3"].grad.float()
    print("real upstream gradient: norm", float(g.norm()), "per-channel |mean| / rms:",
          float((g.mean(dim=(0, 2, 3)).abs() / g.pow(2).mean(dim=(0, 2, 3)).sqrt()).median()))


def run_ref(dtype):
    t = lambda k: sd[k].detach().to(dtype).requires_grad_(True)
    W = {k: t(f"{p}.{k}") for k in ("conv1.weight", "bn1.weight", "bn1.bias", "conv2.weight", "bn2.weight", "bn2.bias", "se.fc.0.weight",
                                    "se.fc.0.bias", "se.fc.2.weight", "se.fc.2.bias")}
    xi = x.detach().clone().to(dtype).requires_grad_(True)
    r1 = TF.relu(TF.conv2d(xi, W["conv1.weight"], None, stride=stride, padding=1))
    b1 = TF.batch_norm(r1, None, None, W["bn1.weight"], W["bn1.bias"], True, 0.1, 1e-5)
    c2 = TF.conv2d(b1, W["conv2.weight"], None, padding=1)
    b2 = TF.batch_norm(c2, None, None, W["bn2.weight"], W["bn2.bias"], True, 0.1, 1e-5)
    y = b2.mean(dim=(2, 3))
    s = torch.sigmoid(TF.linear(TF.relu(TF.linear(y, W["se.fc.0.weight"], W["se.fc.0.bias"])), W["se.fc.2.weight"], W["se.fc.2.bias"]))
    se = b2 * s[:, :, None, None]
    out = TF.relu(se + xi)
    inter = dict(r1=r1, b1=b1, c2=c2, b2=b2, se=se)
    for v in inter.values():
        v.retain_grad()
    out.backward(g.to(dtype))
    res = {"d_" + k: v.grad.permute(0, 2, 3, 1) for k, v in inter.items()}
    res["d_x"] = xi.grad.permute(0, 2, 3, 1)
    res.update({"dW_" + k: v.grad for k, v in W.items()})
    res["out"] = out.detach().permute(0, 2, 3, 1)
    res["r1"] = r1.detach().permute(0, 2, 3, 1)
    return res


r32, r64 = run_ref(torch.float32), run_ref(torch.float64)
blk = getattr(model.audio_encoder.feat_extractor, f"layer{li}")[bi]
model.to(dev).train()
if len(sys.argv) > 4 and sys.argv[4] == "full":
    # the block INSIDE the full HIP training step: its intermediates' gradients against the float64 block fed with the float64
    # network's own block input / upstream gradient
    from emotiongestures_amd.train import nets
    nets.DEBUG_TAPS = {id(blk): {}}
    pose, _e, _s, pred, _t = model(torch.from_numpy(inp["spec"]).to(dev), torch.from_numpy(inp["text"]).to(dev), torch.from_numpy(inp["pre_pose"]).to(dev), None)
    tg = torch.from_numpy((hash_unit("train.target_pose", 2 * 34 * 126, 0) - 0.5).astype(np.float32).reshape(2, 34, 126)).to(dev)
    lab = torch.from_numpy(inp["label"]).argmax(1).to(dev)
    F.add(F.smooth_l1_loss(pose, tg, 1.0, 100.0), F.cross_entropy(pred, lab)).backward()
    t = nets.DEBUG_TAPS[id(blk)]
    rel = lambda a, b: float((a.double().cpu() - b.double()).norm() / b.double().norm().clamp_min(1e-30))
    print("FULL-network taps of the block vs float64:")
    print("  forward  x", rel(t["x"], x.permute(0, 2, 3, 1)), " out", rel(t["out"], r64["out"]))
    print("  upstream gradient d_out", rel(t["out"].grad, g.permute(0, 2, 3, 1)))
    for k in ("se", "b2", "c2", "b1", "r1", "x"):
        print(f"  d_{k:4s}", rel(t[k].grad, r64["d_" + k]))
    m_h, m_r = (t["r1"].detach().cpu() > 0), (r64["r1"] > 0)
    print("  relu mask: elements", m_h.numel(), "differing", int((m_h != m_r).sum()), " |r1| forward err", rel(t["r1"], r64["r1"]),
          " fraction active", float(m_r.float().mean()))
    dyv = t["r1"].grad.detach().cpu().double()
    print("  ||d_r1 on differing mask positions|| / ||d_r1||", float((dyv * (m_h != m_r)).norm() / dyv.norm()))
    print("  dW_conv1", rel(blk.conv1.weight.grad, r64["dW_conv1.weight"]), " dW_conv2", rel(blk.conv2.weight.grad, r64["dW_conv2.weight"]))
    sys.exit(0)
xi = x.permute(0, 2, 3, 1).contiguous().to(dev).requires_grad_(True)
xa, xb = F.fork(xi)
r1 = F.conv3x3(xa, blk.conv1.weight, None, blk.stride, relu=True)
b1 = F.batch_norm(r1, blk.bn1)
c2 = F.conv3x3(b1, blk.conv2.weight)
b2 = F.batch_norm(c2, blk.bn2)
se = F.se_layer(b2, blk.se.fc[0], blk.se.fc[2])
out = F.relu(F.add(se, xb))
inter = dict(r1=r1, b1=b1, c2=c2, b2=b2, se=se)
for v in inter.values():
    v.retain_grad()
out.backward(g.permute(0, 2, 3, 1).contiguous().to(dev))
got = {"d_" + k: v.grad for k, v in inter.items()}
got["d_x"] = xi.grad
for k in ("conv1.weight", "bn1.weight", "bn1.bias", "conv2.weight", "bn2.weight", "bn2.bias"):
    mod, attr = k.split(".")
    got["dW_" + k] = getattr(getattr(blk, mod), attr).grad
got["dW_se.fc.0.weight"], got["dW_se.fc.2.weight"] = blk.se.fc[0].weight.grad, blk.se.fc[2].weight.grad
got["out"] = out.detach()
rel = lambda a, b: float((a.double().cpu() - b.double()).norm() / b.double().norm().clamp_min(1e-30))
print(f"{'tensor':22s} {'hip vs f64':>11s} {'torch32 vs f64':>15s} {'norm':>10s}")
for k in ("out", "d_se", "d_b2", "dW_bn2.weight", "dW_bn2.bias", "dW_se.fc.2.weight", "dW_se.fc.0.weight", "d_c2", "dW_conv2.weight", "d_b1", "dW_bn1.weight",
          "dW_bn1.bias", "d_r1", "dW_conv1.weight", "d_x"):
    print(f"{k:22s} {rel(got[k], r64[k]):11.2e} {rel(r32[k], r64[k]):15.2e} {float(r64[k].norm()):10.3e}")
