#!/usr/bin/env python3
"""Generate tests/golden/grads.npz: loss values and per-parameter gradient fingerprints from the REFERENCE's own classes
under torch autograd on CPU (build container only: needs /root/reference, which never ships).

    python tests/golden/make_golden_grad.py

Cases (SURVEY.md §8c "a train()-mode BN/LN forward+backward golden with dropout p forced to 0"):
  gen   Full_model.Models_spatial_memory.Transformer, TED shapes, B = 2, .train(), every nn.Dropout p = 0;
        loss = 100 * smooth_l1(pose, target) + cross_entropy(emotion_prediction, label)       (BASELINE configs[2], SURVEY §8d cfg 3)
  emo   model.audio_emotion_classifer.EmotionNet, B = 2, .train();
        loss = 100 * FocalLoss(alpha, gamma=2)(logits, label) with the script's own FocalLoss    (train_audio_classifier_K_fold.py:89-105,168)
Weights / inputs come from emotiongestures_amd.synth (integer hash); stand-ins only for imports unused on these paths
(torch_dct, torchvision*, umap, fasttext).  Per parameter the file holds: L2 norm (float64), sum (float64) and a strided sample of
up to 64 gradient values; parameters that receive no gradient are listed in `<case>/nograd`.
"""
import os
import sys
import types
from types import SimpleNamespace

import numpy as np
import torch
import torch.nn as nn
import torch.nn.functional as F

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
REF = "/root/reference"

from emotiongestures_amd.synth import hash_unit, load_synth_weights, synth_inputs  # noqa: E402

NS = 64


def stub():
    for name in ("torch_dct", "torchvision", "torchvision.models", "torchvision.utils", "torchvision.transforms", "umap", "fasttext"):
        if name not in sys.modules:
            sys.modules[name] = types.ModuleType(name)
    tv = sys.modules["torchvision"]
    tv.utils, tv.transforms, tv.models = sys.modules["torchvision.utils"], sys.modules["torchvision.transforms"], sys.modules["torchvision.models"]
    sys.modules["torchvision.utils"].save_image = None
    if REF not in sys.path:
        sys.path.insert(0, REF)


def fingerprint(out, case, model):
    nograd = []
    for k, p in model.named_parameters():
        if p.grad is None:
            nograd.append(k)
            continue
        g = p.grad.detach().reshape(-1).double().numpy()
        stride = max(1, g.size // NS)
        out[f"{case}/g/{k}/norm"] = np.float64(np.linalg.norm(g))
        out[f"{case}/g/{k}/sum"] = np.float64(g.sum())
        out[f"{case}/g/{k}/sample"] = g[::stride][:NS].astype(np.float32)
    out[f"{case}/nograd"] = np.array(nograd)


def train_targets(batch, frames, pose_dim, seed):
    """Target poses in +-0.5 and integer emotion labels (the one-hot of synth_inputs)."""
    t = (hash_unit("train.target_pose", batch * frames * pose_dim, seed) - 0.5).astype(np.float32).reshape(batch, frames, pose_dim)
    return t


def gen_case(out, seed=0, batch=2):
    stub()
    from Full_model.Models_spatial_memory import Transformer
    args = SimpleNamespace(chunk=4, hidden_size=300, n_layers=3, freeze_wordembed=False, wordembed_dim=300, dropout_prob=0.1)
    lang = SimpleNamespace(n_words=200, word_embedding_weights=None)
    m = Transformer(args, lang, frames=34, pose_dim=126, prior_frames=4, d_word_vec=512, d_model=512, d_inner=2048, n_layers=3,
                    n_head=8, d_k=64, d_v=64)
    load_synth_weights(m, seed)
    m.train()
    for mod in m.modules():
        if isinstance(mod, nn.Dropout):
            mod.p = 0.0
    inp = synth_inputs(batch, 34, 126, 4, seed=seed)
    target = torch.from_numpy(train_targets(batch, 34, 126, seed))
    label = torch.from_numpy(inp["label"]).argmax(1)
    pose, emo, sem, pred, txt = m(torch.from_numpy(inp["spec"]), torch.from_numpy(inp["text"]), torch.from_numpy(inp["pre_pose"]), None)
    loss = 100.0 * F.smooth_l1_loss(pose, target) + F.cross_entropy(pred, label)
    loss.backward()
    out["gen/loss"] = np.float64(loss.item())
    out["gen/pose"] = pose.detach().numpy()
    out["gen/emotion_prediction"] = pred.detach().numpy()
    out["gen/meta"] = np.asarray([batch, seed], np.int64)
    bn = m.audio_encoder.feat_extractor.layer2[0].bn1
    out["gen/bn_running_mean"] = bn.running_mean.numpy().copy()         # one train-mode forward from the synthetic buffers
    out["gen/bn_running_var"] = bn.running_var.numpy().copy()
    fingerprint(out, "gen", m)
    print("gen loss", loss.item(), "params with grad", sum(p.grad is not None for p in m.parameters()), "without", len(out["gen/nograd"]))


def genmem_case(out, seed=3, batch=4):
    """Full_model.Models_memory.Transformer (SP_Memory_Net_v1 gate + the batch-coupled TM_Memory_Net, Models_memory.py:233-251,282-293) in
    train() mode at a FIXED batch of 4 (TM sums over the batch), dropout p = 0, same loss as `gen`."""
    stub()
    from Full_model.Models_memory import Transformer
    args = SimpleNamespace(chunk=4, hidden_size=300, n_layers=3, freeze_wordembed=False, wordembed_dim=300, dropout_prob=0.1)
    lang = SimpleNamespace(n_words=200, word_embedding_weights=None)
    m = Transformer(args, lang, frames=34, pose_dim=126, prior_frames=4, d_word_vec=512, d_model=512, d_inner=2048, n_layers=3,
                    n_head=8, d_k=64, d_v=64)
    load_synth_weights(m, seed)
    m.train()
    for mod in m.modules():
        if isinstance(mod, nn.Dropout):
            mod.p = 0.0
    inp = synth_inputs(batch, 34, 126, 4, seed=seed)
    target = torch.from_numpy(train_targets(batch, 34, 126, seed))
    label = torch.from_numpy(inp["label"]).argmax(1)
    pose, emo, sem, pred, txt = m(torch.from_numpy(inp["spec"]), torch.from_numpy(inp["text"]), torch.from_numpy(inp["pre_pose"]), None)
    loss = 100.0 * F.smooth_l1_loss(pose, target) + F.cross_entropy(pred, label)
    loss.backward()
    out["genmem/loss"] = np.float64(loss.item())
    out["genmem/pose"] = pose.detach().numpy()
    out["genmem/emotion_prediction"] = pred.detach().numpy()
    out["genmem/meta"] = np.asarray([batch, seed], np.int64)
    # only the prior / memory encoder differs from the `gen` case: keep its gradients plus a few downstream / upstream witnesses
    keep = ("prior_seq_encoder.", "post_projector.6.", "decoder.layer_stack.0.enc_attn.w_qs", "audio_encoder.fc2.", "emotion_classifer_header.6.")
    sub = {}
    fingerprint(sub, "genmem", m)
    for k, v in sub.items():
        if k == "genmem/nograd" or any(k.startswith("genmem/g/" + pre) for pre in keep):
            out[k] = v
    print("genmem loss", loss.item(), "kept", sum(1 for k in out if k.startswith("genmem/g/") and k.endswith("/norm")), "gradient fingerprints; without grad", len(out["genmem/nograd"]))


def emo_case(out, seed=31, batch=2):
    stub()
    os.chdir(REF)
    from model.audio_emotion_classifer import EmotionNet
    sys.path.insert(0, os.path.join(ROOT, "tests", "golden"))
    from make_golden_emotion_net import emotion_input
    net = EmotionNet()
    load_synth_weights(net, seed)
    net.train()
    x = torch.from_numpy(emotion_input(batch, seed))
    label = torch.tensor([2, 5][:batch])
    alpha = [0.5, 2.0][:batch]          # multiplies the per-sample loss vector (the script's list alpha broadcasts over the batch axis)

    # the script's own FocalLoss (train_audio_classifier_K_fold.py:89-105); the script cannot be imported (matplotlib / sklearn /
    # lmdb imports and module-level code), so the 12-line class is instantiated from its source text
    src = open(os.path.join(REF, "train_audio_classifier_K_fold.py")).read()
    a = src.index("class FocalLoss(nn.Module):")
    b = src.index("def train_K_fold(")
    ns = {"nn": nn, "torch": torch}
    exec(src[a:b], ns)
    crit = ns["FocalLoss"](alpha=torch.tensor(alpha), gamma=2, reduction="mean")
    logits = net(x)
    loss = crit(logits, label) * 100
    loss.backward()
    out["emo/loss"] = np.float64(loss.item())
    out["emo/logits"] = logits.detach().numpy()
    out["emo/label"] = label.numpy()
    out["emo/alpha"] = np.asarray(alpha, np.float32)
    fingerprint(out, "emo", net)
    print("emo loss", loss.item())


def cvae_case(out, seed=7, n=3):
    """MLP_Reconstruct_v3 (CAVE/BEAT_CVAE.py:312-424) in train() mode, dropout p = 0, reparameterize's randn_like replaced by a
    fixed eps; loss = smooth_l1(recon, x) + KLD(mu, logvar) (the reference has the module but no training loss: a standard VAE
    objective exercises every parameter of Encoder / fc_mu / fc_var / Decoder)."""
    stub()
    from CAVE.BEAT_CVAE import MLP_Reconstruct_v3
    m = MLP_Reconstruct_v3()
    load_synth_weights(m, seed)
    m.train()
    for mod in m.modules():
        if isinstance(mod, nn.Dropout):
            mod.p = 0.0
    inp = synth_inputs(n, frames=60, seed=seed)
    x, y = torch.from_numpy(inp["sampled"]), torch.from_numpy(inp["label"])
    eps = torch.from_numpy(synth_inputs(n, seed=seed + 1)["z"])
    real = torch.randn_like
    torch.randn_like = lambda t, **k: eps.clone()
    try:
        rec, mu, logvar = m(x, y)
    finally:
        torch.randn_like = real
    kld = torch.mean(-0.5 * torch.sum(1 + logvar - mu ** 2 - logvar.exp(), dim=1), dim=0)
    loss = F.smooth_l1_loss(rec, x) + kld
    loss.backward()
    out["cvae/loss"] = np.float64(loss.item())
    out["cvae/mu"] = mu.detach().numpy()
    out["cvae/logvar"] = logvar.detach().numpy()
    out["cvae/meta"] = np.asarray([n, seed], np.int64)
    fingerprint(out, "cvae", m)
    print("cvae loss", loss.item(), "params without grad", len(out["cvae/nograd"]))


def main():
    torch.manual_seed(0)
    torch.set_num_threads(8)
    out = {}
    gen_case(out)
    genmem_case(out)
    emo_case(out)
    cvae_case(out)
    path = os.path.join(ROOT, "tests", "golden", "grads.npz")
    np.savez_compressed(path, **out)
    print("wrote", path, os.path.getsize(path) // 1024, "KiB")


if __name__ == "__main__":
    main()
