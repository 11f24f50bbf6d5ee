"""CPU oracle for the EmotionGesture audio->gesture hot path.  TEST INFRASTRUCTURE ONLY.

This file is the checker, never the product: only ``tests/``, ``__graft_entry__.smoke()`` and
``bench.py``'s ``cpu_baseline`` leg may import it.  The shipped path (``emotiongestures_amd``)
never routes through it and fails loudly when the HIP library is missing.

It is an independent, vectorised fp32 restatement (torch CPU functional ops + numpy) of the
reference's eval-mode arithmetic.  Every function cites the reference file:line it follows
(paths relative to the upstream repo).  It takes a flat ``state_dict``-style mapping with the
reference's own key names, so the same synthetic weights drive the reference (when the golden
vectors are generated, ``tests/golden/make_golden.py``), this oracle and the HIP path.

Pinning: ``tests/test_oracle_golden.py`` checks this file against ``tests/golden/*.npz``, which
were produced by importing and running the reference's own classes in the build container.
Exception -- ``melspectrogram``: the reference delegates to librosa (utils/train_utils_BEAT.py:186-190),
which is neither vendored, pinned nor installed here, and has no in-repo caller or test, so that
one function is **parity unpinned**; it follows librosa's documented defaults (see its docstring).
"""
from __future__ import annotations

import math
from dataclasses import dataclass
from typing import Dict, Mapping, Optional

import numpy as np
import torch
import torch.nn.functional as F

SD = Mapping[str, torch.Tensor]


@dataclass
class GenCfg:
    """Constructor arguments of Transformer (Full_model/Models_spatial_memory.py:474-477) as
    used by the eval script (test_emotion_gesture_diversity_iterative.py:135)."""
    frames: int = 34
    pose_dim: int = 126
    prior_frames: int = 4
    chunk: int = 4
    d_model: int = 512
    d_inner: int = 2048
    n_layers: int = 3
    n_head: int = 8
    d_k: int = 64
    d_v: int = 64
    tcn_layers: int = 3          # args.n_layers
    tcn_kernel: int = 2
    variant: str = "spatial"     # "spatial" = Models_spatial_memory.py, "memory" = Models_memory.py


# --------------------------------------------------------------------------------------------
# small helpers
# --------------------------------------------------------------------------------------------

_BN_TRAINING = False


class bn_training:
    """Context: BatchNorm layers use BATCH statistics, as nn.BatchNorm does in train() mode (the configuration of the gradient
    goldens: model.train() with every Dropout p forced to 0).  Running buffers are not updated by the oracle."""

    def __enter__(self):
        global _BN_TRAINING
        self.prev, _BN_TRAINING = _BN_TRAINING, True

    def __exit__(self, *a):
        global _BN_TRAINING
        _BN_TRAINING = self.prev


# ---- train-mode nn.Dropout at the reference's sites ----------------------------------------------------------------------
# The oracle never DRAWS a mask: a caller injects them (`with dropout_masks(fn)`), keyed by the NAME the nn.Dropout module has in the reference's
# model (`named_modules()`), e.g. "encoder.layer_stack.0.slf_attn.attention.dropout" (Modules.py:11,21).  With nothing injected every site is
# the identity: eval(), or the p = 0 gradient-parity configuration.  tests/golden/make_golden_dropout_grad.py drives the REFERENCE's own modules
# with the same masks under the same names, which pins the placements below.
_DROPOUT = None


class dropout_masks:
    """`fn(site, x) -> tensor of x's shape holding keep / (1 - p) per element, or None (identity)`."""

    def __init__(self, fn):
        self.fn = fn

    def __enter__(self):
        global _DROPOUT
        self.prev, _DROPOUT = _DROPOUT, self.fn

    def __exit__(self, *a):
        global _DROPOUT
        _DROPOUT = self.prev


def _drop(site: str, x: torch.Tensor) -> torch.Tensor:
    if _DROPOUT is None:
        return x
    m = _DROPOUT(site, x)
    return x if m is None else x * m


# The library's mask stream (emotiongestures_amd/csrc/common.h mix32 / dropout_keep, train/functional.py next_dropout_offset), restated in integer
# numpy so that the masks of a training step can be computed WITHOUT the GPU: element i of a site lives at counter `offset + i` (row-major flat
# index of the tensor the nn.Dropout sees), keep <=> hash(seed, counter) >= p * 2^32; a site of n elements advances the stream by n rounded up
# to 1024.  Bit-exact integer work: tests/test_gpu_training.py compares it with eg_dropout on the GPU element for element.
_M32 = np.uint64(0xFFFFFFFF)


def _mix32(h: np.ndarray) -> np.ndarray:
    h = h ^ (h >> np.uint64(16))
    h = (h * np.uint64(0x85EBCA6B)) & _M32
    h = h ^ (h >> np.uint64(13))
    h = (h * np.uint64(0xC2B2AE35)) & _M32
    return h ^ (h >> np.uint64(16))


def dropout_keep_mask(seed: int, offset: int, numel: int, p: float) -> np.ndarray:
    """bool[numel]: the library's keep decisions for one nn.Dropout(p) site at stream position `offset` (no device epoch: eager steps)."""
    ctr = np.arange(numel, dtype=np.uint64) + np.uint64(offset)
    thr = np.uint64(int(float(np.float32(p)) * 4294967296.0))
    lo = _mix32((ctr & _M32) ^ np.uint64(seed & 0xFFFFFFFF))
    h = _mix32((lo + (ctr >> np.uint64(32)) * np.uint64(0x9E3779B9) + np.uint64(0x6A09E667)) & _M32)
    return h >= thr


def dropout_site_plan(cfg: "GenCfg", batch: int, p_model: float = 0.2, p_attn: float = 0.1):
    """The nn.Dropout sites of Transformer.forward (spatial variant, train() mode) that lie on the path to the losses, in the order the library's
    train-mode forward visits them (emotiongestures_amd/train/nets.py generator_forward), as (reference module name, tensor shape, p).
    p: Models_spatial_memory.py:477 (dropout=0.2 -> Encoder / Decoder / MultiHeadAttention / PositionwiseFeedForward), Modules.py:8
    (attn_dropout=0.1), the literal nn.Dropout(0.2) of the Sequentials (:107,316,490,511,530-534).  Not listed: the text branch's Dropouts
    (tcn.py, :161 -- the text embedding feeds neither loss, :577,616) and SP_Memory_Net_v2's (its result is discarded, :276-295)."""
    b, f, d, h = batch, cfg.frames, cfg.d_model, cfg.n_head
    plan = [("audio_encoder.dropout", (b, f, d), 0.2), ("prior_seq_encoder.post_header.1", (b, f, d), 0.2),
            ("emotion_proj.1", (b, f, d), 0.2), ("semantic_proj.1", (b, f, d), 0.2), ("encoder.dropout", (b, f, d), p_model)]
    for stack, attn in (("encoder", "slf_attn"), ("decoder", "enc_attn")):
        for l in range(cfg.n_layers):
            q = f"{stack}.layer_stack.{l}"
            plan += [(f"{q}.{attn}.attention.dropout", (b, h, f, f), p_attn), (f"{q}.{attn}.dropout", (b, f, d), p_model),
                     (f"{q}.pos_ffn.dropout", (b, f, d), p_model)]
    plan += [("post_projector.1", (b, f, 4 * d), 0.2), ("post_projector.3", (b, f, d), 0.2), ("post_projector.5", (b, f, cfg.pose_dim), 0.2)]
    return plan


def dropout_plan_masks(plan, seed: int):
    """-> ({site: float32 mask tensor keep / (1 - p)}, [(offset, numel)] in plan order): the library's masks for that plan from stream position 0."""
    masks, where, off = {}, [], 0
    for site, shape, p in plan:
        n = int(np.prod(shape))
        keep = dropout_keep_mask(seed, off, n, p)
        inv = np.float32(1.0) / (np.float32(1.0) - np.float32(p))          # the kernels' fp32 1 / (1 - p)
        masks[site] = torch.from_numpy((keep.astype(np.float32) * inv).reshape(shape))
        where.append((off, n))
        off += (n + 1023) // 1024 * 1024
    return masks, where


def _bn(sd: SD, p: str, x: torch.Tensor, eps: float = 1e-5) -> torch.Tensor:
    """BatchNorm{1,2}d: eval mode = running statistics + affine (torch default eps 1e-5); inside `bn_training()` = batch statistics."""
    if _BN_TRAINING:
        return F.batch_norm(x, None, None, sd[p + ".weight"], sd[p + ".bias"], True, 0.1, eps)
    shape = [1, -1] + [1] * (x.dim() - 2)
    inv = torch.rsqrt(sd[p + ".running_var"] + eps)
    return (x - sd[p + ".running_mean"].view(shape)) * (inv * sd[p + ".weight"]).view(shape) \
        + sd[p + ".bias"].view(shape)


def _lin(sd: SD, p: str, x: torch.Tensor) -> torch.Tensor:
    return F.linear(x, sd[p + ".weight"], sd.get(p + ".bias"))


def _ln(sd: SD, p: str, x: torch.Tensor) -> torch.Tensor:
    return F.layer_norm(x, (x.shape[-1],), sd[p + ".weight"], sd[p + ".bias"], 1e-6)


# --------------------------------------------------------------------------------------------
# a2-a4  audio tower
# --------------------------------------------------------------------------------------------

def se_basic_block(sd: SD, p: str, x: torch.Tensor, stride: int) -> torch.Tensor:
    """SEBasicBlock.forward, Full_model/ResNetBlocks.py:21-37 (ReLU precedes bn1);
    SELayer.forward :92-96."""
    out = F.conv2d(x, sd[p + ".conv1.weight"], None, stride=stride, padding=1)
    out = _bn(sd, p + ".bn1", F.relu(out))
    out = _bn(sd, p + ".bn2", F.conv2d(out, sd[p + ".conv2.weight"], None, padding=1))
    y = out.mean(dim=(2, 3))
    y = torch.sigmoid(_lin(sd, p + ".se.fc.2", F.relu(_lin(sd, p + ".se.fc.0", y))))
    out = out * y[:, :, None, None]
    if (p + ".downsample.0.weight") in sd:
        res = _bn(sd, p + ".downsample.1", F.conv2d(x, sd[p + ".downsample.0.weight"], None, stride=stride))
    else:
        res = x
    return F.relu(out + res)


def resnetse(sd: SD, p: str, x: torch.Tensor, layers=(3, 4, 6), taps: Optional[dict] = None) -> torch.Tensor:
    """ResNetSE.forward, Full_model/ResNetSE34V2.py:62-74; _make_layer :40-55."""
    x = _bn(sd, p + ".bn1", F.relu(F.conv2d(x, sd[p + ".conv1.weight"], sd[p + ".conv1.bias"], padding=1)))
    if taps is not None:
        taps["stem"] = x
    for li, n in enumerate(layers):
        for bi in range(n):
            stride = 2 if (li > 0 and bi == 0) else 1
            x = se_basic_block(sd, f"{p}.layer{li + 1}.{bi}", x, stride)
        if taps is not None:
            taps[f"layer{li + 1}"] = x
    return x


def audio_encoder(sd: SD, p: str, spec: torch.Tensor, taps: Optional[dict] = None) -> torch.Tensor:
    """Audio_ResNetEncoder.forward, Full_model/Models_spatial_memory.py:118-133.
    ``spec`` is [B,1,n_mels,T]; channel c of final_conv1 becomes time step c."""
    x = resnetse(sd, p + ".feat_extractor", spec, taps=taps)
    x = _bn(sd, p + ".bn1", F.conv2d(x, sd[p + ".final_conv1.weight"], sd[p + ".final_conv1.bias"], padding=1))
    b, f = x.shape[:2]
    x = x.reshape(b, f, -1)
    if taps is not None:
        taps["audio_map"] = x
    return _lin(sd, p + ".fc2", _drop(p + ".dropout", _lin(sd, p + ".fc1", x)))          # fc1 -> Dropout(0.2) -> fc2, :128-130


# --------------------------------------------------------------------------------------------
# a12  text branch
# --------------------------------------------------------------------------------------------

def weight_norm_weight(v: torch.Tensor, g: torch.Tensor) -> torch.Tensor:
    """torch.nn.utils.weight_norm (dim=0): w = g * v / ||v||, norm over dims (1,2) per out channel
    (Full_model/tcn.py:19-24)."""
    return v * (g / v.flatten(1).norm(dim=1).view(-1, 1, 1))


def text_encoder_tcn(sd: SD, p: str, text: torch.Tensor, cfg: GenCfg) -> torch.Tensor:
    """TextEncoderTCN.forward, Full_model/Models_spatial_memory.py:171-179; TemporalBlock.forward
    Full_model/tcn.py:43-47 (causal: pad d then chomp d from the right, :12,54-58)."""
    x = F.embedding(text, sd[p + ".embedding.weight"]).transpose(1, 2)          # [B,300,L]
    length = x.shape[-1]
    for i in range(cfg.tcn_layers):
        d = 2 ** i
        pad = (cfg.tcn_kernel - 1) * d
        q = f"{p}.tcn.network.{i}"
        out = x
        for c in ("conv1", "conv2"):
            w = weight_norm_weight(sd[f"{q}.{c}.weight_v"], sd[f"{q}.{c}.weight_g"])
            out = F.conv1d(out, w, sd[f"{q}.{c}.bias"], padding=pad, dilation=d)[:, :, :length]
            out = F.relu(out)
        if (q + ".downsample.weight") in sd:
            res = F.conv1d(x, sd[q + ".downsample.weight"], sd[q + ".downsample.bias"])
        else:
            res = x
        x = F.relu(out + res)
    y = _lin(sd, p + ".fc1.0", x).transpose(1, 2)          # Linear over the time axis
    return _lin(sd, p + ".decoder", y).contiguous()


# --------------------------------------------------------------------------------------------
# a5  prior / memory encoder
# --------------------------------------------------------------------------------------------

def pred_conv(sd: SD, p: str, x: torch.Tensor) -> torch.Tensor:
    """Prior_MemoryEncoder.pred_conv, Full_model/Models_spatial_memory.py:366-373: the prior frames
    are the conv *channels*, the pose_dim axis is the conv length."""
    y = _bn(sd, p + ".2", F.relu(F.conv1d(x, sd[p + ".0.weight"], sd[p + ".0.bias"], padding=1)))
    return _bn(sd, p + ".5", F.relu(F.conv1d(y, sd[p + ".3.weight"], sd[p + ".3.bias"], padding=1)))


def sp_memory_v1(sd: SD, p: str, initial: torch.Tensor, pred: torch.Tensor, cfg: GenCfg) -> torch.Tensor:
    """SP_Memory_Net_v1.forward, Full_model/Models_memory.py:233-251, vectorised: the per-(b,c)
    torch.mm of a [1,D] by a [D,1] is an inner product."""
    b = initial.shape[0]
    mem = initial[:, cfg.prior_frames - cfg.chunk:, :].reshape(b, -1)
    mem = _lin(sd, p + ".spatial_chunk_encoder.2", _drop(p + ".spatial_chunk_encoder.1", _lin(sd, p + ".spatial_chunk_encoder.0", mem)))   # [B,D]
    head = pred[:, :cfg.chunk, :]
    s = torch.sigmoid((head * mem[:, None, :]).sum(-1, keepdim=True))
    out = pred.clone()
    out[:, :cfg.chunk, :] = s * head + (1 - s) * mem[:, None, :]
    return out


def tm_memory(sd: SD, p: str, initial: torch.Tensor, pred: torch.Tensor, cfg: GenCfg) -> torch.Tensor:
    """TM_Memory_Net.forward, Full_model/Models_memory.py:282-293.  NB: the two torch.mm calls
    contract over the *batch* axis (:288-289), so clips in one batch are coupled."""
    b = initial.shape[0]
    mem = initial[:, cfg.prior_frames - cfg.chunk:, :].reshape(b, -1)
    mem = _lin(sd, p + ".temporal_chunk_encoder.2", _drop(p + ".temporal_chunk_encoder.1", _lin(sd, p + ".temporal_chunk_encoder.0", mem)))     # [B,D]
    pe = pred[:, :cfg.chunk, :].reshape(b, -1)
    pe = _lin(sd, p + ".temporal_memory_encoder.2", _drop(p + ".temporal_memory_encoder.1", _lin(sd, p + ".temporal_memory_encoder.0", pe)))     # [B,chunk]
    score = mem @ (mem.t() @ pe)
    w = torch.softmax(score, dim=1)
    out = pred.clone()
    head = pred[:, :cfg.chunk, :]
    out[:, :cfg.chunk, :] = head + head * w[:, :, None]
    return out


def prior_memory_encoder(sd: SD, p: str, prior: torch.Tensor, cfg: GenCfg) -> torch.Tensor:
    """Prior_MemoryEncoder.forward: Full_model/Models_spatial_memory.py:378-390 (SP_Memory_Net_v2
    :276-295 returns its input unchanged -- its writes go to a clone) and
    Full_model/Models_memory.py:336-346 (SP_v1 then TM)."""
    pred = pred_conv(sd, p + ".pred_conv", prior)
    if cfg.variant == "memory":
        pred = sp_memory_v1(sd, p + ".spatial_memory", prior, pred, cfg)
        pred = tm_memory(sd, p + ".temporal_memory", prior, pred, cfg)
    out = torch.cat((prior, pred), 1)
    return _lin(sd, p + ".post_header.2", _drop(p + ".post_header.1", _lin(sd, p + ".post_header.0", out)))       # Linear -> Dropout(0.2) -> Linear, :360-364


# --------------------------------------------------------------------------------------------
# a7-a10  transformer
# --------------------------------------------------------------------------------------------

def multi_head_attention(sd: SD, p: str, q: torch.Tensor, k: torch.Tensor, v: torch.Tensor, cfg: GenCfg, mask: Optional[torch.Tensor] = None):
    """MultiHeadAttention.forward, Full_model/SubLayers.py:30-59; ScaledDotProductAttention.forward,
    Full_model/Modules.py:13-23 (q is divided by sqrt(d_k) before the product).  mask (None on the gesture path): unsqueezed for the head
    axis (SubLayers.py:44-45), then `attn.masked_fill(mask == 0, -1e9)` before the softmax (Modules.py:18-19)."""
    b, lq, lk = q.shape[0], q.shape[1], k.shape[1]
    h, dk, dv = cfg.n_head, cfg.d_k, cfg.d_v
    residual = q
    qh = F.linear(q, sd[p + ".w_qs.weight"]).view(b, lq, h, dk).transpose(1, 2)
    kh = F.linear(k, sd[p + ".w_ks.weight"]).view(b, lk, h, dk).transpose(1, 2)
    vh = F.linear(v, sd[p + ".w_vs.weight"]).view(b, lk, h, dv).transpose(1, 2)
    scores = torch.matmul(qh / (dk ** 0.5), kh.transpose(2, 3))
    if mask is not None:
        scores = scores.masked_fill(mask.unsqueeze(1) == 0, -1e9)
    attn = torch.softmax(scores, dim=-1)
    o = torch.matmul(_drop(p + ".attention.dropout", attn), vh).transpose(1, 2).contiguous().view(b, lq, -1)      # Modules.py:21
    o = _drop(p + ".dropout", F.linear(o, sd[p + ".fc.weight"])) + residual                                          # SubLayers.py:54-55
    return _ln(sd, p + ".layer_norm", o), attn


def positionwise_ffn(sd: SD, p: str, x: torch.Tensor) -> torch.Tensor:
    """PositionwiseFeedForward.forward, Full_model/SubLayers.py:74-84."""
    return _ln(sd, p + ".layer_norm", _drop(p + ".dropout", _lin(sd, p + ".w_2", F.relu(_lin(sd, p + ".w_1", x)))) + x)      # :79-80


def encoder(sd: SD, p: str, x: torch.Tensor, cfg: GenCfg, taps: Optional[dict] = None, tag: str = "enc") -> torch.Tensor:
    """Encoder.forward, Full_model/Models_spatial_memory.py:413-436; PositionalEncoding.forward :46-48;
    EncoderLayer.forward Full_model/Layers.py:18-22."""
    x = _drop(p + ".dropout", x + sd[p + ".position_enc.pos_table"][:, :x.shape[1]])        # :422
    for l in range(cfg.n_layers):
        q = f"{p}.layer_stack.{l}"
        x, _ = multi_head_attention(sd, q + ".slf_attn", x, x, x, cfg)
        x = positionwise_ffn(sd, q + ".pos_ffn", x)
        if taps is not None:
            taps[f"{tag}{l}"] = x
    return x


def decoder(sd: SD, p: str, trg: torch.Tensor, enc_out: torch.Tensor, cfg: GenCfg, taps: Optional[dict] = None) -> torch.Tensor:
    """Decoder.forward, Full_model/Models_spatial_memory.py:455-469; DecoderLayer.forward
    Full_model/Layers.py:50-58 (cross-attention + FFN only; no positional encoding)."""
    x = trg
    for l in range(cfg.n_layers):
        q = f"{p}.layer_stack.{l}"
        x, _ = multi_head_attention(sd, q + ".enc_attn", x, enc_out, enc_out, cfg)
        x = positionwise_ffn(sd, q + ".pos_ffn", x)
        if taps is not None:
            taps[f"dec{l}"] = x
    return x


# --------------------------------------------------------------------------------------------
# a13  generator wiring
# --------------------------------------------------------------------------------------------

def generator_forward(sd: SD, cfg: GenCfg, spec: torch.Tensor, text: torch.Tensor, prior: torch.Tensor,
                      sampled: Optional[torch.Tensor] = None, taps: Optional[dict] = None):
    """Transformer.forward, Full_model/Models_spatial_memory.py:566-616 (Models_memory.py:521-565).
    Returns (pose, emotion_feature, semantic_feature, emotion_prediction, text_embedding)."""
    text_embedding = text_encoder_tcn(sd, "text_encoder", text, cfg)
    feat = audio_encoder(sd, "audio_encoder", spec.unsqueeze(1), taps=taps)
    pr = prior_memory_encoder(sd, "prior_seq_encoder", prior, cfg)
    emo = _lin(sd, "emotion_proj.2", _drop("emotion_proj.1", _lin(sd, "emotion_proj.0", feat)))              # :488-491
    sem = _lin(sd, "semantic_proj.2", _drop("semantic_proj.1", _lin(sd, "semantic_proj.0", feat)))           # :509-512
    h = emo.reshape(emo.shape[0], -1)
    for i in (0, 2, 4):
        h = F.relu(_lin(sd, f"emotion_classifer_header.{i}", h))
    emo_pred = _lin(sd, "emotion_classifer_header.6", h)
    fusion = (sampled if sampled is not None else emo) + sem
    fusion = _lin(sd, "fusion_proj.2", F.relu(_lin(sd, "fusion_proj.0", fusion)))
    enc = encoder(sd, "encoder", fusion, cfg, taps=taps)
    dec = decoder(sd, "decoder", pr, enc, cfg, taps=taps)
    pose = dec
    for i in (0, 2, 4, 6):                              # Linear, Dropout(0.2), Linear, Dropout(0.2), Linear, Dropout(0.2), Linear (:528-536)
        pose = _lin(sd, f"post_projector.{i}", pose if i == 0 else _drop(f"post_projector.{i - 1}", pose))
    if taps is not None:
        taps.update(audio_feat=feat, prior_enc=pr, fusion=fusion)
    return pose, emo, sem, emo_pred, text_embedding


def generator_train_loss(sd: SD, cfg: GenCfg, spec, text, prior, target_pose, label):
    """One training objective of BASELINE configs[2] (SURVEY.md §8d cfg 3): 100 * SmoothL1/Huber(pose, target) + CE(emotion
    logits, label), model in train() mode (batch-statistics BatchNorm), every nn.Dropout the identity unless the caller injects masks
    (`with dropout_masks(...)`).  Returns (loss, pose, emotion logits)."""
    with bn_training():
        pose, _emo, _sem, pred, _txt = generator_forward(sd, cfg, spec, text, prior, None)
    loss = 100.0 * F.smooth_l1_loss(pose, target_pose) + F.cross_entropy(pred, label)
    return loss, pose, pred


def cvae_train_loss(sd: SD, x, y, eps, beta: float = 1.0):
    """A VAE objective on MLP_Reconstruct_v3.forward (CAVE/BEAT_CVAE.py:403-424) in train() mode: smooth_l1(reconstruction, x) +
    beta * mean_b(-0.5 * sum_j(1 + logvar - mu^2 - exp(logvar))).  (The reference ships the module, not its training loss.)"""
    with bn_training():
        rec, mu, logvar = cvae_forward(sd, x, y, eps)
    kld = torch.mean(-0.5 * torch.sum(1 + logvar - mu ** 2 - logvar.exp(), dim=1), dim=0)
    return F.smooth_l1_loss(rec, x) + beta * kld, rec, mu, logvar


def emotion_net_train_loss(sd: SD, mfcc, label, alpha, gamma: float = 2.0):
    """train_audio_classifier_K_fold.py:163-168 with FocalLoss :89-105: 100 * mean(alpha * (1 - pt)^gamma * CE), alpha a
    per-sample weight vector (that is how `self.alpha * ...` broadcasts), EmotionNet in train() mode."""
    with bn_training():
        logits = emotion_net(sd, mfcc)
    ce = F.cross_entropy(logits, label, reduction="none")
    pt = torch.exp(-ce)
    return 100.0 * torch.mean(alpha * (1 - pt) ** gamma * ce), logits


# --------------------------------------------------------------------------------------------
# a14  emotion CVAE (v3)
# --------------------------------------------------------------------------------------------

def _lrelu_bn(sd: SD, p: str, x: torch.Tensor) -> torch.Tensor:
    return _bn(sd, p, F.leaky_relu(x, 0.2))


def cvae_decode(sd: SD, zy: torch.Tensor) -> torch.Tensor:
    """fusion_z_posterior + Decoder, CAVE/BEAT_CVAE.py:355-369,377-381,444-446."""
    n = zy.shape[0]
    z = _lin(sd, "fusion_z_posterior.2", _lin(sd, "fusion_z_posterior.0", zy)).reshape(n, 4, 128)
    x = F.conv_transpose1d(z, sd["Decoder.0.weight"], sd["Decoder.0.bias"], stride=2, padding=1, output_padding=1)
    x = _lrelu_bn(sd, "Decoder.2", x)
    x = F.conv_transpose1d(x, sd["Decoder.3.weight"], sd["Decoder.3.bias"], stride=2, padding=1, output_padding=1)
    x = _lrelu_bn(sd, "Decoder.5", x)
    x = _lrelu_bn(sd, "Decoder.8", F.conv1d(x, sd["Decoder.6.weight"], sd["Decoder.6.bias"], padding=1))
    x = _lrelu_bn(sd, "Decoder.11", F.conv1d(x, sd["Decoder.9.weight"], sd["Decoder.9.bias"], padding=1))
    return F.conv1d(x, sd["Decoder.12.weight"], sd["Decoder.12.bias"], padding=1)


def cvae_sample(sd: SD, y: torch.Tensor, z: torch.Tensor) -> torch.Tensor:
    """MLP_Reconstruct_v3.sample, CAVE/BEAT_CVAE.py:427-447, with the latent draw ``z`` made an
    explicit argument (the reference draws torch.randn(n,32) on the CPU generator, :441)."""
    post_y = _lin(sd, "Posterior_Y_embedding.2", _lin(sd, "Posterior_Y_embedding.0", y))
    return cvae_decode(sd, torch.cat([z, post_y], dim=1))


def cvae_encode(sd: SD, x: torch.Tensor):
    """Encoder + fc_mu / fc_var, CAVE/BEAT_CVAE.py:318-332,344-353,408-412."""
    h = _lrelu_bn(sd, "Encoder.2", F.conv1d(x, sd["Encoder.0.weight"], sd["Encoder.0.bias"], padding=1))
    h = _lrelu_bn(sd, "Encoder.5", F.conv1d(h, sd["Encoder.3.weight"], sd["Encoder.3.bias"], padding=1))
    h = _lrelu_bn(sd, "Encoder.8", F.conv1d(h, sd["Encoder.6.weight"], sd["Encoder.6.bias"], stride=2, padding=2))
    h = _lrelu_bn(sd, "Encoder.11", F.conv1d(h, sd["Encoder.9.weight"], sd["Encoder.9.bias"], stride=2, padding=2))
    h = h.reshape(x.shape[0], -1)
    mu = _lin(sd, "fc_mu.2", _lin(sd, "fc_mu.0", h))
    logvar = _lin(sd, "fc_var.2", _lin(sd, "fc_var.0", h))
    return mu, logvar


def cvae_forward(sd: SD, x: torch.Tensor, y: torch.Tensor, eps: torch.Tensor):
    """MLP_Reconstruct_v3.forward, CAVE/BEAT_CVAE.py:403-424; reparameterize :389-399 with the
    noise ``eps`` explicit."""
    mu, logvar = cvae_encode(sd, x)
    z = eps * torch.exp(0.5 * logvar) + mu
    post_y = _lin(sd, "Posterior_Y_embedding.2", _lin(sd, "Posterior_Y_embedding.0", y))
    return cvae_decode(sd, torch.cat([z, post_y], dim=1)), mu, logvar


# --------------------------------------------------------------------------------------------
# a1  mel front-end  (PARITY UNPINNED: librosa is an un-vendored, un-pinned dependency)
# --------------------------------------------------------------------------------------------

def _hz_to_mel_slaney(f):
    f = np.asarray(f, dtype=np.float64)
    f_sp = 200.0 / 3
    mels = f / f_sp
    min_log_hz = 1000.0
    min_log_mel = min_log_hz / f_sp
    logstep = math.log(6.4) / 27.0
    with np.errstate(divide="ignore", invalid="ignore"):
        log_t = min_log_mel + np.log(np.maximum(f, 1e-30) / min_log_hz) / logstep
    return np.where(f >= min_log_hz, log_t, mels)


def _mel_to_hz_slaney(m):
    m = np.asarray(m, dtype=np.float64)
    f_sp = 200.0 / 3
    min_log_hz = 1000.0
    min_log_mel = min_log_hz / f_sp
    logstep = math.log(6.4) / 27.0
    return np.where(m >= min_log_mel, min_log_hz * np.exp(logstep * (m - min_log_mel)), f_sp * m)


def mel_filterbank(sr: int = 16000, n_fft: int = 1024, n_mels: int = 128) -> np.ndarray:
    """librosa.filters.mel defaults: fmin 0, fmax sr/2, htk=False (Slaney scale), norm='slaney'.
    Returns float32 [n_mels, 1 + n_fft//2] like librosa (dtype=np.float32)."""
    n_bins = 1 + n_fft // 2
    fftfreqs = np.linspace(0.0, sr / 2.0, n_bins)
    mel_pts = np.linspace(_hz_to_mel_slaney(0.0), _hz_to_mel_slaney(sr / 2.0), n_mels + 2)
    hz = _mel_to_hz_slaney(mel_pts)
    fdiff = np.diff(hz)
    ramps = hz[:, None] - fftfreqs[None, :]
    w = np.zeros((n_mels, n_bins), dtype=np.float64)
    for i in range(n_mels):
        lower = -ramps[i] / fdiff[i]
        upper = ramps[i + 2] / fdiff[i + 1]
        w[i] = np.maximum(0.0, np.minimum(lower, upper))
    enorm = 2.0 / (hz[2:n_mels + 2] - hz[:n_mels])
    w *= enorm[:, None]
    return w.astype(np.float32)


def hann_periodic(n: int = 1024) -> np.ndarray:
    """scipy.signal.get_window('hann', n, fftbins=True), as librosa.stft uses."""
    return (0.5 - 0.5 * np.cos(2.0 * np.pi * np.arange(n) / n)).astype(np.float32)


def melspectrogram(audio: np.ndarray, sr: int = 16000, n_fft: int = 1024, hop: int = 512,
                   n_mels: int = 128, top_db: float = 80.0, out_frames: Optional[int] = None) -> np.ndarray:
    """extract_melspectrogram, utils/train_utils_BEAT.py:186-190 = librosa.feature.melspectrogram(
    n_fft=1024, hop_length=512, power=2) -> librosa.power_to_db(ref=np.max) -> float16, followed by
    the loader's slice [:, :expected_len] (data_loader/lmdb_loader_BEAT_full.py:229).

    librosa defaults restated: centred STFT, zero ("constant") padding of n_fft//2 each side
    (librosa >= 0.10), periodic Hann window, 1 + len//hop frames; Slaney mel basis;
    power_to_db: 10*log10(max(amin=1e-10, S)) - 10*log10(max(amin, max S)), floored at max - top_db.
    ``audio`` is [B, n]; returns float32 holding fp16-rounded dB, [B, n_mels, frames]."""
    audio = np.asarray(audio, dtype=np.float32)
    b, n = audio.shape
    pad = n_fft // 2
    x = np.pad(audio.astype(np.float64), ((0, 0), (pad, pad)))
    n_frames = 1 + n // hop
    win = hann_periodic(n_fft).astype(np.float64)
    idx = np.arange(n_fft)[None, :] + hop * np.arange(n_frames)[:, None]
    frames = x[:, idx] * win                                     # [B, frames, n_fft]
    power = np.abs(np.fft.rfft(frames, axis=-1)) ** 2            # [B, frames, 513]
    mel = np.einsum("mk,bfk->bmf", mel_filterbank(sr, n_fft, n_mels).astype(np.float64), power)
    db = 10.0 * np.log10(np.maximum(1e-10, mel))
    ref = np.maximum(1e-10, mel.reshape(b, -1).max(axis=1))
    db -= 10.0 * np.log10(ref)[:, None, None]
    db = np.maximum(db, db.reshape(b, -1).max(axis=1)[:, None, None] - top_db)
    if out_frames is not None:
        db = db[:, :, :out_frames]
    return db.astype(np.float16).astype(np.float32)


def spectrogram_length(n_frames: int, fps: float) -> int:
    """calc_spectrogram_length_from_motion_length, utils/train_utils_BEAT.py:193-195."""
    return int(round((n_frames / fps * 16000 - 1024) / 512 + 1))


# --------------------------------------------------------------------------------------------
# a16 / f1 / f2  eval harness pieces
# --------------------------------------------------------------------------------------------

def fgd_autoencoder(sd: SD, x: torch.Tensor):
    """MLP_Reconstruct.forward, model/FGD.py:63-70 (eval: Dropout identity) -> (reconstruction, 512-d latent)."""
    lat = x
    for i in (0, 2, 4):
        lat = _lin(sd, f"Encoder.{i}", lat)
    out = lat
    for i in (0, 2, 4):
        out = _lin(sd, f"Decoder.{i}", out)
    return out, lat


def skeleton_classifier(sd: SD, pose: torch.Tensor, cfg: Optional[GenCfg] = None):
    """skeleton_classifer.Models.Transformer.forward, skeleton_classifer/Models.py:256-283 -> (logits, mid_feature)."""
    cfg = cfg or GenCfg()
    b = pose.shape[0]
    x = _lin(sd, "prior_seq_encoder.fc2", _lin(sd, "prior_seq_encoder.fc1", pose))
    enc = encoder(sd, "encoder", x, cfg)
    h = enc.reshape(b, -1)
    for i in (0, 2, 4, 6):
        h = F.relu(_lin(sd, f"post_projector.{i}", h))
    return _lin(sd, "post_projector.8", h), enc


def emotion_net(sd: SD, mfcc: torch.Tensor) -> torch.Tensor:
    """EmotionNet.forward, model/audio_emotion_classifer.py:39-49: 4-stage ResNetSE (model/emotion_ResNetSE34V2.py:57-71,
    blocks [3,4,6,3]) on [B,128,128] -> [B,256,16,16] -> flatten (NCHW order) -> 5 x (Linear, ReLU) -> Linear(64, 8)."""
    x = resnetse(sd, "emotion_encoder", mfcc.unsqueeze(1), layers=(3, 4, 6, 3))
    h = x.reshape(x.shape[0], -1)
    for i in (0, 2, 4, 6, 8):
        h = F.relu(_lin(sd, f"emotion_eocder_fc.{i}", h))
    return _lin(sd, "last_fc", h)


def motion_ae(sd: SD, pose: torch.Tensor):
    """MotionAE.forward, model/motion_ae.py:125-130 -> (reconstruction [B,34,D], latent [B,latent]).  Encoder :54-61:
    3 x (Conv1d -> BN -> LeakyReLU(0.2)) [k3, k3, k4 s2] -> Conv1d k3 -> flatten -> Linear/BN x2 -> Linear; decoder :107-116:
    Linear/BN -> Linear -> [B,4,34] -> 2 x (ConvTranspose1d k3 -> BN -> LeakyReLU(0.2)) -> Conv1d k3 x2.
    nn.LeakyReLU(True) in the dense stacks has slope 1.0 (identity)."""
    def bn(p, x):
        return F.batch_norm(x, sd[p + ".running_mean"], sd[p + ".running_var"], sd[p + ".weight"], sd[p + ".bias"], False, 0.0, 1e-5)
    x = pose.reshape(pose.shape[0], pose.shape[1], -1).transpose(1, 2)
    for i, stride in ((0, 1), (1, 1), (2, 2)):
        q = f"encoder.net.{i}"
        x = F.leaky_relu(bn(q + ".1", F.conv1d(x, sd[q + ".0.weight"], sd[q + ".0.bias"], stride=stride)), 0.2)
    x = F.conv1d(x, sd["encoder.net.3.weight"], sd["encoder.net.3.bias"]).flatten(1)
    x = bn("encoder.out_net.1", _lin(sd, "encoder.out_net.0", x))
    x = bn("encoder.out_net.4", _lin(sd, "encoder.out_net.3", x))
    z = _lin(sd, "encoder.out_net.6", x)
    h = bn("decoder.pre_net.1", _lin(sd, "decoder.pre_net.0", z))
    h = _lin(sd, "decoder.pre_net.3", h).view(z.shape[0], 4, -1)
    for i in (0, 3):
        q = f"decoder.net.{i}"
        h = F.leaky_relu(bn(f"decoder.net.{i + 1}", F.conv_transpose1d(h, sd[q + ".weight"], sd[q + ".bias"])), 0.2)
    h = F.conv1d(h, sd["decoder.net.6.weight"], sd["decoder.net.6.bias"])
    h = F.conv1d(h, sd["decoder.net.7.weight"], sd["decoder.net.7.bias"])
    return h.transpose(1, 2), z


# a15  training-side types, forward only
def calc_motion(motion: torch.Tensor) -> torch.Tensor:
    """calc_motion, test_emotion_gesture_diversity_iterative.py:41-44 (frame-to-frame offsets)."""
    return motion[:, 1:] - motion[:, :-1]


def motion_discriminator(sd: SD, x: torch.Tensor, cfg: GenCfg) -> torch.Tensor:
    """Motion_Discriminator.forward, Full_model/Models_spatial_memory.py:658-669 (same class in Models_memory.py):
    encoder (self-attention + FFN layers) -> Linear+ReLU per frame -> flatten -> 6-layer ReLU MLP -> [B, 1] (no sigmoid)."""
    b = x.shape[0]
    h = encoder(sd, "encoder", x, cfg)
    h = F.relu(_lin(sd, "fc1.0", h)).reshape(b, -1)
    for i in (0, 2, 4, 6, 8):
        h = F.relu(_lin(sd, f"fc2.{i}", h))
    return _lin(sd, "fc2.10", h)


def pose_discriminator(sd: SD, x: torch.Tensor, cfg: GenCfg) -> torch.Tensor:
    """Pose_Discriminator.forward, Full_model/Models_spatial_memory.py:698-702: encoder -> Linear(282, 64) -> (Dropout) -> Linear(64, 1)
    -> sigmoid; one probability per frame, [B, T, 1]."""
    h = encoder(sd, "encoder", x, cfg)
    return torch.sigmoid(_lin(sd, "fc.2", _lin(sd, "fc.0", h)))


def softmax_contrastive(face: np.ndarray, audio: np.ndarray):
    """SoftmaxContrastiveLoss.forward / .evaluate, test_emotion_gesture_diversity_iterative.py:80-127, in float64 numpy:
    rows L2-normalised (F.normalize eps 1e-12), cross = clamp(1 / (pairwise L2 + 1e-8), min 1e-8),
    loss = cross_entropy(cross, arange(n)), acc = mean(argmax_j cross[i, j] == i).  Returns (loss, acc, cross)."""
    f = np.asarray(face, np.float64)
    a = np.asarray(audio, np.float64)
    f = f / np.maximum(np.linalg.norm(f, axis=1, keepdims=True), 1e-12)
    a = a / np.maximum(np.linalg.norm(a, axis=1, keepdims=True), 1e-12)
    dist = np.linalg.norm(f[:, None, :] - a[None, :, :], axis=2)
    cross = np.maximum(1.0 / (dist + 1e-8), 1e-8)
    m = cross.max(axis=1, keepdims=True)
    lse = m[:, 0] + np.log(np.exp(cross - m).sum(axis=1))
    n = cross.shape[0]
    loss = float(np.mean(lse - cross[np.arange(n), np.arange(n)]))
    acc = float(np.mean(cross.argmax(axis=1) == np.arange(n)))
    return loss, acc, cross


def adjust_lr_value(init_lr: float, epoch: int) -> float:
    """adjust_lr, test_emotion_gesture_diversity_iterative.py:64-78: piecewise-constant schedule (epochs past 150 are
    undefined upstream: `base_lr` is unbound there)."""
    if epoch <= 15:
        return init_lr
    if epoch <= 50:
        return init_lr * 0.2
    if epoch <= 80:
        return init_lr * 0.01
    if epoch <= 100:
        return init_lr * 0.005
    if epoch <= 150:
        return init_lr * 0.001
    raise ValueError("adjust_lr is undefined past epoch 150 upstream")


def frechet_distance(mu1, sigma1, mu2, sigma2):
    """calculate_frechet_distance, model/FHD_score.py:159-217 (float64, scipy.linalg.sqrtm)."""
    from scipy import linalg
    diff = np.asarray(mu1) - np.asarray(mu2)
    covmean, _ = linalg.sqrtm(np.asarray(sigma1).dot(np.asarray(sigma2)), disp=False)
    if np.iscomplexobj(covmean):
        if not np.allclose(np.diagonal(covmean).imag, 0, atol=1e-3):
            return 100
        covmean = covmean.real
    return diff.dot(diff) + np.trace(sigma1) + np.trace(sigma2) - 2 * np.trace(covmean)


def diversity_score(activations: np.ndarray, frames: int = 60):
    """diversity_score / calculate_diversity, model/FHD_score.py:247-286 (consumes np.random like upstream: two randint(5)
    draws per repeat, 10 repeats)."""
    from scipy import stats
    act = activations.reshape(-1, frames, 512)
    window = np.empty((10, 1))
    for i in range(10):
        first = np.random.randint(0, len(act), 5)
        second = np.random.randint(0, len(act), 5)
        d = 0.0
        for a_, b_ in zip(first, second):
            d += float(torch.dist(torch.from_numpy(act[a_]), torch.from_numpy(act[b_])))
        window[i] = np.float32(d / 5)
    mean, std = np.mean(window, axis=0), np.std(window, axis=0)
    interval = stats.norm.interval(0.95, mean, std)
    return (interval[0] + interval[1]) / 2, interval


# --------------------------------------------------------------------------------------------
# convenience
# --------------------------------------------------------------------------------------------

def to_torch_sd(sd_np: Mapping[str, np.ndarray]) -> Dict[str, torch.Tensor]:
    return {k: torch.from_numpy(np.ascontiguousarray(v)) for k, v in sd_np.items()}
