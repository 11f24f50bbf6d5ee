"""Train-mode forwards of the reference's networks, composed from the HIP operators of `functional.py`.

* `emotion_net_forward`  -- EmotionNet (model/audio_emotion_classifer.py:17-49), the one network the reference ships a training
  loop for (train_audio_classifier_K_fold.py:109-200).
* `generator_forward`    -- Transformer.forward of Full_model/Models_spatial_memory.py:566-616 in train() mode: BatchNorm on
  batch statistics (running buffers updated), every other layer as in eval.  Dropout layers act with p = 0 by default (SURVEY.md §8c:
  the gradient-parity configuration); `model.train_dropout = True` activates the reference's Dropout placements on a counter-based
  mask stream (see `_dp` below).

The modules are the parameter trees of `emotiongestures_amd.modules` (same names / shapes as the reference's state_dict);
activations are NHWC in the audio tower and row-major [rows, features] elsewhere.
"""
from __future__ import annotations

import torch

from .. import ops
from . import functional as F


# Dropout: the gradient-parity configuration (tests, goldens) runs with every Dropout at p = 0.  With
# `model.train_dropout = True` the reference's placements are active with its probabilities, on the library's own mask stream
# (functional.manual_seed) -- including ScaledDotProductAttention's dropout on the probabilities (Modules.py:21, p = 0.1), applied inside
# the attention kernels from the same counter-based mask (forward and backward recompute it; nothing is stored).
_P = {"on": False}


def _dp(x, p):
    return F.dropout(x, p) if _P["on"] else x


def _dropout_on(model) -> None:
    """Entry of a network's train-mode forward: activate the Dropout placements when the model asks for them.  The mask epoch is NOT advanced
    here but once per optimiser step by the step driver (functional.begin_dropout_step): a forward's backward kernels re-read the epoch from
    device memory when they run, so a second forward before that backward must not move it."""
    _P["on"] = bool(getattr(model, "train_dropout", False))


# Cut sites of a segmented training step (train/graph.SegmentedStep): at a named site the activation is detached and a fresh leaf carries the
# forward on, so the backward runs in phases -- loss.backward() stops at the leaves, then each cut continues with x.backward(leaf.grad) --
# and the gradient buckets completed by one phase are all-reduced while the next phase computes.  No arithmetic: a detach and a .grad hand-over.
# A site is only valid where every parameter's gradient and the leaf's gradient are completed within ONE phase: the tower output is (everything
# upstream of it is the tower alone), and so are the inputs of the tower's stages ("layer2", "layer3": a chain, the stage's first block is the only
# consumer); the encoder output is not (the emotion head reaches emotion_proj / fc1 / fc2 past it in another phase).
_CUTS = {"ctx": None}


class CutContext:
    def __init__(self, names=("tower",)):
        self.names, self.cuts = tuple(names), []

    def reset(self):
        self.cuts = []

    def cut(self, name, x):
        if name not in self.names or not x.requires_grad:
            return x
        leaf = x.detach().requires_grad_(True)
        self.cuts.append((name, x, leaf))
        return leaf


def _cut(name, x):
    ctx = _CUTS["ctx"]
    return x if ctx is None else ctx.cut(name, x)


def _fork_n(x, n):
    """n uses of x, gradients summed pairwise on the HIP add kernel."""
    outs = []
    cur = x
    for _ in range(n - 1):
        a, cur = F.fork(cur)
        outs.append(a)
    outs.append(cur)
    return outs


# ---- audio tower (Full_model/ResNetSE34V2.py:62-74, ResNetBlocks.py:21-37) -------------------------------------------------
DEBUG_TAPS = None       # tools/debug_block_grad.py: {id(block): dict} receives the block's intermediates with retain_grad()


FUSE_BLOCK = True           # False: the operator-by-operator block (A/B switch of tests and tools)
TAP_FUSED = None            # tests: a dict that receives conv1's ReLU output and the block output of the FUSED block (ReLU masks)


def se_basic_block(blk, x):
    if not FUSE_BLOCK or DEBUG_TAPS is not None:
        return _se_basic_block_unfused(blk, x)
    # fused data flow: the convolutions emit pooling partials (BatchNorm means and the SE pooling without extra passes), bn1's backward
    # carries conv1's ReLU mask, and everything behind conv2 -- bn2, SE gate, scale, residual add, ReLU -- is one operator
    # conv1 -> ReLU -> bn1 (:24-26); the block input's second consumer (the residual branch) goes through the convolution's passthrough
    # output: the two input gradients are summed in conv1's input-gradient launch
    sub = blk.downsample is not None and blk.stride != 1       # the shortcut is the strided 1x1 convolution: it reads x[:, ::s, ::s] only
    # identity shortcut: its gradient dout * [out > 0] is never written -- the tail leaves (dout, its ReLU bits) in `link` and conv1's input-gradient
    # epilogue masks dout itself
    link = {} if (blk.downsample is None and F.LAZY_SHORTCUT_GRAD and x.requires_grad and x.shape[-1] % 32 == 0 and
                  blk.conv1.weight.shape[0] == x.shape[-1] and blk.stride == 1) else None
    r1, gap1, xb = F.conv3x3(x, blk.conv1.weight, None, blk.stride, relu=True, want_gap=True, defer_mask=True, passthrough="sub" if sub else True,
                             res_link=link)
    # bn1 is not applied here: conv2 (and its weight-gradient kernel) apply it as a per-channel affine while staging r1 (split-bf16 modes)
    b1 = F.batch_norm(r1, blk.bn1, gap=gap1, relu_input=True, defer_apply=True)
    c2, gap2 = F.conv3x3(b1, blk.conv2.weight, want_gap=True)
    if blk.downsample is not None:
        res = F.batch_norm(F.conv1x1(xb, blk.downsample[0].weight, 1 if sub else blk.stride), blk.downsample[1])      # sub: xb is already the strided map
    else:
        res = xb
    out = F.se_block_tail(c2, gap2, res, blk.bn2, blk.se.fc[0], blk.se.fc[2], res_link=link)
    if TAP_FUSED is not None:
        if r1.requires_grad:
            r1.retain_grad()            # the (masked) gradient at conv1's ReLU output: lets a test weigh a mask element it decides differently
        TAP_FUSED.update(r1=r1, out=out.detach())
    return out


def _se_basic_block_unfused(blk, x):
    xa, xb = F.fork(x)
    r1 = F.conv3x3(xa, blk.conv1.weight, None, blk.stride, relu=True)           # conv1 -> ReLU -> bn1 (ReLU precedes BN, :24-26)
    b1 = F.batch_norm(r1, blk.bn1)
    c2 = F.conv3x3(b1, blk.conv2.weight)
    b2 = F.batch_norm(c2, blk.bn2)
    se = F.se_layer(b2, blk.se.fc[0], blk.se.fc[2])
    if blk.downsample is not None:
        res = F.batch_norm(F.conv1x1(xb, blk.downsample[0].weight, blk.stride), blk.downsample[1])
    else:
        res = xb
    out = F.relu(F.add(se, res))
    if DEBUG_TAPS is not None and id(blk) in DEBUG_TAPS:
        t = dict(x=x, r1=r1, b1=b1, c2=c2, b2=b2, se=se, out=out)
        for v in t.values():
            if v.requires_grad:
                v.retain_grad()
        DEBUG_TAPS[id(blk)].update(t)
    return out


def resnetse_forward(enc, spec):
    """spec [B,H,W] -> NHWC feature map."""
    x = spec.unsqueeze(-1).contiguous()
    # conv -> ReLU -> BatchNorm (ResNetSE34V2.py:64-66): the ReLU's backward mask rides on bn1's backward reduction / apply, as in the blocks
    x = F.batch_norm(F.conv3x3(x, enc.conv1.weight, enc.conv1.bias, 1, relu=True, defer_mask=True), enc.bn1, relu_input=True)
    for name, layer in (("layer1", enc.layer1), ("layer2", enc.layer2), ("layer3", enc.layer3), ("layer4", getattr(enc, "layer4", ()))):
        if name != "layer1" and len(layer):
            x = _cut(name, x)           # stage inputs are valid cut sites: one consumer chain, every upstream gradient belongs to the later phase
        for blk in layer:
            x = se_basic_block(blk, x)
    return x


def emotion_net_forward(model, mfcc):
    """EmotionNet.forward in train() mode -> logits [B, 8]."""
    feat = resnetse_forward(model.emotion_encoder, mfcc)                        # [B,16,16,256]
    h = feat.permute(0, 3, 1, 2).reshape(feat.shape[0], -1)                     # feature.view(B,-1) of the NCHW map (:44)
    if F.FUSE_BLOCKS:           # five Linear + ReLU and last_fc as one chain: the ReLU backwards are gates of the input-gradient products
        F.flush_batch_counters()
        return F.linear_chain(h, [model.emotion_eocder_fc[i] for i in (0, 2, 4, 6, 8)] + [model.last_fc], relu_between=True)
    for i in (0, 2, 4, 6, 8):
        lin = model.emotion_eocder_fc[i]
        h = F.linear(h, lin.weight, lin.bias, relu=True)
    F.flush_batch_counters()
    return F.linear(h, model.last_fc.weight, model.last_fc.bias)


# ---- transformer blocks (Full_model/SubLayers.py:30-59,74-84; Layers.py:18-22,50-58) --------------------------------------
# Default: one fused autograd node per block (functional.mha_block / ffn_block / linear_chain: fused Q|K|V, Dropout + residual in the GEMM epilogue,
# ReLU backward as an epilogue gate).  functional.FUSE_BLOCKS = False, or a width the fused LayerNorm backward does not take (Pose_Discriminator's 282),
# composes the same blocks operator by operator below -- the arithmetic per element is the same.
def weight_image_plan(model):
    """(groups, skip) for optim.FlatParams.enable_weight_images: which attention weights are used as ONE matrix by the fused blocks.
    Encoder self attention: Q|K|V (the three single images are never read); a decoder's `enc_attn`: K|V (Q alone); a decoder's `slf_attn` holds
    parameters the forward never touches (Layers.py:50-58): no image at all."""
    from ..modules import Decoder, Encoder
    groups, skip = [], []
    for mod in model.modules():
        if isinstance(mod, Encoder):
            for layer in mod.layer_stack:
                a = layer.slf_attn
                if F.blocks_fusable(a.w_qs.weight.shape[1]):
                    groups.append((a.w_qs.weight, a.w_ks.weight, a.w_vs.weight))
                    skip += [a.w_qs.weight, a.w_ks.weight, a.w_vs.weight]
        elif isinstance(mod, Decoder):
            for layer in mod.layer_stack:
                a, u = layer.enc_attn, layer.slf_attn
                if F.blocks_fusable(a.w_qs.weight.shape[1]):
                    groups.append((a.w_ks.weight, a.w_vs.weight))
                    skip += [a.w_ks.weight, a.w_vs.weight]
                skip += [u.w_qs.weight, u.w_ks.weight, u.w_vs.weight, u.fc.weight]
    return groups, skip


def mha_forward(m, xq, xk, xv):
    """MultiHeadAttention.forward: LN(fc(attention(q Wq, k Wk, v Wv)) + q).  xq / xk / xv are separate uses of the inputs."""
    xq_p, xq_r = F.fork(xq)
    q = F.linear(xq_p, m.w_qs.weight)
    k = F.linear(xk, m.w_ks.weight)
    v = F.linear(xv, m.w_vs.weight)
    o = F.attention(q, k, v, m.n_head, m.attention.dropout_p if _P["on"] else 0.0)     # dropout on the probabilities (Modules.py:21)
    return F.layer_norm(F.add(_dp(F.linear(o, m.fc.weight), m.dropout_p), xq_r), m.layer_norm)      # q = self.dropout(self.fc(q)) (SubLayers.py:54)


def ffn_forward(f, x):
    xa, xr = F.fork(x)
    h = F.linear(xa, f.w_1.weight, f.w_1.bias, relu=True)
    return F.layer_norm(F.add(_dp(F.linear(h, f.w_2.weight, f.w_2.bias), f.dropout_p), xr), f.layer_norm)         # SubLayers.py:79


class _AddRows(torch.autograd.Function):
    """x + table[row % period] (positional table is a buffer: gradient passes straight through)."""

    @staticmethod
    def forward(ctx, x, table):
        return ops.add_rows(x.detach().contiguous(), table, period=x.shape[1])

    @staticmethod
    def backward(ctx, g):
        return g, None


def _p(p):
    return p if _P["on"] else 0.0


def encoder_forward(enc, x):
    x = _dp(_AddRows.apply(x, enc.position_enc.pos_table[0, :x.shape[1]].contiguous()), enc.dropout_p)        # Models_spatial_memory.py:422
    fused = F.blocks_fusable(x.shape[-1])
    for layer in enc.layer_stack:
        if fused:
            a = layer.slf_attn
            x = F.mha_block(a, x, None, _p(a.attention.dropout_p), _p(a.dropout_p))
            x = F.ffn_block(layer.pos_ffn, x, _p(layer.pos_ffn.dropout_p))
        else:
            a, b, c = _fork_n(x, 3)
            x = ffn_forward(layer.pos_ffn, mha_forward(layer.slf_attn, a, b, c))
    return x


def motion_discriminator_forward(md, x):
    """Motion_Discriminator.forward (Models_spatial_memory.py:658-669) in train() mode: encoder over the motion offsets, per-frame
    Linear + ReLU, 6-layer ReLU MLP -> [B, 1] logit.  x may require a gradient (the generator's adversarial term)."""
    B, T, D = x.shape
    _dropout_on(md)
    try:
        enc = encoder_forward(md.encoder, x)
    finally:
        _P["on"] = False
    h = F.linear(enc.reshape(B * T, D), md.fc1[0].weight, md.fc1[0].bias, relu=True).reshape(B, -1)
    if F.FUSE_BLOCKS:
        return F.linear_chain(h, [md.fc2[i] for i in (0, 2, 4, 6, 8, 10)], relu_between=True)
    for i in (0, 2, 4, 6, 8):
        h = F.linear(h, md.fc2[i].weight, md.fc2[i].bias, relu=True)
    return F.linear(h, md.fc2[10].weight, md.fc2[10].bias)


def pose_discriminator_forward(pd, x, dropout=True):
    """Pose_Discriminator.forward (Models_spatial_memory.py:698-702): encoder -> Linear -> Dropout(0.2) -> Linear -> sigmoid, one probability
    per frame [B, T, 1].  x may require a gradient (the generator's adversarial term).  dropout=False: the eval() forward (same operators)."""
    B, T, D = x.shape
    _dropout_on(pd)
    if not dropout:
        _P["on"] = False
    try:
        enc = encoder_forward(pd.encoder, x)
        h = _dp(F.linear(enc.reshape(B * T, D), pd.fc[0].weight, pd.fc[0].bias), 0.2)
    finally:
        _P["on"] = False
    return F.sigmoid(F.linear(h, pd.fc[2].weight, pd.fc[2].bias)).reshape(B, T, 1)


def decoder_forward(dec, trg, enc_out):
    x = trg
    if F.blocks_fusable(enc_out.shape[-1]):
        uses = _fork_n(enc_out, len(dec.layer_stack))                            # one use per layer: K|V is one product
        for i, layer in enumerate(dec.layer_stack):                              # enc_attn + pos_ffn only (Layers.py:50-58)
            a = layer.enc_attn
            x = F.mha_block(a, x, uses[i], _p(a.attention.dropout_p), _p(a.dropout_p))
            x = F.ffn_block(layer.pos_ffn, x, _p(layer.pos_ffn.dropout_p))
        return x
    uses = _fork_n(enc_out, 2 * len(dec.layer_stack))
    for i, layer in enumerate(dec.layer_stack):
        x = ffn_forward(layer.pos_ffn, mha_forward(layer.enc_attn, x, uses[2 * i], uses[2 * i + 1]))
    return x


def _seq_linear(seq, idx, x, relu_between=False, drop=0.0):
    """Linear chain; `drop`: the nn.Dropout between consecutive Linears of the reference's Sequential (no ReLU there)."""
    if F.FUSE_BLOCKS:
        return F.linear_chain(x, [seq[i] for i in idx], relu_between, _p(drop))
    for j, i in enumerate(idx):
        x = F.linear(x, seq[i].weight, seq[i].bias, relu=relu_between and j + 1 < len(idx))
        if drop and j + 1 < len(idx):
            x = _dp(x, drop)
    return x


def prior_encoder_forward(pe, prior):
    """Prior_MemoryEncoder.forward.  Spatial variant (Models_spatial_memory.py:378-390): SP_Memory_Net_v2 returns its input (:276-295), so its
    parameters receive no gradient and it is not evaluated here (its BatchNorm running buffers stay untouched).  Memory variant
    (Models_memory.py:336-346): SP_Memory_Net_v1's sigmoid gate (:233-251), then TM_Memory_Net (:282-293), whose score mem (mem^T pe) sums over
    the BATCH -- the clips of a training batch are coupled exactly as in the reference."""
    x = prior.transpose(1, 2).contiguous()                                       # [B, L = pose_dim, C = prior frames]
    c0, b0, c1, b1 = pe.pred_conv[0], pe.pred_conv[2], pe.pred_conv[3], pe.pred_conv[5]
    h = F.batch_norm(F.relu(F.conv1d_cl(x, c0.weight, c0.bias, 1, 1, 1)), b0)
    h = F.batch_norm(F.relu(F.conv1d_cl(h, c1.weight, c1.bias, 1, 1, 1)), b1)
    pred = h.transpose(1, 2)                                                     # [B, frames - prior, pose_dim]
    if hasattr(pe, "temporal_memory"):
        B, P = prior.shape[0], prior.shape[1]
        sm, tm = pe.spatial_memory, pe.temporal_memory
        chunk = sm.chunk_length
        tail = prior[:, P - chunk:, :].reshape(B, -1).contiguous()               # initial_feature[:, prior_frames - chunk:, :] (:237, :285); an input: no gradient
        mem = _seq_linear(sm.spatial_chunk_encoder, (0, 2), tail, drop=0.2)      # [B, pose_dim]
        pred = F.sp_memory_gate(mem, pred.contiguous(), chunk)                   # spatial (:341)
        pa, pb = F.fork(pred)
        mem2 = _seq_linear(tm.temporal_chunk_encoder, (0, 2), tail, drop=0.2)    # [B, pose_dim]
        enc = _seq_linear(tm.temporal_memory_encoder, (0, 2), pa[:, :chunk, :].reshape(B, -1), drop=0.2)       # [B, chunk]
        ma, mb = F.fork(mem2)
        tt = F.linear(enc.t(), ma.t())                                           # (mem^T enc)^T = enc^T mem: [chunk, pose_dim]  (:288, the sum over the batch)
        score = F.linear(mb, tt)                                                 # mem (mem^T enc): [B, chunk]  (:289)
        pred = F.tm_memory_scale(score, pb, chunk)                               # temporal (:290-292)
    out = torch.cat((prior, pred), 1)                                            # [B, frames, pose_dim]
    return _seq_linear(pe.post_header, (0, 2), out.contiguous(), drop=0.2)


def audio_encoder_forward(ae, spec):
    """Audio_ResNetEncoder.forward (Models_spatial_memory.py:118-133)."""
    x = _cut("tower", resnetse_forward(ae.feat_extractor, spec))                 # [B,32,31,128]; the tower's backward is 60 % of the step and holds 6 % of the gradient bytes
    x = F.batch_norm(F.conv3x3(x, ae.final_conv1.weight, ae.final_conv1.bias), ae.bn1)          # [B,H,W,F]
    B, H, W, Fr = x.shape
    x = x.permute(0, 3, 1, 2).reshape(B, Fr, H * W)                              # channel c becomes time step c (:124)
    if F.FUSE_BLOCKS:
        return F.linear_chain(x, [ae.fc1, ae.fc2], False, _p(0.2))                                          # :128-130
    return F.linear(_dp(F.linear(x, ae.fc1.weight, ae.fc1.bias), 0.2), ae.fc2.weight, ae.fc2.bias)


def text_encoder_forward_nograd(te, text):
    """TextEncoderTCN.forward (Models_spatial_memory.py:171-179) on the inference kernels, without gradient: the returned
    text_embedding feeds neither the pose nor the emotion head, so the reference's loss never reaches these parameters."""
    with torch.no_grad():
        lv = [(b.conv1.weight_v, b.conv1.weight_g, b.conv1.bias, b.conv2.weight_v, b.conv2.weight_g, b.conv2.bias) for b in te.tcn.network]
        ver = tuple(t._version for tup in lv for t in tup) + (str(text.device),)
        cache = getattr(te, "_train_pack", None)
        if cache is None or cache[0] != ver:
            cache = (ver, ops.pack_tcn_weights(lv, text.device))
            te._train_pack = cache
        emb = te.embedding.weight.detach()[text].contiguous()                   # gather (data movement)
        y = ops.tcn_forward(emb, cache[1], len(lv), "f32")                       # [B, L, C]
        B, Ln, Cc = y.shape
        fc = te.fc1[0]
        z = F.raw_linear(y.transpose(1, 2).reshape(B * Cc, Ln).contiguous(), fc.weight.detach(), fc.bias.detach())       # Linear over time
        z = z.view(B, Cc, Ln).transpose(1, 2).reshape(B * Ln, Cc).contiguous()
        return F.raw_linear(z, te.decoder.weight.detach(), te.decoder.bias.detach()).view(B, Ln, -1)


def generator_forward(model, input_spectrum, text, prior_seq, sampled_emotion_feature=None):
    """Transformer.forward in train() mode, both variants (Models_spatial_memory.py:566-616, Models_memory.py:521-565: they differ in the prior /
    memory encoder only).  The text branch does not feed the pose or the emotion head (:577,616), so it is evaluated without gradient on the
    inference kernels."""
    _dropout_on(model)
    # `model.aux_stream` (a torch.cuda.Stream, optional): the text branch (no gradient, feeds neither head) and the prior-pose branch (meets the
    # rest only at the decoder) run there beside the audio tower: one fork, one join in front of the decoder.  The prior branch's backward then
    # runs on that stream too (autograd replays a node on its forward's stream and orders the gradient hand-over with events).  The host-side
    # call order -- and with it the dropout stream's offsets -- is the one-stream order.
    aux = getattr(model, "aux_stream", None) if text.is_cuda else None
    try:
        if aux is not None:
            cur = torch.cuda.current_stream(text.device)
            aux.wait_stream(cur)
            with torch.cuda.stream(aux):
                text_embedding = text_encoder_forward_nograd(model.text_encoder, text)
        else:
            text_embedding = text_encoder_forward_nograd(model.text_encoder, text)
        spectrum_feature = audio_encoder_forward(model.audio_encoder, input_spectrum)
        if aux is not None:
            with torch.cuda.stream(aux):
                prior = prior_encoder_forward(model.prior_seq_encoder, prior_seq)
        else:
            prior = prior_encoder_forward(model.prior_seq_encoder, prior_seq)
        sa, sb = F.fork(spectrum_feature)
        emotion_feature = _seq_linear(model.emotion_proj, (0, 2), sa, drop=0.2)
        semantic_feature = _seq_linear(model.semantic_proj, (0, 2), sb, drop=0.2)
        B = emotion_feature.shape[0]
        if sampled_emotion_feature is None:
            e_cls, e_fus = F.fork(emotion_feature)
        else:
            e_cls, e_fus = emotion_feature, sampled_emotion_feature
        emotion_prediction = _seq_linear(model.emotion_classifer_header, (0, 2, 4, 6), e_cls.reshape(B, -1), relu_between=True)
        fusion = _seq_linear(model.fusion_proj, (0, 2), F.add(e_fus, semantic_feature), relu_between=True)
        enc_out = encoder_forward(model.encoder, fusion)
        if aux is not None:
            cur.wait_stream(aux)
            text_embedding.record_stream(cur)           # allocated on the side stream, handed to the caller's
            prior.record_stream(cur)
        dec_out = decoder_forward(model.decoder, prior, enc_out)
        pose = _seq_linear(model.post_projector, (0, 2, 4, 6), dec_out, drop=0.2)
    finally:
        _P["on"] = False
        F.flush_batch_counters()
    return pose, emotion_feature, semantic_feature, emotion_prediction, text_embedding


# ---- emotion CVAE (CAVE/BEAT_CVAE.py:312-424) ---------------------------------------------------------------------------------
def _cl(x_ncl):
    return x_ncl.transpose(1, 2).contiguous()


def cvae_forward(vae, Input, y, eps):
    """MLP_Reconstruct_v3.forward in train() mode -> (reconstruction [n, frames, d_model], mu [n,32], logvar [n,32]).
    Conv1d / ConvTranspose1d run channels-last ([n, L, C]); LeakyReLU(0.2) precedes BatchNorm1d as upstream (:318-332,355-369);
    `eps` replaces torch.randn_like(std) of reparameterize (:398)."""
    E, D = vae.Encoder, vae.Decoder
    _dropout_on(vae)            # the Dropout(0.2) inside the four Linear pairs (:336,345,350,380), active with `vae.train_dropout = True`
    try:
        h = _cl(Input)                                                     # [n, L = d_model, C = frames]
        for ci, bi, st, pd in ((0, 2, 1, 1), (3, 5, 1, 1), (6, 8, 2, 2), (9, 11, 2, 2)):
            h = F.batch_norm(F.leaky_relu(F.conv1d_cl(h, E[ci].weight, E[ci].bias, st, pd, 1), 0.2), E[bi])
        n = h.shape[0]
        latent = h.transpose(1, 2).reshape(n, -1)                          # NCL flatten (:409)
        la, lb = F.fork(latent)
        mu = _seq_linear(vae.fc_mu, (0, 2), la, drop=0.2)
        logvar = _seq_linear(vae.fc_var, (0, 2), lb, drop=0.2)
        mu_z, mu_out = F.fork(mu)
        lv_z, lv_out = F.fork(logvar)
        z = F.reparameterize(mu_z, lv_z, eps)
        py = vae.Posterior_Y_embedding
        post_y = _seq_linear(py, (0, 2), y, drop=0.2)
        zc = torch.cat((z, post_y), 1)
        fz = vae.fusion_z_posterior
        zc = _seq_linear(fz, (0, 2), zc, drop=0.2)
    finally:
        _P["on"] = False
    h = _cl(zc.reshape(n, 4, -1))                                      # [n, L = d_model/4, C = 4]
    h = F.batch_norm(F.leaky_relu(F.conv_transpose1d_cl(h, D[0].weight, D[0].bias), 0.2), D[2])
    h = F.batch_norm(F.leaky_relu(F.conv_transpose1d_cl(h, D[3].weight, D[3].bias), 0.2), D[5])
    h = F.batch_norm(F.leaky_relu(F.conv1d_cl(h, D[6].weight, D[6].bias, 1, 1, 1), 0.2), D[8])
    h = F.batch_norm(F.leaky_relu(F.conv1d_cl(h, D[9].weight, D[9].bias, 1, 1, 1), 0.2), D[11])
    h = F.conv1d_cl(h, D[12].weight, D[12].bias, 1, 1, 1)
    F.flush_batch_counters()
    return h.transpose(1, 2), mu_out, lv_out
