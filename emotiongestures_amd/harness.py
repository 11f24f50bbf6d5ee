"""Evaluation harness of the audio->gesture path.

Mirrors what the reference's eval loop does around the generator
(test_emotion_gesture_diversity_iterative.py:191-261): CVAE sample -> generator -> pose, then the FGD auto-encoder
features of predicted and target poses (model/FGD.py:26-82), their Frechet distance and diversity score
(model/FHD_score.py:159-217,247-311), MPJRE, pose L2 and the emotion accuracy of a skeleton classifier
(skeleton_classifer/Models.py:199-283).  The beat-alignment score is left out: it is a per-sample librosa routine
(model/Beat_score_v2.py) and librosa is neither vendored nor installed.

Every network forward runs on the GPU through libemogest_hip.so (the FGD encoder and the classifier are built from the
same Linear / attention / LayerNorm operators as the generator); the Gaussian statistics and the matrix square root
stay float64 numpy/scipy on the host exactly as upstream.
"""
from __future__ import annotations

from typing import Dict, Iterable, Optional, Tuple

import numpy as np
import torch
import torch.nn as nn

from . import _lib as L
from . import ops
from .modules import Encoder, Linear, ReplicaAware, _eval_only, _seq, replica_forward

__all__ = ["MLP_Reconstruct", "SkeletonTransformer", "Prior_Encoder", "compute_acc", "l2_distance_pose", "mpjre", "calc_motion",
           "calculate_frechet_distance", "calculate_diversity", "diversity_score", "evaluate"]


class _PackCache:
    """Packed (EG_PACK_LINEAR) copies of nn.Linear weights, rebuilt only when the parameter changes."""

    def __init__(self):
        self._c: Dict[int, tuple] = {}

    def get(self, lin: Linear, device, pad_k: int = 0):
        """pad_k > 0: the weight gets that many zero columns first (K not a multiple of 4, e.g. pose_dim 126 / 282)."""
        key = id(lin)
        ver = (lin.weight._version, str(device), pad_k)
        hit = self._c.get(key)
        if hit is None or hit[0] != ver:
            w = lin.weight if not pad_k else torch.cat([lin.weight, lin.weight.new_zeros(lin.weight.shape[0], pad_k)], 1)
            hit = (ver, ops.pack_linear_weight(w, device), w.detach())
            self._c[key] = hit
        return hit[1]

    def padded_weight(self, lin: Linear):
        return self._c[id(lin)][2]


def _affine_chain(cache: _PackCache, x: torch.Tensor, layers, relu_between: bool, precision: str) -> torch.Tensor:
    """x [rows, K] through Linear layers (optionally ReLU between them, never after the last)."""
    for i, lin in enumerate(layers):
        last = i == len(layers) - 1
        k = lin.weight.shape[1]
        if k % 4:                         # e.g. pose_dim 282/126: pad K with zero columns (layout plumbing)
            pad = (-k) % 4
            x = torch.cat([x, x.new_zeros(x.shape[0], pad)], 1)
            packed = cache.get(lin, x.device, pad_k=pad)
            x = ops.linear(x, cache.padded_weight(lin), lin.bias, relu=relu_between and not last, precision=precision, packed=packed)
        elif x.shape[0] <= 64 and k >= 8192:
            x = ops.linear_splitk(x, lin.weight, lin.bias, relu=relu_between and not last, splits=max(1, k // 512), precision=precision,
                                  packed=cache.get(lin, x.device))
        else:
            x = ops.linear(x, lin.weight, lin.bias, relu=relu_between and not last, precision=precision, packed=cache.get(lin, x.device))
    return x


class MLP_Reconstruct(ReplicaAware, nn.Module):
    """model/FGD.py:26-82: per-frame pose auto-encoder whose 512-d latent is the FGD feature.  ``pose_dim`` is 282 upstream
    (hard-coded :32,58); Dropout layers are identity in eval."""

    def __init__(self, bath=True, *, pose_dim=282, precision="f32"):
        super().__init__()
        self.Encoder = _seq(Linear(pose_dim, 512), None, Linear(512, 512), None, Linear(512, 512))
        self.Decoder = _seq(Linear(512, 512), None, Linear(512, 512), None, Linear(512, pose_dim))
        self.precision = precision
        self._cache = _PackCache()

    @replica_forward
    def forward(self, Input):
        _eval_only(self)
        shp = Input.shape
        x = Input.reshape(-1, shp[-1]).contiguous()
        latent = _affine_chain(self._cache, x, [self.Encoder[0], self.Encoder[2], self.Encoder[4]], False, self.precision)
        out = _affine_chain(self._cache, latent, [self.Decoder[0], self.Decoder[2], self.Decoder[4]], False, self.precision)
        return out.view(*shp[:-1], out.shape[-1]), latent.view(*shp[:-1], 512)


class Prior_Encoder(nn.Module):
    """skeleton_classifer/Models.py:88-116"""

    def __init__(self, pose_dim, d_model):
        super().__init__()
        self.fc1, self.fc2 = Linear(pose_dim, d_model), Linear(d_model, d_model)


class SkeletonTransformer(ReplicaAware, nn.Module):
    """skeleton_classifer/Models.py:199-283: emotion classifier on a pose sequence -> (logits [B,8], mid_feature [B,T,d])."""

    def __init__(self, class_dim=8, pose_dim=242, src_pad_idx=1, trg_pad_idx=1, d_word_vec=64, d_model=64, d_inner=512,
                 n_layers=3, n_head=8, d_k=32, d_v=32, dropout=0.2, n_position=60, *, precision="f32"):
        super().__init__()
        assert d_model == d_word_vec
        self.d_model = d_model
        self.prior_seq_encoder = Prior_Encoder(pose_dim, d_model)
        self.post_projector = _seq(Linear(n_position * d_model, d_model * 4), None, Linear(d_model * 4, d_model), None,
                                   Linear(d_model, 128), None, Linear(128, 64), None, Linear(64, class_dim))
        self.encoder = Encoder(n_position=n_position, d_word_vec=d_word_vec, d_model=d_model, d_inner=d_inner, n_layers=n_layers,
                               n_head=n_head, d_k=d_k, d_v=d_v, pad_idx=src_pad_idx, dropout=dropout)
        for p in self.parameters():
            if p.dim() > 1:
                nn.init.xavier_uniform_(p)
        self.precision = precision
        self._cache = _PackCache()

    @replica_forward
    def forward(self, prior_seq):
        _eval_only(self)
        B, T, D = prior_seq.shape
        for layer in self.encoder.layer_stack:
            layer.slf_attn.precision = layer.pos_ffn.precision = self.precision
        x = _affine_chain(self._cache, prior_seq.reshape(B * T, D).contiguous(), [self.prior_seq_encoder.fc1, self.prior_seq_encoder.fc2],
                          False, self.precision).view(B, T, self.d_model)
        enc, *_ = self.encoder(x, None)
        mid = enc
        head = [self.post_projector[i] for i in (0, 2, 4, 6, 8)]
        logits = _affine_chain(self._cache, enc.reshape(B, -1).contiguous(), head, True, self.precision)
        return logits, mid


# ---- training-side types on the path, forward only ---------------------------------------------------
class Motion_Discriminator(ReplicaAware, nn.Module):
    """Full_model/Models_spatial_memory.py:620-669 (identical class in Models_memory.py): encoder over the motion offsets,
    per-frame Linear+ReLU, 6-layer ReLU MLP -> [B, 1] logit (no sigmoid).  Its forward needs pose_dim == d_word_vec == d_model
    (the encoder adds a d_word_vec-wide table to the raw offsets and fc1 consumes the d_model-wide encoder output as if it were
    pose_dim wide), which the upstream defaults (128 vs 282) violate; the constructor refuses such a combination up front.
    eval(): the inference kernels; train(): the differentiable operators of emotiongestures_amd/train (gradient goldens from the reference:
    tests/golden/adv_grads.npz)."""

    def __init__(self, frames=59, pose_dim=282, src_pad_idx=1, trg_pad_idx=1, d_word_vec=128, d_model=128, d_inner=1024, n_layers=2,
                 n_head=8, d_k=64, d_v=64, dropout=0.2, n_position=59, *, precision="f32"):
        super().__init__()
        if not (pose_dim == d_word_vec == d_model):
            raise ValueError(f"Motion_Discriminator: forward needs pose_dim == d_word_vec == d_model (got {pose_dim}, {d_word_vec}, "
                             f"{d_model}); the upstream defaults fail at the first tensor add")
        self.d_model, self.frames = d_model, frames
        self.encoder = Encoder(n_position=n_position, d_word_vec=d_word_vec, d_model=d_model, d_inner=d_inner, n_layers=n_layers,
                               n_head=n_head, d_k=d_k, d_v=d_v, pad_idx=src_pad_idx, dropout=dropout)
        self.fc1 = _seq(Linear(pose_dim, 64), None)
        self.fc2 = _seq(Linear(frames * 64, 2048), None, Linear(2048, 1024), None, Linear(1024, 256), None, Linear(256, 64), None,
                        Linear(64, 16), None, Linear(16, 1))
        for p in self.parameters():
            if p.dim() > 1:
                nn.init.xavier_uniform_(p)
        self.precision = precision
        self._cache = _PackCache()

    @replica_forward
    def forward(self, x):
        B, T, D = x.shape
        if T != self.frames or D != self.d_model:
            raise ValueError(f"Motion_Discriminator.forward: expected [B, {self.frames}, {self.d_model}], got {tuple(x.shape)}")
        if self.training:           # train() mode: differentiable HIP operators (emotiongestures_amd/train), gradients for its parameters and for x
            from .train import nets
            return nets.motion_discriminator_forward(self, x)
        for layer in self.encoder.layer_stack:
            layer.slf_attn.precision = layer.pos_ffn.precision = self.precision
        enc, *_ = self.encoder(x.contiguous(), None)
        h = ops.linear(enc.reshape(B * T, D).contiguous(), self.fc1[0].weight, self.fc1[0].bias, relu=True, precision=self.precision,
                       packed=self._cache.get(self.fc1[0], x.device))
        head = [self.fc2[i] for i in (0, 2, 4, 6, 8, 10)]
        return _affine_chain(self._cache, h.reshape(B, -1).contiguous(), head, True, self.precision)


class Pose_Discriminator(ReplicaAware, nn.Module):
    """Full_model/Models_spatial_memory.py:671-704: encoder over a pose sequence -> per-frame Linear(282, 64) -> Dropout(0.2) -> Linear(64, 1)
    -> sigmoid: one real / fake probability per frame, [B, T, 1].  The head is hard-coded to 282 inputs and consumes the d_model-wide encoder
    output directly, and the encoder adds a d_word_vec-wide positional table to the raw poses: the forward is consistent only for
    d_word_vec == d_model == pose_dim (282 upstream), which the upstream defaults (128) violate -- refused up front, as Motion_Discriminator.
    `pose_dim` is a keyword of this build (upstream: the literal 282).  eval(): the inference kernels; train(): the differentiable operators
    (train/nets.pose_discriminator_forward), the 0.2 Dropout active with `train_dropout`."""

    def __init__(self, src_pad_idx=1, trg_pad_idx=1, d_word_vec=128, d_model=128, d_inner=1024, n_layers=3, n_head=8, d_k=64, d_v=64,
                 dropout=0.2, n_position=60, *, pose_dim=282, precision="f32"):
        super().__init__()
        if not (pose_dim == d_word_vec == d_model):
            raise ValueError(f"Pose_Discriminator: forward needs d_word_vec == d_model == {pose_dim} (got {d_word_vec}, {d_model}); "
                             "the upstream defaults fail at the first tensor add")
        self.d_model = d_model
        self.encoder = Encoder(n_position=n_position, d_word_vec=d_word_vec, d_model=d_model, d_inner=d_inner, n_layers=n_layers,
                               n_head=n_head, d_k=d_k, d_v=d_v, pad_idx=src_pad_idx, dropout=dropout)
        self.fc = _seq(Linear(pose_dim, 64), None, Linear(64, 1))
        for p in self.parameters():
            if p.dim() > 1:
                nn.init.xavier_uniform_(p)
        self.precision = precision

    @replica_forward
    def forward(self, x):
        B, T, D = x.shape
        if D != self.d_model:
            raise ValueError(f"Pose_Discriminator.forward: expected [B, T, {self.d_model}], got {tuple(x.shape)}")
        from .train import functional as TF
        from .train import nets
        if self.training:
            return nets.pose_discriminator_forward(self, x)
        # eval(): the same HIP operators without a tape and with every Dropout off.  (The fused inference blocks -- eg_multi_head_attention,
        # eg_positionwise_ffn -- stream 16-byte-aligned rows; a 282-wide model is not, so the encoder runs operator by operator here.)
        with TF.precision(self.precision if self.precision in ("f32", "bf16x3") else "f32"), torch.no_grad():
            return nets.pose_discriminator_forward(self, x, dropout=False)


class SoftmaxContrastiveLoss(nn.Module):
    """test_emotion_gesture_diversity_iterative.py:80-127 on the GPU (eg_contrastive_loss: one workgroup per row, fixed-order
    reductions).  forward -> scalar loss tensor (differentiable when an input requires a gradient: eg_contrastive_loss_backward);
    evaluate -> (accuracy, cross_dist [n, n])."""

    def _run(self, face_feat, audio_feat, want_cross):
        if face_feat.dim() != 2 or face_feat.shape != audio_feat.shape:
            raise ValueError(f"SoftmaxContrastiveLoss: expected two [n, d] tensors, got {tuple(face_feat.shape)} and {tuple(audio_feat.shape)}")
        if not face_feat.is_cuda or not audio_feat.is_cuda:
            raise RuntimeError("SoftmaxContrastiveLoss runs on the GPU only (no CPU fallback)")
        lib = L.load()
        f, a = face_feat.detach().float().contiguous(), audio_feat.detach().float().contiguous()
        n, d = f.shape
        dev = f.device
        cross = torch.empty(n, n, device=dev) if want_cross else None
        res = torch.empty(2, device=dev)
        nbytes = int(lib.eg_contrastive_workspace_bytes(n))
        ws = torch.empty(nbytes, dtype=torch.uint8, device=dev)
        p = lambda t: None if t is None else t.data_ptr()
        L.check(lib.eg_contrastive_loss(p(f), p(a), n, d, p(cross), res[0:1].data_ptr(), res[1:2].data_ptr(), p(ws), nbytes,
                                        torch.cuda.current_stream(dev).cuda_stream), "eg_contrastive_loss")
        return res[0], res[1], cross

    @torch.no_grad()
    def evaluate(self, face_feat, audio_feat, mode="max"):
        if mode != "max":
            raise ValueError(mode)
        _loss, acc, cross = self._run(face_feat, audio_feat, True)
        return acc, cross

    def forward(self, face_feat, audio_feat, contrastive_device, mode="max"):
        if mode != "max":
            raise ValueError(mode)
        dev = torch.device(contrastive_device)
        if torch.is_grad_enabled() and (face_feat.requires_grad or audio_feat.requires_grad):
            from .train import functional as TF
            return TF.contrastive_loss(face_feat.to(dev), audio_feat.to(dev)).reshape(())
        return self._run(face_feat.to(dev), audio_feat.to(dev), False)[0]


def adjust_lr(optimizer, init_lr, epoch, decay_rate=0.1, decay_epoch=4):
    """test_emotion_gesture_diversity_iterative.py:64-78: piecewise-constant learning rate written into every param group
    (`decay_rate` / `decay_epoch` are accepted and ignored, as upstream).  Past epoch 150 upstream dies on an unbound local;
    here that is a ValueError."""
    if epoch <= 15:
        lr = init_lr
    elif epoch <= 50:
        lr = init_lr * 0.2
    elif epoch <= 80:
        lr = init_lr * 0.01
    elif epoch <= 100:
        lr = init_lr * 0.005
    elif epoch <= 150:
        lr = init_lr * 0.001
    else:
        raise ValueError(f"adjust_lr: the schedule is defined for epochs 0..150 (got {epoch})")
    for param_group in optimizer.param_groups:
        param_group["lr"] = lr


def set_requires_grad(nets, requires_grad=False):
    """test_emotion_gesture_diversity_iterative.py:51-62"""
    if not isinstance(nets, list):
        nets = [nets]
    for net in nets:
        if net is not None:
            for param in net.parameters():
                param.requires_grad = requires_grad


# ---- metrics (host side, as upstream) ---------------------------------------------------------------------------------
def compute_acc(input_label: torch.Tensor, out: torch.Tensor) -> torch.Tensor:
    """test_emotion_gesture_diversity_iterative.py:35-39"""
    pred = out.topk(1, 1)[1].squeeze(1)
    return 100 * torch.true_divide(torch.sum(pred == input_label), input_label.size(0))


def calc_motion(motion: torch.Tensor) -> torch.Tensor:
    """test_emotion_gesture_diversity_iterative.py:41-44 (frame-to-frame offsets; 60 upstream = n_frames)."""
    n = motion.shape[1]
    return motion[:, 1:n, :] - motion[:, : n - 1, :]


def l2_distance_pose(fake: np.ndarray, gt: np.ndarray) -> float:
    """test_emotion_gesture_diversity_iterative.py:46-49"""
    return float(np.mean(np.linalg.norm(gt - fake, axis=-1)))


def mpjre(target: torch.Tensor, pred: torch.Tensor) -> float:
    """Mean per-joint rotation error term, :223 (mean |target - pred| over 6-d groups)."""
    b = target.shape[0]
    return float(torch.mean(torch.absolute(target.reshape(b, -1, 6) - pred.reshape(b, -1, 6))))


def calculate_frechet_distance(mu1, sigma1, mu2, sigma2, eps=1e-6, imaginary="return_100"):
    """model/FHD_score.py:159-217: ||mu1-mu2||^2 + Tr(C1 + C2 - 2 sqrt(C1 C2)), float64, scipy sqrtm.  Any ValueError raised
    inside upstream's try block (a non-negligible imaginary part of the square root, or scipy rejecting the product, e.g. NaN / inf entries) returns
    100 as that file does (`imaginary="return_100"`); model/embedding_space_evaluator.py:156-209 carries the same formula without the try, so
    the ValueError propagates there (`imaginary="raise"`) -- its one difference, so that copy delegates here."""
    from scipy import linalg

    mu1, mu2 = np.atleast_1d(mu1), np.atleast_1d(mu2)
    sigma1, sigma2 = np.atleast_2d(sigma1), np.atleast_2d(sigma2)
    assert mu1.shape == mu2.shape and sigma1.shape == sigma2.shape
    diff = mu1 - mu2
    try:        # upstream's try covers the square root, the singular-product retry and the trace: ANY ValueError in there returns 100 (:196-213)
        covmean, _ = linalg.sqrtm(sigma1.dot(sigma2), disp=False)
        if not np.isfinite(covmean).all():
            offset = np.eye(sigma1.shape[0]) * eps
            covmean = linalg.sqrtm((sigma1 + offset).dot(sigma2 + offset))
        if np.iscomplexobj(covmean):
            if not np.allclose(np.diagonal(covmean).imag, 0, atol=1e-3):
                raise ValueError("Imaginary component {}".format(np.max(np.abs(covmean.imag))))
            covmean = covmean.real
        tr_covmean = np.trace(covmean)
    except ValueError:
        if imaginary == "raise":
            raise
        return 100
    return diff.dot(diff) + np.trace(sigma1) + np.trace(sigma2) - 2 * tr_covmean


class FrechetAccumulator:
    """Running first and second moments of feature rows ON THE DEVICE ("(sum x, sum x x^T) accumulators"):
    the Frechet distance of model/FHD_score.py:159-217 needs only the mean and covariance of the [N, 512] feature rows, so a
    long evaluation (or several ranks) never has to hold or gather the rows themselves.

    push(rows [N, D] on the GPU):  s += colsum(x - c);  S += (x - c)^T (x - c)   (eg_colsum / eg_gemm_tn, fp32, fixed-order sums;
    c = a fixed shift -- the first batch's column mean by default -- keeps S - s s^T / n well conditioned).
    stats() -> (mu, sigma) in float64 on the host, the same estimator as np.mean / np.cov(rowvar=False).
    all_reduce(): sums (n, s, S) over the ranks of a process group (D + D^2 + 1 numbers per rank); every rank
    must use the same shift (pass `shift=` explicitly, e.g. zeros)."""

    def __init__(self, dim: int = 512, device="cuda", shift: Optional[torch.Tensor] = None):
        self.dim, self.device = dim, torch.device(device)
        self.n = 0
        self.s = torch.zeros(dim, device=self.device)
        self.S = torch.zeros(dim, dim, device=self.device)
        self.shift = None if shift is None else shift.to(self.device, torch.float32).contiguous()

    def push(self, rows: torch.Tensor) -> None:
        from .train import functional as TFn
        x = rows.detach().reshape(-1, self.dim)
        if not x.is_cuda:
            raise L.EgError("FrechetAccumulator.push: rows must be on the GPU (no CPU fallback)")
        x = x.float().contiguous()
        if self.shift is None:
            self.shift = TFn.raw_ew(TFn.EW_SCALE, TFn.raw_colsum(x)[0], None, 1.0 / x.shape[0])
        xc = ops.add_rows(x, TFn.raw_ew(TFn.EW_SCALE, self.shift, None, -1.0).view(1, -1), period=1)
        self.s = TFn.raw_ew(TFn.EW_ADD, self.s, TFn.raw_colsum(xc)[0])
        TFn.raw_gemm_tn(xc, xc, out=self.S, accumulate=True)
        self.n += x.shape[0]

    def all_reduce(self, group=None) -> None:
        import torch.distributed as dist
        n = torch.tensor([float(self.n)], device=self.device, dtype=torch.float64)
        for t in (n, self.s, self.S):
            dist.all_reduce(t, op=dist.ReduceOp.SUM, group=group)
        self.n = int(round(float(n.item())))

    def stats(self):
        n = self.n
        if n < 2:
            raise ValueError("FrechetAccumulator.stats: need at least two rows")
        s, S = self.s.double().cpu().numpy(), self.S.double().cpu().numpy()
        mu = self.shift.double().cpu().numpy() + s / n
        sigma = (S - np.outer(s, s) / n) / (n - 1)
        return mu, (sigma + sigma.T) / 2


def calculate_diversity(activations: np.ndarray, labels, diversity_times: int = 5) -> np.float32:
    """model/FHD_score.py:270-286: mean L2 distance of `diversity_times` random pairs (np.random.randint twice, in that order)."""
    num = len(labels)
    first = np.random.randint(0, num, diversity_times)
    second = np.random.randint(0, num, diversity_times)
    acc = 0.0
    for i, j in zip(first, second):
        acc += float(np.sqrt(((activations[i] - activations[j]) ** 2).sum()))
    return np.float32(acc / diversity_times)


def diversity_score(activations: np.ndarray, frames: int = 60):
    """model/FHD_score.py:247-268: 10 repeats of calculate_diversity -> centre and 95 % normal interval."""
    from scipy import stats

    act = activations.reshape(-1, frames, 512)
    window = np.empty((10, 1))
    for i in range(10):
        window[i] = calculate_diversity(act, act)
    mean, std = np.mean(window, axis=0), np.std(window, axis=0)
    interval = stats.norm.interval(0.95, mean, std)
    return (interval[0] + interval[1]) / 2, interval


# ---- the eval loop ---------------------------------------------------------------------------------------------------
def evaluate(generator, vae, fgd: MLP_Reconstruct, classifier: Optional[SkeletonTransformer], batches: Iterable[dict],
             n_pre_poses: int, device="cuda", z_list: Optional[list] = None) -> Dict[str, float]:
    """One pass of test_model's hot loop (:191-261) over an iterable of batches, each a dict with
    ``spec [B,128,T]``, ``text [B,60]``, ``pose_seq [B,F,D]`` (target; the first n_pre_poses frames are the prior) and
    ``label [B,8]`` one-hot.  ``z_list`` optionally fixes the CVAE latents per batch (default: torch.randn on the CPU
    generator as upstream).  Returns the metrics of the summary line (:261) except the beat score."""
    pred_feats, tgt_feats, l2s, rots, accs = [], [], [], [], []
    frames = None
    with torch.no_grad():
        for bi, batch in enumerate(batches):
            pose_seq = batch["pose_seq"].to(device)
            frames = pose_seq.shape[1]
            pre_pose = pose_seq[:, :n_pre_poses].contiguous()
            label = batch["label"].to(device)
            sampled = vae.sample(label, z=None if z_list is None else z_list[bi])                                    # :203
            pred_pose, _, _, _, _ = generator(batch["spec"].to(device), batch["text"].to(device), pre_pose, sampled)  # :205
            if classifier is not None:
                logits, _ = classifier(pred_pose)                                                                    # :217
                accs.append(float(compute_acc(torch.max(label, 1)[1], logits)))
            rots.append(mpjre(pose_seq, pred_pose))                                                                 # :223
            _, pf = fgd(pred_pose)                                                                                   # :226-229
            _, tf = fgd(pose_seq)
            pred_feats.append(pf.reshape(-1, 512).cpu().numpy().astype(np.float64))
            tgt_feats.append(tf.reshape(-1, 512).cpu().numpy().astype(np.float64))
            l2s.append(l2_distance_pose(pred_pose.cpu().numpy().astype(np.float32), pose_seq.cpu().numpy().astype(np.float32)))
    pred_arr, tgt_arr = np.concatenate(pred_feats), np.concatenate(tgt_feats)
    fid = calculate_frechet_distance(np.mean(pred_arr, axis=0), np.cov(pred_arr, rowvar=False),
                                     np.mean(tgt_arr, axis=0), np.cov(tgt_arr, rowvar=False))                       # :250-255
    div, interval = diversity_score(pred_arr, frames)                                                               # :256
    out = {"pose_l2": float(np.mean(l2s)), "rotation_deg": float(np.mean(rots)) * 57.2958, "fgd": float(np.real(fid)),
           "diversity": float(np.ravel(div)[0]), "diversity_lo": float(np.ravel(interval[0])[0]), "diversity_hi": float(np.ravel(interval[1])[0])}
    if accs:
        out["emotion_acc"] = float(np.mean(accs))
    return out
