"""Host mirror of the reference's nn.Module surface for the audio->gesture path.

Same class names, constructor arguments, ``state_dict`` keys, forward signatures and return
tuples as the reference, so its evaluation flow
(test_emotion_gesture_diversity_iterative.py:135-205) runs unchanged -- but ``forward`` enqueues
HIP kernels through libemogest_hip.so instead of calling ATen.  Parameter-holding leaves
(``Linear``, ``Conv2d`` ...) deliberately have no ``forward``: nothing here can silently fall back
to an eager PyTorch op.

``Transformer`` (spatial variant) and ``EmotionNet`` also run in ``train()`` mode (emotiongestures_amd/train/: fp32 HIP forward
and backward operators under torch.autograd); the other modules are eval-only and raise in ``train()`` mode.
"""
from __future__ import annotations

import copy
import functools
import math
import threading
from typing import Optional

import numpy as np
import torch
import torch.nn as nn

from . import _lib as L
from . import ops
from .engine import CvaeEngine, GeneratorEngine

__all__ = [
    "Linear", "Conv1d", "Conv2d", "ConvTranspose1d", "BatchNorm", "LayerNormParams", "WeightNormConv1d",
    "SELayer", "SEBasicBlock", "ResNetSE", "ScaledDotProductAttention", "MultiHeadAttention",
    "PositionwiseFeedForward", "EncoderLayer", "DecoderLayer", "TemporalBlock", "TemporalConvNet",
    "PositionalEncoding", "Audio_ResNetEncoder", "TextEncoderTCN", "SP_Memory_Net_v1", "SP_Memory_Net_v2",
    "TM_Memory_Net", "Prior_MemoryEncoder", "Encoder", "Decoder", "Transformer", "MLP_Reconstruct_v3",
]

_DEFAULT_PRECISION = "f32"


def set_default_precision(p: str) -> None:
    """Arithmetic mode of new modules' conv/GEMM kernels: 'f32' (parity), 'bf16x3', 'bf16'."""
    global _DEFAULT_PRECISION
    L.precision_code(p)
    _DEFAULT_PRECISION = p


def _eval_only(m: nn.Module) -> None:
    if m.training:
        raise NotImplementedError(
            f"{type(m).__name__}: only the eval-mode forward is implemented on the HIP path (call .eval()); "
            "there is no PyTorch fallback")


# ------------------------------------------------------------------------------------------------
# nn.DataParallel support (the reference's caller wraps every model in it when device_count() > 1:
# test_emotion_gesture_diversity_iterative.py:137-138,150-151,160-161,169-170; train_audio_classifier_K_fold.py:129-130)
# ------------------------------------------------------------------------------------------------
class _DpState:
    """Per-ORIGIN state that must outlive DataParallel's per-forward replicas: a lock, the per-device engines of the arena-based
    modules and the per-device parameter shadows of the operator-composed ones.  Copies (deepcopy / pickle) start empty."""

    def __init__(self):
        self.lock = threading.RLock()
        self.engines = {}          # str(device) -> [engine, key]
        self.shadows = {}          # str(device) -> [module copy on that device, weights version, lock]

    def __deepcopy__(self, memo):
        return _DpState()

    def __reduce__(self):
        return (_DpState, ())


def _tensor_device(args, kwargs):
    for a in list(args) + list(kwargs.values()):
        if isinstance(a, torch.Tensor) and a.is_cuda:
            return a.device
    return None


def _weights_version_of(m: nn.Module):
    # (_version, data_ptr): `param.data = new_tensor` keeps or collides version counters but moves the storage
    return tuple((t._version, t.data_ptr()) for t in list(m.parameters()) + list(m.buffers()))


class ReplicaAware:
    """Mixin for the modules the reference's caller hands to ``nn.DataParallel``.

    ``torch.nn.parallel.replicate`` rebuilds every replica per forward with ``_parameters == {}`` (the broadcast copies are plain
    attributes), so a replica has no ``parameters()`` / ``state_dict()`` and nothing cached on it survives the call.  A replica
    therefore keeps a reference to its ORIGIN, and the work is done by state that lives on the origin (``_DpState``):

    * arena modules (``Transformer``, ``MLP_Reconstruct_v3``): one engine + packed arena per device, keyed by the origin's weight
      version -- packed and uploaded once per device and weight version, not per forward (``engine(device)``);
    * operator-composed modules (FGD auto-encoder, skeleton classifier, EmotionNet, MotionAE): the origin itself on its own
      device, a cached parameter shadow (deep copy, refreshed when the origin's weights change) on the others, so their packed
      weight images persist as well.

    Replicas of one device run under that device's lock (DataParallel runs replicas on threads; two of them on one GPU share
    workspaces).  train() mode under DataParallel is refused: the training layer is one trainer per process
    (``bench.py --gpus N`` / ``train.loops.train_k_fold`` are the data-parallel paths)."""

    def _replicate_for_data_parallel(self):
        replica = super()._replicate_for_data_parallel()
        replica.__dict__["_dp_origin"] = self.__dict__.get("_dp_origin") or self
        return replica

    def _origin(self):
        return self.__dict__.get("_dp_origin") or self

    def _dp(self) -> _DpState:
        o = self._origin()
        st = o.__dict__.get("_dp_state")
        if st is None:
            st = o.__dict__.setdefault("_dp_state", _DpState())
        return st

    def _refuse_replica_training(self):
        if self.training and self.__dict__.get("_dp_origin") is not None:
            raise RuntimeError(
                f"{type(self).__name__}: train() mode under nn.DataParallel is not supported by the HIP path (one trainer per process); "
                "use one process per GPU: `python bench.py --gpus N --train` or emotiongestures_amd.train.loops.train_k_fold under "
                "torch.distributed.run (gradient all-reduce over RCCL)")

    def _device_twin(self, dev):
        """The module that computes for `dev`: the origin on its own device, else a cached shadow of it on `dev`.  -> (module, lock)"""
        o = self._origin()
        st = self._dp()
        home = next(o.parameters()).device
        if dev is None or dev == home:
            return o, st.lock
        with st.lock:
            ver = _weights_version_of(o)
            ent = st.shadows.get(str(dev))
            if ent is None:
                twin = copy.deepcopy(o).to(dev)
                twin.train(o.training)
                ent = st.shadows[str(dev)] = [twin, ver, threading.RLock()]
            elif ent[1] != ver:
                with torch.no_grad():
                    for dst, src in zip(list(ent[0].parameters()) + list(ent[0].buffers()), list(o.parameters()) + list(o.buffers())):
                        dst.copy_(src)                    # in place: the shadow's packed images see a new _version and refresh
                ent[1] = ver
            ent[0].train(o.training)
            return ent[0], ent[2]


def replica_forward(fwd):
    """Decorator for the forward of an operator-composed ReplicaAware module: a DataParallel replica runs the origin's (or its
    per-device shadow's) forward under that device's lock."""
    @functools.wraps(fwd)
    def wrapper(self, *args, **kwargs):
        if self.__dict__.get("_dp_origin") is None:
            return fwd(self, *args, **kwargs)
        self._refuse_replica_training()
        twin, lock = self._device_twin(_tensor_device(args, kwargs))
        with lock:
            return fwd(twin, *args, **kwargs)
    return wrapper


# ------------------------------------------------------------------------------------------------
# parameter holders (same names/shapes/init as the torch layers the reference uses; no forward)
# ------------------------------------------------------------------------------------------------
class _Holder(nn.Module):
    def forward(self, *a, **k):
        raise RuntimeError(f"{type(self).__name__} only holds parameters; it is computed by the fused HIP path of "
                           "its parent module (no eager fallback)")


class Linear(_Holder):
    def __init__(self, in_features, out_features, bias=True):
        super().__init__()
        self.in_features, self.out_features = in_features, out_features
        self.weight = nn.Parameter(torch.empty(out_features, in_features))
        self.bias = nn.Parameter(torch.empty(out_features)) if bias else None
        nn.init.kaiming_uniform_(self.weight, a=math.sqrt(5))
        if bias:
            bound = 1 / math.sqrt(in_features)
            nn.init.uniform_(self.bias, -bound, bound)


class _ConvNd(_Holder):
    def __init__(self, shape, fan_in, bias):
        super().__init__()
        self.weight = nn.Parameter(torch.empty(*shape))
        self.bias = nn.Parameter(torch.empty(shape[0] if not getattr(self, "_transposed", False) else shape[1])) if bias else None
        nn.init.kaiming_uniform_(self.weight, a=math.sqrt(5))
        if bias:
            bound = 1 / math.sqrt(fan_in)
            nn.init.uniform_(self.bias, -bound, bound)


class Conv2d(_ConvNd):
    def __init__(self, cin, cout, kernel_size=3, stride=1, padding=0, bias=True):
        self.stride, self.padding, self.kernel_size = stride, padding, kernel_size
        super().__init__((cout, cin, kernel_size, kernel_size), cin * kernel_size * kernel_size, bias)


class Conv1d(_ConvNd):
    def __init__(self, cin, cout, kernel_size=3, stride=1, padding=0, bias=True):
        self.stride, self.padding, self.kernel_size = stride, padding, kernel_size
        super().__init__((cout, cin, kernel_size), cin * kernel_size, bias)


class ConvTranspose1d(_ConvNd):
    _transposed = True

    def __init__(self, cin, cout, kernel_size=3, stride=2, padding=1, output_padding=1):
        self.stride, self.padding, self.kernel_size = stride, padding, kernel_size
        super().__init__((cin, cout, kernel_size), cout * kernel_size, True)


class BatchNorm(_Holder):
    def __init__(self, n):
        super().__init__()
        self.weight, self.bias = nn.Parameter(torch.ones(n)), nn.Parameter(torch.zeros(n))
        self.register_buffer("running_mean", torch.zeros(n))
        self.register_buffer("running_var", torch.ones(n))
        self.register_buffer("num_batches_tracked", torch.tensor(0, dtype=torch.long))

    def affine(self, eps=1e-5):
        scale = self.weight / torch.sqrt(self.running_var + eps)
        return scale, self.bias - self.running_mean * scale


class LayerNormParams(_Holder):
    def __init__(self, d, eps=1e-6):
        super().__init__()
        self.eps = eps
        self.weight, self.bias = nn.Parameter(torch.ones(d)), nn.Parameter(torch.zeros(d))


class Embedding(_Holder):
    def __init__(self, n, d):
        super().__init__()
        self.weight = nn.Parameter(torch.randn(n, d))


class WeightNormConv1d(_Holder):
    """weight_norm(nn.Conv1d) as torch.nn.utils.weight_norm registers it: bias, weight_g, weight_v
    (Full_model/tcn.py:19-24)."""

    def __init__(self, cin, cout, k):
        super().__init__()
        self.bias = nn.Parameter(torch.empty(cout))
        v = torch.empty(cout, cin, k).normal_(0, 0.01)                 # tcn.py:38-39
        self.weight_g = nn.Parameter(v.flatten(1).norm(dim=1).view(-1, 1, 1).clone())
        self.weight_v = nn.Parameter(v)
        bound = 1 / math.sqrt(cin * k)
        nn.init.uniform_(self.bias, -bound, bound)


class _Placeholder(nn.Module):
    """Dropout / activation slots inside nn.Sequential: keeps the reference's child indices."""

    def forward(self, x):
        raise RuntimeError("placeholder slot; computed by the fused HIP path of the parent module")


def _seq(*mods):
    return nn.Sequential(*[m if m is not None else _Placeholder() for m in mods])


# ------------------------------------------------------------------------------------------------
# audio tower
# ------------------------------------------------------------------------------------------------
class SELayer(nn.Module):
    """Full_model/ResNetBlocks.py:81-96"""

    def __init__(self, channel, reduction=8):
        super().__init__()
        self.fc = _seq(Linear(channel, channel // reduction), None, Linear(channel // reduction, channel), None)


class SEBasicBlock(nn.Module):
    """Full_model/ResNetBlocks.py:7-37.  forward takes/returns NCHW like the reference."""
    expansion = 1

    def __init__(self, inplanes, planes, stride=1, downsample=None, reduction=8):
        super().__init__()
        self.conv1 = Conv2d(inplanes, planes, 3, stride, 1, bias=False)
        self.bn1 = BatchNorm(planes)
        self.conv2 = Conv2d(planes, planes, 3, 1, 1, bias=False)
        self.bn2 = BatchNorm(planes)
        self.se = SELayer(planes, reduction)
        self.downsample = downsample
        self.stride = stride if isinstance(stride, int) else stride[0]
        self.precision = _DEFAULT_PRECISION

    def _packed(self, device):
        """Packed conv weights / folded BatchNorm vectors, rebuilt only when a parameter or buffer changes."""
        ver = (str(device),) + tuple(t._version for t in list(self.parameters()) + list(self.buffers()))
        if getattr(self, "_pack", None) is None or self._pack[0] != ver:
            s1, t1 = self.bn1.affine()
            s2, t2 = self.bn2.affine()
            c1 = ops.conv3x3_pack(self.conv1.weight, None, s1, t1, device)
            c2 = ops.conv3x3_pack(self.conv2.weight, None, s2, t2, device)
            ds = None
            if self.downsample is not None:
                a, b = self.downsample[1].affine()
                ds = (a.detach().to(device).contiguous(), b.detach().to(device).contiguous())
            self._pack = (ver, c1, c2, ds)
        return self._pack[1:]

    def forward_nhwc(self, x):
        _eval_only(self)
        c1, c2, ds = self._packed(x.device)
        h = ops.conv3x3(x, self.conv1.weight, stride=self.stride, relu=True, precision=self.precision, packed=c1)
        y, gap = ops.conv3x3(h, self.conv2.weight, want_gap=True, precision=self.precision, packed=c2)
        gate = ops.se_gate(gap, self.se.fc[0].weight, self.se.fc[0].bias, self.se.fc[2].weight, self.se.fc[2].bias,
                           y.shape[1] * y.shape[2])
        if self.downsample is not None:
            return ops.se_residual_relu(y, gate, x, self.downsample[0].weight, ds[0], ds[1], stride=self.stride)
        return ops.se_residual_relu(y, gate, x)

    def forward(self, x):
        return self.forward_nhwc(x.permute(0, 2, 3, 1).contiguous()).permute(0, 3, 1, 2).contiguous()


class ResNetSE(nn.Module):
    """Full_model/ResNetSE34V2.py:13-74"""

    def __init__(self, block, layers, num_filters):
        super().__init__()
        self.inplanes = num_filters[0]
        self.conv1 = Conv2d(1, num_filters[0], 3, 1, 1)
        self.bn1 = BatchNorm(num_filters[0])
        self.layer1 = self._make_layer(block, num_filters[0], layers[0])
        self.layer2 = self._make_layer(block, num_filters[1], layers[1], stride=(2, 2))
        self.layer3 = self._make_layer(block, num_filters[2], layers[2], stride=(2, 2))
        if len(num_filters) > 3:        # model/emotion_ResNetSE34V2.py:26 (the audio emotion classifier's 256-channel stage)
            self.layer4 = self._make_layer(block, num_filters[3], layers[3], stride=(2, 2))
        for m in self.modules():
            if isinstance(m, Conv2d):
                nn.init.kaiming_normal_(m.weight, mode="fan_out", nonlinearity="relu")

    def _make_layer(self, block, planes, blocks, stride=1):
        downsample = None
        s = stride if isinstance(stride, int) else stride[0]
        if s != 1 or self.inplanes != planes * block.expansion:
            downsample = _seq(Conv2d(self.inplanes, planes * block.expansion, 1, s, 0, bias=False),
                              BatchNorm(planes * block.expansion))
        layers = [block(self.inplanes, planes, s, downsample)]
        self.inplanes = planes * block.expansion
        layers += [block(self.inplanes, planes) for _ in range(1, blocks)]
        return nn.Sequential(*layers)

    def forward_nhwc(self, spec):
        """spec [B,H,W] -> NHWC feature map."""
        _eval_only(self)
        s, t = self.bn1.affine()
        x = ops.stem_conv(spec, self.conv1.weight, self.conv1.bias, s, t)
        for layer in (self.layer1, self.layer2, self.layer3, getattr(self, "layer4", ())):
            for blk in layer:
                x = blk.forward_nhwc(x)
        return x

    def forward(self, x):
        return self.forward_nhwc(x[:, 0]).permute(0, 3, 1, 2).contiguous()


# ------------------------------------------------------------------------------------------------
# transformer blocks
# ------------------------------------------------------------------------------------------------
class ScaledDotProductAttention(nn.Module):
    """Full_model/Modules.py:5-23 (the gesture path always passes mask = None; a mask is honoured: masked_fill(mask == 0, -1e9))."""

    def __init__(self, temperature, attn_dropout=0.1):
        super().__init__()
        self.temperature = temperature
        self.dropout_p = float(attn_dropout)        # Modules.py:11; acts only in train() mode with train_dropout (train/nets.py)

    def forward(self, q, k, v, mask=None):
        _eval_only(self)
        b, h, lq, d = q.shape
        if abs(self.temperature - d ** 0.5) > 1e-6:
            raise NotImplementedError("temperature must be sqrt(d_k)")
        flat = lambda t: t.transpose(1, 2).reshape(t.shape[0], t.shape[2], h * d)
        out, attn = ops.attention(flat(q), flat(k), flat(v), h, want_attn=True, mask=mask)
        return out.view(b, lq, h, d).transpose(1, 2), attn


class MultiHeadAttention(nn.Module):
    """Full_model/SubLayers.py:9-59"""

    def __init__(self, n_head, d_model, d_k, d_v, dropout=0.1):
        super().__init__()
        self.n_head, self.d_k, self.d_v = n_head, d_k, d_v
        self.w_qs = Linear(d_model, n_head * d_k, bias=False)
        self.w_ks = Linear(d_model, n_head * d_k, bias=False)
        self.w_vs = Linear(d_model, n_head * d_v, bias=False)
        self.fc = Linear(n_head * d_v, d_model, bias=False)
        self.attention = ScaledDotProductAttention(temperature=d_k ** 0.5)
        self.layer_norm = LayerNormParams(d_model, eps=1e-6)
        self.dropout_p = float(dropout)              # SubLayers.py:26
        self.precision = _DEFAULT_PRECISION

    def forward(self, q, k, v, mask=None):
        _eval_only(self)
        if k is not v and not torch.equal(k, v):
            raise NotImplementedError("HIP MultiHeadAttention: k is v, as on the reference path")
        ws = (self.w_qs.weight, self.w_ks.weight, self.w_vs.weight, self.fc.weight)
        ver = (str(q.device),) + tuple(w._version for w in ws)
        if getattr(self, "_pack", None) is None or self._pack[0] != ver:        # packed once per weight version (standalone use)
            self._pack = (ver, [ops.pack_linear_weight(w, q.device)[0] for w in ws])
        if mask is not None:
            # the fused block (eg_multi_head_attention) has no mask argument -- the path never passes one; with a mask the same kernels run
            # operator by operator: projections, masked attention (SubLayers.py:44-47: the head axis is broadcast), output projection +
            # residual, LayerNorm
            B, Lq, D = q.shape
            pk = self._pack[1]
            lin = lambda x, w, p, res=None: ops.linear(x.reshape(-1, x.shape[-1]).contiguous(), w, None, res1=res, precision=self.precision,
                                                       packed=(p, (w.shape[0] + 63) // 64 * 64, (w.shape[1] + 63) // 64 * 64))
            qp = lin(q, ws[0], pk[0]).view(B, Lq, -1)
            kp = lin(k, ws[1], pk[1]).view(B, k.shape[1], -1)
            vp = lin(v, ws[2], pk[2]).view(B, v.shape[1], -1)
            o, attn = ops.attention(qp, kp, vp, self.n_head, want_attn=True, precision=self.precision, mask=mask)
            pre = lin(o, ws[3], pk[3], res=q.reshape(-1, D).contiguous())
            return ops.layernorm(pre, self.layer_norm.weight, self.layer_norm.bias, eps=1e-6).view(B, Lq, D), attn
        return ops.multi_head_attention(q, k, *ws, self.layer_norm.weight, self.layer_norm.bias, self.n_head, self.precision,
                                        packed=self._pack[1])


class PositionwiseFeedForward(nn.Module):
    """Full_model/SubLayers.py:64-84"""

    def __init__(self, d_in, d_hid, dropout=0.1):
        super().__init__()
        self.w_1, self.w_2 = Linear(d_in, d_hid), Linear(d_hid, d_in)
        self.layer_norm = LayerNormParams(d_in, eps=1e-6)
        self.dropout_p = float(dropout)              # SubLayers.py:72
        self.precision = _DEFAULT_PRECISION

    def forward(self, x):
        _eval_only(self)
        ver = (str(x.device), self.w_1.weight._version, self.w_2.weight._version)
        if getattr(self, "_pack", None) is None or self._pack[0] != ver:
            self._pack = (ver, (ops.pack_linear_weight(self.w_1.weight, x.device)[0], ops.pack_linear_weight(self.w_2.weight, x.device)[0]))
        return ops.positionwise_ffn(x, self.w_1.weight, self.w_1.bias, self.w_2.weight, self.w_2.bias, self.layer_norm.weight,
                                    self.layer_norm.bias, self.precision, packed=self._pack[1])


class EncoderLayer(nn.Module):
    """Full_model/Layers.py:10-22"""

    def __init__(self, d_model, d_inner, n_head, d_k, d_v, dropout=0.1):
        super().__init__()
        self.slf_attn = MultiHeadAttention(n_head, d_model, d_k, d_v, dropout=dropout)
        self.pos_ffn = PositionwiseFeedForward(d_model, d_inner, dropout=dropout)

    def forward(self, enc_input, slf_attn_mask=None):
        out, attn = self.slf_attn(enc_input, enc_input, enc_input, mask=slf_attn_mask)
        return self.pos_ffn(out), attn


class DecoderLayer(nn.Module):
    """Full_model/Layers.py:41-58: slf_attn parameters exist but only enc_attn + pos_ffn run."""

    def __init__(self, d_model, d_inner, n_head, d_k, d_v, dropout=0.1):
        super().__init__()
        self.slf_attn = MultiHeadAttention(n_head, d_model, d_k, d_v, dropout=dropout)
        self.enc_attn = MultiHeadAttention(n_head, d_model, d_k, d_v, dropout=dropout)
        self.pos_ffn = PositionwiseFeedForward(d_model, d_inner, dropout=dropout)

    def forward(self, dec_input, enc_output, slf_attn_mask=None, dec_enc_attn_mask=None):
        out, attn = self.enc_attn(dec_input, enc_output, enc_output, mask=dec_enc_attn_mask)
        return self.pos_ffn(out), None, attn


class PositionalEncoding(nn.Module):
    """Full_model/Models_spatial_memory.py:25-57 (table in float64, then fp32)."""

    def __init__(self, d_hid, n_position=200):
        super().__init__()
        self.register_buffer("pos_table", self._table(n_position, d_hid))
        self.register_buffer("pos_table2", self._table(n_position, d_hid))

    @staticmethod
    def _table(n_position, d_hid):
        pos = np.arange(n_position, dtype=np.float64)[:, None]
        j = np.arange(d_hid)
        t = pos / np.power(10000, 2 * (j // 2) / d_hid)[None, :]
        t[:, 0::2] = np.sin(t[:, 0::2])
        t[:, 1::2] = np.cos(t[:, 1::2])
        return torch.FloatTensor(t).unsqueeze(0)


class Encoder(nn.Module):
    """Full_model/Models_spatial_memory.py:395-436"""

    def __init__(self, d_word_vec, n_layers, n_head, d_k, d_v, d_model, d_inner, pad_idx, dropout=0.1, n_position=200,
                 use_wscale=True):
        super().__init__()
        self.position_embeddings = Embedding(n_position, d_model)
        self.position_enc = PositionalEncoding(d_word_vec, n_position=n_position)
        self.layer_stack = nn.ModuleList([EncoderLayer(d_model, d_inner, n_head, d_k, d_v, dropout=dropout) for _ in range(n_layers)])
        self.layer_norm = LayerNormParams(d_model, eps=1e-6)
        self.dropout_p = float(dropout)              # Models_spatial_memory.py:407

    def forward(self, src_seq, src_mask, return_attns=False, global_feature=False):
        _eval_only(self)
        x = ops.add_rows(src_seq, self.position_enc.pos_table[0, :src_seq.size(1)].contiguous(), period=src_seq.size(1))
        attns = []
        for layer in self.layer_stack:
            x, a = layer(x, slf_attn_mask=src_mask)
            attns += [a] if return_attns else []
        return (x, attns) if return_attns else (x,)


class Decoder(nn.Module):
    """Full_model/Models_spatial_memory.py:438-469"""

    def __init__(self, d_word_vec, n_layers, n_head, d_k, d_v, d_model, d_inner, pad_idx, n_position=200, dropout=0.1):
        super().__init__()
        self.position_enc = PositionalEncoding(d_word_vec, n_position=n_position)
        self.layer_stack = nn.ModuleList([DecoderLayer(d_model, d_inner, n_head, d_k, d_v, dropout=dropout) for _ in range(n_layers)])
        self.layer_norm = LayerNormParams(d_model, eps=1e-6)

    def forward(self, trg_seq, trg_mask, enc_output, src_mask, return_attns=False):
        _eval_only(self)
        x, attns = trg_seq, []
        for layer in self.layer_stack:
            x, _, a = layer(x, enc_output, slf_attn_mask=trg_mask, dec_enc_attn_mask=src_mask)
            attns += [a] if return_attns else []
        return (x, [], attns) if return_attns else (x, attns)


# ------------------------------------------------------------------------------------------------
# TCN text branch
# ------------------------------------------------------------------------------------------------
class TemporalBlock(nn.Module):
    """Full_model/tcn.py:16-47.  state_dict keeps the reference's duplicated keys (conv1.* == net.0.*)."""

    def __init__(self, n_inputs, n_outputs, kernel_size, stride, dilation, padding, dropout=0.2):
        super().__init__()
        if n_inputs != n_outputs or kernel_size != 2 or stride != 1:
            raise NotImplementedError("HIP TemporalBlock: equal in/out channels, kernel 2, stride 1 (the EmotionGesture config)")
        self.conv1 = WeightNormConv1d(n_inputs, n_outputs, kernel_size)
        self.conv2 = WeightNormConv1d(n_outputs, n_outputs, kernel_size)
        self.net = _seq(self.conv1, None, None, None, self.conv2, None, None, None)
        self.downsample = None
        self.dilation = dilation


class TemporalConvNet(nn.Module):
    """Full_model/tcn.py:49-64; forward takes/returns [B, C, L] like the reference."""

    def __init__(self, num_inputs, num_channels, kernel_size=2, dropout=0.2):
        super().__init__()
        layers = []
        for i, ch in enumerate(num_channels):
            cin = num_inputs if i == 0 else num_channels[i - 1]
            layers.append(TemporalBlock(cin, ch, kernel_size, stride=1, dilation=2 ** i, padding=(kernel_size - 1) * 2 ** i,
                                        dropout=dropout))
        self.network = nn.Sequential(*layers)
        self.precision = _DEFAULT_PRECISION

    def forward(self, x):
        _eval_only(self)
        lv = [(b.conv1.weight_v, b.conv1.weight_g, b.conv1.bias, b.conv2.weight_v, b.conv2.weight_g, b.conv2.bias)
              for b in self.network]
        packed = ops.pack_tcn_weights(lv, x.device)
        y = ops.tcn_forward(x.transpose(1, 2).contiguous(), packed, len(lv), self.precision)
        return y.transpose(1, 2).contiguous()


class TextEncoderTCN(nn.Module):
    """Full_model/Models_spatial_memory.py:143-179"""

    def __init__(self, args, n_words, embed_size=300, pre_trained_embedding=None, kernel_size=2, dropout=0.3, emb_dropout=0.1):
        super().__init__()
        self.embedding = Embedding(n_words, embed_size)
        if pre_trained_embedding is not None:
            assert pre_trained_embedding.shape[0] == n_words and pre_trained_embedding.shape[1] == embed_size
            self.embedding.weight.data.copy_(torch.FloatTensor(pre_trained_embedding))
            self.embedding.weight.requires_grad_(not args.freeze_wordembed)
        self.tcn = TemporalConvNet(embed_size, [args.hidden_size] * args.n_layers, kernel_size, dropout=dropout)
        self.decoder = Linear(args.hidden_size, 512)
        self.decoder.bias.data.fill_(0)
        self.decoder.weight.data.normal_(0, 0.01)
        self.fc1 = _seq(Linear(60, 60))


# ------------------------------------------------------------------------------------------------
# prior / memory encoder, audio encoder (parameter trees; computed inside Transformer.forward)
# ------------------------------------------------------------------------------------------------
class SP_Memory_Net_v1(nn.Module):
    """Full_model/Models_memory.py:215-251"""

    def __init__(self, args, prior_frames, pred_frames, pose_dim, d_model):
        super().__init__()
        self.chunk_length = args.chunk
        self.spatial_chunk_encoder = _seq(Linear(args.chunk * pose_dim, pose_dim), None, Linear(pose_dim, pose_dim))


class SP_Memory_Net_v2(nn.Module):
    """Full_model/Models_spatial_memory.py:255-295: parameters exist, output == input (writes go to a clone)."""

    def __init__(self, args, prior_frames, pred_frames, pose_dim, d_model):
        super().__init__()
        self.chunk_length = args.chunk
        self.spatial_chunk_encoder = _seq(Conv1d(args.chunk, 1, 3, 1, 1), None, BatchNorm(1), Conv1d(1, 1, 3, 1, 1), None, BatchNorm(1))


class TM_Memory_Net(nn.Module):
    """Full_model/Models_memory.py:263-293"""

    def __init__(self, args, prior_frames, pred_frames, pose_dim, d_model):
        super().__init__()
        self.chunk_length = args.chunk
        self.temporal_chunk_encoder = _seq(Linear(args.chunk * pose_dim, pose_dim), None, Linear(pose_dim, pose_dim))
        self.temporal_memory_encoder = _seq(Linear(args.chunk * pose_dim, args.chunk), None, Linear(args.chunk, args.chunk))


class Prior_MemoryEncoder(nn.Module):
    """Full_model/Models_spatial_memory.py:341-390 (variant 'spatial') / Models_memory.py:299-346 ('memory')."""

    def __init__(self, args, prior_frames, frames, pose_dim, d_model, variant="spatial"):
        super().__init__()
        self.post_header = _seq(Linear(pose_dim, d_model), None, Linear(d_model, d_model))
        self.pred_length = frames - prior_frames
        pl = self.pred_length
        self.pred_conv = _seq(Conv1d(prior_frames, pl, 3, 1, 1), None, BatchNorm(pl), Conv1d(pl, pl, 3, 1, 1), None, BatchNorm(pl))
        if variant == "spatial":
            self.spatial_memory = SP_Memory_Net_v2(args, prior_frames, pl, pose_dim, d_model)
        else:
            self.spatial_memory = SP_Memory_Net_v1(args, prior_frames, pl, pose_dim, d_model)
            self.temporal_memory = TM_Memory_Net(args, prior_frames, pl, pose_dim, d_model)


class Audio_ResNetEncoder(nn.Module):
    """Full_model/Models_spatial_memory.py:92-133"""

    def __init__(self, frames, d_model, fc_in=32 * 31):
        super().__init__()
        num_filters = [32, 64, 128]
        self.feat_extractor = ResNetSE(SEBasicBlock, [3, 4, 6], num_filters)
        self.final_conv1 = Conv2d(num_filters[2], frames, 3, 1, 1)
        self.bn1 = BatchNorm(frames)
        self.fc1 = Linear(fc_in, d_model)          # 32*31 upstream (:105); derived from the spectrogram size here
        self.fc2 = Linear(d_model, d_model)


# ------------------------------------------------------------------------------------------------
# the generator
# ------------------------------------------------------------------------------------------------
class Transformer(ReplicaAware, nn.Module):
    """Full_model/Models_spatial_memory.py:471-616 and Full_model/Models_memory.py:426-565 (``_variant``).

    Extra keyword-only arguments (not in the reference, defaults reproduce it): ``spec_len`` /
    ``n_mels`` size the audio tower for spectrograms other than 128x124 (Appendix B of SURVEY.md);
    ``precision`` picks the MFMA arithmetic ('f32' parity mode, 'bf16x3', 'bf16')."""
    _variant = "spatial"

    def __init__(self, args, lang_model, frames=60, pose_dim=282, prior_frames=10, src_pad_idx=1, trg_pad_idx=1,
                 d_word_vec=64, d_model=64, d_inner=512, n_layers=3, n_head=8, d_k=32, d_v=32, dropout=0.2, n_position=60,
                 *, spec_len=124, n_mels=128, precision=None):
        super().__init__()
        assert d_model == d_word_vec, "To facilitate the residual connections, the dimensions of all module outputs shall be the same."
        self.d_model = d_model
        self.src_pad_idx, self.trg_pad_idx = src_pad_idx, trg_pad_idx
        h3 = ((n_mels - 1) // 2) // 2 + 1
        w3 = ((spec_len - 1) // 2) // 2 + 1
        self.audio_encoder = Audio_ResNetEncoder(frames, d_model, fc_in=h3 * w3)
        self.text_encoder = TextEncoderTCN(args, lang_model.n_words, args.wordembed_dim,
                                           pre_trained_embedding=lang_model.word_embedding_weights, dropout=args.dropout_prob)
        self.emotion_proj = _seq(Linear(d_model, d_model), None, Linear(d_model, d_model))
        self.emotion_classifer_header = _seq(Linear(frames * d_model, d_model), None, Linear(d_model, 256), None,
                                             Linear(256, 64), None, Linear(64, 8))
        self.semantic_proj = _seq(Linear(d_model, d_model), None, Linear(d_model, d_model))
        self.fusion_proj = _seq(Linear(d_model, d_model), None, Linear(d_model, d_model))
        self.prior_seq_encoder = Prior_MemoryEncoder(args, prior_frames, frames, pose_dim, d_model, variant=self._variant)
        self.post_projector = _seq(Linear(d_model, d_model * 4), None, Linear(d_model * 4, d_model), None,
                                   Linear(d_model, pose_dim), None, Linear(pose_dim, pose_dim))
        self.encoder = Encoder(n_position=n_position, d_word_vec=d_word_vec, d_model=d_model, d_inner=d_inner, n_layers=n_layers,
                               n_head=n_head, d_k=d_k, d_v=d_v, pad_idx=src_pad_idx, dropout=dropout)
        self.decoder = Decoder(n_position=n_position, d_word_vec=d_word_vec, d_model=d_model, d_inner=d_inner, n_layers=n_layers,
                               n_head=n_head, d_k=d_k, d_v=d_v, pad_idx=trg_pad_idx, dropout=dropout)
        for p in self.parameters():                      # Models_spatial_memory.py:557-559
            if p.dim() > 1:
                nn.init.xavier_uniform_(p)
        if d_k != d_v:
            raise NotImplementedError("HIP path: d_k == d_v")
        if n_position < frames:
            raise ValueError(f"n_position={n_position} < frames={frames}: the positional table would be too short "
                             "(the reference hard-codes n_position=60, Models_spatial_memory.py:477)")
        self._cfg = dict(frames=frames, pose_dim=pose_dim, prior_frames=prior_frames, chunk=args.chunk, d_model=d_model,
                         d_inner=d_inner, n_layers=n_layers, n_head=n_head, d_k=d_k, n_mels=n_mels, spec_len=spec_len,
                         text_len=60, n_words=lang_model.n_words, embed_dim=args.wordembed_dim, tcn_hidden=args.hidden_size,
                         tcn_layers=args.n_layers, variant=self._variant, n_position=n_position)
        self.precision = precision or _DEFAULT_PRECISION
        self.keep_taps = False
        self.concurrent = False          # fork the text / prior branches onto side streams (engine option)
        self.fold_affine = False         # fold the Dropout-only Linear chains at pack time (fewer launches / FLOPs; off for parity runs)
        self.train_dropout = False       # train(): activate the reference's Dropout layers (default: p = 0, the gradient-parity configuration)
        self.shared_chip = False         # several batches in flight on this GPU (ClipPipeline sets it): GEMM tiles chosen for CU time, not stand-alone latency
        self.fuse_se = True              # identity SE blocks: gate from conv1's output moments, tail in conv2's epilogue (same arithmetic order per element)

    # ---- engine management: repack the arena only when weights / device / mode changed ----
    def _weights_version(self):
        return _weights_version_of(self._origin())

    @staticmethod
    def _engine_slot(dev, shared_chip, concurrent) -> str:
        # the launch-policy classes (stand-alone / several batches in flight; branches on side streams or not) keep SEPARATE engines per device, so
        # that a 1-lane and an N-lane ClipPipeline on one generator do not evict each other's engine (and with it the arena their captured graphs
        # point into)
        return str(dev) + ("#shared" if shared_chip else "") + ("#branches" if concurrent else "")

    def engine_peek(self, shared_chip=None, concurrent=None) -> Optional[GeneratorEngine]:
        """The engine of the module's own device for that launch-policy class as it stands (None before its first use); never packs."""
        o = self._origin()
        sc = self.shared_chip if shared_chip is None else bool(shared_chip)
        cc = self.concurrent if concurrent is None else bool(concurrent)
        ent = self._dp().engines.get(self._engine_slot(next(o.parameters()).device, sc, cc))
        return None if ent is None else ent[0]

    @property
    def _engine(self) -> Optional[GeneratorEngine]:
        """The engine of the module's own device (None before the first forward)."""
        return self.engine_peek()

    def engine(self, device=None, shared_chip=None, concurrent=None) -> GeneratorEngine:
        """The engine (packed weight arena + workspaces) for `device` (default: the parameters' device).  One per device (and launch-policy
        class: `shared_chip` / `concurrent`, default the module's own flags; ClipPipeline passes its own per call and never writes the module's), kept on
        the ORIGIN module, so nn.DataParallel's per-forward replicas reuse it: weights and the version key come from the origin's
        parameters, the arena is packed and uploaded once per device and weight version."""
        o = self._origin()
        dev = torch.device(device) if device is not None else next(o.parameters()).device
        if dev.type != "cuda":
            raise L.EgError("emotiongestures_amd.Transformer runs only on a GPU (model.to('cuda')); there is no CPU fallback")
        if dev.index is None:
            dev = torch.device("cuda", torch.cuda.current_device())
        st = self._dp()
        with st.lock:
            sc = self.shared_chip if shared_chip is None else bool(shared_chip)
            cc = self.concurrent if concurrent is None else bool(concurrent)
            slot = self._engine_slot(dev, sc, cc)
            mode = (str(dev), self.precision, self.keep_taps, cc, self.fold_affine, self.fuse_se, sc)
            key = (mode, self._weights_version())
            ent = st.engines.get(slot)
            if ent is None or ent[1] != key:
                if ent is None or ent[1][0] != mode:
                    ent = st.engines[slot] = [GeneratorEngine(precision=self.precision, keep_taps=self.keep_taps, concurrent=cc,
                                                              fold_affine=self.fold_affine, fuse_se=self.fuse_se, shared_chip=sc,
                                                              **self._cfg), None]
                ent[0].load_weights(o.state_dict(), dev)
                ent[1] = key
            return ent[0]

    def forward(self, input_spectrum, text, prior_seq, sampled_emotion_feature=None, *, slot=0):
        if self.training:           # train() mode: differentiable HIP operators, BatchNorm on batch statistics (train/nets.py)
            self._refuse_replica_training()
            from .train import nets
            return nets.generator_forward(self, input_spectrum, text, prior_seq, sampled_emotion_feature)
        dev = input_spectrum.device if self.__dict__.get("_dp_origin") is not None and input_spectrum.is_cuda else None
        return self.engine(dev).forward(input_spectrum, text, prior_seq, sampled_emotion_feature, slot=slot)

    def forward_draws(self, input_spectrum, prior_seq, sampled_emotion_features, *, slot=0):
        """Diversity sampling (BASELINE config 5): sampled [B,R,frames,d_model] -> pose [B,R,frames,pose_dim]."""
        _eval_only(self)
        return self.engine().forward_draws(input_spectrum, prior_seq, sampled_emotion_features, slot=slot)


class TransformerMemory(Transformer):
    _variant = "memory"


# ------------------------------------------------------------------------------------------------
# emotion CVAE
# ------------------------------------------------------------------------------------------------
class MLP_Reconstruct_v3(ReplicaAware, nn.Module):
    """CAVE/BEAT_CVAE.py:312-460.  ``frames`` (60 upstream, hard-coded :320,365-368) is a keyword here."""

    def __init__(self, bath=True, *, frames=60, d_model=512):
        super().__init__()
        f, q = frames, d_model // 4
        self.Encoder = _seq(Conv1d(f, 32, 3, 1, 1), None, BatchNorm(32), Conv1d(32, 16, 3, 1, 1), None, BatchNorm(16),
                            Conv1d(16, 8, 5, 2, 2), None, BatchNorm(8), Conv1d(8, 4, 5, 2, 2), None, BatchNorm(4))
        self.Posterior_Y_embedding = _seq(Linear(8, 16), None, Linear(16, 32))
        self.fc_mu = _seq(Linear(4 * q, 128), None, Linear(128, 32))
        self.fc_var = _seq(Linear(4 * q, 128), None, Linear(128, 32))
        self.Decoder = _seq(ConvTranspose1d(4, 8), None, BatchNorm(8), ConvTranspose1d(8, 16), None, BatchNorm(16),
                            Conv1d(16, 32, 3, 1, 1), None, BatchNorm(32), Conv1d(32, f, 3, 1, 1), None, BatchNorm(f),
                            Conv1d(f, f, 3, 1, 1))
        self.fusion_z_posterior = _seq(Linear(64, 128), None, Linear(128, 4 * q))
        self._frames, self._d_model = frames, d_model

    @property
    def _engine(self) -> Optional[CvaeEngine]:
        ent = self._dp().engines.get(str(next(self._origin().parameters()).device))
        return None if ent is None else ent[0]

    def engine(self, device=None) -> CvaeEngine:
        """One engine + arena per device on the origin module (see Transformer.engine)."""
        o = self._origin()
        dev = torch.device(device) if device is not None else next(o.parameters()).device
        if dev.type != "cuda":
            raise L.EgError("emotiongestures_amd MLP_Reconstruct_v3 runs only on a GPU; there is no CPU fallback")
        if dev.index is None:
            dev = torch.device("cuda", torch.cuda.current_device())
        st = self._dp()
        with st.lock:
            key = (str(dev), _weights_version_of(o))
            ent = st.engines.get(str(dev))
            if ent is None or ent[1] != key:
                if ent is None:
                    ent = st.engines[str(dev)] = [CvaeEngine(self._frames, self._d_model), None]
                ent[0].load_weights(o.state_dict(), dev)
                ent[1] = key
            return ent[0]

    def reparameterize(self, mu, logvar):
        return ops.reparameterize(mu, logvar, torch.randn_like(mu))          # :389-399

    def sample(self, y, z=None, *, slot=0):
        """:427-447.  The latent is drawn with torch.randn(n, 32) on the CPU generator exactly as upstream (:441)
        unless ``z`` is given (parity tests pass it explicitly).  ``slot`` selects a private workspace (ClipPipeline lane)."""
        _eval_only(self)
        if z is None:
            z = torch.randn(*[y.shape[0], 32])
        return self.engine(y.device if y.is_cuda else None).sample(y, z.to(y.device), slot=slot)

    def forward(self, Input, y, eps=None):
        """:403-424.  eval(): running-statistics BatchNorm on the fused engine; train(): differentiable HIP operators with
        batch statistics (emotiongestures_amd/train/nets.py)."""
        if eps is None:
            eps = torch.randn(Input.shape[0], 32, device=Input.device)
        if self.training:
            self._refuse_replica_training()
            from .train import nets
            return nets.cvae_forward(self, Input, y, eps.to(Input.device))
        return self.engine(Input.device if Input.is_cuda else None).forward(Input, y, eps)
