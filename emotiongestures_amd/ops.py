"""Tensor-level wrappers of the block operators in include/emogest.h.

Used by the module-level host mirrors and by the per-kernel parity tests.  They take GPU torch
tensors, pack weights into the library's layouts where needed, call the C ABI on the current HIP
stream and return GPU tensors.  No arithmetic of the path is done in torch here.
"""
from __future__ import annotations

import ctypes as C
from typing import Optional

import numpy as np
import torch

from . import _lib as L
from . import packing
from .engine import _need_cuda, _ptr, _stream


def _dev(t):
    return t.device


def pack_linear_weight(w: torch.Tensor, device) -> torch.Tensor:
    """nn.Linear weight [N,K] -> EG_PACK_LINEAR image on `device` (rows and K padded to 64)."""
    n, k = w.shape
    npad, kpad = (n + 63) // 64 * 64, (k + 63) // 64 * 64
    if w.is_cuda and w.device == torch.device(device) and w.dtype == torch.float32:
        # already resident: build the image on the device (bit-identical to the host packer, tests/test_gpu_training.py; the host path costs
        # seconds for EmotionNet's 65536 x 4096 layer, e.g. at every validation after a training step)
        lib = L.load()
        wd = w.detach().contiguous()
        img = torch.empty(int(lib.eg_linear_packed_floats(n, k)), dtype=torch.float32, device=wd.device)
        L.check(lib.eg_pack_linear_device(_ptr(wd), k, n, k, 0, _ptr(img), _stream(wd.device)), "eg_pack_linear_device")
        return img, npad, kpad
    return torch.from_numpy(packing._pack_linear(w.detach().cpu().float(), npad, kpad)).to(device), npad, kpad


def pack_conv3x3_weight(w: torch.Tensor, device):
    opad = (w.shape[0] + 15) // 16 * 16
    if w.is_cuda and w.device == torch.device(device) and w.dtype == torch.float32 and w.shape[1] % 8 == 0:
        lib = L.load()
        wd = w.detach().contiguous()
        img = torch.empty(int(lib.eg_conv3x3_packed_floats(wd.shape[1], opad)), dtype=torch.float32, device=wd.device)
        L.check(lib.eg_pack_conv3x3_device(_ptr(wd), wd.shape[0], wd.shape[1], 0, _ptr(img), _stream(wd.device)), "eg_pack_conv3x3_device")
        return img, opad
    return torch.from_numpy(packing._pack_conv3x3(w.detach().cpu().float(), opad)).to(device), opad


def _padvec(v: Optional[torch.Tensor], npad: int, device, fill=0.0):
    if v is None:
        return None
    out = torch.full((npad,), fill, dtype=torch.float32, device=device)
    out[: v.numel()] = v.detach().to(device, torch.float32).reshape(-1)
    return out


def conv3x3(x_nhwc, weight_oihw, bias=None, scale=None, shift=None, stride=1, relu=False, nchw_out=False, want_gap=False,
            precision="f32", packed=None):
    """Conv2d 3x3 pad 1 + fused epilogue (Full_model/ResNetBlocks.py:12,14).  x NHWC [B,H,W,Cin].
    ``packed`` = (weight image, padded bias, padded scale, padded shift) from conv3x3_pack(): skips the per-call host packing."""
    lib = L.load()
    x = _need_cuda(x_nhwc, "x")
    dev = x.device
    B, H, W, Cin = x.shape
    Cout = weight_oihw.shape[0]
    if packed is not None:
        wp, bias_p, scale_p, shift_p = packed
    else:
        wp, bias_p, scale_p, shift_p = conv3x3_pack(weight_oihw, bias, scale, shift, dev)
    Ho, Wo = (H - 1) // stride + 1, (W - 1) // stride + 1
    y = torch.empty((B, Cout, Ho * Wo) if nchw_out else (B, Ho, Wo, Cout), device=dev)
    gap = None
    tiles = lib.eg_conv3x3_gap_tiles(H, W, Cin, Cout, stride)
    if want_gap:
        gap = torch.empty(B, tiles, Cout, device=dev)
    L.check(lib.eg_conv3x3(_ptr(x), _ptr(wp), _ptr(bias_p), _ptr(scale_p), _ptr(shift_p), _ptr(y), _ptr(gap), B, H, W, Cin, Cout,
                           stride, int(relu), int(nchw_out), L.precision_code(precision), _stream(dev)), "eg_conv3x3")
    return (y, gap) if want_gap else y


def conv3x3_pack(weight_oihw, bias, scale, shift, device):
    """Device-resident operands of conv3x3 (weight images + channel vectors padded to the packed channel count)."""
    wp, opad = pack_conv3x3_weight(weight_oihw, device)
    return wp, _padvec(bias, opad, device), _padvec(scale, opad, device), _padvec(shift, opad, device)


def stem_conv(x, weight, bias, scale, shift):
    lib = L.load()
    x = _need_cuda(x, "x")
    dev = x.device
    B, H, W = x.shape
    Cc = weight.shape[0]
    w9 = weight.detach().reshape(Cc, 9).t().contiguous().to(dev, torch.float32)
    y = torch.empty(B, H, W, Cc, device=dev)
    L.check(lib.eg_stem_conv(_ptr(x), _ptr(w9), _ptr(_need_cuda(bias, "bias")), _ptr(_need_cuda(scale, "scale")),
                             _ptr(_need_cuda(shift, "shift")), _ptr(y), B, H, W, Cc, _stream(dev)), "eg_stem_conv")
    return y


def se_gate(gap_partial, w1, b1, w2, b2, hw: int):
    lib = L.load()
    g = _need_cuda(gap_partial, "gap")
    dev = g.device
    B, tiles, Cc = g.shape
    gate = torch.empty(B, Cc, device=dev)
    L.check(lib.eg_se_gate(_ptr(g), tiles, _ptr(_need_cuda(w1, "w1")), _ptr(_need_cuda(b1, "b1")), _ptr(_need_cuda(w2, "w2")),
                           _ptr(_need_cuda(b2, "b2")), _ptr(gate), B, Cc, hw, _stream(dev)), "eg_se_gate")
    return gate


def se_residual_relu(y, gate, x_in, ds_weight=None, ds_scale=None, ds_shift=None, stride=1):
    lib = L.load()
    y, gate, x_in = _need_cuda(y, "y"), _need_cuda(gate, "gate"), _need_cuda(x_in, "x_in")
    dev = y.device
    B, Ho, Wo, Cc = y.shape
    _, Hi, Wi, Cin = x_in.shape
    dsw = None
    if ds_weight is not None:
        dsw = ds_weight.detach().reshape(Cc, Cin).t().contiguous().to(dev, torch.float32)
        ds_scale, ds_shift = _need_cuda(ds_scale, "ds_scale"), _need_cuda(ds_shift, "ds_shift")
    out = torch.empty_like(y)
    L.check(lib.eg_se_residual_relu(_ptr(y), _ptr(gate), _ptr(x_in), _ptr(dsw), _ptr(ds_scale), _ptr(ds_shift), _ptr(out), B, Ho, Wo,
                                    Cc, Hi, Wi, Cin, stride, _stream(dev)), "eg_se_residual_relu")
    return out


def linear(x, weight, bias=None, res1=None, res2=None, relu=False, a_shift=0, a_seq=0, precision="f32", packed=None):
    """y = epi(x @ weight.T): x [M,K] (row-major, K%4==0), weight nn.Linear [N,K]."""
    lib = L.load()
    x = _need_cuda(x, "x")
    dev = x.device
    M, K = x.shape
    N = weight.shape[0]
    wp, npad, kpad = packed if packed is not None else pack_linear_weight(weight, dev)
    bias_p = _padvec(bias, npad, dev)
    y = torch.empty(M, N, device=dev)
    r1 = _need_cuda(res1, "res1") if res1 is not None else None
    r2 = _need_cuda(res2, "res2") if res2 is not None else None
    L.check(lib.eg_linear(_ptr(x), K, _ptr(wp), kpad, _ptr(bias_p), _ptr(r1), _ptr(r2), N, _ptr(y), N, M, N, K, int(relu), a_shift,
                          a_seq, L.precision_code(precision), _stream(dev)), "eg_linear")
    return y


def linear_splitk(x, weight, bias=None, relu=False, splits=8, precision="f32", packed=None):
    lib = L.load()
    x = _need_cuda(x, "x")
    dev = x.device
    M, K = x.shape
    N = weight.shape[0]
    wp, npad, kpad = packed if packed is not None else pack_linear_weight(weight, dev)
    bias_p = _padvec(bias, npad, dev)
    y = torch.empty(M, N, device=dev)
    part = torch.empty(splits + 1, M, N, device=dev)
    L.check(lib.eg_linear_splitk(_ptr(x), K, _ptr(wp), kpad, _ptr(bias_p), _ptr(y), N, M, N, K, int(relu), splits, _ptr(part),
                                 L.precision_code(precision), _stream(dev)), "eg_linear_splitk")
    return y


def layernorm(x, gamma, beta, eps=1e-6):
    lib = L.load()
    x = _need_cuda(x, "x")
    dev = x.device
    rows, d = x.reshape(-1, x.shape[-1]).shape
    y = torch.empty_like(x)
    L.check(lib.eg_layernorm(_ptr(x), _ptr(_need_cuda(gamma, "gamma")), _ptr(_need_cuda(beta, "beta")), _ptr(y), rows, d, eps,
                             _stream(dev)), "eg_layernorm")
    return y


def _attention_mask_bytes(mask, B, Lq, Lk, dev):
    """The reference's mask ([B, Lq, Lk], [B, 1, Lk] or anything broadcastable to one of them; `mask == 0` = masked, Modules.py:18-19) as the
    contiguous uint8 array eg_attention_masked reads: returns (bytes [B, 1 or Lq, Lk], batch stride, query stride)."""
    m = mask
    if m.dim() == 4:                      # already unsqueezed for the head axis (SubLayers.py:44-45)
        if m.shape[1] != 1:
            raise NotImplementedError("attention mask: per-head masks are not used by the reference (the head axis is broadcast)")
        m = m[:, 0]
    if m.dim() == 2:
        m = m[:, None, :]
    rows = Lq if m.shape[1] != 1 else 1
    m = (m != 0).expand(B, rows, Lk).to(device=dev, dtype=torch.uint8).contiguous()
    return m, rows * Lk, (Lk if rows > 1 else 0)


def attention(q, k, v, heads: int, want_attn=False, precision="f32", mask=None):
    """q [B,Lq,H*64], k/v [B,Lk,H*64] -> out [B,Lq,H*64] (Full_model/Modules.py:13-23); both products on MFMA in `precision`.
    mask: the reference's optional mask (masked_fill(mask == 0, -1e9) before the softmax)."""
    lib = L.load()
    q, k, v = _need_cuda(q, "q"), _need_cuda(k, "k"), _need_cuda(v, "v")
    dev = q.device
    B, Lq, D = q.shape
    Lk = k.shape[1]
    out = torch.empty_like(q)
    attn = torch.empty(B, heads, Lq, Lk, device=dev) if want_attn else None
    if mask is not None:
        mb, sb, sq = _attention_mask_bytes(mask, B, Lq, Lk, dev)
        L.check(lib.eg_attention_masked(_ptr(q), D, _ptr(k), D, _ptr(v), D, _ptr(mb), sb, sq, _ptr(out), D, _ptr(attn), B, heads, Lq, Lk,
                                        D // heads, L.precision_code(precision), _stream(dev)), "eg_attention_masked")
        return (out, attn) if want_attn else out
    L.check(lib.eg_attention(_ptr(q), D, _ptr(k), D, _ptr(v), D, _ptr(out), D, _ptr(attn), B, heads, Lq, Lk, D // heads,
                             L.precision_code(precision), _stream(dev)), "eg_attention")
    return (out, attn) if want_attn else out


def multi_head_attention(xq, xkv, wq, wk, wv, wo, ln_g, ln_b, heads: int, precision="f32", want_attn=True, packed=None):
    """``packed`` = the four pack_linear_weight(...)[0] images of (wq, wk, wv, wo): skips the per-call host packing."""
    lib = L.load()
    xq, xkv = _need_cuda(xq, "q"), _need_cuda(xkv, "kv")
    dev = xq.device
    B, Lq, D = xq.shape
    Lk = xkv.shape[1]
    packs = packed if packed is not None else [pack_linear_weight(w, dev)[0] for w in (wq, wk, wv, wo)]
    nbytes = lib.eg_mha_workspace_bytes(B, Lq, Lk, D, heads)
    ws = torch.empty(int(nbytes), dtype=torch.uint8, device=dev)
    out = torch.empty_like(xq)
    attn = torch.empty(B, heads, Lq, Lk, device=dev) if want_attn else None
    L.check(lib.eg_multi_head_attention(_ptr(xq), _ptr(xkv), *[_ptr(p) for p in packs], _ptr(_need_cuda(ln_g, "g")),
                                        _ptr(_need_cuda(ln_b, "b")), _ptr(out), _ptr(attn), B, Lq, Lk, D, heads,
                                        L.precision_code(precision), _ptr(ws), nbytes, _stream(dev)), "eg_multi_head_attention")
    return out, attn


def positionwise_ffn(x, w1, b1, w2, b2, ln_g, ln_b, precision="f32", packed=None):
    lib = L.load()
    x = _need_cuda(x, "x")
    dev = x.device
    d = x.shape[-1]
    rows = x.numel() // d
    di = w1.shape[0]
    p1, p2 = packed if packed is not None else (pack_linear_weight(w1, dev)[0], pack_linear_weight(w2, dev)[0])
    nbytes = lib.eg_ffn_workspace_bytes(rows, d, di)
    ws = torch.empty(int(nbytes), dtype=torch.uint8, device=dev)
    out = torch.empty_like(x)
    L.check(lib.eg_positionwise_ffn(_ptr(x), _ptr(p1), _ptr(_need_cuda(b1, "b1")), _ptr(p2), _ptr(_need_cuda(b2, "b2")),
                                    _ptr(_need_cuda(ln_g, "g")), _ptr(_need_cuda(ln_b, "b")), _ptr(out), rows, d, di,
                                    L.precision_code(precision), _ptr(ws), nbytes, _stream(dev)), "eg_positionwise_ffn")
    return out


def pack_tcn_weights(levels, device):
    """levels: list of (v1, g1, b1, v2, g2, b2) weight-norm tensors per TemporalBlock (tcn.py:18-24)."""
    chunks = []
    for (v1, g1, b1, v2, g2, b2) in levels:
        for v, g, b in ((v1, g1, b1), (v2, g2, b2)):
            v, g = v.detach().cpu().float(), g.detach().cpu().float()
            w = v * (g / v.flatten(1).norm(dim=1).view(-1, 1, 1))
            c = w.shape[0]
            npad, cpad = (c + 63) // 64 * 64, (c + 63) // 64 * 64
            for tap in range(2):
                chunks.append(packing._pack_linear(w[:, :, tap].contiguous(), npad, cpad))
            bb = np.zeros(npad, np.float32)
            bb[:c] = b.detach().cpu().float().numpy()
            chunks.append(bb)
    return torch.from_numpy(np.concatenate(chunks)).to(device)


def tcn_forward(x_blc, packed_w, levels: int, precision="f32"):
    """x [B, L, C] channels-last -> [B, L, C] (Full_model/tcn.py:63)."""
    lib = L.load()
    x = _need_cuda(x_blc, "x")
    dev = x.device
    B, Ln, Cc = x.shape
    cpad = (Cc + 63) // 64 * 64
    xp = torch.zeros(B, Ln, cpad, device=dev)
    xp[:, :, :Cc] = x
    y = torch.zeros(B, Ln, cpad, device=dev)
    ws = torch.empty(4 * B * Ln * cpad, dtype=torch.float32, device=dev)
    L.check(lib.eg_tcn_forward(_ptr(xp), _ptr(packed_w), _ptr(y), B, Ln, Cc, levels, L.precision_code(precision), _ptr(ws),
                               ws.numel() * 4, _stream(dev)), "eg_tcn_forward")
    return y[:, :, :Cc].contiguous()


def reparameterize(mu, logvar, eps):
    lib = L.load()
    mu, logvar, eps = _need_cuda(mu, "mu"), _need_cuda(logvar, "logvar"), _need_cuda(eps, "eps")
    z = torch.empty_like(mu)
    L.check(lib.eg_reparameterize(_ptr(mu), _ptr(logvar), _ptr(eps), _ptr(z), mu.numel(), _stream(mu.device)), "eg_reparameterize")
    return z


def sigmoid(a):
    """torch.sigmoid on the library's elementwise kernel (eg_elementwise op 6)."""
    lib = L.load()
    a = _need_cuda(a, "a").contiguous()
    out = torch.empty_like(a)
    L.check(lib.eg_elementwise(_ptr(a), None, _ptr(out), a.numel(), 6, 0.0, _stream(a.device)), "eg_elementwise")
    return out


def add_rows(a, table, period=0):
    """a [.., rows, d] + table[row % period] (period 0: plain add)."""
    lib = L.load()
    a, table = _need_cuda(a, "a"), _need_cuda(table, "table")
    d = a.shape[-1]
    out = torch.empty_like(a)
    L.check(lib.eg_add_rows(_ptr(a), _ptr(table), _ptr(out), a.numel() // d, d, period, _stream(a.device)), "eg_add_rows")
    return out


def conv1d(x, weight, bias, stride=1, padding=0, leaky=False, scale=None, shift=None):
    """nn.Conv1d on [n, cin, l] (contiguous) with an optional LeakyReLU(0.2) [+ per-channel affine] epilogue (eg_conv1d)."""
    lib = L.load()
    x = _need_cuda(x, "x").contiguous()
    dev = x.device
    n, cin, lin = x.shape
    cout, _, k = weight.shape
    lout = (lin + 2 * padding - k) // stride + 1
    y = torch.empty(n, cout, lout, device=dev)
    w = weight.detach().float().contiguous().to(dev)
    b = (bias.detach().float() if bias is not None else torch.zeros(cout)).contiguous().to(dev)
    sc = scale.detach().float().contiguous().to(dev) if scale is not None else None
    sh = shift.detach().float().contiguous().to(dev) if shift is not None else None
    L.check(lib.eg_conv1d(_ptr(x), _ptr(w), _ptr(b), _ptr(sc), _ptr(sh), _ptr(y), n, cin, cout, lin, k, stride, padding, int(leaky),
                          _stream(dev)), "eg_conv1d")
    return y
