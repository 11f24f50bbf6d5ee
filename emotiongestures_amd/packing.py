"""Host-side weight packing: reference ``state_dict`` -> the library's weight arena.

The C library publishes the arena layout (``eg_*_weight_entry``, include/emogest.h EgPackKind); this
module fills it on the CPU, once per ``load_state_dict``, and the result is uploaded as one fp32
device buffer.  One-time plumbing, not on the per-batch path.
"""
from __future__ import annotations

import ctypes as C
from typing import Mapping

import numpy as np
import torch

from . import _lib as L

BN_EPS = 1e-5       # torch.nn.BatchNorm{1,2}d default, which the reference keeps


def _t(sd: Mapping[str, torch.Tensor], key: str) -> torch.Tensor:
    if "|" in key:                      # several tensors concatenated along dim 0 (fused projections)
        return torch.cat([_t(sd, k) for k in key.split("|")], 0)
    if key not in sd:
        raise KeyError(f"state_dict is missing '{key}' required by the HIP weight manifest")
    return sd[key].detach().to("cpu", torch.float32)


def _bf16_split_bits(w: torch.Tensor):
    """hi = bf16_rne(w), lo = bf16_rne(w - hi): the same split the kernels apply to activations."""
    hi = w.to(torch.bfloat16)
    lo = (w - hi.to(torch.float32)).to(torch.bfloat16)
    return hi.view(torch.int16).numpy(), lo.view(torch.int16).numpy()


def _with_bf16_images(f32_img: np.ndarray, hi_bits: np.ndarray, lo_bits: np.ndarray) -> np.ndarray:
    out = np.concatenate([f32_img.reshape(-1).view(np.int16), hi_bits.reshape(-1), lo_bits.reshape(-1)])
    return out.view(np.float32)


def _pack_linear(w: torch.Tensor, npad: int, kpad: int) -> np.ndarray:
    """fp32 image [npad][kpad] row-major, then bf16 hi / lo images tile-planar [npad/64][kpad/8][64 rows][8]
    (one K-step's 8 octets of a 64-row tile are 8 contiguous 1-KiB pieces for global_load_lds)."""
    n, k = w.shape
    assert npad % 64 == 0 and kpad % 64 == 0
    p = torch.zeros(npad, kpad, dtype=torch.float32)
    p[:n, :k] = w
    t = p.reshape(npad // 64, 64, kpad // 8, 8).permute(0, 2, 1, 3).contiguous()
    hi, lo = _bf16_split_bits(t)
    return _with_bf16_images(p.numpy(), hi, lo)


def _pack_conv3x3(w: torch.Tensor, opad: int) -> np.ndarray:
    o, i = w.shape[:2]
    p = torch.zeros(opad, i, 3, 3, dtype=torch.float32)
    p[:o] = w
    t = p.permute(2, 3, 1, 0).reshape(9, i, opad)                        # [tap][ci][co]
    f32 = t.reshape(9, i // 4, 4, opad).permute(0, 1, 3, 2).contiguous()    # [tap][ci/4][co][4]
    oct_ = t.reshape(9, i // 8, 8, opad).permute(0, 1, 3, 2).contiguous()   # [tap][ci/8][co][8]
    hi, lo = _bf16_split_bits(oct_)
    return _with_bf16_images(f32.numpy(), hi, lo)


def _bn_affine(sd, prefix: str):
    scale = _t(sd, prefix + ".weight") / torch.sqrt(_t(sd, prefix + ".running_var") + BN_EPS)
    shift = _t(sd, prefix + ".bias") - _t(sd, prefix + ".running_mean") * scale
    return scale, shift


def _pad1(v: torch.Tensor, npad: int) -> np.ndarray:
    out = np.zeros(npad, dtype=np.float32)
    out[: v.numel()] = v.reshape(-1).numpy()
    return out


def _fold_chain(sd, chain: str):
    """"last@...@first": y = W_last(... (W_first x + b_first) ...) + b_last as one affine map, folded in float64."""
    names = chain.split("@")[::-1]                  # application order
    w = _t(sd, names[0] + ".weight").double()
    b = _t(sd, names[0] + ".bias").double()
    for nme in names[1:]:
        wi, bi = _t(sd, nme + ".weight").double(), _t(sd, nme + ".bias").double()
        w, b = wi @ w, wi @ b + bi
    return w, b


def _fold(sd, key: str):
    parts = [_fold_chain(sd, c) for c in key.split("|")]
    return torch.cat([p[0] for p in parts], 0).float(), torch.cat([p[1] for p in parts], 0).float()


def pack_entry(sd: Mapping[str, torch.Tensor], e: L.EgWeightEntry) -> np.ndarray:
    key, kind, d = e.key.decode(), e.kind, list(e.dims)
    if kind in (L.PACK_RAW, L.PACK_CONV1D):
        out = _t(sd, key).reshape(-1).numpy()
    elif kind == L.PACK_LINEAR:
        out = _pack_linear(_t(sd, key), d[2], d[3])
    elif kind == L.PACK_VEC_PAD:
        out = _pad1(_t(sd, key), d[1])
    elif kind == L.PACK_CONV3X3:
        out = _pack_conv3x3(_t(sd, key), d[2])
    elif kind == L.PACK_BN_SCALE:
        out = _pad1(_bn_affine(sd, key)[0], d[1])
    elif kind == L.PACK_BN_SHIFT:
        out = _pad1(_bn_affine(sd, key)[1], d[1])
    elif kind == L.PACK_CONV1X1:
        w = _t(sd, key)
        out = w.reshape(w.shape[0], w.shape[1]).t().contiguous().reshape(-1).numpy()
    elif kind == L.PACK_STEM:
        w = _t(sd, key)
        out = w.reshape(w.shape[0], 9).t().contiguous().reshape(-1).numpy()
    elif kind == L.PACK_WN_TAP:
        g, v = _t(sd, key + ".weight_g"), _t(sd, key + ".weight_v")
        w = v * (g / v.flatten(1).norm(dim=1).view(-1, 1, 1))          # torch weight_norm, dim=0 (tcn.py:19)
        npad = (d[0] + 63) // 64 * 64
        out = _pack_linear(w[:, :, d[2]].contiguous(), npad, d[3])
    elif kind == L.PACK_POS_TABLE:
        out = _t(sd, key)[0, : d[0], :].contiguous().reshape(-1).numpy()
    elif kind == L.PACK_LINEAR_T:
        out = _t(sd, key).t().contiguous().reshape(-1).numpy()
    elif kind == L.PACK_LINEAR_FOLD:
        out = _pack_linear(_fold(sd, key)[0], d[2], d[3])
    elif kind == L.PACK_BIAS_FOLD:
        out = _pad1(_fold(sd, key)[1], d[1])
    else:
        raise L.EgError(f"unknown pack kind {kind} for '{key}'")
    out = np.ascontiguousarray(out, dtype=np.float32)
    if out.size != e.numel:
        raise L.EgError(f"packing '{key}' (kind {kind}) produced {out.size} floats, manifest says {e.numel}")
    return out


def manifest(handle, num_fn, entry_fn):
    lib = L.load()
    n = getattr(lib, num_fn)(handle)
    out = []
    for i in range(n):
        e = L.EgWeightEntry()
        L.check(getattr(lib, entry_fn)(handle, i, C.byref(e)), entry_fn)
        out.append(e)
    return out


def build_arena(sd: Mapping[str, torch.Tensor], entries, total_floats: int) -> torch.Tensor:
    """CPU fp32 arena (pinned if possible) filled per the manifest."""
    arena = np.zeros(int(total_floats), dtype=np.float32)
    for e in entries:
        arena[e.offset: e.offset + e.numel] = pack_entry(sd, e)
    return torch.from_numpy(arena)


def strip_module_prefix(sd: Mapping[str, torch.Tensor]):
    """The reference's loaders strip the DataParallel prefix
    (test_emotion_gesture_diversity_iterative.py:149,159,168)."""
    return {(k[7:] if k.startswith("module.") else k): v for k, v in sd.items()}
