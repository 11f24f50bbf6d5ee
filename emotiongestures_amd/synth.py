"""Platform-exact synthetic weights and inputs.

There is no checkpoint and no dataset for this path (the reference's are private:
test_emotion_gesture_diversity_iterative.py:149,159,168,337-338), so every test, the golden
generator, ``bench.py`` and ``smoke()`` regenerate weights and inputs from the same
integer-hash stream.  Only uint32/uint64 integer arithmetic and a handful of exactly-rounded
fp32 multiplies are used, so the values are bit-identical on every machine (no libm, no torch RNG).

The value at flat index ``i`` of the tensor called ``key`` is::

    h = mix32(fnv1a32(key) ^ seed*0x9E3779B9 ^ (i * 0x85EBCA6B))        (uint32)
    u = (h >> 8) * 2**-24                                              in [0, 1), exact in fp32
    v = lo + u * (hi - lo)                                             fp32
"""
from __future__ import annotations

import math
from typing import Dict, Iterable, Mapping, Optional, Tuple

import numpy as np

__all__ = [
    "hash_uniform", "synth_tensor_for_key", "synth_state_dict", "synth_inputs", "synth_audio",
    "EMOTIONS",
]

# train_audio_classifier_K_fold.py:65
EMOTIONS = ["neutral", "happiness", "anger", "sadness", "contempt", "surprise", "fear", "disgust"]

_M32 = np.uint64(0xFFFFFFFF)


def _fnv1a32(text: str) -> int:
    h = 0x811C9DC5
    for b in text.encode("utf-8"):
        h ^= b
        h = (h * 0x01000193) & 0xFFFFFFFF
    return h


def _mix32(x: np.ndarray) -> np.ndarray:
    """lowbias32-style avalanche on a uint64 array holding uint32 values."""
    x = x & _M32
    x ^= x >> np.uint64(16)
    x = (x * np.uint64(0x7FEB352D)) & _M32
    x ^= x >> np.uint64(15)
    x = (x * np.uint64(0x846CA68B)) & _M32
    x ^= x >> np.uint64(16)
    return x


def hash_unit(key: str, n: int, seed: int = 0) -> np.ndarray:
    """n fp32 values in [0,1), deterministic in (key, seed)."""
    base = np.uint64((_fnv1a32(key) ^ ((seed * 0x9E3779B9) & 0xFFFFFFFF)) & 0xFFFFFFFF)
    idx = np.arange(n, dtype=np.uint64)
    h = _mix32(base ^ ((idx * np.uint64(0x85EBCA6B)) & _M32))
    h = _mix32(h + np.uint64(0x6A09E667))
    return ((h >> np.uint64(8)).astype(np.float32) * np.float32(2.0 ** -24)).astype(np.float32)


def hash_uniform(key: str, shape: Iterable[int], lo: float, hi: float, seed: int = 0) -> np.ndarray:
    shape = tuple(int(s) for s in shape)
    n = int(np.prod(shape)) if shape else 1
    u = hash_unit(key, n, seed)
    v = np.float32(lo) + u * np.float32(hi - lo)
    return v.astype(np.float32).reshape(shape)


def _fans(shape: Tuple[int, ...]) -> Tuple[int, int]:
    # torch.nn.init._calculate_fan_in_and_fan_out semantics (what the reference's
    # xavier_uniform_ sweep sees, Full_model/Models_spatial_memory.py:557-559)
    rf = 1
    for s in shape[2:]:
        rf *= s
    return shape[1] * rf, shape[0] * rf


def synth_tensor_for_key(key: str, shape: Tuple[int, ...], seed: int = 0) -> Optional[np.ndarray]:
    """Value for one ``state_dict`` entry, chosen from the key's role.  Returns None for
    entries that keep their constructor value (sinusoid tables, step counters)."""
    leaf = key.rsplit(".", 1)[-1]
    if leaf in ("pos_table", "pos_table2", "num_batches_tracked"):
        return None
    if leaf == "running_var":
        return hash_uniform(key, shape, 0.5, 1.5, seed)
    if leaf == "running_mean":
        return hash_uniform(key, shape, -0.1, 0.1, seed)
    if leaf == "weight_g":                       # weight-norm gain (Full_model/tcn.py:19)
        return hash_uniform(key, shape, 0.7, 1.3, seed)
    if leaf == "weight_v":
        return hash_uniform(key, shape, -1.0, 1.0, seed)
    if len(shape) <= 1:
        if leaf == "weight":                     # BatchNorm / LayerNorm scale
            return hash_uniform(key, shape, 0.9, 1.1, seed)
        return hash_uniform(key, shape, -0.05, 0.05, seed)   # biases
    if "embedding" in key or "embeddings" in key:
        return hash_uniform(key, shape, -0.5, 0.5, seed)
    fan_in, fan_out = _fans(tuple(shape))
    bound = math.sqrt(6.0 / float(fan_in + fan_out))
    return hash_uniform(key, shape, -bound, bound, seed)


def synth_state_dict(shapes: Mapping[str, Tuple[int, ...]], seed: int = 0) -> Dict[str, np.ndarray]:
    """``shapes``: key -> shape (e.g. ``{k: tuple(v.shape) for k, v in module.state_dict().items()}``)."""
    out = {}
    for k, shp in shapes.items():
        v = synth_tensor_for_key(k, tuple(shp), seed)
        if v is not None:
            out[k] = v
    return out


def load_synth_weights(module, seed: int = 0):
    """Fill a torch module (reference class or our mirror: same keys) in place."""
    import torch

    sd = module.state_dict()
    new = synth_state_dict({k: tuple(v.shape) for k, v in sd.items()}, seed)
    with torch.no_grad():
        for k, v in new.items():
            sd[k].copy_(torch.from_numpy(v).to(sd[k].dtype))
    return module


def synth_audio(batch: int, n_samples: int = 64000, seed: int = 0) -> np.ndarray:
    """Speech-like 16 kHz audio: per-clip harmonic stack with 4 Hz AM plus -30 dB noise
    (BASELINE.md §3).  sin() is evaluated in float64 and rounded to fp32: the mel oracle and
    the HIP mel kernel both consume this fp32 array, so libm ulp differences between
    machines only perturb the *input*, never the parity comparison."""
    t = np.arange(n_samples, dtype=np.float64) / 16000.0
    out = np.empty((batch, n_samples), dtype=np.float32)
    for b in range(batch):
        u = hash_unit(f"audio/{b}", 16, seed).astype(np.float64)
        f0 = 90.0 + 160.0 * u[0]
        sig = np.zeros_like(t)
        for hnum in range(1, 6):
            sig += (0.6 ** hnum) * np.sin(2 * np.pi * f0 * hnum * t + 6.28 * u[hnum])
        am = 0.55 + 0.45 * np.sin(2 * np.pi * 4.0 * t + 6.28 * u[7])
        noise = hash_uniform(f"audio/noise/{b}", (n_samples,), -1.0, 1.0, seed).astype(np.float64)
        sig = 0.3 * am * sig + 0.0316 * noise
        out[b] = sig.astype(np.float32)
    return out


def synth_inputs(batch: int, frames: int = 34, pose_dim: int = 126, prior_frames: int = 4,
                 spec_len: int = 124, n_mels: int = 128, text_len: int = 60, n_words: int = 200,
                 d_model: int = 512, seed: int = 0) -> Dict[str, np.ndarray]:
    """Generator inputs with the layout of test_emotion_gesture_diversity_iterative.py:193-205.

    spec: fp16-rounded dB values in [-80, 0] (data_loader/lmdb_loader_BEAT_full.py:229,242);
    text: word indices; pre_pose: +-0.5; label: one-hot over 8 emotions; z: CVAE latent in
    [-2, 2] (explicit, so no RNG has to match); sampled: a stand-in emotion feature map."""
    spec = hash_uniform("in/spec", (batch, n_mels, spec_len), -80.0, 0.0, seed)
    spec = spec.astype(np.float16).astype(np.float32)
    text = (hash_unit("in/text", batch * text_len, seed) * np.float32(n_words)).astype(np.int64)
    text = np.minimum(text, n_words - 1).reshape(batch, text_len)
    pre_pose = hash_uniform("in/pre_pose", (batch, prior_frames, pose_dim), -0.5, 0.5, seed)
    lab = (hash_unit("in/label", batch, seed) * np.float32(8)).astype(np.int64) % 8
    onehot = np.zeros((batch, 8), dtype=np.float32)
    onehot[np.arange(batch), lab] = 1.0
    z = hash_uniform("in/z", (batch, 32), -2.0, 2.0, seed)
    sampled = hash_uniform("in/sampled", (batch, frames, d_model), -1.0, 1.0, seed)
    return {"spec": spec, "text": text, "pre_pose": pre_pose, "label": onehot, "z": z,
            "sampled": sampled}


def digest(x, n: int = 2048) -> Dict[str, np.ndarray]:
    """Compact fingerprint of a tensor for golden files: a strided sample of <= n elements
    plus two global statistics (float64 accumulation)."""
    a = np.ascontiguousarray(np.asarray(x, dtype=np.float32)).reshape(-1)
    stride = max(1, a.size // n)
    return {"sample": a[::stride][:n].copy(), "mean": np.float64(a.astype(np.float64).mean()),
            "absmean": np.float64(np.abs(a.astype(np.float64)).mean()),
            "shape": np.asarray(x.shape, dtype=np.int64)}


def synth_clip(seed: int = 7, duration: float = 11.3, fps_in: int = 30, joints: int = 47, n_words: int = 24, start_time: float = 3.0) -> dict:
    """One upstream-shaped clip dict (data_loader/data_preprocessor_expressive.py:70-77): skeleton [T, joints, 3] at fps_in,
    raw 16 kHz audio, a synthetic whole-clip fp16 dB spectrogram [128, 1 + n // 512] and timed words (absolute times)."""
    n_skel = int(duration * fps_in)
    skel = (hash_unit("clip.skeletons", n_skel * joints * 3, seed) * 2 - 1).astype(np.float32).reshape(n_skel, joints, 3)
    n_audio = int(duration * 16000)
    audio = ((hash_unit("clip.audio_raw", n_audio, seed) * 2 - 1) * 0.3).astype(np.float32)
    n_spec = 1 + n_audio // 512
    spec = (-80.0 * hash_unit("clip.audio_feat", 128 * n_spec, seed)).astype(np.float16).reshape(128, n_spec)
    starts = np.sort(hash_unit("clip.words", n_words, seed).astype(np.float64) * duration)
    words = [["w%d" % i, float(start_time + s), float(start_time + min(duration, s + 0.25))] for i, s in enumerate(starts)]
    return {"skeletons": skel, "audio_feat": spec, "audio_raw": audio, "words": words, "start_frame_no": 100,
            "end_frame_no": 100 + n_skel, "start_time": start_time, "end_time": start_time + duration}
