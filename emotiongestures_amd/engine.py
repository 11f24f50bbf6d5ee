"""Device-side engines: a packed weight arena + workspace + one C-ABI call per batch.

PyTorch is used here only for device memory (arena / workspace / outputs come from its caching
allocator) and for the current HIP stream; every FLOP of the path runs in libemogest_hip.so.
"""
from __future__ import annotations

import ctypes as C
import functools
import threading
from typing import Dict, Mapping, Optional

import torch

from . import _lib as L
from . import packing


def _ptr(t: Optional[torch.Tensor]):
    return None if t is None else C.c_void_p(t.data_ptr())


def _stream(device) -> C.c_void_p:
    return C.c_void_p(torch.cuda.current_stream(device).cuda_stream)


def _need_cuda(t: torch.Tensor, name: str, dtype=torch.float32) -> torch.Tensor:
    if not t.is_cuda:
        raise L.EgError(f"{name}: the HIP path needs a GPU tensor (got {t.device}); there is no CPU fallback")
    if t.dtype != dtype:
        t = t.to(dtype)
    return t.contiguous()


def _locked(fn):
    """Serialise the host-side enqueue of one engine: a call carves the engine's workspace and enqueues ~200 kernels on the current stream;
    two threads on one device (nn.DataParallel replicas, device_ids with repeats) must not interleave theirs."""
    @functools.wraps(fn)
    def wrapper(self, *a, **k):
        with self._lock:
            return fn(self, *a, **k)
    return wrapper


class GeneratorEngine:
    """Host handle for eg_generator_* (Transformer.forward, Full_model/Models_spatial_memory.py:566-616)."""

    def __init__(self, *, frames=34, pose_dim=126, prior_frames=4, chunk=4, d_model=512, d_inner=2048, n_layers=3,
                 n_head=8, d_k=64, n_mels=128, spec_len=124, text_len=60, n_words=200, embed_dim=300, tcn_hidden=300,
                 tcn_layers=3, variant="spatial", precision="f32", n_position=60, keep_taps=False, concurrent=False, fold_affine=False, fuse_se=True,
                 shared_chip=False):
        lib = L.load()
        cfg = L.EgGeneratorConfig()
        L.check(lib.eg_generator_default_config(C.byref(cfg)), "eg_generator_default_config")
        cfg.frames, cfg.pose_dim, cfg.prior_frames, cfg.chunk = frames, pose_dim, prior_frames, chunk
        cfg.d_model, cfg.d_inner, cfg.n_layers, cfg.n_head, cfg.d_k = d_model, d_inner, n_layers, n_head, d_k
        cfg.n_mels, cfg.spec_len, cfg.text_len, cfg.n_words = n_mels, spec_len, text_len, n_words
        cfg.embed_dim, cfg.tcn_hidden, cfg.tcn_layers = embed_dim, tcn_hidden, tcn_layers
        cfg.variant = {"spatial": 0, "memory": 1}[variant] if isinstance(variant, str) else int(variant)
        cfg.precision = L.precision_code(precision)
        cfg.n_position = max(n_position, frames)
        cfg.reserved[0] = 1 if keep_taps else 0
        cfg.reserved[1] = 1 if concurrent else 0
        cfg.reserved[2] = 1 if fold_affine else 0
        cfg.reserved[3] = 0 if fuse_se else 1
        cfg.reserved[4] = 1 if shared_chip else 0     # several batches in flight (ClipPipeline): GEMM tiles chosen for CU time, not latency
        self.cfg = cfg
        h = C.c_void_p()
        L.check(lib.eg_generator_create(C.byref(cfg), C.byref(h)), "eg_generator_create")
        self._h = h
        self._lib = lib
        self.entries = packing.manifest(h, "eg_generator_num_weights", "eg_generator_weight_entry")
        self.arena_floats = lib.eg_generator_arena_floats(h)
        self.arena: Optional[torch.Tensor] = None
        self._ws: Dict[tuple, torch.Tensor] = {}
        self._lock = threading.RLock()
        self.uploads = 0                 # arena packs + uploads so far (tests: once per device and weight version)

    def __del__(self):
        try:
            if getattr(self, "_h", None):
                self._lib.eg_generator_destroy(self._h)
                self._h = None
        except Exception:
            pass

    # ---- weights ----
    def load_weights(self, sd: Mapping[str, torch.Tensor], device) -> None:
        cpu = packing.build_arena(packing.strip_module_prefix(sd), self.entries, self.arena_floats)
        with self._lock:
            self.arena = cpu.to(device)
            self._ws.clear()
            self.uploads += 1

    def _workspace(self, key, nbytes: int, device) -> torch.Tensor:
        ws = self._ws.get(key)
        if ws is None or ws.numel() < nbytes or ws.device != torch.device(device):
            ws = torch.empty(int(nbytes), dtype=torch.uint8, device=device)
            self._ws[key] = ws
        return ws

    # ---- Transformer.forward ----
    @_locked
    def forward(self, spec, text, prior, sampled=None, want_aux=True, slot=0):
        if self.arena is None:
            raise L.EgError("GeneratorEngine.forward before load_weights")
        dev = self.arena.device
        c = self.cfg
        spec = _need_cuda(spec, "input_spectrum")
        text = _need_cuda(text, "text", torch.int64)
        prior = _need_cuda(prior, "prior_seq")
        B = spec.shape[0]
        if tuple(spec.shape) != (B, c.n_mels, c.spec_len):
            raise L.EgError(f"input_spectrum shape {tuple(spec.shape)} != (B,{c.n_mels},{c.spec_len})")
        if tuple(text.shape) != (B, c.text_len):
            raise L.EgError(f"text shape {tuple(text.shape)} != (B,{c.text_len})")
        if tuple(prior.shape) != (B, c.prior_frames, c.pose_dim):
            raise L.EgError(f"prior_seq shape {tuple(prior.shape)} != (B,{c.prior_frames},{c.pose_dim})")
        if sampled is not None:
            sampled = _need_cuda(sampled, "sampled_emotion_feature")
            if tuple(sampled.shape) != (B, c.frames, c.d_model):
                raise L.EgError(f"sampled_emotion_feature shape {tuple(sampled.shape)} != (B,{c.frames},{c.d_model})")
        ws_bytes = self._lib.eg_generator_workspace_bytes(self._h, B)
        ws = self._workspace(("fwd", B) if slot == 0 else ("fwd", B, slot), ws_bytes, dev)   # one workspace per concurrent slot
        pose = torch.empty(B, c.frames, c.pose_dim, device=dev)
        emo = torch.empty(B, c.frames, c.d_model, device=dev) if want_aux else None
        sem = torch.empty(B, c.frames, c.d_model, device=dev) if want_aux else None
        pred = torch.empty(B, 8, device=dev) if want_aux else None
        txt = torch.empty(B, c.text_len, 512, device=dev) if want_aux else None
        L.check(self._lib.eg_generator_forward(self._h, _ptr(self.arena), B, _ptr(spec), _ptr(text), _ptr(prior), _ptr(sampled),
                                               _ptr(pose), _ptr(emo), _ptr(sem), _ptr(pred), _ptr(txt), _ptr(ws), ws_bytes,
                                               _stream(dev)), "eg_generator_forward")
        return pose, emo, sem, pred, txt

    @_locked
    def forward_draws(self, spec, prior, sampled, slot=0):
        """BASELINE config 5: sampled [B, R, frames, d_model] -> pose [B, R, frames, pose_dim].  slot: a private workspace (one per step in flight)."""
        dev = self.arena.device
        c = self.cfg
        spec, prior, sampled = _need_cuda(spec, "spec"), _need_cuda(prior, "prior"), _need_cuda(sampled, "sampled")
        B, R = sampled.shape[0], sampled.shape[1]
        ws_bytes = self._lib.eg_generator_draws_workspace_bytes(self._h, B, R)
        ws = self._workspace(("draws", B, R) if slot == 0 else ("draws", B, R, slot), ws_bytes, dev)
        pose = torch.empty(B, R, c.frames, c.pose_dim, device=dev)
        L.check(self._lib.eg_generator_forward_draws(self._h, _ptr(self.arena), B, R, _ptr(spec), _ptr(prior), _ptr(sampled),
                                                     _ptr(pose), _ptr(ws), ws_bytes, _stream(dev)), "eg_generator_forward_draws")
        return pose

    def tap(self, name: str, batch: int) -> torch.Tensor:
        """Copy of an intermediate of the last forward(batch) (parity tests)."""
        ws = self._ws[("fwd", batch)]
        p, n = C.c_void_p(), C.c_int64()
        L.check(self._lib.eg_generator_tap(self._h, batch, _ptr(ws), name.encode(), C.byref(p), C.byref(n)), "eg_generator_tap")
        off = p.value - ws.data_ptr()
        return ws[off: off + 4 * n.value].view(torch.float32).clone()


class CvaeEngine:
    """Host handle for eg_cvae_* (MLP_Reconstruct_v3, CAVE/BEAT_CVAE.py:312-460)."""

    def __init__(self, frames=60, d_model=512):
        lib = L.load()
        cfg = L.EgCvaeConfig()
        L.check(lib.eg_cvae_default_config(C.byref(cfg)), "eg_cvae_default_config")
        cfg.frames, cfg.d_model = frames, d_model
        self.cfg = cfg
        h = C.c_void_p()
        L.check(lib.eg_cvae_create(C.byref(cfg), C.byref(h)), "eg_cvae_create")
        self._h, self._lib = h, lib
        self.entries = packing.manifest(h, "eg_cvae_num_weights", "eg_cvae_weight_entry")
        self.arena_floats = lib.eg_cvae_arena_floats(h)
        self.arena = None
        self._ws = {}
        self._lock = threading.RLock()
        self.uploads = 0

    def __del__(self):
        try:
            if getattr(self, "_h", None):
                self._lib.eg_cvae_destroy(self._h)
                self._h = None
        except Exception:
            pass

    def load_weights(self, sd, device):
        with self._lock:
            self.arena = packing.build_arena(packing.strip_module_prefix(sd), self.entries, self.arena_floats).to(device)
            self._ws.clear()
            self.uploads += 1

    def _workspace(self, n, device, slot=0):
        nbytes = self._lib.eg_cvae_workspace_bytes(self._h, n)
        ws = self._ws.get((n, slot))                # one workspace per concurrent slot (ClipPipeline lane)
        if ws is None or ws.device != torch.device(device):
            ws = torch.empty(int(nbytes), dtype=torch.uint8, device=device)
            self._ws[(n, slot)] = ws
        return ws, nbytes

    @_locked
    def sample(self, y, z, slot=0):
        dev = self.arena.device
        y, z = _need_cuda(y, "y"), _need_cuda(z, "z")
        n = y.shape[0]
        ws, nbytes = self._workspace(n, dev, slot)
        out = torch.empty(n, self.cfg.frames, self.cfg.d_model, device=dev)
        L.check(self._lib.eg_cvae_sample(self._h, _ptr(self.arena), n, _ptr(y), _ptr(z), _ptr(out), _ptr(ws), nbytes,
                                         _stream(dev)), "eg_cvae_sample")
        return out

    @_locked
    def forward(self, x, y, eps):
        dev = self.arena.device
        x, y, eps = _need_cuda(x, "x"), _need_cuda(y, "y"), _need_cuda(eps, "eps")
        n = x.shape[0]
        ws, nbytes = self._workspace(n, dev)
        rec = torch.empty(n, self.cfg.frames, self.cfg.d_model, device=dev)
        mu, logvar = torch.empty(n, 32, device=dev), torch.empty(n, 32, device=dev)
        L.check(self._lib.eg_cvae_forward(self._h, _ptr(self.arena), n, _ptr(x), _ptr(y), _ptr(eps), _ptr(rec), _ptr(mu),
                                          _ptr(logvar), _ptr(ws), nbytes, _stream(dev)), "eg_cvae_forward")
        return rec, mu, logvar


class MelFrontEnd:
    """extract_melspectrogram on the GPU (utils/train_utils_BEAT.py:186-190)."""

    def __init__(self, device):
        import numpy as np
        lib = L.load()
        fb, win, tw = np.zeros(513 * 128, np.float32), np.zeros(1024, np.float32), np.zeros(1024, np.float32)
        band = np.zeros(256, np.int32)
        L.check(lib.eg_mel_tables(fb.ctypes.data_as(C.c_void_p), win.ctypes.data_as(C.c_void_p), tw.ctypes.data_as(C.c_void_p),
                                  band.ctypes.data_as(C.c_void_p)), "eg_mel_tables")
        self.fb, self.win, self.tw, self.band = (torch.from_numpy(a).to(device) for a in (fb, win, tw, band))
        self._lib, self.device, self._ws = lib, torch.device(device), {}

    def __call__(self, audio: torch.Tensor, out_frames: Optional[int] = None, slot: int = 0) -> torch.Tensor:
        audio = _need_cuda(audio, "audio")
        B, n = audio.shape
        n_frames = 1 + n // 512
        out_frames = n_frames if out_frames is None else out_frames
        nbytes = self._lib.eg_mel_workspace_bytes(B, n)
        ws = self._ws.get((B, n, slot))             # one workspace per concurrent slot (ClipPipeline lane)
        if ws is None:
            ws = torch.empty(int(nbytes), dtype=torch.uint8, device=self.device)
            self._ws[(B, n, slot)] = ws
        spec = torch.empty(B, 128, out_frames, device=self.device)
        L.check(self._lib.eg_melspectrogram(_ptr(audio), B, n, _ptr(self.fb), _ptr(self.win), _ptr(self.tw), _ptr(self.band), _ptr(spec), out_frames,
                                            _ptr(ws), nbytes, _stream(self.device)), "eg_melspectrogram")
        return spec
