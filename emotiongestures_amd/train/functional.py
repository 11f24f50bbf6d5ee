"""Differentiable operators of the training path: every forward AND backward is a HIP kernel of libemogest_hip.so.

torch.autograd only sequences the backward (the reference trains through autograd too: train_audio_classifier_K_fold.py:155-175);
it performs no arithmetic here: fan-outs go through `fork` (its backward adds on the HIP elementwise kernel), losses scale their
own gradients.  torch is used for device memory and for pure data movement (zero fill, strided copies, permutes of weights).
Everything is fp32 (EG_PREC_F32: v_mfma_f32_16x16x4_f32) -- gradient parity with the reference's fp32 autograd is the bar
(tests/test_gpu_training.py, tolerance 1e-4 per-parameter relative L2).  No CPU fallback: CPU tensors raise.
"""
from __future__ import annotations

import ctypes as C
from typing import Optional

import torch

from .. import _lib as L
from ..engine import _ptr, _stream

F32 = L.EG_PREC_F32
_WS = {}


_PREC = {"conv": F32, "gemm": F32}


def set_precision(name: str) -> None:
    """Arithmetic of the convolutions (forward, input and weight gradient) and of the Linear products (forward, input gradient) of the
    training path: "f32" (fp32 MFMA; the gradient-parity configuration) or "bf16x3" (3-term split-bf16 MFMA with fp32 accumulation)."""
    if name not in ("f32", "bf16x3"):
        raise ValueError(f"train precision {name!r}: expected 'f32' or 'bf16x3'")
    _PREC["conv"] = _PREC["gemm"] = F32 if name == "f32" else L.EG_PREC_BF16X3


def get_precision() -> str:
    return "f32" if _PREC["conv"] == F32 else "bf16x3"


class precision:
    """`with functional.precision("bf16x3"): ...` -- the arithmetic of the training operators for a block, restored on exit also when the block
    raises.  The training layer keeps its configuration (precision, scratch buffers, registered weight images, the dropout stream, the cut
    context of a segmented step) per PROCESS: one trainer per process, as the one-process-per-GPU design has it.  It cannot be per thread:
    autograd runs the backward operators on its own device thread, which must see the forward's configuration.  (The C ABI underneath is
    stateless and re-entrant; this note is about the Python layer above it.)"""

    def __init__(self, name: str):
        self.name, self.prev = name, None

    def __enter__(self):
        self.prev = get_precision()
        set_precision(self.name)
        return self

    def __exit__(self, *exc):
        set_precision(self.prev)
        return False


def reset_state() -> None:
    """Drop every piece of per-process state of the training layer: fp32 precision, no registered weight images, no scratch buffers,
    host-side dropout stream at seed 0 (device epoch off), no pending BatchNorm counters.  For `finally:` blocks of drivers that train
    several models in one process (bench.py's legs, the K-fold loop)."""
    set_precision("f32")
    register_weight_images(None)
    _WS.clear()
    _DROP["epoch"] = None
    _DROP["log"] = None
    manual_seed(0)
    _PENDING_COUNTERS.clear()


def _lib():
    return L.load()


def _chk(t: torch.Tensor, name="tensor") -> torch.Tensor:
    if not t.is_cuda:
        raise L.EgError(f"{name}: the training path runs only on a GPU (got {t.device}); there is no CPU fallback")
    t = t.detach()
    if t.dtype != torch.float32:
        t = t.float()
    return t.contiguous()


_CONST = {}


def _const_vec(dev, n: int, value: float) -> torch.Tensor:
    key = (str(dev), n, value)
    t = _CONST.get(key)
    if t is None:
        t = _CONST[key] = torch.full((n,), value, dtype=torch.float32, device=dev)
    return t


def grad_out(param, shape=None) -> torch.Tensor:
    """Where a backward kernel writes the gradient of `param`: its slice of the flat gradient buffer (optim.flatten_parameters) when it has
    one that nothing wrote yet this step -- autograd then adopts that view as .grad and the collection copy disappears -- else fresh memory
    (first use wins; a second use of a shared parameter is accumulated by autograd as usual)."""
    slot = getattr(param, "_eg_slot", None)
    if slot is not None and id(param) not in param._eg_fp.written:
        param._eg_fp.written.add(id(param))
        return slot.view(shape if shape is not None else slot.shape)        # a new view object: autograd may take it over without cloning
    return torch.empty(shape if shape is not None else param.shape, dtype=torch.float32, device=param.device)


def _scratch(dev, floats: int, tag="ws") -> torch.Tensor:
    """Reusable scratch (partials of split-K / column reductions), one buffer per (device, tag, STREAM): use on a stream is ordered by the
    stream; two branches of a step running on different streams (bench.py: the emotion CVAE beside the generator's backward) never share one."""
    key = (str(dev), tag, torch.cuda.current_stream(dev).cuda_stream if torch.device(dev).type == "cuda" else 0)
    buf = _WS.get(key)
    if buf is None or buf.numel() < floats:
        buf = torch.empty(max(int(floats), 1 << 16), dtype=torch.float32, device=dev)
        _WS[key] = buf
    return buf


def _pad_cols(t: torch.Tensor, mult=4) -> torch.Tensor:
    """[R, K] -> [R, Kp] zero padded to a multiple of `mult` (pure data movement, one launch)."""
    k = t.shape[1]
    kp = (k + mult - 1) // mult * mult
    if kp == k:
        return t
    if not t.is_cuda:
        raise L.EgError("_pad_cols: the HIP path needs a GPU tensor")
    t = t.contiguous()
    out = torch.empty(t.shape[0], kp, dtype=torch.float32, device=t.device)
    L.check(_lib().eg_pad_cols(_ptr(t), _ptr(out), t.shape[0], k, kp, _stream(t.device)), "eg_pad_cols")
    return out


# ---- raw (non-autograd) kernels ---------------------------------------------------------------------------------------------
_IMAGES = {"reg": []}


class _ImageRegistry:
    """Every registered optim.WeightImages (one per flattened model: generator, discriminator, ...), searched in registration order."""

    def __init__(self, sets):
        self.sets = sets

    def lookup(self, w, kind, flag):
        f32_now = _PREC["gemm" if kind == 0 else "conv"] == F32
        for reg in self.sets:
            if f32_now and getattr(reg, "bf16_only", False):
                continue                # images refreshed without their fp32 head (WeightImages built in a split-bf16 mode): the fp32 kernels pack per use
            hit = reg.lookup(w, kind, flag)
            if hit is not None:
                return hit
        return None


def register_weight_images(images) -> None:
    """Add the optim.WeightImages of a model being trained (several models may be registered at once: the adversarial setup trains a
    generator and a Motion_Discriminator side by side); None clears the registry (every use then packs its own image)."""
    if images is None:
        _IMAGES["reg"] = []
    elif all(images is not r for r in _IMAGES["reg"]):
        _IMAGES["reg"] = _IMAGES["reg"] + [images]


def unregister_weight_images(images) -> None:
    _IMAGES["reg"] = [r for r in _IMAGES["reg"] if r is not images]


def _image_registry():
    return _ImageRegistry(_IMAGES["reg"]) if _IMAGES["reg"] else None


def _pack_linear(w, transpose=False):
    """[N,K] fp32 (or its transpose) -> eg_linear's split-bf16 weight image; returns (image, ldw).  Resident image (refreshed once per
    optimiser step) when the weight is a registered parameter, else one launch."""
    lib = _lib()
    r, c = w.shape
    n, k = (c, r) if transpose else (r, c)
    reg = _image_registry()
    if reg is not None:
        img = reg.lookup(w, 0, int(transpose))
        if img is not None:
            return img, (k + 63) // 64 * 64
    img = torch.empty(int(lib.eg_linear_packed_floats(n, k)), dtype=torch.float32, device=w.device)
    L.check(lib.eg_pack_linear_device(_ptr(w), c, n, k, int(transpose), _ptr(img), _stream(w.device)), "eg_pack_linear_device")
    return img, (k + 63) // 64 * 64


def _linear_ex(x, lda, w, ldw, bias, res, y, M, N, K, relu, prec, gate=None, drop=None, splits=0, x_img=None, y_img=None):
    """One eg_linear_ex call (include/emogest.h: EgLinearArgs).  drop = (p, seed, offset, epoch tensor | None) of a Dropout site or None.
    x_img: X as pre-split bf16 images of width K (then `x` is not read); y_img: receives Y as images of width N."""
    a = L.EgLinearArgs()
    a.x, a.w, a.bias, a.res1, a.res2, a.y = (x.data_ptr() if x is not None else None), w.data_ptr(), (bias.data_ptr() if bias is not None else None), \
        (res.data_ptr() if res is not None else None), None, y.data_ptr()
    if x_img is not None:
        a.x_images, a.k_x = x_img.data_ptr(), K
    if y_img is not None:
        a.y_images, a.y_k = y_img.data_ptr(), N
    a.gate_src = gate.data_ptr() if gate is not None else None
    a.lda, a.ldw, a.ldr, a.ldc, a.ldg = lda, ldw, N, N, N
    a.m, a.n, a.k, a.relu, a.precision, a.splits = M, N, K, int(relu), prec, int(splits)
    a.drop_p = 0.0
    if drop is not None and drop[0] > 0.0:
        a.drop_p, a.drop_seed, a.drop_offset = float(drop[0]), int(drop[1]), int(drop[2])
        a.drop_epoch = drop[3].data_ptr() if drop[3] is not None else None
    part = None
    if splits >= 2:
        part = _scratch(y.device, splits * M * N, "splitk")
        a.partial = part.data_ptr()
    L.check(_lib().eg_linear_ex(C.byref(a), _stream(y.device)), "eg_linear_ex")
    return y


# From this many rows up the fused blocks chain their products through pre-split images.  OFF by default (measured, round 5): at 128 clips per step
# (4352 rows) the chained step takes 28.12 ms against 27.86 ms without -- stand-alone the pre-split product is only 4-17 us ahead of the fp32-input
# kernel per launch at these shapes (tools/bench_ops.py gemm: 14.0 / 29.6 / 39.8 / 40.6 us against 18.4 / 35.4 / 46.0 / 57.7 us for 4352 x {512, 1536, 2048} x 512
# and 4352 x 512 x 2048), and writing the images (LayerNorm outputs, the 2048-wide hidden and its gradient: ~36 MB per FFN each way) costs more than that.
PRESPLIT_ROWS = 1 << 30          # off (measured slower at every batch size tried); tests lower it to cover the image entry points


def presplit_ok(rows: int, k: int) -> bool:
    """Do the fused blocks hand activations of `rows` x `k` to their consumer as pre-split bf16 (hi, lo) images (the producing LayerNorm / GEMM epilogue
    writes them, the consuming product reads both operands by LDS-DMA)?  Split-bf16 arithmetic only, `rows >= PRESPLIT_ROWS`."""
    return _PREC["gemm"] != F32 and rows >= PRESPLIT_ROWS and k % 64 == 0 and k > 0


def new_images(rows: int, k: int, dev) -> torch.Tensor:
    """Storage of the bf16 (hi, lo) tile-planar images of a [rows, k] activation (eg_split_tiles layout): 4 bytes per element of the padded tile grid."""
    return torch.empty(((rows + 63) // 64) * 64 * k, dtype=torch.float32, device=dev)


def images_of(t):
    """The images a producing block attached to its output tensor (None: the consumer reads fp32)."""
    return getattr(t, "_eg_img", None)


def raw_linear(x, w, bias=None, relu=False, res=None, w_transposed=False, gate=None, drop=None, x_img=None, want_img=False):
    """y[M,N] = epi(x[M,K] w[N,K]^T) on the MFMA GEMM (eg_linear_ex): fp32, or split-bf16 under set_precision("bf16x3").
    epi: + bias; `gate` ([M,N]): ReLU backward (v = gate > 0 ? v : 0); `drop` = (p, seed, offset, epoch): nn.Dropout on the product from the
    counter hash; + res; relu.  w_transposed: w is [K,N] and the product is x w (the input gradient of a Linear).
    x_img: x as pre-split images (the pre-split product: split-bf16 only); want_img: also return y as images -> (y, y_img)."""
    if _PREC["gemm"] != F32:
        if x_img is None:
            x = _pad_cols(x)
        M, K = x.shape
        N = w.shape[1] if w_transposed else w.shape[0]
        wimg, ldw = _pack_linear(w, w_transposed)
        y = torch.empty(M, N, dtype=torch.float32, device=x.device)
        y_img = new_images(M, N, x.device) if want_img else None
        splits = 0
        if x_img is None and y_img is None:
            if M <= 128 and K >= 4096:
                splits = min(64, K // 1024)
            elif K >= 1024:
                # few output tiles and a deep K (the FFN's second product and the first one's input gradient at 16 clips: 544 x 512 x 2048 = 72 tiles
                # on 256 CUs, 64 serial K-steps, 29.7 us): split K until ~256 workgroups exist, fold with the epilogue pass
                tiles = ((M + 63) // 64) * ((N + 63) // 64)
                if tiles <= 96:
                    splits = min(K // 512, max(2, 256 // tiles))
        _linear_ex(x, K, wimg, ldw, bias, res, y, M, N, K, relu, _PREC["gemm"], gate, drop, splits, x_img, y_img)
        return (y, y_img) if want_img else y
    if w_transposed:
        w = raw_transpose(w)
    x, w = _pad_cols(x), _pad_cols(w)
    M, K = x.shape
    N = w.shape[0]
    y = torch.empty(M, N, dtype=torch.float32, device=x.device)
    splits = min(64, K // 1024) if (M <= 128 and K >= 4096) else 0      # short and deep (emotion_classifer_header.0: K = frames * d_model)
    _linear_ex(x, K, w, K, bias, res, y, M, N, K, relu, F32, gate, drop, splits)
    return (y, None) if want_img else y


def raw_linear_img(x, w, bias=None, relu=False, res=None, w_transposed=False, gate=None, drop=None, x_img=None, want_img=False):
    """raw_linear that ALWAYS returns (y, images of y or None)."""
    out = raw_linear(x, w, bias, relu, res, w_transposed, gate, drop, x_img, want_img)
    return out if want_img else (out, None)


def raw_transpose(x):
    lib = _lib()
    r, c = x.shape
    y = torch.empty(c, r, dtype=torch.float32, device=x.device)
    L.check(lib.eg_transpose(_ptr(x), c, r, c, _ptr(y), r, _stream(x.device)), "eg_transpose")
    return y


def raw_gemm_tn(a, b, out=None, accumulate=False):
    """c[m,n] = sum_k a[k,m] b[k,n]"""
    lib = _lib()
    k, m = a.shape
    n = b.shape[1]
    c = out if out is not None else torch.empty(m, n, dtype=torch.float32, device=a.device)
    need = max(lib.eg_gemm_tn_workspace_floats(m, n, k), m * n if accumulate else 0)
    ws = _scratch(a.device, need, "tn") if need else None
    L.check(lib.eg_gemm_tn(_ptr(a), m, _ptr(b), n, _ptr(c), n, m, n, k, _ptr(ws), ws.numel() if ws is not None else 0, int(accumulate),
                           _stream(a.device)), "eg_gemm_tn")
    return c


def raw_colsum(a, b=None, want0=True, want1=False, out0=None, out1=None):
    lib = _lib()
    rows, c = a.shape
    ws = _scratch(a.device, lib.eg_colreduce_workspace_floats(c), "col")
    o0 = (out0 if out0 is not None else torch.empty(c, dtype=torch.float32, device=a.device)) if want0 else None
    o1 = (out1 if out1 is not None else torch.empty(c, dtype=torch.float32, device=a.device)) if want1 else None
    L.check(lib.eg_colsum(_ptr(a), _ptr(b), _ptr(o0), _ptr(o1), rows, c, _ptr(ws), _stream(a.device)), "eg_colsum")
    return o0, o1


def raw_ew(op, a, b=None, s=0.0):
    lib = _lib()
    y = torch.empty_like(a)
    L.check(lib.eg_elementwise(_ptr(a), _ptr(b), _ptr(y), a.numel(), op, float(s), _stream(a.device)), "eg_elementwise")
    return y


EW_RELU, EW_RELU_BWD, EW_LEAKY, EW_LEAKY_BWD, EW_ADD, EW_SCALE, EW_SIGMOID, EW_SIGMOID_BWD, EW_MUL, EW_AXPY, EW_EXP, EW_SCALE_DEV = range(12)


def raw_linear_backward(x, w, dy, need_dx=True, w_param=None, b_param=None, want_db=True):
    """dx = dy w;  dw = dy^T x;  db = colsum(dy).  w_param / b_param: the parameters themselves (gradients go to their flat slices).
    Under set_precision("bf16x3") dw and db come from ONE launch of the split-bf16 MFMA kernel (csrc/lingrad.hip; + a fixed-order reduce when
    the rows are split); the fp32 configuration keeps the fp32 TN GEMM and the column sum."""
    in_place = w_param is not None and x.shape[1] == w.shape[1]
    if _PREC["gemm"] != F32:
        lib = _lib()
        R, N = dy.shape
        K = x.shape[1]
        dev = dy.device
        dw = grad_out(w_param) if in_place else torch.empty(N, K, dtype=torch.float32, device=dev)
        db = (grad_out(b_param) if b_param is not None else torch.empty(N, dtype=torch.float32, device=dev)) if want_db else None
        need = int(lib.eg_linear_wgrad_mfma_workspace_floats(R, N, K))
        ws = _scratch(dev, need, "tn") if need else None
        L.check(lib.eg_linear_wgrad_mfma(_ptr(dy), N, _ptr(x), K, _ptr(dw), K, _ptr(db), R, N, K, _ptr(ws), ws.numel() if ws is not None else 0,
                                         _stream(dev)), "eg_linear_wgrad_mfma")
    else:
        dw = raw_gemm_tn(dy, x, out=grad_out(w_param) if in_place else None)
        db = raw_colsum(dy, out0=grad_out(b_param) if b_param is not None else None)[0] if want_db else None
    dx = raw_linear(dy, w, w_transposed=True) if need_dx else None      # [M,N] x [N,K]
    return dx, dw, db


# ---- autograd Functions -------------------------------------------------------------------------------------------------------
class _Linear(torch.autograd.Function):
    @staticmethod
    def forward(ctx, x, w, b, relu):
        xs = x.shape
        x2 = _chk(x, "x").reshape(-1, xs[-1])
        wd = _chk(w, "weight")
        if wd.dim() == 4:           # a 1x1 convolution's [Cout, Cin, 1, 1] (conv1x1): the same memory as the [Cout, Cin] matrix
            wd = wd.view(wd.shape[0], wd.shape[1])
        y = raw_linear(x2, wd, _chk(b) if b is not None else None, relu)
        ctx.save_for_backward(x2, wd, y if relu else None)
        ctx.has_b, ctx.xs, ctx.need_dx = b is not None, xs, x.requires_grad
        ctx.params = (w, b)
        return y.view(*xs[:-1], wd.shape[0])

    @staticmethod
    def backward(ctx, dy):
        x2, w, y = ctx.saved_tensors
        dy2 = _chk(dy).reshape(-1, w.shape[0])
        if y is not None:
            dy2 = raw_ew(EW_RELU_BWD, dy2, y)
        dx, dw, db = raw_linear_backward(x2, w, dy2, ctx.need_dx, ctx.params[0], ctx.params[1] if ctx.has_b else None, want_db=ctx.has_b)
        return (dx.view(ctx.xs) if dx is not None else None), dw, (db if ctx.has_b else None), None


def linear(x, w, b=None, relu=False):
    """nn.Linear (+ fused ReLU)"""
    return _Linear.apply(x, w, b, relu)


class _Fork(torch.autograd.Function):
    @staticmethod
    def forward(ctx, x):
        ctx.set_materialize_grads(False)            # an unused branch hands None back, not a zero-filled map
        return x.view_as(x), x.view_as(x)

    @staticmethod
    def backward(ctx, ga, gb):
        if ga is None:
            return gb
        if gb is None:
            return ga
        return raw_ew(EW_ADD, _chk(ga), _chk(gb))


def fork(x):
    """Two uses of one tensor; the backward sums the two gradients on the HIP elementwise kernel (not torch's accumulate)."""
    a, b = _Fork.apply(x)
    img = images_of(x)
    if img is not None:             # both uses may read the row through the producer's images
        a._eg_img = b._eg_img = img
    return a, b


class _Add(torch.autograd.Function):
    @staticmethod
    def forward(ctx, a, b):
        return raw_ew(EW_ADD, _chk(a), _chk(b))

    @staticmethod
    def backward(ctx, g):
        return g, g


def add(a, b):
    return _Add.apply(a, b)


class _Act(torch.autograd.Function):
    @staticmethod
    def forward(ctx, x, slope):
        xd = _chk(x)
        ctx.save_for_backward(xd)
        ctx.slope = slope
        return raw_ew(EW_RELU, xd) if slope == 0.0 else raw_ew(EW_LEAKY, xd, None, slope)

    @staticmethod
    def backward(ctx, dy):
        (x,) = ctx.saved_tensors
        return (raw_ew(EW_RELU_BWD, _chk(dy), x) if ctx.slope == 0.0 else raw_ew(EW_LEAKY_BWD, _chk(dy), x, ctx.slope)), None


def relu(x):
    return _Act.apply(x, 0.0)


class _Sigmoid(torch.autograd.Function):
    @staticmethod
    def forward(ctx, x):
        y = raw_ew(EW_SIGMOID, _chk(x))
        ctx.save_for_backward(y)
        return y

    @staticmethod
    def backward(ctx, dy):
        (y,) = ctx.saved_tensors
        return raw_ew(EW_SIGMOID_BWD, _chk(dy), y)          # dy * y * (1 - y)


def sigmoid(x):
    return _Sigmoid.apply(x)


def leaky_relu(x, slope=0.2):
    return _Act.apply(x, float(slope))


_DROP = {"seed": 0, "offset": 0, "epoch": None}


def manual_seed(seed: int) -> None:
    """Seed of the dropout mask stream (counter-based; every dropout call advances the counter by its element count).  Also rewinds the
    device-resident epoch (if one is in use) to 0: the same seed then reproduces the same masks."""
    _DROP["seed"], _DROP["offset"] = int(seed) & 0xFFFFFFFF, 0
    if _DROP.get("epoch") is not None:
        _DROP["epoch"].zero_()


def use_device_dropout_epoch(dev) -> torch.Tensor:
    """Keep a per-step mask epoch on the device (an int32 counter mixed into the seed by the kernels).  `begin_dropout_step()` then rewinds
    the host-side offset and increments the counter with a launch, so a step captured into a hipGraph draws a fresh mask at every replay
    although its host scalars are frozen.  Returns the counter tensor."""
    ep = _DROP.get("epoch")
    if ep is None or ep.device != torch.device(dev):
        ep = _DROP["epoch"] = torch.zeros(1, dtype=torch.int32, device=dev)
    return ep


def begin_dropout_step() -> None:
    """Start of a training STEP in device-epoch mode (no-op otherwise): offsets restart at 0, the device epoch advances by one.
    Called once per optimiser step by the step drivers (train/graph.GraphedStep and SegmentedStep wrap it around the step they capture) --
    never by a network's forward: the dropout kernels read the epoch from device memory when they EXECUTE, so every forward and backward
    of one step must see the same value.  Two dropout-active forwards before one backward (a generator and a Motion_Discriminator, or the
    generator called twice) therefore share the step's epoch and draw distinct masks from growing offsets.  Without a driver (eager use of
    the device epoch) the offsets simply keep growing, which is equally collision-free."""
    ep = _DROP.get("epoch")
    if ep is not None:
        _DROP["offset"] = 0
        L.check(_lib().eg_counter_add(_ptr(ep), 1, _stream(ep.device)), "eg_counter_add")


def next_dropout_offset(numel: int) -> int:
    off = _DROP["offset"]
    _DROP["offset"] = off + (int(numel) + 1023) // 1024 * 1024
    log = _DROP.get("log")
    if log is not None:             # tests: the (offset, numel) of every Dropout site of a step, in call order (record_dropout_sites)
        log.append((off, int(numel)))
    return off


def record_dropout_sites(on: bool = True):
    """Start (-> the list that will fill) / stop recording the (stream offset, element count) of every Dropout site the forward visits.  The masks
    themselves are a pure function of (seed, offset + flat index, p): `dropout_mask` returns one, oracle.dropout_keep_mask restates it in numpy."""
    _DROP["log"] = [] if on else None
    return _DROP["log"]


def dropout_mask(p: float, seed: int, offset: int, numel: int, device) -> torch.Tensor:
    """keep / (1 - p) per element of the site at stream position `offset`: eg_dropout applied to ones (what every fused epilogue multiplies by)."""
    ones = torch.ones(int(numel), dtype=torch.float32, device=device)
    y = torch.empty_like(ones)
    L.check(_lib().eg_dropout_dev(_ptr(ones), _ptr(y), ones.numel(), float(p), int(seed) & 0xFFFFFFFF, int(offset), None, _stream(ones.device)), "eg_dropout")
    return y


class _Dropout(torch.autograd.Function):
    @staticmethod
    def forward(ctx, x, p, seed, offset, epoch):
        lib = _lib()
        xd = _chk(x)
        y = torch.empty_like(xd)
        L.check(lib.eg_dropout_dev(_ptr(xd), _ptr(y), xd.numel(), float(p), seed, offset, _ptr(epoch), _stream(xd.device)), "eg_dropout")
        ctx.cfg = (float(p), seed, offset, epoch)
        return y

    @staticmethod
    def backward(ctx, dy):
        lib = _lib()
        p, seed, offset, epoch = ctx.cfg
        d = _chk(dy)
        dx = torch.empty_like(d)
        L.check(lib.eg_dropout_dev(_ptr(d), _ptr(dx), d.numel(), p, seed, offset, _ptr(epoch), _stream(d.device)), "eg_dropout")
        return dx, None, None, None, None


def dropout(x, p: float):
    """nn.Dropout(p) in train() mode (p = 0: identity, no launch)."""
    if p <= 0.0:
        return x
    if x.is_cuda and torch.cuda.is_current_stream_capturing() and _DROP.get("epoch") is None:
        # the (seed, offset) pair is a host scalar: frozen into a captured graph, every replay would apply the SAME mask
        raise L.EgError("dropout inside a stream capture needs the device-resident mask epoch (functional.use_device_dropout_epoch); "
                        "train/graph.GraphedStep enables it when told the model trains with dropout (stochastic=True)")
    return _Dropout.apply(x, p, _DROP["seed"], next_dropout_offset(x.numel()), _DROP.get("epoch"))


def _pack_conv(w, flip=False):
    """OIHW -> the image eg_conv3x3 reads (fp32 + bf16 hi/lo), built on the device in one launch; flip: the dgrad filter."""
    lib = _lib()
    co, ci = w.shape[:2]
    cie, coe = (co, ci) if flip else (ci, co)
    reg = _image_registry()
    if reg is not None:
        hit = reg.lookup(w, 1, int(flip))
        if hit is not None:
            return hit
    img = torch.empty(int(lib.eg_conv3x3_packed_floats(cie, (coe + 15) // 16 * 16)), dtype=torch.float32, device=w.device)
    L.check(lib.eg_pack_conv3x3_device(_ptr(w), co, ci, int(flip), _ptr(img), _stream(w.device)), "eg_pack_conv3x3_device")
    return img


PAD_WGRAD = True            # False: ragged-width weight gradients on the fp32 implicit GEMM (A/B switch of the tests)
S2_DGRAD = __import__("os").environ.get("EG_S2_DGRAD", "1") != "0"      # False: stride-2 input gradients through the column product + col2im (A/B switch)
S2_WGRAD = __import__("os").environ.get("EG_S2_WGRAD", "1") != "0"      # False: stride-2 weight gradients on the fp32 implicit GEMM (A/B switch)


class _Conv3x3(torch.autograd.Function):
    """nn.Conv2d(k=3, pad=1, stride s) on NHWC activations; weight OIHW as in the reference's state_dict."""

    @staticmethod
    def forward(ctx, x, w, b, stride, relu, want_gap=False, defer_mask=False, passthrough=False, in_scale=None, in_shift=None, res_link=None):
        lib = _lib()
        ctx.set_materialize_grads(False)            # no zero-filled "gradients" for the non-differentiable pooling partials / an unused alias
        xd, wd = _chk(x, "x"), _chk(w, "weight")
        ctx.in_affine = None
        if in_scale is not None:
            # x is the PRE-normalisation map of a deferred BatchNorm (batch_norm(defer_apply=True)): its per-channel affine is applied while the
            # convolution (and later its weight-gradient kernel) stages the operand -- the normalised map never exists in memory
            if not (_PREC["conv"] != F32 and stride == 1 and xd.shape[-1] % 32 == 0 and wd.shape[0] % 32 == 0 and b is None and not relu):
                raise L.EgError("conv3x3: a deferred BatchNorm in front needs a split-bf16, stride-1, bias-free convolution with channels % 32 == 0")
            ctx.in_affine = (in_scale, in_shift)
        gap = None
        B, H, W, Ci = xd.shape
        Co = wd.shape[0]
        Ho, Wo = (H - 1) // stride + 1, (W - 1) // stride + 1
        dev = xd.device
        if Ci % 32 == 0:
            coutp = (Co + 15) // 16 * 16
            wp = _pack_conv(wd)
            prec = _PREC["conv"]
            bp = None
            if b is not None:
                bp = torch.zeros(coutp, dtype=torch.float32, device=dev)
                bp[:Co].copy_(_chk(b))
            if Co % 4 == 0:
                y = torch.empty(B, Ho, Wo, Co, dtype=torch.float32, device=dev)
                tiles = int(lib.eg_conv3x3_gap_tiles(H, W, Ci, Co, stride)) if want_gap else 0
                if ctx.in_affine is not None:
                    if want_gap:
                        if 256 % Co:
                            raise L.EgError("conv3x3: a deferred BatchNorm in front needs Cout to divide 256 for the pooling partials")
                        gap = torch.empty(2, B, tiles, Co, dtype=torch.float32, device=dev)
                    L.check(lib.eg_conv3x3_sq_in_affine(_ptr(xd), _ptr(in_scale), _ptr(in_shift), _ptr(wp), None, _ptr(y), _ptr(gap),
                                                        _ptr(gap[1]) if gap is not None else None, B, H, W, Ci, Co, 1, 0, prec, _stream(dev)),
                            "eg_conv3x3_sq_in_affine")
                elif want_gap and prec != F32 and 256 % Co == 0:
                    # split-bf16 modes: the epilogue also emits the per-tile sums of squares -- gap is then [2, B, tiles, Co] (plane 0 = the sums every
                    # consumer reads through the base pointer, plane 1 = the squares): train-mode BatchNorm needs no pass over y for its variance
                    gap = torch.empty(2, B, tiles, Co, dtype=torch.float32, device=dev)
                    L.check(lib.eg_conv3x3_sq(_ptr(xd), _ptr(wp), _ptr(bp), _ptr(y), _ptr(gap), _ptr(gap[1]), B, H, W, Ci, Co, stride, int(relu), prec,
                                              _stream(dev)), "eg_conv3x3_sq")
                else:
                    if want_gap:        # per-(clip, tile) channel sums of y from the conv epilogue: BatchNorm's mean / the SE pooling for free
                        gap = torch.empty(B, tiles, Co, dtype=torch.float32, device=dev)
                    L.check(lib.eg_conv3x3(_ptr(xd), _ptr(wp), _ptr(bp), None, None, _ptr(y), _ptr(gap), B, H, W, Ci, Co, stride, int(relu), 0, prec,
                                           _stream(dev)), "eg_conv3x3")
            else:           # ragged channel count (final_conv1: 128 -> frames): channel-major epilogue, then back to NHWC
                yc = torch.empty(B, Co, Ho * Wo, dtype=torch.float32, device=dev)
                L.check(lib.eg_conv3x3(_ptr(xd), _ptr(wp), _ptr(bp), None, None, _ptr(yc), None, B, H, W, Ci, Co, stride, int(relu), 1, prec,
                                       _stream(dev)), "eg_conv3x3")
                y = yc.view(B, Co, Ho, Wo).permute(0, 2, 3, 1).contiguous()
        elif Ci == 1 and stride == 1 and relu and b is not None and Co % 4 == 0 and Co <= 128 and 256 % (Co // 4) == 0:
            # the stem (ResNetSE34V2.py:64-66: conv -> ReLU, BatchNorm follows as its own operator): the inference stem kernel with an identity affine
            w9 = raw_transpose(wd.view(Co, 9))                                  # [9][Co] tap-major
            one, zero = _const_vec(dev, Co, 1.0), _const_vec(dev, Co, 0.0)
            y = torch.empty(B, Ho, Wo, Co, dtype=torch.float32, device=dev)
            L.check(lib.eg_stem_conv(_ptr(xd), _ptr(w9), _ptr(_chk(b)), _ptr(one), _ptr(zero), _ptr(y), B, H, W, Co, _stream(dev)), "eg_stem_conv")
        else:               # other thin inputs: im2col rows through the GEMM
            col = torch.empty(B * Ho * Wo, 9 * Ci, dtype=torch.float32, device=dev)
            L.check(lib.eg_im2col3x3(_ptr(xd), _ptr(col), B, H, W, Ci, stride, 0, _stream(dev)), "eg_im2col3x3")
            wm = wd.permute(0, 2, 3, 1).reshape(Co, 9 * Ci).contiguous()
            y = raw_linear(col, wm, _chk(b) if b is not None else None, relu).view(B, Ho, Wo, Co)
        if want_gap and gap is None:
            raise ValueError("conv3x3(want_gap=True): only the NHWC tower convolutions (Cin % 32 == 0, Cout % 4 == 0) emit pooling partials")
        ctx.save_for_backward(xd, wd, y if (relu and not defer_mask) else None)      # defer_mask: the consumer (batch_norm(relu_input=True)) applies it
        ctx.stride, ctx.has_b, ctx.need_dx = stride, b is not None, x.requires_grad
        ctx.params = (w, b)
        ctx.passthrough = bool(passthrough)
        ctx.pass_sub = passthrough == "sub"
        # res_link (a dict shared with the block's se_block_tail): the tail's backward leaves (dout, ReLU bits) there instead of writing the identity
        # shortcut's gradient dout * [out > 0] as a map; this backward masks dout in the input-gradient epilogue (eg_conv3x3_res_masked)
        ctx.res_link = res_link if (passthrough is True and stride == 1 and Ci == Co and Ci % 32 == 0) else None
        if res_link is not None and ctx.res_link is None:
            raise L.EgError("conv3x3(res_link=...): only with passthrough=True on a square stride-1 convolution with channels % 32 == 0")
        outs = (y,)
        if want_gap:
            ctx.mark_non_differentiable(gap)
            outs += (gap,)
        if ctx.pass_sub:
            # the second consumer is the stride-2 1x1 shortcut (ResNetSE34V2.py:43-47), which reads x[:, ::s, ::s] only: hand it that quarter map; its
            # gradient comes back on the quarter grid and the stride-2 input-gradient kernel adds it to the pixels it belongs to -- no zero-filled
            # full-resolution map is written and re-read
            xs = torch.empty(B, Ho, Wo, Ci, dtype=torch.float32, device=dev)
            L.check(lib.eg_subsample(_ptr(xd), _ptr(xs), B, H, W, Ci, stride, 0, _stream(dev)), "eg_subsample")
            outs += (xs,)
        elif passthrough:       # a second use of x (the block's residual branch): its gradient comes back into THIS backward and is added in the
            outs += (x.view_as(x),)                 # dgrad launch's epilogue instead of a separate map-sized add (fork)
        return outs if len(outs) > 1 else y

    @staticmethod
    def backward(ctx, dy, *rest):
        lib = _lib()
        dres = rest[-1] if ctx.passthrough else None
        x, w, y = ctx.saved_tensors
        B, H, W, Ci = x.shape
        Co = w.shape[0]
        lazy = ctx.res_link.pop("masked", None) if ctx.res_link is not None else None        # (dout, bits) left by the block's tail
        if lazy is not None and (dres is not None or dy is None):
            raise L.EgError("conv3x3: the masked shortcut gradient cannot be combined with another gradient of the alias / a missing dy")
        if dy is None:              # only the alias was used downstream
            if dres is not None and ctx.pass_sub:
                full = torch.empty_like(x)
                L.check(lib.eg_subsample(_ptr(_chk(dres)), _ptr(full), B, H, W, Ci, ctx.stride, 1, _stream(x.device)), "eg_subsample")
                dres = full
            return (_chk(dres) if dres is not None else None), None, None, None, None, None, None, None, None, None, None
        dyd = _chk(dy)
        if y is not None:
            dyd = raw_ew(EW_RELU_BWD, dyd, y)
        Ho, Wo = dyd.shape[1], dyd.shape[2]
        dev = x.device
        dy2 = dyd.view(B * Ho * Wo, Co)
        dw = grad_out(ctx.params[0], (Co, Ci, 3, 3))
        dwm = None
        if ctx.in_affine is not None:   # the deferred BatchNorm's affine re-applied while x is staged (forward checked the shape constraints)
            need = lib.eg_conv3x3_wgrad_mfma_workspace_floats(B, H, W, Ci, Co)
            ws = _scratch(dev, need, "wgrad")
            L.check(lib.eg_conv3x3_wgrad_mfma_oihw_in_affine(_ptr(x), _ptr(ctx.in_affine[0]), _ptr(ctx.in_affine[1]), _ptr(dy2), _ptr(dw), B, H, W, Ci, Co,
                                                             _ptr(ws), ws.numel(), _stream(dev)), "eg_conv3x3_wgrad_mfma (in-affine)")
        elif _PREC["conv"] != F32 and ctx.stride == 1 and Ci % 32 == 0 and Co % 32 == 0:       # split-bf16 MFMA weight gradient, written OIHW
            need = lib.eg_conv3x3_wgrad_mfma_workspace_floats(B, H, W, Ci, Co)                # straight into the flat gradient slice
            ws = _scratch(dev, need, "wgrad")
            L.check(lib.eg_conv3x3_wgrad_mfma_oihw(_ptr(x), _ptr(dy2), _ptr(dw), B, H, W, Ci, Co, _ptr(ws), ws.numel(), _stream(dev)),
                    "eg_conv3x3_wgrad_mfma")
        elif _PREC["conv"] != F32 and ctx.stride == 1 and Ci % 32 == 0 and PAD_WGRAD:
            # a ragged output width (final_conv1: 128 -> frames = 34 / 60 / 120): the same MFMA kernel on dy zero-padded to a multiple of 32 channels --
            # the padded rows of dW come out zero and are dropped (one pad pass over dy and a small copy instead of the fp32 implicit GEMM:
            # 274 -> ~60 us at 128 clips)
            Cop = (Co + 31) // 32 * 32
            dyp = _pad_cols(dy2, 32)
            dwp = torch.empty(Cop, Ci, 3, 3, dtype=torch.float32, device=dev)
            need = lib.eg_conv3x3_wgrad_mfma_workspace_floats(B, H, W, Ci, Cop)
            ws = _scratch(dev, need, "wgrad")
            L.check(lib.eg_conv3x3_wgrad_mfma_oihw(_ptr(x), _ptr(dyp), _ptr(dwp), B, H, W, Ci, Cop, _ptr(ws), ws.numel(), _stream(dev)),
                    "eg_conv3x3_wgrad_mfma")
            dw.copy_(dwp[:Co])                                              # data movement into the flat gradient slice
        elif _PREC["conv"] == L.EG_PREC_BF16X3 and ctx.stride == 2 and Ci % 4 == 0 and S2_WGRAD:
            # stride-2 entry convolution: dW = dY^T im2col(x) on the split-bf16 Linear weight-gradient kernel, the window gathered while x is staged
            # (was the fp32 implicit GEMM below: 263 us per layer at 128 clips)
            dwm = torch.empty(Co, 9 * Ci, dtype=torch.float32, device=dev)
            need = int(lib.eg_linear_wgrad_mfma_workspace_floats(B * Ho * Wo, Co, 9 * Ci))
            ws = _scratch(dev, need, "lingrad") if need else None
            L.check(lib.eg_conv3x3_wgrad_gather_mfma(_ptr(x), _ptr(dy2), _ptr(dwm), B, H, W, Ci, Co, ctx.stride, _ptr(ws), ws.numel() if ws is not None else 0,
                                                     _stream(dev)), "eg_conv3x3_wgrad_gather_mfma")
        elif Ci % 4 == 0:           # implicit GEMM over the output pixels (no im2col buffer)
            dwm = torch.empty(Co, 9 * Ci, dtype=torch.float32, device=dev)
            need = lib.eg_gemm_tn_workspace_floats(Co, 9 * Ci, B * Ho * Wo)
            ws = _scratch(dev, need, "tn") if need else None
            L.check(lib.eg_conv3x3_wgrad(_ptr(x), _ptr(dy2), _ptr(dwm), B, H, W, Ci, Co, ctx.stride, _ptr(ws), ws.numel() if ws is not None else 0,
                                         _stream(dev)), "eg_conv3x3_wgrad")
        else:                       # the stem (1 input channel)
            col = torch.empty(B * Ho * Wo, 9 * Ci, dtype=torch.float32, device=dev)
            L.check(lib.eg_im2col3x3(_ptr(x), _ptr(col), B, H, W, Ci, ctx.stride, 0, _stream(dev)), "eg_im2col3x3")
            dwm = raw_gemm_tn(dy2, col)                                     # [Co, (kh,kw,ci)]
        if dwm is not None:
            dw.copy_(dwm.view(Co, 3, 3, Ci).permute(0, 3, 1, 2))           # (kh, kw, ci) -> OIHW, straight into the flat gradient slice
        db = raw_colsum(dy2, out0=grad_out(ctx.params[1]))[0] if ctx.has_b else None
        dx = None
        if ctx.need_dx and ctx.stride == 1 and Ci == Co and Ci % 32 == 0:
            # input gradient of a square stride-1 conv = the forward kernel on the 180-degree rotated, transposed filter
            # (F.conv2d's dgrad); no 9x im2col intermediate
            wp = _pack_conv(w, flip=True)                                       # w'[ci][co][kh][kw] = w[co][ci][2-kh][2-kw]
            dx = torch.empty_like(x)
            res = _chk(dres) if dres is not None else None                      # the other consumer's gradient: added in the epilogue
            if lazy is not None:                                                # ... masked there from (dout, ReLU bits): never a map of its own
                L.check(lib.eg_conv3x3_res_masked(_ptr(dyd), _ptr(wp), _ptr(lazy[0]), _ptr(lazy[1]), _ptr(dx), B, H, W, Co, Ci, _PREC["conv"], _stream(dev)),
                        "eg_conv3x3_res_masked (dgrad)")
                lazy = None
            else:
                L.check(lib.eg_conv3x3_se(_ptr(dyd), _ptr(wp), None, None, None, None, _ptr(res), _ptr(dx), None, B, H, W, Co, Ci, 1, 0, 0, _PREC["conv"],
                                          _stream(dev)), "eg_conv3x3 (dgrad)")
            dres = None
        elif ctx.need_dx and _PREC["conv"] != F32 and ctx.stride == 1 and Ci == 128 and Co <= 64 and PAD_WGRAD:
            # final_conv1 (128 -> frames): the same rotated-filter convolution on dy and the filter zero-padded to 64 output channels, instead of
            # the [pixels, 9 * 128] column product + col2im (585 MB written and re-read per 128-clip step)
            dyp = _pad_cols(dy2, 64).view(B, Ho, Wo, 64)
            wpad = torch.zeros(64, Ci, 3, 3, dtype=torch.float32, device=dev)
            wpad[:Co].copy_(w)
            wp = _pack_conv(wpad, flip=True)
            dx = torch.empty_like(x)
            res = _chk(dres) if dres is not None else None
            L.check(lib.eg_conv3x3_se(_ptr(dyp), _ptr(wp), None, None, None, None, _ptr(res), _ptr(dx), None, B, H, W, 64, Ci, 1, 0, 0, _PREC["conv"],
                                      _stream(dev)), "eg_conv3x3 (dgrad, padded)")
            dres = None
        elif ctx.need_dx and ctx.stride == 2 and _PREC["conv"] == L.EG_PREC_BF16X3 and S2_DGRAD and (Ci, Co) in ((32, 64), (64, 128), (128, 256)):
            # stride-2 entry convolution of a stage: the phase-decomposed MFMA input gradient (9 tap products per four dx pixels), the shortcut's
            # quarter-grid gradient added in its epilogue -- instead of the [pixels, 9 Ci] column product + col2im (+ the zero-filled scatter)
            wp = _pack_conv(w, flip=True)
            dx = torch.empty_like(x)
            rq = _chk(dres) if (dres is not None and ctx.pass_sub) else None
            L.check(lib.eg_conv3x3_dgrad_s2(_ptr(dyd), _ptr(wp), _ptr(rq), _ptr(dx), B, H, W, Ci, Co, _PREC["conv"], _stream(dev)), "eg_conv3x3_dgrad_s2")
            if rq is not None:
                dres = None
        elif ctx.need_dx:
            wmat_t = w.permute(2, 3, 1, 0).reshape(9 * Ci, Co).contiguous()   # [(kh,kw,ci), co] = Wmat^T
            dcol = raw_linear(dy2, wmat_t)                                    # [P, 9 Ci]
            dx = torch.empty_like(x)
            L.check(lib.eg_im2col3x3(_ptr(dcol), _ptr(dx), B, H, W, Ci, ctx.stride, 1, _stream(dev)), "eg_col2im3x3")
        if lazy is not None and ctx.need_dx:
            raise L.EgError("conv3x3: the masked shortcut gradient was left for an input-gradient path without the fused epilogue")
        if dres is not None and ctx.pass_sub:       # the shortcut's quarter-grid gradient on a path without the fused epilogue: scatter, then add
            full = torch.empty_like(x)
            L.check(lib.eg_subsample(_ptr(_chk(dres)), _ptr(full), B, H, W, Ci, ctx.stride, 1, _stream(dev)), "eg_subsample")
            dres = full
        if dres is not None:        # passthrough on a path without the fused epilogue (or no dx wanted): the plain add
            dx = raw_ew(EW_ADD, dx, _chk(dres)) if dx is not None else _chk(dres)
        return dx, dw, db, None, None, None, None, None, None, None, None


def conv3x3(x_nhwc, w_oihw, b=None, stride=1, relu=False, want_gap=False, defer_mask=False, passthrough=False, res_link=None):
    """want_gap: also return the per-(clip, tile) channel sums of the output; defer_mask: the ReLU's backward is applied by the consumer;
    passthrough: also return an alias of the input for its second consumer (the residual branch) -- the two gradients are then summed in the
    input-gradient launch's epilogue instead of by a `fork`; passthrough="sub" (stride > 1): that consumer is the strided 1x1 shortcut, the extra
    output is x[:, ::stride, ::stride] and its gradient comes back on that grid."""
    aff = getattr(x_nhwc, "_eg_in_affine", None)        # a deferred BatchNorm produced x: (scale, shift) to apply while staging
    if aff is not None:
        return _Conv3x3.apply(x_nhwc, w_oihw, b, stride, relu, want_gap, defer_mask, passthrough, aff[0], aff[1], res_link)
    return _Conv3x3.apply(x_nhwc, w_oihw, b, stride, relu, want_gap, defer_mask, passthrough, None, None, res_link)


class _Subsample(torch.autograd.Function):
    @staticmethod
    def forward(ctx, x, stride):
        lib = _lib()
        xd = _chk(x)
        B, H, W, Cc = xd.shape
        Ho, Wo = (H - 1) // stride + 1, (W - 1) // stride + 1
        y = torch.empty(B, Ho, Wo, Cc, dtype=torch.float32, device=xd.device)
        L.check(lib.eg_subsample(_ptr(xd), _ptr(y), B, H, W, Cc, stride, 0, _stream(xd.device)), "eg_subsample")
        ctx.shape, ctx.stride = (B, H, W, Cc), stride
        return y

    @staticmethod
    def backward(ctx, dy):
        lib = _lib()
        B, H, W, Cc = ctx.shape
        dyd = _chk(dy)
        dx = torch.empty(B, H, W, Cc, dtype=torch.float32, device=dyd.device)
        L.check(lib.eg_subsample(_ptr(dyd), _ptr(dx), B, H, W, Cc, ctx.stride, 1, _stream(dyd.device)), "eg_subsample")
        return dx, None


def conv1x1(x_nhwc, w_oi11, stride=1):
    """nn.Conv2d(k=1, stride s, bias=False): the downsample shortcut (ResNetSE34V2.py:43-47)."""
    xs = _Subsample.apply(x_nhwc, stride) if stride != 1 else x_nhwc
    return linear(xs, w_oi11)           # the parameter itself: its gradient lands in its flat slice, its weight image is the resident one


def _running_stats_written(bn):
    """The kernel updated running_mean / running_var through raw pointers: bump their version counters (the inference engines key their
    folded BatchNorm vectors on them), and count the batch as nn.BatchNorm does."""
    torch.autograd.graph.increment_version(bn.running_mean)
    torch.autograd.graph.increment_version(bn.running_var)
    _PENDING_COUNTERS.append(bn.num_batches_tracked)
    if len(_PENDING_COUNTERS) >= 512:       # direct users of batch_norm() that never flush: bound the list (the nets flush once per forward)
        flush_batch_counters()


_PENDING_COUNTERS = []


def flush_batch_counters() -> None:
    """num_batches_tracked += 1 for every BatchNorm that ran since the last flush, as ONE multi-tensor launch (43 separate increments per
    generator step otherwise).  train/nets.py calls it at the end of each network's forward; direct users of batch_norm() call it themselves
    (or read the counters only after a forward of the nets)."""
    if _PENDING_COUNTERS:
        torch._foreach_add_(_PENDING_COUNTERS, 1)
        _PENDING_COUNTERS.clear()


class _BatchNorm(torch.autograd.Function):
    """nn.BatchNorm{1,2}d in train() mode over the last (channel) axis; updates the running buffers in place."""

    @staticmethod
    def forward(ctx, x, gamma, beta, run_mean, run_var, momentum, eps, gap=None, relu_input=False, defer=False):
        lib = _lib()
        xd, g, b = _chk(x), _chk(gamma), _chk(beta)
        Cc = xd.shape[-1]
        rows = xd.numel() // Cc
        dev = xd.device
        mean, rstd = torch.empty(Cc, device=dev), torch.empty(Cc, device=dev)
        ws = _scratch(dev, lib.eg_colreduce_workspace_floats(Cc), "col")
        if defer:
            # statistics only; the apply is handed to the consuming convolution as one affine per channel (conv3x3 reads `_eg_in_affine` off the
            # returned alias): the normalised map is neither written nor read -- one write + one read of the map less per block, forward
            aff = torch.empty(2, Cc, device=dev)
            L.check(lib.eg_bn_train_stats_sq(_ptr(gap), _ptr(gap[1]), gap.shape[2], gap.shape[1], _ptr(g), _ptr(b), _ptr(mean), _ptr(rstd), _ptr(run_mean),
                                             _ptr(run_var), _ptr(aff[0]), _ptr(aff[1]), rows, Cc, float(momentum), float(eps), _ptr(ws), _stream(dev)),
                    "eg_bn_train_stats_sq")
            ctx.save_for_backward(xd, g, mean, rstd)
            ctx.relu_input = bool(relu_input)
            ctx.params = (gamma, beta)
            ctx.mark_non_differentiable(aff)
            return x.view_as(x), aff
        y = torch.empty_like(xd)
        if gap is not None and gap.dim() == 4:     # sums and sums of squares from the producing convolution: statistics without touching the map
            L.check(lib.eg_bn_train_forward_sq(_ptr(xd), _ptr(gap), _ptr(gap[1]), gap.shape[2], gap.shape[1], _ptr(g), _ptr(b), _ptr(y), _ptr(mean),
                                               _ptr(rstd), None, _ptr(run_mean), _ptr(run_var), rows, Cc, float(momentum), float(eps), _ptr(ws),
                                               _stream(dev)), "eg_bn_train_forward_sq")
        elif gap is not None:       # mean from the producing convolution's pooling partials: one pass (centred squares) instead of two
            L.check(lib.eg_bn_train_forward_gap(_ptr(xd), _ptr(gap), gap.shape[1], gap.shape[0], _ptr(g), _ptr(b), _ptr(y), _ptr(mean), _ptr(rstd), None,
                                                _ptr(run_mean), _ptr(run_var), rows, Cc, float(momentum), float(eps), _ptr(ws), _stream(dev)),
                    "eg_bn_train_forward_gap")
        else:
            L.check(lib.eg_bn_train_forward(_ptr(xd), _ptr(g), _ptr(b), _ptr(y), _ptr(mean), _ptr(rstd), _ptr(run_mean), _ptr(run_var), rows, Cc,
                                            float(momentum), float(eps), _ptr(ws), _stream(dev)), "eg_bn_train_forward")
        ctx.save_for_backward(xd, g, mean, rstd)
        ctx.relu_input = bool(relu_input)
        ctx.params = (gamma, beta)
        return y

    @staticmethod
    def backward(ctx, dy, _daff=None):
        lib = _lib()
        x, g, mean, rstd = ctx.saved_tensors
        Cc = x.shape[-1]
        rows = x.numel() // Cc
        dev = x.device
        dyd = _chk(dy)
        dx, dg, db = torch.empty_like(x), grad_out(ctx.params[0]), grad_out(ctx.params[1])
        ws = _scratch(dev, lib.eg_colreduce_workspace_floats(Cc), "col")
        L.check(lib.eg_bn_train_backward(_ptr(x), _ptr(dyd), _ptr(g), _ptr(mean), _ptr(rstd), _ptr(dx), _ptr(dg), _ptr(db), rows, Cc, int(ctx.relu_input),
                                         _ptr(ws), _stream(dev)), "eg_bn_train_backward")
        return dx, dg, db, None, None, None, None, None, None, None


DEFER_BN_APPLY = __import__("os").environ.get("EG_DEFER_BN", "1") != "0"       # False: every BatchNorm writes its output map (A/B switch)
# Deferral pays only on large maps: at 128 clips per step 27.36 vs 27.48 ms (13 launches and 2.9 GB less), at 16 clips 8.12 vs 8.07 ms (the affine in
# the convolution's and the weight-gradient kernel's staging costs more than the 6 us apply launch it replaces).  Threshold in elements of the map.
DEFER_BN_MIN_NUMEL = int(__import__("os").environ.get("EG_DEFER_BN_MIN_NUMEL", str(12 << 20)))


def batch_norm(x_channels_last, bn, momentum=0.1, eps=1e-5, gap=None, relu_input=False, defer_apply=False):
    """`bn` = a BatchNorm parameter holder (weight, bias, running_mean, running_var, num_batches_tracked).  gap: pooling partials of the
    convolution that produced x (conv3x3(want_gap=True)); relu_input: x = relu(.) whose mask this backward applies (conv3x3(defer_mask=True)).
    defer_apply: the ONLY consumer is a stride-1, bias-free conv3x3 (a block's conv2): in the split-bf16 modes the normalisation is then not
    applied here but handed to that convolution as a per-channel affine (returned tensor = an alias of x carrying `_eg_in_affine`; conv3x3 and its
    weight-gradient kernel apply it while staging): nobody else may read the returned tensor's values."""
    Cc = x_channels_last.shape[-1]
    if (defer_apply and DEFER_BN_APPLY and _PREC["conv"] != F32 and gap is not None and gap.dim() == 4 and Cc % 32 == 0 and 256 % Cc == 0
            and x_channels_last.dim() == 4 and x_channels_last.numel() >= DEFER_BN_MIN_NUMEL):
        y, aff = _BatchNorm.apply(x_channels_last, bn.weight, bn.bias, bn.running_mean, bn.running_var, momentum, eps, gap, relu_input, True)
        y._eg_in_affine = (aff[0], aff[1])
        _running_stats_written(bn)
        return y
    y = _BatchNorm.apply(x_channels_last, bn.weight, bn.bias, bn.running_mean, bn.running_var, momentum, eps, gap, relu_input)
    _running_stats_written(bn)
    return y


SE_TAIL_RELU_BITS = __import__("os").environ.get("EG_SE_TAIL_BITS", "1") != "0"       # False: the tail's backward reads the block output for its ReLU mask (A/B)
LAZY_SHORTCUT_GRAD = __import__("os").environ.get("EG_LAZY_SHORTCUT", "1") != "0"     # False: the tail writes the identity shortcut's gradient as a map (A/B)


class _SEBlockTail(torch.autograd.Function):
    """relu(se(bn2(c2)) + res) of SEBasicBlock.forward (ResNetBlocks.py:28-36) as one operator: bn2's statistics from conv2's pooling
    partials + one centred pass, the SE gate per clip, one fused output pass; bn2's output is never stored (the backward recomputes it)."""

    @staticmethod
    def forward(ctx, c2, gap, res, gamma, beta, run_mean, run_var, w1, b1, w2, b2, momentum, eps, res_link=None):
        lib = _lib()
        x, r = _chk(c2), _chk(res)
        B, H, W, Cc = x.shape
        hw, dev = H * W, x.device
        g, bt, w1d, b1d, w2d, b2d = _chk(gamma), _chk(beta), _chk(w1), _chk(b1), _chk(w2), _chk(b2)
        mean, rstd, clip = torch.empty(Cc, device=dev), torch.empty(Cc, device=dev), torch.empty(B, Cc, device=dev)
        ws = _scratch(dev, lib.eg_colreduce_workspace_floats(Cc), "col")
        st = _stream(dev)
        if gap.dim() == 4:          # conv2 emitted sums and sums of squares: bn2's statistics and the SE pooling without a pass over c2
            L.check(lib.eg_bn_train_forward_sq(None, _ptr(gap), _ptr(gap[1]), gap.shape[2], B, None, None, None, _ptr(mean), _ptr(rstd), _ptr(clip),
                                               _ptr(run_mean), _ptr(run_var), B * hw, Cc, float(momentum), float(eps), _ptr(ws), st), "eg_bn_train_forward_sq")
        else:
            L.check(lib.eg_bn_train_forward_gap(_ptr(x), _ptr(gap), gap.shape[1], B, None, None, None, _ptr(mean), _ptr(rstd), _ptr(clip), _ptr(run_mean),
                                                _ptr(run_var), B * hw, Cc, float(momentum), float(eps), _ptr(ws), st), "eg_bn_train_forward_gap")
        pooled, h, gate = torch.empty(B, Cc, device=dev), torch.empty(B, Cc // 8, device=dev), torch.empty(B, Cc, device=dev)
        L.check(lib.eg_se_gate_train_forward(_ptr(clip), _ptr(mean), _ptr(rstd), _ptr(g), _ptr(bt), _ptr(w1d), _ptr(b1d), _ptr(w2d), _ptr(b2d), _ptr(pooled),
                                             _ptr(h), _ptr(gate), B, hw, Cc, st), "eg_se_gate_train_forward")
        out = torch.empty_like(x)
        # the tail's ReLU mask as bits (one nibble per float4): the two backward passes read it instead of the whole `out` map
        bits = torch.empty(x.numel() // 32, dtype=torch.int32, device=dev) if (SE_TAIL_RELU_BITS and x.numel() % 32 == 0) else None
        L.check(lib.eg_se_tail_forward(_ptr(x), _ptr(r), _ptr(mean), _ptr(rstd), _ptr(g), _ptr(bt), _ptr(gate), _ptr(out), _ptr(bits), B, hw, Cc, st), "eg_se_tail_forward")
        ctx.save_for_backward(x, out if bits is None else bits, mean, rstd, clip, pooled, h, gate, g, bt, w1d, w2d)
        ctx.has_bits = bits is not None
        # res_link: `res` is the block input's alias out of conv1 (an identity shortcut); with the bit mask its gradient is not written as a map --
        # (dout, bits) go into the link and conv1's input-gradient epilogue masks dout itself (1.4 GB less written per 128-clip step)
        ctx.res_link = res_link if bits is not None else None
        ctx.params = (gamma, beta, w1, b1, w2, b2)
        return out

    @staticmethod
    def backward(ctx, dout):
        lib = _lib()
        x, out, mean, rstd, clip, pooled, h, gate, g, bt, w1, w2 = ctx.saved_tensors
        bits = None
        if ctx.has_bits:
            bits, out = out, None
        B, H, W, Cc = x.shape
        hw, dev, Ch = H * W, x.device, Cc // 8
        d = _chk(dout)
        st = _stream(dev)
        ws = _scratch(dev, lib.eg_colreduce_workspace_floats(Cc), "col")
        small = torch.empty(6, B, Cc, device=dev)                                   # s1, s2raw, dz2, dgap_hw, u1, u2
        s1, s2, dz2, dgap, u1, u2 = small.unbind(0)
        dz1 = torch.empty(B, Ch, device=dev)
        L.check(lib.eg_se_tail_backward_reduce(_ptr(d), _ptr(out), _ptr(bits), _ptr(x), _ptr(mean), _ptr(s1), _ptr(s2), B, hw, Cc, _ptr(ws), st),
                "eg_se_tail_backward_reduce")
        L.check(lib.eg_se_gate_train_backward(_ptr(s1), _ptr(s2), _ptr(clip), _ptr(mean), _ptr(rstd), _ptr(g), _ptr(bt), _ptr(gate), _ptr(h), _ptr(w1), _ptr(w2),
                                              _ptr(dz2), _ptr(dz1), _ptr(dgap), _ptr(u1), _ptr(u2), B, hw, Cc, st), "eg_se_gate_train_backward")
        m1, m2 = torch.empty(2, Cc, device=dev).unbind(0)
        dg, db, dw1, db1, dw2, db2 = (grad_out(p) for p in ctx.params)
        L.check(lib.eg_se_tail_backward_finish(_ptr(u1), _ptr(u2), _ptr(dz2), _ptr(dz1), _ptr(h), _ptr(pooled), _ptr(dg), _ptr(db), _ptr(m1), _ptr(m2),
                                               _ptr(dw1), _ptr(db1), _ptr(dw2), _ptr(db2), B, hw, Cc, st), "eg_se_tail_backward_finish")
        dc2 = torch.empty_like(x)
        dres = torch.empty_like(x) if ctx.res_link is None else None
        if ctx.res_link is not None:
            ctx.res_link["masked"] = (d, bits)
        L.check(lib.eg_se_tail_backward_apply(_ptr(d), _ptr(out), _ptr(bits), _ptr(x), _ptr(mean), _ptr(rstd), _ptr(g), _ptr(gate), _ptr(dgap), _ptr(m1), _ptr(m2),
                                              _ptr(dc2), _ptr(dres), B, hw, Cc, st), "eg_se_tail_backward_apply")
        return dc2, None, dres, dg, db, None, None, dw1, db1, dw2, db2, None, None, None


def se_block_tail(c2, gap, res, bn, fc0, fc2, momentum=0.1, eps=1e-5, res_link=None):
    """res_link: the dict also given to the conv3x3(passthrough=True, res_link=...) whose alias output `res` is (see _Conv3x3.forward)."""
    out = _SEBlockTail.apply(c2, gap, res, bn.weight, bn.bias, bn.running_mean, bn.running_var, fc0.weight, fc0.bias, fc2.weight, fc2.bias, momentum, eps,
                             res_link)
    _running_stats_written(bn)
    return out


class _SELayer(torch.autograd.Function):
    """SELayer.forward (ResNetBlocks.py:92-96): y * sigmoid(W2 relu(W1 mean_hw(y) + b1) + b2), y NHWC."""

    @staticmethod
    def forward(ctx, y, w1, b1, w2, b2):
        lib = _lib()
        yd = _chk(y)
        B, H, W, Cc = yd.shape
        dev = yd.device
        gap = torch.empty(B, Cc, device=dev)
        ws = _scratch(dev, lib.eg_colreduce_workspace_floats(Cc), "col")
        L.check(lib.eg_seg_mean(_ptr(yd), _ptr(gap), B, H * W, Cc, 1.0 / (H * W), _ptr(ws), _stream(dev)), "eg_seg_mean")
        w1d, b1d, w2d, b2d = _chk(w1), _chk(b1), _chk(w2), _chk(b2)
        h = raw_linear(gap, w1d, b1d, relu=True)
        gate = raw_ew(EW_SIGMOID, raw_linear(h, w2d, b2d))
        out = torch.empty_like(yd)
        L.check(lib.eg_se_scale(_ptr(yd), _ptr(gate), None, _ptr(out), B, H * W, Cc, _stream(dev)), "eg_se_scale")
        ctx.save_for_backward(yd, gap, h, gate, w1d, w2d)
        return out

    @staticmethod
    def backward(ctx, dout):
        lib = _lib()
        y, gap, h, gate, w1, w2 = ctx.saved_tensors
        B, H, W, Cc = y.shape
        dev = y.device
        do = _chk(dout)
        dgate = torch.empty(B, Cc, device=dev)
        ws = _scratch(dev, lib.eg_colreduce_workspace_floats(Cc), "col")
        L.check(lib.eg_seg_dot(_ptr(do), _ptr(y), _ptr(dgate), B, H * W, Cc, _ptr(ws), _stream(dev)), "eg_seg_dot")
        dz2 = raw_ew(EW_SIGMOID_BWD, dgate, gate)
        dh, dw2, db2 = raw_linear_backward(h, w2, dz2)
        dz1 = raw_ew(EW_RELU_BWD, dh, h)
        dgap, dw1, db1 = raw_linear_backward(gap, w1, dz1)
        dgap = raw_ew(EW_SCALE, dgap, None, 1.0 / (H * W))
        dy = torch.empty_like(y)
        L.check(lib.eg_se_scale(_ptr(do), _ptr(gate), _ptr(dgap), _ptr(dy), B, H * W, Cc, _stream(dev)), "eg_se_scale")
        return dy, dw1, db1, dw2, db2


def se_layer(y_nhwc, fc0, fc2):
    return _SELayer.apply(y_nhwc, fc0.weight, fc0.bias, fc2.weight, fc2.bias)


class _LayerNorm(torch.autograd.Function):
    @staticmethod
    def forward(ctx, x, g, b, eps):
        lib = _lib()
        xd, gd, bd = _chk(x), _chk(g), _chk(b)
        D = xd.shape[-1]
        rows = xd.numel() // D
        y = torch.empty_like(xd)
        L.check(lib.eg_layernorm(_ptr(xd), _ptr(gd), _ptr(bd), _ptr(y), rows, D, float(eps), _stream(xd.device)), "eg_layernorm")
        ctx.save_for_backward(xd, gd)
        ctx.eps = eps
        ctx.params = (g, b)
        return y

    @staticmethod
    def backward(ctx, dy):
        lib = _lib()
        x, g = ctx.saved_tensors
        D = x.shape[-1]
        rows = x.numel() // D
        dyd = _chk(dy)
        dx, t = torch.empty_like(x), torch.empty_like(x)
        L.check(lib.eg_layernorm_backward(_ptr(x), _ptr(dyd), _ptr(g), _ptr(dx), _ptr(t), rows, D, float(ctx.eps), _stream(x.device)),
                "eg_layernorm_backward")
        db, dg = raw_colsum(dyd.view(rows, D), t.view(rows, D), want1=True, out0=grad_out(ctx.params[1]), out1=grad_out(ctx.params[0]))   # t = xhat
        return dx, dg, db, None


def layer_norm(x, ln):
    return _LayerNorm.apply(x, ln.weight, ln.bias, ln.eps)


class _Attention(torch.autograd.Function):
    """ScaledDotProductAttention (Modules.py:13-23) on [B, L, H*64] projections: forward and backward on the fp32 matrix pipe, with nn.Dropout(p)
    on the probabilities (`attn = self.dropout(F.softmax(attn, dim=-1))`, :21) from the counter-based mask stream when p > 0."""

    @staticmethod
    def forward(ctx, q, k, v, heads, p, seed, offset, epoch):
        lib = _lib()
        qd, kd, vd = _chk(q), _chk(k), _chk(v)
        B, Lq, D = qd.shape
        Lk = kd.shape[1]
        dev = qd.device
        out = torch.empty_like(qd)
        attn = torch.empty(B, heads, Lq, Lk, device=dev)
        L.check(lib.eg_attention_train(_ptr(qd), D, _ptr(kd), D, _ptr(vd), D, _ptr(out), D, _ptr(attn), B, heads, Lq, Lk, D // heads, float(p), seed,
                                       offset, _ptr(epoch), _stream(dev)), "eg_attention_train")
        ctx.save_for_backward(qd, kd, vd, attn)
        ctx.cfg = (heads, float(p), seed, offset, epoch)
        return out

    @staticmethod
    def backward(ctx, dout):
        lib = _lib()
        q, k, v, attn = ctx.saved_tensors
        heads, p, seed, offset, epoch = ctx.cfg
        B, Lq, D = q.shape
        Lk = k.shape[1]
        do = _chk(dout)
        dq, dk, dv = torch.empty_like(q), torch.empty_like(k), torch.empty_like(v)
        L.check(lib.eg_attention_backward_train(_ptr(q), D, _ptr(k), D, _ptr(v), D, _ptr(attn), _ptr(do), D, _ptr(dq), D, _ptr(dk), D, _ptr(dv), D, B,
                                                heads, Lq, Lk, D // heads, p, seed, offset, _ptr(epoch), _stream(q.device)), "eg_attention_backward_train")
        return dq, dk, dv, None, None, None, None, None


def attention(q, k, v, heads, dropout_p: float = 0.0):
    """dropout_p: nn.Dropout on the attention probabilities (train mode); the mask stream is the one of `dropout()`."""
    if dropout_p > 0.0:
        if q.is_cuda and torch.cuda.is_current_stream_capturing() and _DROP.get("epoch") is None:
            raise L.EgError("attention dropout inside a stream capture needs the device-resident mask epoch (functional.use_device_dropout_epoch)")
        B, Lq = q.shape[0], q.shape[1]
        off = next_dropout_offset(B * heads * Lq * k.shape[1])
        return _Attention.apply(q, k, v, heads, dropout_p, _DROP["seed"], off, _DROP.get("epoch"))
    return _Attention.apply(q, k, v, heads, 0.0, 0, 0, None)


# ---- fused transformer blocks (verdict r04 item 2: fewer, fatter launches) ---------------------------------------------------------------
# One autograd node per MultiHeadAttention / PositionwiseFeedForward / Linear chain instead of one per operator.  What that buys:
#  * Q|K|V (self attention) and K|V (cross attention) are ONE product each way: the three weights are adjacent rows of the flat parameter buffer
#    (`fused_rows`), so the forward, the input gradient (K = 1536 / 1024) and the weight gradient are one launch each instead of three, the
#    attention kernels read / write the [rows, 3D] buffer through row strides, and the fan-out adds of the old `fork`s disappear;
#  * nn.Dropout + residual ride in the producing GEMM's epilogue (eg_linear_ex: the mask is the stateless counter hash); in the backward the
#    LayerNorm kernel emits the gradient twice -- plain for the residual, Dropout-masked for the branch -- and its affine gradients from the same
#    pass; ReLU's backward is a gate in the epilogue of the input-gradient product, the residual's gradient an epilogue add.
# Same arithmetic per element as the operator-by-operator composition (same kernels, same K order), so the gradient goldens hold unchanged.
FUSE_BLOCKS = True          # False: train/nets.py composes the blocks operator by operator (the A/B switch of the tests)


def drop_site(p: float, numel: int, like: torch.Tensor):
    """(p, seed, offset, epoch) of ONE nn.Dropout call site for a tensor of `numel` elements, None when p == 0.  Advances the mask stream."""
    if p <= 0.0:
        return None
    if like.is_cuda and torch.cuda.is_current_stream_capturing() and _DROP.get("epoch") is None:
        raise L.EgError("dropout inside a stream capture needs the device-resident mask epoch (functional.use_device_dropout_epoch); "
                        "train/graph.GraphedStep enables it when told the model trains with dropout (stochastic=True)")
    return (float(p), _DROP["seed"], next_dropout_offset(numel), _DROP.get("epoch"))


def fused_rows(ws):
    """[n_i, k] matrices -> (one [sum n_i, k] matrix, is_view).  A view when they lie back to back in one storage (parameters flattened by
    optim.flatten_parameters in registration order: w_qs, w_ks, w_vs), else a concatenation (data movement)."""
    w0 = ws[0]
    k = w0.shape[1]
    ptr, ok = w0.data_ptr(), True
    for w in ws:
        ok = ok and w.is_contiguous() and w.shape[1] == k and w.data_ptr() == ptr and w.untyped_storage().data_ptr() == w0.untyped_storage().data_ptr()
        ptr += w.numel() * 4
    n = sum(w.shape[0] for w in ws)
    if ok:
        return torch.as_strided(w0, (n, k), (k, 1)), True
    return torch.cat([w.reshape(-1, k) for w in ws], 0), False


def fused_grad_out(params, k):
    """Destination of the weight gradient of a fused product: ONE [sum n_i, k] view over the parameters' adjacent slices of the flat gradient
    buffer when there is one (nothing wrote them yet this step), else fresh memory.  -> (matrix, per-parameter views to hand to autograd)."""
    slots = [getattr(p, "_eg_slot", None) for p in params]
    n = sum(p.shape[0] for p in params)
    if all(sl is not None for sl in slots) and all(id(p) not in p._eg_fp.written for p in params):
        ptr, ok = slots[0].data_ptr(), True
        for sl in slots:
            ok = ok and sl.data_ptr() == ptr
            ptr += sl.numel() * 4
        if ok:
            for p in params:
                p._eg_fp.written.add(id(p))
            return torch.as_strided(slots[0], (n, k), (k, 1)), [sl.view(sl.shape) for sl in slots]
    m = torch.empty(n, k, dtype=torch.float32, device=params[0].device)
    outs, r = [], 0
    for p in params:
        outs.append(m[r:r + p.shape[0]].view(p.shape))
        r += p.shape[0]
    return m, outs


def raw_wgrad(x, dy, dw, db=None):
    """dw[N,K] = dy[R,N]^T x[R,K] (+ db = colsum(dy)) into the given tensors: one split-bf16 MFMA launch (csrc/lingrad.hip) under "bf16x3",
    the fp32 TN GEMM + column sum under "f32"."""
    lib = _lib()
    R, N = dy.shape
    K = x.shape[1]
    if _PREC["gemm"] != F32:
        need = int(lib.eg_linear_wgrad_mfma_workspace_floats(R, N, K))
        ws = _scratch(dy.device, need, "tn") if need else None
        L.check(lib.eg_linear_wgrad_mfma(_ptr(dy), N, _ptr(x), K, _ptr(dw), K, _ptr(db), R, N, K, _ptr(ws), ws.numel() if ws is not None else 0,
                                         _stream(dy.device)), "eg_linear_wgrad_mfma")
    else:
        raw_gemm_tn(dy, x, out=dw)
        if db is not None:
            raw_colsum(dy, out0=db)
    return dw, db


def _ln_forward(x2, g, b, eps, want_img=False):
    """-> (y, images of y or None).  The images feed the next block's first product (presplit_ok decides)."""
    y = torch.empty_like(x2)
    img = new_images(x2.shape[0], x2.shape[1], x2.device) if (want_img and presplit_ok(x2.shape[0], x2.shape[1])) else None
    L.check(_lib().eg_layernorm_img(_ptr(x2), _ptr(g), _ptr(b), _ptr(y), _ptr(img), x2.shape[0], x2.shape[1], float(eps), _stream(x2.device)), "eg_layernorm")
    return y, img


def _ln_backward_ex(pre, dy2, g, eps, site, g_param, b_param, want_img=False):
    """-> (d pre, d pre through the Dropout `site` (the same tensor when the site is off), dgamma, dbeta, images of the branch gradient or None):
    eg_layernorm_backward_ex."""
    lib = _lib()
    rows, D = pre.shape
    dpre = torch.empty_like(pre)
    dbr = torch.empty_like(pre) if site is not None else None
    dg, db = grad_out(g_param), grad_out(b_param)
    ws = _scratch(pre.device, int(lib.eg_layernorm_backward_ex_workspace_floats(rows, D)), "lnx")
    img = new_images(rows, D, pre.device) if (want_img and presplit_ok(rows, D)) else None
    p, seed, off, ep = site if site is not None else (0.0, 0, 0, None)
    L.check(lib.eg_layernorm_backward_ex(_ptr(pre), _ptr(dy2), _ptr(g), _ptr(dpre), _ptr(dbr), _ptr(dg), _ptr(db), rows, D, float(eps), float(p), int(seed),
                                         int(off), _ptr(ep), _ptr(ws), _ptr(img), _stream(pre.device)), "eg_layernorm_backward_ex")
    return dpre, (dbr if dbr is not None else dpre), dg, db, img


def blocks_fusable(d_model: int) -> bool:
    return FUSE_BLOCKS and d_model % 64 == 0 and d_model <= 1024


class _MHABlock(torch.autograd.Function):
    """MultiHeadAttention.forward (SubLayers.py:30-59): LN(dropout(fc(attention(q Wq, k Wk, v Wv))) + q) as one node; xkv None = self attention."""

    @staticmethod
    def forward(ctx, xq, xkv, wq, wk, wv, wfc, g, b, heads, p_attn, p_fc, eps, xq_img, xkv_img):
        lib = _lib()
        B, Lq, D = xq.shape
        xq2 = _chk(xq).reshape(B * Lq, D)
        selfa = xkv is None
        if xq_img is not None and not presplit_ok(B * Lq, D):
            xq_img = None
        wqd, wkd, wvd, wfd, gd, bd = _chk(wq), _chk(wk), _chk(wv), _chk(wfc), _chk(g), _chk(b)
        dev = xq2.device
        Dq = wqd.shape[0]               # heads x d_k: the projection width (Motion_Discriminator: 8 x 64 = 512 over a 128-wide model)
        if selfa:
            Lk, xkv2 = Lq, None
            wcat, _ = fused_rows([wqd, wkd, wvd])
            qkv = raw_linear(xq2, wcat, x_img=xq_img)                     # [rows, 3D]; X through the images the producing LayerNorm emitted
            q, k, v, ldq, ldk = qkv, qkv[:, Dq:], qkv[:, 2 * Dq:], 3 * Dq, 3 * Dq
            saved_proj = (qkv,)
        else:
            Lk = xkv.shape[1]
            xkv2 = _chk(xkv).reshape(B * Lk, D)
            wcat, _ = fused_rows([wkd, wvd])
            if xkv_img is not None and not presplit_ok(B * Lk, D):
                xkv_img = None
            qb = raw_linear(xq2, wqd, x_img=xq_img)
            kv = raw_linear(xkv2, wcat, x_img=xkv_img)                    # [rows_k, 2D]
            q, k, v, ldq, ldk = qb, kv, kv[:, Dq:], Dq, 2 * Dq
            saved_proj = (qb, kv)
        o = torch.empty(B * Lq, Dq, device=dev)
        attn = torch.empty(B, heads, Lq, Lk, device=dev)
        site_a = drop_site(p_attn, B * heads * Lq * Lk, xq2)
        pa, sa, oa, ea = site_a if site_a is not None else (0.0, 0, 0, None)
        L.check(lib.eg_attention_train(q.data_ptr(), ldq, k.data_ptr(), ldk, v.data_ptr(), ldk, _ptr(o), Dq, _ptr(attn), B, heads, Lq, Lk, Dq // heads,
                                       float(pa), sa, oa, _ptr(ea), _stream(dev)), "eg_attention_train")
        site_f = drop_site(p_fc, B * Lq * D, xq2)
        pre = raw_linear(o, wfd, res=xq2, drop=site_f)                    # dropout(fc(.)) + residual in the product's epilogue (SubLayers.py:54)
        y, yimg = _ln_forward(pre, gd, bd, eps, want_img=True)
        ctx.save_for_backward(xq2, xkv2, attn, o, pre, wqd, wkd, wvd, wfd, gd, *saved_proj)
        ctx.cfg = (B, Lq, Lk, D, Dq, heads, selfa, site_a, site_f, float(eps), xq.requires_grad, (xkv is not None and xkv.requires_grad))
        ctx.params = (wq, wk, wv, wfc, g, b)
        ctx.set_materialize_grads(False)
        if yimg is None:
            return y.view(B, Lq, D), None
        ctx.mark_non_differentiable(yimg)
        return y.view(B, Lq, D), yimg

    @staticmethod
    def backward(ctx, dy, _dimg=None):
        lib = _lib()
        xq2, xkv2, attn, o, pre, wq, wk, wv, wfc, g = ctx.saved_tensors[:10]
        proj = ctx.saved_tensors[10:]
        B, Lq, Lk, D, Dq, heads, selfa, site_a, site_f, eps, need_dxq, need_dxkv = ctx.cfg
        pq, pk, pv, pfc, pg, pb = ctx.params
        dev = pre.device
        dy2 = _chk(dy).reshape(B * Lq, D)
        dpre, dfc, dg, db, dfc_img = _ln_backward_ex(pre, dy2, g, eps, site_f, pg, pb, want_img=True)
        do = raw_linear(dfc, wfc, w_transposed=True, x_img=dfc_img)
        dwfc, _ = raw_wgrad(o, dfc, grad_out(pfc))
        pa, sa, oa, ea = site_a if site_a is not None else (0.0, 0, 0, None)
        if selfa:
            (qkv,) = proj
            dqkv = torch.empty_like(qkv)
            L.check(lib.eg_attention_backward_train(qkv.data_ptr(), 3 * Dq, qkv[:, Dq:].data_ptr(), 3 * Dq, qkv[:, 2 * Dq:].data_ptr(), 3 * Dq, _ptr(attn), _ptr(do), Dq,
                                                    dqkv.data_ptr(), 3 * Dq, dqkv[:, Dq:].data_ptr(), 3 * Dq, dqkv[:, 2 * Dq:].data_ptr(), 3 * Dq, B, heads, Lq, Lk,
                                                    Dq // heads, float(pa), sa, oa, _ptr(ea), _stream(dev)), "eg_attention_backward_train")
            wcat, _ = fused_rows([wq, wk, wv])
            dx = raw_linear(dqkv, wcat, w_transposed=True, res=dpre) if need_dxq else None      # the residual's gradient rides in the epilogue
            dwm, dws = fused_grad_out((pq, pk, pv), D)
            raw_wgrad(xq2, dqkv, dwm)
            return (dx.view(B, Lq, D) if dx is not None else None), None, dws[0], dws[1], dws[2], dwfc, dg, db, None, None, None, None, None, None
        qb, kv = proj
        dq, dkv = torch.empty_like(qb), torch.empty_like(kv)
        L.check(lib.eg_attention_backward_train(_ptr(qb), Dq, kv.data_ptr(), 2 * Dq, kv[:, Dq:].data_ptr(), 2 * Dq, _ptr(attn), _ptr(do), Dq, _ptr(dq), Dq,
                                                dkv.data_ptr(), 2 * Dq, dkv[:, Dq:].data_ptr(), 2 * Dq, B, heads, Lq, Lk, Dq // heads, float(pa), sa, oa, _ptr(ea),
                                                _stream(dev)), "eg_attention_backward_train")
        wcat, _ = fused_rows([wk, wv])
        dxq = raw_linear(dq, wq, w_transposed=True, res=dpre) if need_dxq else None
        dxkv = raw_linear(dkv, wcat, w_transposed=True) if need_dxkv else None
        dwq, _ = raw_wgrad(xq2, dq, grad_out(pq))
        dwm, dws = fused_grad_out((pk, pv), D)
        raw_wgrad(xkv2, dkv, dwm)
        return ((dxq.view(B, Lq, D) if dxq is not None else None), (dxkv.view(B, Lk, D) if dxkv is not None else None), dwq, dws[0], dws[1], dwfc, dg, db,
                None, None, None, None, None, None)


def mha_block(m, xq, xkv=None, p_attn=0.0, p_fc=0.0):
    """m: a MultiHeadAttention parameter holder.  xkv None: self attention (k = v = q = xq); else k = v = xkv."""
    dq = m.w_qs.weight.shape[0]
    if m.w_ks.weight.shape[0] != dq or m.w_vs.weight.shape[0] != dq or dq % m.n_head or m.fc.weight.shape[1] != dq:
        # the block slices ONE fused Q|K|V product by the projection width and hands n_head x (width / n_head) heads to the attention kernel for
        # K and V alike (the reference's only configuration: d_k = d_v = 64, SubLayers.py:17-28); anything else must take nets.mha_forward
        raise L.EgError(f"mha_block: needs n_head*d_k == n_head*d_v (w_qs / w_ks / w_vs rows {dq} / {m.w_ks.weight.shape[0]} / "
                        f"{m.w_vs.weight.shape[0]}, fc columns {m.fc.weight.shape[1]}, heads {m.n_head}); use the unfused path")
    y, yimg = _MHABlock.apply(xq, xkv, m.w_qs.weight, m.w_ks.weight, m.w_vs.weight, m.fc.weight, m.layer_norm.weight, m.layer_norm.bias, m.n_head,
                              float(p_attn), float(p_fc), m.layer_norm.eps, images_of(xq), images_of(xkv) if xkv is not None else None)
    if yimg is not None:
        y._eg_img = yimg            # the consumer's first product reads the row through these images (large batches, split-bf16)
    return y


class _FFNBlock(torch.autograd.Function):
    """PositionwiseFeedForward.forward (SubLayers.py:74-84): LN(dropout(w_2(relu(w_1 x))) + x) as one node: 3 launches forward, 6 backward."""

    @staticmethod
    def forward(ctx, x, w1, b1, w2, b2, g, b, p_drop, eps, x_img):
        xs = x.shape
        D = xs[-1]
        x2 = _chk(x).reshape(-1, D)
        w1d, b1d, w2d, b2d, gd, bd = _chk(w1), _chk(b1), _chk(w2), _chk(b2), _chk(g), _chk(b)
        chain = x_img is not None and presplit_ok(x2.shape[0], D) and presplit_ok(x2.shape[0], w1d.shape[0])
        # large batches, split-bf16: x arrives as images (from the producing LayerNorm), the hidden leaves its product as fp32 (the backward's gate and
        # weight gradient read it) AND as images for w_2
        h, h_img = raw_linear_img(x2, w1d, b1d, relu=True, x_img=x_img if chain else None, want_img=chain)
        site = drop_site(p_drop, x2.numel(), x2)
        pre = raw_linear(h, w2d, b2d, res=x2, drop=site, x_img=h_img)     # x = self.dropout(x); x += residual (SubLayers.py:79)
        y, yimg = _ln_forward(pre, gd, bd, eps, want_img=True)
        ctx.save_for_backward(x2, h, pre, w1d, w2d, gd)
        ctx.cfg = (xs, site, float(eps), x.requires_grad)
        ctx.params = (w1, b1, w2, b2, g, b)
        ctx.set_materialize_grads(False)
        if yimg is None:
            return y.view(xs), None
        ctx.mark_non_differentiable(yimg)
        return y.view(xs), yimg

    @staticmethod
    def backward(ctx, dy, _dimg=None):
        x2, h, pre, w1, w2, g = ctx.saved_tensors
        xs, site, eps, need_dx = ctx.cfg
        p1, pb1, p2, pb2, pg, pb = ctx.params
        dy2 = _chk(dy).reshape(-1, xs[-1])
        rows = x2.shape[0]
        chain = presplit_ok(rows, xs[-1]) and presplit_ok(rows, h.shape[1])
        dpre, dff, dg, db, dff_img = _ln_backward_ex(pre, dy2, g, eps, site, pg, pb, want_img=chain)
        # ReLU backward as the epilogue gate; the gated gradient as fp32 (for dW_1) and as images (for the next input-gradient product)
        dh, dh_img = raw_linear_img(dff, w2, w_transposed=True, gate=h, x_img=dff_img, want_img=chain and need_dx)
        dw2, db2 = raw_wgrad(h, dff, grad_out(p2), grad_out(pb2))
        dx = raw_linear(dh, w1, w_transposed=True, res=dpre, x_img=dh_img) if need_dx else None     # + the residual's gradient
        dw1, db1 = raw_wgrad(x2, dh, grad_out(p1), grad_out(pb1))
        return (dx.view(xs) if dx is not None else None), dw1, db1, dw2, db2, dg, db, None, None, None


def ffn_block(f, x, p_drop=0.0):
    y, yimg = _FFNBlock.apply(x, f.w_1.weight, f.w_1.bias, f.w_2.weight, f.w_2.bias, f.layer_norm.weight, f.layer_norm.bias, float(p_drop), f.layer_norm.eps,
                              images_of(x))
    if yimg is not None:
        y._eg_img = yimg
    return y


class _LinearChain(torch.autograd.Function):
    """Linear -> [ReLU] -> [Dropout] -> Linear -> ... (the reference's nn.Sequential MLPs: Models_spatial_memory.py:488-536, BEAT_CVAE.py:336-380): ReLU and
    Dropout between consecutive layers are epilogue options of the producing product forward, and of the NEXT layer's input-gradient product
    backward -- no elementwise launch either way."""

    @staticmethod
    def forward(ctx, x, relu_between, drop_p, n_layers, x_img, *wb):
        xs = x.shape
        h = _chk(x).reshape(-1, xs[-1])
        ws = [_chk(wb[2 * i]) for i in range(n_layers)]
        bs = [(_chk(wb[2 * i + 1]) if wb[2 * i + 1] is not None else None) for i in range(n_layers)]
        ins, sites = [], []
        rows = h.shape[0]
        img = x_img if (x_img is not None and presplit_ok(rows, h.shape[1])) else None
        for i in range(n_layers):
            last = i + 1 == n_layers
            ins.append(h)
            site = drop_site(drop_p, h.shape[0] * ws[i].shape[0], h) if (drop_p > 0.0 and not last) else None
            sites.append(site)
            # large batches, split-bf16: a layer whose output width suits (a multiple of 64) hands it to the next one as images
            want = (not last) and presplit_ok(rows, ws[i].shape[0]) and ws[i].shape[0] == ws[i + 1].shape[1]
            h, img = raw_linear_img(h, ws[i], bs[i], relu=relu_between and not last, drop=site, x_img=img, want_img=want)
        ctx.save_for_backward(*ins, *ws)
        ctx.cfg = (xs, n_layers, bool(relu_between), sites, x.requires_grad)
        ctx.params = wb
        return h.view(*xs[:-1], ws[-1].shape[0])

    @staticmethod
    def backward(ctx, dy):
        xs, n, relu_between, sites, need_dx = ctx.cfg
        ins, ws = ctx.saved_tensors[:n], ctx.saved_tensors[n:]
        g = _chk(dy).reshape(-1, ws[-1].shape[0])
        grads = [None] * (2 * n)
        rows, gimg = g.shape[0], None
        for i in range(n - 1, -1, -1):
            pw, pbias = ctx.params[2 * i], ctx.params[2 * i + 1]
            same = ins[i].shape[1] == ws[i].shape[1]
            dw = grad_out(pw) if same else torch.empty(ws[i].shape[0], ins[i].shape[1], dtype=torch.float32, device=g.device)
            db = grad_out(pbias) if pbias is not None else None
            raw_wgrad(ins[i], g, dw, db)
            grads[2 * i], grads[2 * i + 1] = (dw if same else dw[:, :ws[i].shape[1]]), db
            if i > 0 or need_dx:
                # the masks of the boundary in FRONT of layer i (ReLU by its saved output = this layer's input; Dropout from that site's scalars)
                gate = ins[i] if (relu_between and i > 0) else None
                # the gradient continues as images when the next input-gradient product (layer i-1's) can take them
                want = i > 0 and (i > 1 or need_dx) and presplit_ok(rows, ws[i].shape[1]) and presplit_ok(rows, ws[i - 1].shape[0]) and \
                    ws[i].shape[1] == ws[i - 1].shape[0]
                use = gimg if (gimg is not None and presplit_ok(rows, g.shape[1])) else None
                g, gimg = raw_linear_img(g, ws[i], w_transposed=True, gate=gate, drop=sites[i - 1] if i > 0 else None, x_img=use, want_img=want)
                if g.shape[1] != (ins[i].shape[1]):
                    g = g[:, :ins[i].shape[1]]
                    gimg = None
        dx = g.reshape(xs) if need_dx else None
        return (dx, None, None, None, None, *grads)


def linear_chain(x, layers, relu_between=False, drop_p=0.0):
    """layers: Linear parameter holders.  x [..., K] -> [..., N_last]."""
    wb = []
    for lin in layers:
        wb += [lin.weight, lin.bias]
    return _LinearChain.apply(x, bool(relu_between), float(drop_p), len(layers), images_of(x), *wb)


class _Conv1dCL(torch.autograd.Function):
    """nn.Conv1d on channels-last activations x [B, L, Cin] with the reference's weight [Cout, Cin, k]: one launch forward, two backward
    (csrc/conv1d_train.hip), gradients written straight into the parameters' flat slices."""

    @staticmethod
    def forward(ctx, x, w, b, stride, pad, dilation):
        lib = _lib()
        xd, wd = _chk(x), _chk(w)
        B, Ln, Ci = xd.shape
        Co, _, k = wd.shape
        Lout = (Ln + 2 * pad - dilation * (k - 1) - 1) // stride + 1
        dev = xd.device
        y = torch.empty(B, Lout, Co, device=dev)
        L.check(lib.eg_conv1d_cl_forward(_ptr(xd), _ptr(wd), _ptr(_chk(b)) if b is not None else None, _ptr(y), B, Ln, Ci, Lout, Co, k, stride, pad, dilation,
                                         _stream(dev)), "eg_conv1d_cl_forward")
        ctx.save_for_backward(xd, wd)
        ctx.cfg = (stride, pad, dilation, Lout, x.requires_grad)
        ctx.params = (w, b)
        return y

    @staticmethod
    def backward(ctx, dy):
        lib = _lib()
        x, w = ctx.saved_tensors
        stride, pad, dilation, Lout, need_dx = ctx.cfg
        B, Ln, Ci = x.shape
        Co, _, k = w.shape
        dyd = _chk(dy)
        st = _stream(x.device)
        wp, bp = ctx.params
        dw = grad_out(wp)
        db = grad_out(bp) if bp is not None else None
        need = int(lib.eg_conv1d_cl_backward_weight_workspace_floats(B, Ci, Lout, Co, k, stride, dilation))
        ws = _scratch(x.device, need, "c1w") if need else None
        L.check(lib.eg_conv1d_cl_backward_weight(_ptr(x), _ptr(dyd), _ptr(dw), _ptr(db), None, B, Ln, Ci, Lout, Co, k, stride, pad, dilation, _ptr(ws),
                                                 ws.numel() if ws is not None else 0, st), "eg_conv1d_cl_backward_weight")
        dx = None
        if need_dx:
            dx = torch.empty_like(x)
            L.check(lib.eg_conv1d_cl_backward_input(_ptr(dyd), _ptr(w), None, _ptr(dx), B, Ln, Ci, Lout, Co, k, stride, pad, dilation, st),
                    "eg_conv1d_cl_backward_input")
        return dx, dw, db, None, None, None


def conv1d_cl(x_blc, w, b=None, stride=1, pad=0, dilation=1):
    return _Conv1dCL.apply(x_blc, w, b, stride, pad, dilation)


class _ConvT1dCL(torch.autograd.Function):
    """nn.ConvTranspose1d(k, stride, padding, output_padding) on channels-last x [B, L, Cin]; weight [Cin, Cout, k] as in the reference's
    state_dict.  The adjoint of a strided conv: forward = that conv's input-gradient kernel (+ bias), input gradient = its forward kernel,
    weight gradient = its weight-gradient kernel with x and dy exchanged."""

    @staticmethod
    def forward(ctx, x, w, b, stride, pad, out_pad):
        lib = _lib()
        xd, wd = _chk(x), _chk(w)
        B, Ln, Ci = xd.shape
        _, Co, k = wd.shape
        Lout = (Ln - 1) * stride - 2 * pad + k + out_pad
        dev = xd.device
        y = torch.empty(B, Lout, Co, device=dev)
        L.check(lib.eg_conv1d_cl_backward_input(_ptr(xd), _ptr(wd), _ptr(_chk(b)) if b is not None else None, _ptr(y), B, Lout, Co, Ln, Ci, k, stride, pad, 1,
                                                _stream(dev)), "eg_conv1d_cl_backward_input (ConvTranspose1d forward)")
        ctx.save_for_backward(xd, wd)
        ctx.cfg = (stride, pad, Lout, x.requires_grad)
        ctx.params = (w, b)
        return y

    @staticmethod
    def backward(ctx, dy):
        lib = _lib()
        x, w = ctx.saved_tensors
        stride, pad, Lout, need_dx = ctx.cfg
        B, Ln, Ci = x.shape
        _, Co, k = w.shape
        dyd = _chk(dy)
        st = _stream(x.device)
        wp, bp = ctx.params
        dw = grad_out(wp)
        db = grad_out(bp) if bp is not None else None
        need = int(lib.eg_conv1d_cl_backward_weight_workspace_floats(B, Co, Ln, Ci, k, stride, 1))
        ws = _scratch(x.device, need, "c1w") if need else None
        if need and db is not None:     # the tiled kernel sums dy-side biases only: the layer's bias gradient (a sum over the x side here) on its own
            raw_colsum(dyd.view(B * Lout, Co), out0=db)
        L.check(lib.eg_conv1d_cl_backward_weight(_ptr(dyd), _ptr(x), _ptr(dw), None, None if need else _ptr(db), B, Lout, Co, Ln, Ci, k, stride, pad, 1,
                                                 _ptr(ws), ws.numel() if ws is not None else 0, st), "eg_conv1d_cl_backward_weight (ConvTranspose1d)")
        dx = None
        if need_dx:
            dx = torch.empty_like(x)
            L.check(lib.eg_conv1d_cl_forward(_ptr(dyd), _ptr(w), None, _ptr(dx), B, Lout, Co, Ln, Ci, k, stride, pad, 1, st),
                    "eg_conv1d_cl_forward (ConvTranspose1d input gradient)")
        return dx, dw, db, None, None, None


def conv_transpose1d_cl(x_blc, w, b=None, stride=2, pad=1, out_pad=1):
    return _ConvT1dCL.apply(x_blc, w, b, stride, pad, out_pad)


class _SPGate(torch.autograd.Function):
    """SP_Memory_Net_v1's gate (Models_memory.py:239-249): for c < chunk  s = sigmoid(<mem_b, pred_bc>), out_bc = s pred_bc + (1 - s) mem_b."""

    @staticmethod
    def forward(ctx, mem, pred, chunk):
        lib = _lib()
        m, p = _chk(mem), _chk(pred)
        B, P, D = p.shape
        out, gate = torch.empty_like(p), torch.empty(B, max(chunk, 1), device=p.device)
        L.check(lib.eg_sp_gate_forward(_ptr(m), _ptr(p), _ptr(out), _ptr(gate), B, P, D, chunk, _stream(p.device)), "eg_sp_gate_forward")
        ctx.save_for_backward(m, p, gate)
        ctx.chunk = chunk
        return out

    @staticmethod
    def backward(ctx, g):
        lib = _lib()
        m, p, gate = ctx.saved_tensors
        B, P, D = p.shape
        gd = _chk(g)
        dp, dm = torch.empty_like(p), torch.empty_like(m)
        L.check(lib.eg_sp_gate_backward(_ptr(m), _ptr(p), _ptr(gate), _ptr(gd), _ptr(dp), _ptr(dm), B, P, D, ctx.chunk, _stream(p.device)), "eg_sp_gate_backward")
        return dm, dp, None


def sp_memory_gate(mem, pred, chunk: int):
    return _SPGate.apply(mem, pred, int(chunk))


class _TMScale(torch.autograd.Function):
    """TM_Memory_Net behind its score (Models_memory.py:290-292): w = softmax(score, dim=1), out_bc = pred_bc (1 + w_bc) for c < chunk."""

    @staticmethod
    def forward(ctx, score, pred, chunk):
        lib = _lib()
        sc, p = _chk(score), _chk(pred)
        B, P, D = p.shape
        out, w = torch.empty_like(p), torch.empty(B, chunk, device=p.device)
        L.check(lib.eg_tm_scale_forward(_ptr(sc), _ptr(p), _ptr(out), _ptr(w), B, P, D, chunk, _stream(p.device)), "eg_tm_scale_forward")
        ctx.save_for_backward(w, p)
        ctx.chunk = chunk
        return out

    @staticmethod
    def backward(ctx, g):
        lib = _lib()
        w, p = ctx.saved_tensors
        B, P, D = p.shape
        gd = _chk(g)
        dp, ds = torch.empty_like(p), torch.empty_like(w)
        L.check(lib.eg_tm_scale_backward(_ptr(w), _ptr(p), _ptr(gd), _ptr(dp), _ptr(ds), B, P, D, ctx.chunk, _stream(p.device)), "eg_tm_scale_backward")
        return ds, dp, None


def tm_memory_scale(score, pred, chunk: int):
    return _TMScale.apply(score, pred, int(chunk))


class _Reparam(torch.autograd.Function):
    """z = eps * exp(0.5 * logvar) + mu (MLP_Reconstruct_v3.reparameterize, CAVE/BEAT_CVAE.py:389-399); eps is an input."""

    @staticmethod
    def forward(ctx, mu, logvar, eps):
        m, lv, e = _chk(mu), _chk(logvar), _chk(eps)
        std = raw_ew(EW_EXP, lv, None, 0.5)
        es = raw_ew(EW_MUL, e, std)
        ctx.save_for_backward(es)
        return raw_ew(EW_ADD, es, m)

    @staticmethod
    def backward(ctx, dz):
        (es,) = ctx.saved_tensors
        d = _chk(dz)
        return d, raw_ew(EW_SCALE, raw_ew(EW_MUL, d, es), None, 0.5), None


def reparameterize(mu, logvar, eps):
    return _Reparam.apply(mu, logvar, eps)


class _KLD(torch.autograd.Function):
    @staticmethod
    def forward(ctx, mu, logvar, scale):
        lib = _lib()
        m, lv = _chk(mu), _chk(logvar)
        n, d = m.shape
        loss, dm, dl = torch.empty(1, device=m.device), torch.empty_like(m), torch.empty_like(lv)
        L.check(lib.eg_kld(_ptr(m), _ptr(lv), _ptr(loss), _ptr(dm), _ptr(dl), n, d, float(scale), _stream(m.device)), "eg_kld")
        ctx.save_for_backward(dm, dl)
        return loss

    @staticmethod
    def backward(ctx, g):
        dm, dl = ctx.saved_tensors
        gs = _chk(g).reshape(-1)            # the incoming scalar stays on the device (no host read-back: keeps the launch queue running ahead)
        return raw_ew(EW_SCALE_DEV, dm, gs), raw_ew(EW_SCALE_DEV, dl, gs), None


def kld_loss(mu, logvar, scale=1.0):
    """scale * mean_b(-0.5 * sum_j(1 + logvar - mu^2 - exp(logvar))): the standard VAE KL term."""
    return _KLD.apply(mu, logvar, scale)


class _Contrastive(torch.autograd.Function):
    """SoftmaxContrastiveLoss.forward (test_emotion_gesture_diversity_iterative.py:112-127, mode 'max') with its gradient."""

    @staticmethod
    def forward(ctx, face, audio):
        lib = _lib()
        f, a = _chk(face), _chk(audio)
        n, d = f.shape
        dev = f.device
        res = torch.empty(2, device=dev)
        nbytes = int(lib.eg_contrastive_workspace_bytes(n))
        ws = torch.empty(nbytes, dtype=torch.uint8, device=dev)
        L.check(lib.eg_contrastive_loss(_ptr(f), _ptr(a), n, d, None, _ptr(res[0:1]), _ptr(res[1:2]), _ptr(ws), nbytes, _stream(dev)), "eg_contrastive_loss")
        ctx.save_for_backward(f, a)
        return res[0:1].clone()

    @staticmethod
    def backward(ctx, g):
        lib = _lib()
        f, a = ctx.saved_tensors
        n, d = f.shape
        dev = f.device
        gf, ga = torch.empty_like(f), torch.empty_like(a)
        nbytes = int(lib.eg_contrastive_backward_workspace_bytes(n))
        ws = torch.empty(nbytes, dtype=torch.uint8, device=dev)
        L.check(lib.eg_contrastive_loss_backward(_ptr(f), _ptr(a), n, d, _ptr(gf), _ptr(ga), _ptr(ws), nbytes, _stream(dev)), "eg_contrastive_loss_backward")
        gs = _chk(g).reshape(-1)
        return raw_ew(EW_SCALE_DEV, gf, gs), raw_ew(EW_SCALE_DEV, ga, gs)


def contrastive_loss(face_feat, audio_feat):
    """SoftmaxContrastiveLoss()(face_feat, audio_feat, device) as a differentiable scalar ([1] tensor)."""
    if face_feat.dim() != 2 or face_feat.shape != audio_feat.shape:
        raise ValueError(f"contrastive_loss: expected two [n, d] tensors, got {tuple(face_feat.shape)} and {tuple(audio_feat.shape)}")
    return _Contrastive.apply(face_feat, audio_feat)


class _SmoothL1(torch.autograd.Function):
    @staticmethod
    def forward(ctx, pred, target, beta, scale):
        lib = _lib()
        p, t = _chk(pred), _chk(target)
        dev = p.device
        loss, dp = torch.empty(1, device=dev), torch.empty_like(p)
        ws = _scratch(dev, 1024, "loss")
        L.check(lib.eg_smooth_l1(_ptr(p), _ptr(t), _ptr(loss), _ptr(dp), p.numel(), float(beta), float(scale), _ptr(ws), _stream(dev)), "eg_smooth_l1")
        ctx.save_for_backward(dp)
        return loss

    @staticmethod
    def backward(ctx, g):
        (dp,) = ctx.saved_tensors
        return raw_ew(EW_SCALE_DEV, dp, _chk(g).reshape(-1)), None, None, None


def smooth_l1_loss(pred, target, beta=1.0, scale=1.0):
    """scale * F.smooth_l1_loss(pred, target, beta=beta) -- nn.HuberLoss(delta=1) has the same values."""
    return _SmoothL1.apply(pred, target, beta, scale)


class _CrossEntropy(torch.autograd.Function):
    @staticmethod
    def forward(ctx, logits, labels, alpha, gamma, scale):
        lib = _lib()
        z = _chk(logits)
        B, Cc = z.shape
        dev = z.device
        lab = labels.to(dev, torch.int64).contiguous()
        loss, dz = torch.empty(1, device=dev), torch.empty_like(z)
        ws = _scratch(dev, max(B, 64), "loss")
        a = _chk(alpha) if alpha is not None else None
        L.check(lib.eg_cross_entropy(_ptr(z), _ptr(lab), _ptr(a), float(gamma), float(scale), _ptr(loss), _ptr(dz), B, Cc, _ptr(ws), _stream(dev)),
                "eg_cross_entropy")
        ctx.save_for_backward(dz)
        return loss

    @staticmethod
    def backward(ctx, g):
        (dz,) = ctx.saved_tensors
        return raw_ew(EW_SCALE_DEV, dz, _chk(g).reshape(-1)), None, None, None, None


def cross_entropy(logits, labels, scale=1.0):
    """scale * nn.CrossEntropyLoss()(logits, labels)"""
    return _CrossEntropy.apply(logits, labels, None, -1.0, scale)


def focal_loss(logits, labels, alpha, gamma=2.0, scale=1.0):
    """scale * FocalLoss(alpha, gamma, 'mean') of train_audio_classifier_K_fold.py:89-105.  `alpha` multiplies the per-sample
    loss vector exactly as upstream's `self.alpha * (1-pt)**self.gamma * ce_loss` does: a scalar, or one weight per SAMPLE
    (upstream passes a list of 8 class weights, which only broadcasts at batch size 8 -- position-wise, not by label)."""
    B = logits.shape[0]
    if not torch.is_tensor(alpha):
        alpha = torch.tensor([float(alpha)] * B if not isinstance(alpha, (list, tuple)) else [float(a) for a in alpha])
    alpha = alpha.to(logits.device, torch.float32).reshape(-1)
    if alpha.numel() == 1:
        alpha = alpha.expand(B).contiguous()
    if alpha.numel() != B:
        raise ValueError(f"focal_loss: alpha has {alpha.numel()} entries for a batch of {B} (it multiplies the per-sample losses)")
    return _CrossEntropy.apply(logits, labels, alpha, float(gamma), scale)
