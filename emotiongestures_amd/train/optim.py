"""Optimiser state and data-parallel gradient reduction for the training path.

* `flatten_parameters(model)`: every trainable parameter becomes a view of ONE contiguous fp32 buffer, and its `.grad` a view of
  a second one (same offsets, 16-byte aligned slices), so that the optimiser is one kernel launch and the gradient all-reduce
  runs on a few large contiguous buckets (RCCL over xGMI is per-link bound: few large messages, MI355X-first).
* `FlatAdam`: torch.optim.Adam's update (betas, eps, L2 weight decay added to the gradient; train_audio_classifier_K_fold.py:128
  uses lr, betas=(0.5, 0.999), weight_decay=1e-5) as one fused HIP kernel over the flat buffers (eg_adam_step).
* `GradBuckets`: clip-level data parallelism: gradients are summed across ranks with `all_reduce` on ~25 MB
  contiguous buckets of the flat gradient buffer, launched from the LAST parameters backwards (the order backward produces
  them) on a side stream so that the reduction of finished buckets overlaps the rest of the backward pass; the sum is then
  divided by the world size with one HIP elementwise launch over the flat gradient (`_scale`).  BatchNorm statistics stay per
  replica, as with the reference's plain nn.BatchNorm under nn.DataParallel.
"""
from __future__ import annotations

import os
from typing import List, Optional

import torch

from .. import _lib as L
from ..engine import _ptr, _stream


class FlatParams:
    def __init__(self, params: List[torch.nn.Parameter], flat: torch.Tensor, grad: torch.Tensor, offsets: List[int]):
        self.params, self.flat, self.grad, self.offsets = params, flat, grad, offsets
        self.index = {id(p): i for i, p in enumerate(params)}
        self.has_grad = [False] * len(params)       # set by collect_one: parameters autograd produced a gradient for
        self.images: Optional[WeightImages] = None  # enable_weight_images(): all weight images of a step in one launch
        self.buffers: List[torch.Tensor] = []       # the model's buffers (BatchNorm running statistics): written by kernels, versions bumped with the parameters
        self.written = set()                        # parameters whose flat gradient slice a backward kernel already wrote this step
        self._dirty, self._zeroed = set(), False        # flatten_parameters() allocates the buffer zeroed and says so
        for p, o in zip(params, offsets):           # the operators of train/functional.py write weight gradients straight into these slices
            p._eg_slot = grad[o:o + p.numel()].view(p.shape)
            p._eg_fp = self

    def zero_grad(self):
        """Gradients start from None: autograd then hands each parameter a fresh gradient tensor (no torch add into an old one);
        `collect` / the bucket hooks copy it into the flat buffer."""
        for p in self.params:
            p.grad = None
        self.written.clear()

    def enable_weight_images(self, groups=(), skip=()):
        """Keep the split-bf16 weight images of all Linear / conv3x3 parameters resident and refresh them once per optimiser step.
        groups / skip: the fused-product plan of the model (`nets.weight_image_plan(model)`), see WeightImages."""
        from . import functional as F
        plan = (tuple(tuple(id(p) for p in g) for g in groups), tuple(id(p) for p in skip))
        if self.images is None:
            self.images = WeightImages(self, groups, skip)
            self._image_plan = plan
        elif plan != self._image_plan:
            raise L.EgError("enable_weight_images: called again with a different groups / skip plan (the images are built once per FlatParams)")
        F.register_weight_images(self.images)
        return self.images

    def bump_versions(self, images_fresh: bool = True):
        """The optimiser and the BatchNorm kernels write parameters / running statistics through raw pointers: tell torch (`_version`) so
        that everything keyed on it -- the inference engines' packed-weight caches -- sees the change.  images_fresh: the caller rebuilt the
        weight images from the values just written (False: they are stale now; lookups miss until the next refresh)."""
        for t, updated in zip(self.params, self.has_grad):
            if updated:                             # parameters without a gradient are skipped by the optimiser (torch.optim.Adam semantics): unchanged
                torch.autograd.graph.increment_version(t)
        for t in self.buffers:
            torch.autograd.graph.increment_version(t)
        if self.images is not None and images_fresh:
            self.images.mark_fresh()                # the caller refreshed them from the values just written (FlatAdam.step / the captured graph)

    def collect_one(self, p: torch.nn.Parameter):
        o = self.offsets[self.index[id(p)]]
        n = p.numel()
        sl = self.grad[o:o + n]
        i = self.index[id(p)]
        if p.grad is None:
            if i in self._dirty or not self._zeroed:     # the flat buffer starts zeroed and nothing writes the slices of gradient-less parameters:
                sl.zero_()                               # zero again only after a step that did write this one (dynamic graphs)
                self._dirty.discard(i)
            self.has_grad[i] = False
            return
        self.has_grad[i] = True
        self._dirty.add(i)
        if p.grad.data_ptr() != sl.data_ptr():
            sl.copy_(p.grad.reshape(-1))                 # data movement into the flat buffer
            p.grad = sl.view(p.shape)

    def collect(self):
        for p in self.params:
            self.collect_one(p)


def flatten_parameters(model: torch.nn.Module) -> FlatParams:
    """Re-home every trainable parameter inside one contiguous buffer (values preserved) and allocate the matching flat
    gradient buffer (same offsets, 16-byte aligned slices)."""
    params = [p for p in model.parameters() if p.requires_grad]
    if not params:
        raise ValueError("no trainable parameters")
    dev = params[0].device
    offsets, total = [], 0
    for p in params:
        offsets.append(total)
        total += (p.numel() + 3) // 4 * 4                 # 16-byte aligned slices
    flat = torch.zeros(total, dtype=torch.float32, device=dev)
    grad = torch.zeros(total, dtype=torch.float32, device=dev)
    for p, o in zip(params, offsets):
        n = p.numel()
        flat[o:o + n].copy_(p.detach().reshape(-1))
        p.data = flat[o:o + n].view(p.shape)
        p.grad = None
    fp = FlatParams(params, flat, grad, offsets)
    fp._zeroed = True
    fp.buffers = [b for b in model.buffers() if b.is_floating_point()]
    return fp


class WeightImages:
    """The split-bf16 weight images of every Linear ([n, k] and its transpose) and 3x3 convolution (filter and its rotated transpose) of
    a flattened model, refreshed with ONE launch per optimiser step (eg_pack_table) instead of one launch per use: parameters and images
    live at fixed addresses, so the table is built once.  `train/functional.py` looks an image up by (storage address, orientation) and
    checks the parameter's version; a miss (reshaped weights, a stale version) falls back to the per-use packer."""

    def __init__(self, fp: "FlatParams", groups=(), skip=()):
        """groups: tuples of 2-D parameters that lie back to back in the flat buffer and are used as ONE matrix (the fused Q|K|V / K|V products of
        train/functional.py: `nets.weight_image_plan(model)`); skip: parameters that are only ever used through a group (no image of their own)."""
        import numpy as np
        lib = L.load()
        dev = fp.flat.device
        if dev.type != "cuda":
            raise L.EgError("WeightImages: needs a GPU")
        from . import functional as F
        # built while the step computes in a split-bf16 mode: the refresh leaves the fp32 head of each image unwritten (flag bit 1 of the table entry,
        # csrc/train.hip pack_table_kernel) -- half the bytes of the launch -- and an fp32 lookup later misses (functional._ImageRegistry.lookup)
        self.bf16_only = F.get_precision() != "f32"
        entries, self.images, self.params = [], {}, []
        first = 0
        skip_ids = {id(p) for p in skip}

        def add(ptr, owners, kinds):
            nonlocal first
            for kind, a, b, c, flag, floats, rows in kinds:
                # zeros, not empty: in bf16_only mode the refresh never writes the fp32 head; a consumer that reads it by mistake must get zeros
                # (a loud all-zero result), not whatever the allocator left there
                img = torch.zeros(floats, dtype=torch.float32, device=dev)
                self.images[(ptr, kind, flag, rows)] = (img, owners)
                entries.append((ptr, img.data_ptr(), kind, a, b, c, flag | (2 if self.bf16_only else 0), first))
                first += int(lib.eg_pack_table_blocks(kind, a, b, flag))

        for p in fp.params:
            if id(p) in skip_ids:
                continue
            kinds = []
            if p.dim() == 2 or (p.dim() == 4 and tuple(p.shape[2:]) == (1, 1)):        # nn.Linear, or a 1x1 convolution used as one
                n, k = p.shape[:2]
                kinds = [(0, n, k, k, 0, int(lib.eg_linear_packed_floats(n, k)), n), (0, k, n, k, 1, int(lib.eg_linear_packed_floats(k, n)), n)]
            elif p.dim() == 4 and tuple(p.shape[2:]) == (3, 3):
                co, ci = p.shape[:2]
                if ci % 8 == 0:
                    kinds.append((1, co, ci, 0, 0, int(lib.eg_conv3x3_packed_floats(ci, (co + 15) // 16 * 16)), co))
                if co % 8 == 0:
                    kinds.append((1, co, ci, 0, 1, int(lib.eg_conv3x3_packed_floats(co, (ci + 15) // 16 * 16)), co))
            add(p.data_ptr(), (p,), kinds)
            if kinds:
                self.params.append(p)
        for grp in groups:
            k = grp[0].shape[1]
            ptr = grp[0].data_ptr()
            for q in grp:
                if q.dim() != 2 or q.shape[1] != k or q.data_ptr() != ptr:
                    raise ValueError("WeightImages: a fused group must be 2-D parameters of one width lying back to back in the flat buffer")
                ptr += q.numel() * 4
            n = sum(q.shape[0] for q in grp)
            add(grp[0].data_ptr(), tuple(grp), [(0, n, k, k, 0, int(lib.eg_linear_packed_floats(n, k)), n), (0, k, n, k, 1, int(lib.eg_linear_packed_floats(k, n)), n)])
            for q in grp:
                if all(q is not r for r in self.params):
                    self.params.append(q)
        self.count, self.total_blocks = len(entries), first
        rec = np.zeros(self.count, dtype=np.dtype([("src", "<u8"), ("img", "<u8"), ("kind", "<i4"), ("a", "<i4"), ("b", "<i4"), ("c", "<i4"), ("flag", "<i4"),
                                                   ("first", "<i4")]))
        for i, e in enumerate(entries):
            rec[i] = e
        assert rec.dtype.itemsize == 40
        self.table = torch.from_numpy(rec.view(np.uint8).reshape(-1).copy()).to(dev)
        self.versions = {}
        self.refresh()

    def refresh(self):
        """Rebuild every image from the current parameter values (one launch) and remember the versions they correspond to."""
        if self.count:
            L.check(L.load().eg_pack_table(_ptr(self.table), self.count, self.total_blocks, _stream(self.table.device)), "eg_pack_table")
        self.mark_fresh()

    def mark_fresh(self):
        self.versions = {id(p): p._version for p in self.params}

    def lookup(self, w: torch.Tensor, kind: int, flag: int):
        hit = self.images.get((w.data_ptr(), kind, flag, w.shape[0]))
        if hit is None:
            return None
        img, owners = hit
        p = owners[0]
        if len(owners) == 1:
            same = tuple(w.shape) == tuple(p.shape) or (p.dim() == 4 and tuple(p.shape[2:]) == (1, 1) and tuple(w.shape) == tuple(p.shape[:2]))
        else:               # a fused group: the [sum n_i, k] matrix over its members
            same = w.dim() == 2 and w.shape[1] == p.shape[1] and w.shape[0] == sum(q.shape[0] for q in owners) and w.is_contiguous()
        if not same or any(self.versions.get(id(q)) != q._version for q in owners):
            return None                 # a different view of that storage, or a parameter changed since the last refresh
        return img


class FlatAdam:
    def __init__(self, fp: FlatParams, lr=1e-3, betas=(0.9, 0.999), eps=1e-8, weight_decay=0.0):
        if not fp.flat.is_cuda:
            raise L.EgError("FlatAdam: the optimiser kernel runs only on a GPU (no CPU fallback)")
        self.fp, self.lr, self.betas, self.eps, self.weight_decay = fp, lr, betas, eps, weight_decay
        self.exp_avg = torch.zeros_like(fp.flat)
        self.exp_avg_sq = torch.zeros_like(fp.flat)
        self.t = 0
        self.step_dev = None        # device-resident step count (train/graph.py: steps replayed from a captured hipGraph)

    def use_device_step(self):
        """Keep the step count on the device from now on: `step()` increments it with a kernel and the update reads it there."""
        if self.step_dev is None:
            self.step_dev = torch.tensor([self.t], dtype=torch.int32, device=self.fp.flat.device)

    def step(self, collected: bool = False):
        fp = self.fp
        if not collected:
            fp.collect()
        self.t += 1
        lib = L.load()
        if self.step_dev is not None:
            L.check(lib.eg_counter_add(_ptr(self.step_dev), 1, _stream(fp.flat.device)), "eg_counter_add")
        # torch.optim.Adam skips parameters whose .grad is None (no weight decay, no moment update): one launch per contiguous
        # run of parameters that did receive a gradient (a handful: the text branch / unused decoder self-attention are skipped)
        runs, lo = [], None
        for i, (p, o) in enumerate(zip(fp.params, fp.offsets)):
            if fp.has_grad[i]:
                if lo is None:
                    lo = o
                hi = o + (p.numel() + 3) // 4 * 4
            elif lo is not None:
                runs.append((lo, hi))
                lo = None
        if lo is not None:
            runs.append((lo, hi))
        for lo, hi in runs:
            hi = min(hi, fp.flat.numel())
            sl = slice(lo, hi)
            if self.step_dev is not None:
                L.check(lib.eg_adam_step_dev(_ptr(fp.flat[sl]), _ptr(fp.grad[sl]), _ptr(self.exp_avg[sl]), _ptr(self.exp_avg_sq[sl]), hi - lo,
                                             float(self.lr), float(self.betas[0]), float(self.betas[1]), float(self.eps), float(self.weight_decay),
                                             _ptr(self.step_dev), _stream(fp.flat.device)), "eg_adam_step_dev")
                continue
            L.check(lib.eg_adam_step(_ptr(fp.flat[sl]), _ptr(fp.grad[sl]), _ptr(self.exp_avg[sl]), _ptr(self.exp_avg_sq[sl]), hi - lo,
                                     float(self.lr), float(self.betas[0]), float(self.betas[1]), float(self.eps), float(self.weight_decay),
                                     self.t, _stream(fp.flat.device)), "eg_adam_step")
        from . import functional as F
        refresh = fp.images is not None and F.get_precision() != "f32"      # the fp32 operators never read the images
        if refresh:
            fp.images.refresh()                     # one launch: every weight image of the next step, from the values just written
        fp.bump_versions(images_fresh=refresh)      # not refreshed: a later bf16x3 use sees a stale version and packs its own image

    def zero_grad(self):
        self.fp.zero_grad()


def stage_splits(model, fp: FlatParams):
    """Flat offsets at which the gradient buckets of a generator should be cut for a backward segmented per tower stage (nets.CutContext names
    "tower", "layer3", "layer2"): the first parameter of layer2, of layer3, of what follows layer3 inside the audio encoder (final_conv1 ...), and of
    the first module after the audio encoder.  Parameters are flattened in registration order, the tower first, so these are the phase boundaries
    of the backward read from the end of the buffer."""
    ae = getattr(model, "audio_encoder", None)
    if ae is None:
        for m in model.modules():
            if hasattr(m, "audio_encoder"):
                ae = m.audio_encoder
                break
    if ae is None:
        return []
    fe = ae.feat_extractor
    firsts = []
    for mod in (fe.layer2, fe.layer3):
        ps = [p for p in mod.parameters() if id(p) in fp.index]
        if ps:
            firsts.append(min(fp.offsets[fp.index[id(p)]] for p in ps))
    tower = [fp.offsets[fp.index[id(p)]] for p in fe.parameters() if id(p) in fp.index]
    enc = [fp.offsets[fp.index[id(p)]] for p in ae.parameters() if id(p) in fp.index]
    for group in (tower, enc):          # first parameter behind the tower / behind the whole audio encoder
        later = [o for o in fp.offsets if group and o > max(group)]
        if later:
            firsts.append(min(later))
    return sorted(set(o for o in firsts if o > 0))


class GradBuckets:
    """Bucketed gradient all-reduce over the flat gradient buffer (sum across ranks, then scale by 1/world)."""

    def __init__(self, fp: FlatParams, bucket_mb: float = 25.0, group=None, split_at=()):
        """split_at: flat offsets (parameter starts) at which a bucket must end even when it is short of `bucket_mb` -- the phase boundaries of a
        segmented backward (train/graph.SegmentedStep with several cuts: a bucket that spanned two phases would only be complete, and reduced,
        after the later one).  `stage_splits(model, fp)` returns the audio tower's."""
        import torch.distributed as dist
        self.fp, self.group = fp, group
        self.world = dist.get_world_size(group) if dist.is_initialized() else 1
        # the collectives are issued when there is someone to reduce with -- or, EG_FORCE_COLLECTIVES=1 with an initialised process group, even at
        # world 1: every all_reduce / staging conversion / stream hand-off of the data-parallel step then really runs (over RCCL on a GPU box),
        # which proves the issue order and the bf16 staging on one GPU before an 8-GPU node does (sum over one rank and x 1/1 are identities)
        self.active = self.world > 1 or (dist.is_initialized() and os.environ.get("EG_FORCE_COLLECTIVES") == "1")
        per = max(1, int(bucket_mb * (1 << 20) / 4))
        n = fp.grad.numel()
        forced = set(int(o) for o in split_at)
        unknown = forced - set(fp.offsets)
        if unknown:
            raise ValueError(f"GradBuckets: split_at offsets {sorted(unknown)} are not parameter starts")
        # bucket boundaries on parameter boundaries, walking from the LAST parameter (whose gradient is ready first) backwards
        self.buckets = []
        end = n
        for o in reversed(fp.offsets):
            if end - o >= per or (o in forced and end > o):
                self.buckets.append((o, end))
                end = o
        if end > 0:
            self.buckets.append((0, end))
        self.param_bucket = [self.bucket_of(o) for o in fp.offsets]
        self.members = [sum(1 for b in self.param_bucket if b == i) for i in range(len(self.buckets))]
        self.stream = torch.cuda.Stream(fp.grad.device) if fp.grad.is_cuda else None
        self._pending: Optional[List[int]] = None
        self._handles = []
        self._unused: Optional[set] = None      # parameters that received no gradient in the first backward (static graph)
        self.launched: List[int] = []           # bucket launch order of the last backward (tests)
        self.deferred = False                   # True: the hooks only collect (a captured hipGraph holds forward + backward); reduce_deferred() follows the replay
        self.payload = "f32"                    # "bf16": buckets travel as bfloat16 (reduce_bucket)
        self._stage = {}

    def bucket_of(self, offset: int) -> int:
        for i, (lo, hi) in enumerate(self.buckets):
            if lo <= offset < hi:
                return i
        raise IndexError(offset)

    # ---- overlapped mode: hooks launch a bucket's all-reduce as soon as its last gradient has been produced ----
    def attach(self):
        """Register post-accumulate-grad hooks: each parameter's fresh gradient is copied into the flat buffer; when a bucket
        is complete its all-reduce starts on the side stream while backward continues (call `begin()` before every backward
        and `finish()` after it)."""
        for p in self.fp.params:
            p.register_post_accumulate_grad_hook(self._hook)
        return self

    def begin(self):
        self._pending = list(self.members)
        self._seen = set()
        self._handles = []
        self.launched = []
        if self._unused:            # known from the first step: their (zero) slices are ready before backward starts
            for i in self._unused:
                self.fp.collect_one(self.fp.params[i])
                self._seen.add(i)
                self._pending[self.param_bucket[i]] -= 1

    def _hook(self, p):
        if self._pending is None:
            return
        self.fp.collect_one(p)
        i = self.fp.index[id(p)]
        self._seen.add(i)
        b = self.param_bucket[i]
        self._pending[b] -= 1
        if self._pending[b] == 0:
            self._launch(b)

    def _launch(self, b):
        import torch.distributed as dist
        self.launched.append(b)
        if not self.active or self.deferred:
            return
        lo, hi = self.buckets[b]
        g = self.fp.grad[lo:hi]
        if self.stream is not None:
            self.stream.wait_stream(torch.cuda.current_stream(g.device))
            with torch.cuda.stream(self.stream):
                self.reduce_bucket(b)
        elif self.payload == "bf16":
            self.reduce_bucket(b)
        else:
            self._handles.append(dist.all_reduce(g, op=dist.ReduceOp.SUM, group=self.group, async_op=True))

    def finish(self):
        """After backward: parameters that received no gradient are zero-filled, their buckets reduced, everything scaled by
        1/world; the caller's stream then owns a complete averaged flat gradient (`FlatAdam.step(collected=True)`)."""
        missing = [i for i in range(len(self.fp.params)) if i not in self._seen]
        if self._unused is None:
            self._unused = set(missing)                 # the networks here have a static graph (e.g. the text branch never gets a gradient)
        for i in missing:
            self.fp.collect_one(self.fp.params[i])      # p.grad is None here: zero slice
            self._pending[self.param_bucket[i]] -= 1
        for b, left in enumerate(self._pending):
            if left == 0 and b not in self.launched:
                self._launch(b)
        for h in self._handles:
            h.wait()
        if self.stream is not None:
            torch.cuda.current_stream(self.fp.grad.device).wait_stream(self.stream)
        self._pending = None
        if not self.deferred:
            self._scale()

    def reduce_deferred(self):
        """After a replayed forward + backward (deferred mode): the bucket all-reduces in backward order on the caller's stream, then 1/world."""
        if self.active:
            for b in range(len(self.buckets)):
                self.reduce_bucket(b)
        self._scale()

    def reduce_bucket(self, b: int):
        """all_reduce(SUM) of bucket b on the CURRENT stream (train/graph.SegmentedStep issues it on its side stream while the next graph
        segment replays).  `payload = "bf16"`: the bucket travels as bfloat16 -- half the bytes on the xGMI links -- through a staging buffer
        (fp32 -> bf16 -> all_reduce -> fp32: two HIP conversion launches; the sum itself is then taken in bf16 by the collective: every
        gradient element carries a relative error of ~2^-9 per addend, measured in tests/test_gpu_training.py)."""
        import torch.distributed as dist
        if not self.active:
            return
        lo, hi = self.buckets[b]
        g = self.fp.grad[lo:hi]
        if self.payload != "bf16":
            dist.all_reduce(g, op=dist.ReduceOp.SUM, group=self.group)
            return
        st = self._stage.get(b)
        if st is None:
            st = self._stage[b] = torch.empty(hi - lo, dtype=torch.bfloat16, device=g.device)
        if g.is_cuda:
            lib = L.load()
            L.check(lib.eg_f32_to_bf16(_ptr(g), _ptr(st), hi - lo, _stream(g.device)), "eg_f32_to_bf16")
            dist.all_reduce(st, op=dist.ReduceOp.SUM, group=self.group)
            L.check(lib.eg_bf16_to_f32(_ptr(st), _ptr(g), hi - lo, 1.0, _stream(g.device)), "eg_bf16_to_f32")
        else:                                               # CPU only exists for the gloo test of the bucket logic
            st.copy_(g)
            dist.all_reduce(st, op=dist.ReduceOp.SUM, group=self.group)
            g.copy_(st)

    # ---- simple mode: everything after backward ----
    def all_reduce(self):
        import torch.distributed as dist
        self.fp.collect()
        if self.active:
            for lo, hi in self.buckets:
                dist.all_reduce(self.fp.grad[lo:hi], op=dist.ReduceOp.SUM, group=self.group)
        self._scale()

    def _scale(self):
        if not self.active:
            return
        g = self.fp.grad
        if g.is_cuda:
            lib = L.load()
            L.check(lib.eg_elementwise(_ptr(g), None, _ptr(g), g.numel(), 5, 1.0 / self.world, _stream(g.device)), "eg_elementwise")
        else:
            g.mul_(1.0 / self.world)                    # CPU only exists for the gloo test of the bucket logic
