"""Whole training step as one captured hipGraph.

A training step is ~2 300 small launches issued through autograd; below ~64 clips per GPU the host (Python + launch calls, ~11 us each)
is slower than the GPU.  `GraphedStep` captures forward + backward + gradient collection + Adam once (torch.cuda.graph: stream capture,
allocations from a private pool) and replays it per step: HIP graphs instead of a tracing compiler, the MI355X-native answer to a
launch-bound loop.  Everything that varies between steps lives in device memory: the inputs (static tensors refreshed with `copy_`), the
Adam step count (`FlatAdam.use_device_step`), BatchNorm's running statistics, and -- `stochastic=True`, for models that train with
`train_dropout` -- the dropout mask epoch (`functional.use_device_dropout_epoch`: a counter the captured step increments itself, so every
replay draws a fresh mask although the host-side (seed, offset) scalars are frozen at capture).  A dropout call inside a capture without
that epoch raises (functional.dropout).
"""
from __future__ import annotations

from typing import Callable, Dict

import torch

from .. import _lib as L
from . import functional as F
from .optim import FlatAdam
from ..pipeline import CAPTURE_MODE          # "thread_local": the RCCL watchdog thread may poll its events while this thread captures


def _drop_stream_scratch(stream) -> None:
    if stream is None:
        return
    sid = stream.cuda_stream
    for k in [k for k in F._WS if k[2] == sid]:
        del F._WS[k]


class GraphedStep:
    def __init__(self, step_fn: Callable[[Dict[str, torch.Tensor]], torch.Tensor], inputs: Dict[str, torch.Tensor], optimizer: FlatAdam = None,
                 warmup: int = 3, device=None, stochastic: bool = False):
        """step_fn(inputs) runs zero_grad, forward, backward, the gradient collection and -- when `optimizer` is given -- optimizer.step(),
        and returns the loss tensor.  `inputs` are the static device tensors the captured graph reads; `run(new_inputs)` copies fresh
        values into them.  Data parallel: capture forward + backward + collection only (optimizer=None, GradBuckets.deferred = True) and run
        `GradBuckets.reduce_deferred()` and the optimiser after each replay -- the collective stays outside the captured graph."""
        dev = optimizer.fp.flat.device if optimizer is not None else torch.device(device)
        if dev.type != "cuda":
            raise L.EgError("GraphedStep: needs a GPU")
        self._epoch_keep = None
        if stochastic:                              # the model trains with dropout: the mask epoch moves to the device BEFORE the warm-up steps
            self._epoch_keep = F.use_device_dropout_epoch(dev)     # the captured kernels hold its raw pointer: owned by the step, whatever reset_state() does later
        self.inputs, self.opt = inputs, optimizer
        if optimizer is not None:
            optimizer.use_device_step()
        if stochastic:                              # one mask epoch per STEP (not per forward): every forward / backward of the step reads the same value
            user_step = step_fn

            def step_fn(i):
                F.begin_dropout_step()
                return user_step(i)
        # Warm-up and capture run on the SAME side stream: functional._scratch keys its buffers by stream, so the warm-up sizes exactly the
        # buffers the capture then bakes into the graph (captured on another stream, every scratch buffer would be allocated inside the capture
        # and the warm-up's copies would stay behind under a dead key).
        side = self._stream = torch.cuda.Stream(dev)
        side.wait_stream(torch.cuda.current_stream(dev))
        with torch.cuda.stream(side):               # warm-up off the capture: lazily sized scratch buffers, kernel attributes, autograd state
            for _ in range(max(1, warmup)):
                self.warmup_loss = step_fn(inputs)  # real steps on `inputs`: a loop that must not train a batch twice takes this as its iteration
        torch.cuda.current_stream(dev).wait_stream(side)
        torch.cuda.synchronize(dev)
        self.graph = torch.cuda.CUDAGraph()
        with torch.cuda.graph(self.graph, stream=side, capture_error_mode=CAPTURE_MODE):
            self.loss = step_fn(inputs)
        if optimizer is not None:
            optimizer.t -= 1                        # the capture recorded a step without executing it
        # The captured kernels hold raw pointers into the shared scratch buffers of train/functional.py (split-K / column-reduction / wgrad
        # partials).  A later, larger eager request replaces such a buffer in the registry; keep the ones this graph was captured with alive
        # for as long as the graph exists, or the caching allocator would hand their memory to someone else under a replay.
        self._scratch_keep = dict(F._WS)
        self._images_refreshed = F.get_precision() != "f32"        # FlatAdam.step refreshes the weight images only for the split-bf16 operators

    def close(self) -> None:
        """Drop the graph and the scratch buffers registered under this step's stream (a loop that builds one step per fold / leg calls this, or
        the process-wide registry keeps every dead step's split-K / weight-gradient partials)."""
        self.graph = None
        self._scratch_keep = None
        _drop_stream_scratch(self._stream)

    def run(self, new_inputs: Dict[str, torch.Tensor] = None) -> torch.Tensor:
        if new_inputs:
            for k, v in new_inputs.items():
                self.inputs[k].copy_(v, non_blocking=True)
        self.graph.replay()
        if self.opt is not None:
            self.opt.t += 1
            self.opt.fp.bump_versions(images_fresh=self._images_refreshed)      # the replay rewrote parameters / running statistics behind torch's back
        return self.loss


class SegmentedStep:
    """A data-parallel training step as a few captured hipGraph segments with the bucket all-reduces BETWEEN them.

    With more than one rank a single graph of forward + backward leaves the whole gradient reduction exposed behind the replay.  Here the
    backward is cut at the tower output and at the inputs of the tower's stages (train/nets.CutContext; `cuts`): segment 0 = forward + the
    backward of everything behind the audio tower (post-projector, decoder, encoder, heads, projections: ~94 % of the gradient bytes, ~40 % of
    the backward time); segment 1 = final_conv1 + layer3's backward, segment 2 = layer2's, segment 3 = layer1's and the stem's.  After a segment
    replays, the buckets it completed are all-reduced on a side stream WHILE the next segment replays on the main stream; only what the LAST
    segment completes (layer1 + stem: 0.2 MB) is reduced with nothing left to hide it.  For that the buckets must end at the phase boundaries:
    build them with `GradBuckets(fp, split_at=optim.stage_splits(model, fp))`.  Then the tail graph (1/world scaling, fused Adam, weight-image
    refresh).  The collectives stay outside
    the graphs (RCCL launches issued by torch.distributed on the side stream), so the same code runs over gloo in the tests.

    loss_fn() runs the train-mode forward and returns the loss tensor (no zero_grad / backward / optimiser calls inside).
    use_graphs=False issues the same phases eagerly (the CPU test of the bucket ordering; also the fallback while debugging).
    stochastic=True: the model's dropout is on (`train_dropout`): the mask epoch moves to the device before the warm-up, so that every replay
    of the segments draws fresh masks (forward and backward of one step share the epoch: it advances once per step, in segment 0).
    after_backward: see __init__."""

    def __init__(self, loss_fn, buckets, optimizer, device=None, cuts=("tower", "layer3", "layer2"), warmup: int = 3, use_graphs: bool = True, stochastic: bool = False,
                 after_backward=None):
        from . import nets
        self.loss_fn, self.gb, self.opt, self.nets = loss_fn, buckets, optimizer, nets
        # after_backward(step): called at the end of phase 0, behind loss.backward() -- where a loss_fn that forked part of the step onto another stream
        # (bench.py: the emotion CVAE's forward + backward beside the generator's) joins it again, inside the captured segment; it may replace step.loss
        self.after_backward = after_backward
        self.ctx = nets.CutContext(cuts)
        self.use_graphs = bool(use_graphs)
        self.dev = torch.device(device) if device is not None else buckets.fp.grad.device
        self._epoch_keep = None
        if stochastic and self.use_graphs and self.dev.type == "cuda":       # the model trains with dropout: device-resident mask epoch, as in GraphedStep
            self._epoch_keep = F.use_device_dropout_epoch(self.dev)         # owned by the step: the captured kernels hold its raw pointer
        self.gb.deferred = True                                   # hooks only collect and record completion order; this class issues the collectives
        self.side = torch.cuda.Stream(self.dev) if self.dev.type == "cuda" else None
        self.ready = []                                           # per phase: buckets completed by that phase (recorded at the first run)
        self.loss = None
        self.graphs = None
        if self.use_graphs:
            if self.dev.type != "cuda":
                raise L.EgError("SegmentedStep(use_graphs=True): needs a GPU")
            optimizer.use_device_step()
            warm = self._stream = torch.cuda.Stream(self.dev)        # warm-up AND capture stream (scratch buffers are keyed by stream, see GraphedStep)
            warm.wait_stream(torch.cuda.current_stream(self.dev))
            with torch.cuda.stream(warm):
                for _ in range(max(1, warmup)):
                    self._eager_step()
            torch.cuda.current_stream(self.dev).wait_stream(warm)
            torch.cuda.synchronize(self.dev)
            self._check_overlap()
            self._capture()
        self.n_segments = (len(self.graphs) - 1) if self.graphs else None

    def _check_overlap(self):
        """The overlap this class exists for needs buckets that END at the phase boundaries: with buckets that span phases
        (`GradBuckets(fp)` without `split_at=optim.stage_splits(model, fp)`) a non-last phase completes no bucket, its segment boundary buys
        nothing, and every reduction lands in the exposed join.  The step stays correct; say so once."""
        idle = [i for i, r in enumerate(self.ready[:-1]) if not r]
        if idle and self.gb.active:
            import warnings
            warnings.warn(f"SegmentedStep: backward phase(s) {idle} of {len(self.ready)} complete no gradient bucket, so their all-reduces cannot overlap the "
                          "next segment; build the buckets with GradBuckets(fp, split_at=optim.stage_splits(model, fp)) or pass fewer cuts",
                          RuntimeWarning, stacklevel=3)
        self.idle_phases = idle

    def close(self) -> None:
        self.graphs = None
        self._scratch_keep = None
        _drop_stream_scratch(getattr(self, "_stream", None))

    # ---- the phases -------------------------------------------------------------------------------------------------
    def _phase0(self):
        F.begin_dropout_step()                                    # device-epoch mode only: one mask epoch per step, advanced here and nowhere else
        self.opt.zero_grad()
        self.gb.begin()
        self.ctx.reset()
        self.nets._CUTS["ctx"] = self.ctx
        try:
            self.loss = self.loss_fn()
        finally:
            self.nets._CUTS["ctx"] = None
        self.loss.backward()
        if self.after_backward is not None:
            self.after_backward(self)
        return list(self.gb.launched)

    def _phase_cut(self, i):
        before = len(self.gb.launched)
        _name, x, leaf = self.ctx.cuts[len(self.ctx.cuts) - 1 - i]
        x.backward(leaf.grad)
        return list(self.gb.launched[before:])

    def _phase_last_collect(self):
        before = len(self.gb.launched)
        self.gb.finish()                                          # deferred: zero-fills gradient-less slices, completes their buckets, no collective, no scaling
        return list(self.gb.launched[before:])

    def _tail(self):
        self.gb._scale()
        self.opt.step(collected=True)

    def _reduce(self, bucket_ids):
        for b in bucket_ids:
            self.gb.reduce_bucket(b)

    def _eager_step(self, exposed=None):
        cur = torch.cuda.current_stream(self.dev) if self.side is not None else None
        ready = [self._phase0()]
        n_cuts = len(self.ctx.cuts)
        for i in range(n_cuts):
            self._overlap(ready[-1], cur)
            ready.append(self._phase_cut(i))
        ready[-1] = ready[-1] + self._phase_last_collect()
        self._join(ready[-1], cur, exposed)
        self._tail()
        self.ready = ready
        return self.loss

    def _overlap(self, bucket_ids, cur):
        """Reduce `bucket_ids` on the side stream; the caller goes on issuing the next phase on the main stream."""
        if self.side is None:
            self._reduce(bucket_ids)
            return
        self.side.wait_stream(cur)
        with torch.cuda.stream(self.side):
            self._reduce(bucket_ids)

    def _join(self, bucket_ids, cur, exposed):
        """After the last phase: wait for the overlapped reductions, reduce what the last phase completed.  `exposed`: list receiving an event
        pair around this (the part of the gradient reduction that is NOT hidden behind compute)."""
        ev = None
        if exposed is not None and self.side is not None:
            ev = (torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True))
            ev[0].record()
        # The last phase's buckets are reduced on the SIDE stream as well, never on the caller's: with async_op=False torch.distributed enqueues the
        # collective on the current stream and its watchdog thread later queries the work's end event -- an event last recorded on the stream this
        # class captures its graphs on makes that query fail while a capture is open ("operation not permitted on an event last recorded in a
        # capturing stream": 3 of 10 forced 1-rank RCCL runs aborted in round 5 before this).
        if self.side is not None:
            self.side.wait_stream(cur)
            with torch.cuda.stream(self.side):
                self._reduce(bucket_ids)
            cur.wait_stream(self.side)
        else:
            self._reduce(bucket_ids)
        if ev is not None:
            ev[1].record()
            exposed.append(ev)

    def exposed_bytes(self) -> int:
        """Gradient bytes whose all-reduce has no later segment to hide behind: the buckets completed by the last phase."""
        if not self.ready:
            return 0
        return int(sum(4 * (self.gb.buckets[b][1] - self.gb.buckets[b][0]) for b in self.ready[-1]))

    # ---- capture / replay -------------------------------------------------------------------------------------------
    def _capture(self):
        pool = torch.cuda.graph_pool_handle()
        graphs, ready = [], []
        g0 = torch.cuda.CUDAGraph()
        with torch.cuda.graph(g0, pool=pool, stream=self._stream, capture_error_mode=CAPTURE_MODE):
            ready.append(self._phase0())
        graphs.append(g0)
        n_cuts = len(self.ctx.cuts)
        for i in range(n_cuts):
            gi = torch.cuda.CUDAGraph()
            with torch.cuda.graph(gi, pool=pool, stream=self._stream, capture_error_mode=CAPTURE_MODE):
                r = self._phase_cut(i)
                if i == n_cuts - 1:
                    r = r + self._phase_last_collect()
            graphs.append(gi)
            ready.append(r)
        if n_cuts == 0:
            ready[-1] = ready[-1] + self._phase_last_collect_captured(pool, graphs)
        gt = torch.cuda.CUDAGraph()
        with torch.cuda.graph(gt, pool=pool, stream=self._stream, capture_error_mode=CAPTURE_MODE):
            self._tail()
        graphs.append(gt)
        self.opt.t -= 1                                           # the capture recorded an optimiser step without executing it
        self.graphs, self.ready = graphs, ready
        self._scratch_keep = dict(F._WS)
        self._images_refreshed = F.get_precision() != "f32"

    def _phase_last_collect_captured(self, pool, graphs):
        g = torch.cuda.CUDAGraph()
        with torch.cuda.graph(g, pool=pool, stream=self._stream, capture_error_mode=CAPTURE_MODE):
            r = self._phase_last_collect()
        graphs.append(g)
        return r

    def run(self, exposed=None):
        if not self.use_graphs:
            return self._eager_step(exposed)
        cur = torch.cuda.current_stream(self.dev)
        body = self.graphs[:-1]
        for i, g in enumerate(body):
            g.replay()
            if i + 1 < len(body) and i < len(self.ready) - 1:
                self._overlap(self.ready[i], cur)
        self._join(self.ready[-1], cur, exposed)
        self.graphs[-1].replay()
        self.opt.t += 1
        self.opt.fp.bump_versions(images_fresh=self._images_refreshed)
        return self.loss
