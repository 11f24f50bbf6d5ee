"""Several independent batches ("steps") in flight on one GPU.

One step of the path (mel -> CVAE sample -> generator) is ~230 launches whose middle third (transformer GEMMs, attention,
LayerNorm at M = B*frames = 2176 rows) cannot fill 256 CUs, while the convolution tower can.  Batches are independent
(SURVEY.md §8e), so `ClipPipeline` keeps `lanes` of them in flight: the lanes share the models (one engine handle and one
read-only weights arena, so the weights stay hot in L2 / Infinity Cache for all of them) and every lane owns its workspaces
(the engines' `slot` argument), static input / output buffers, one captured hipGraph of the whole step and one stream.  Lanes are replayed
round-robin; the GPU's workgroup dispatcher then overlaps the low-occupancy phase of one batch with the convolution phase of
another (measured on MI355X, B=64, bf16x3: 17.5k clips/s with one step in flight, 22.1k with four).

No CPU fallback: everything here needs the HIP library and a GPU.
"""
from __future__ import annotations

from typing import Callable, Iterable, Iterator, List, Optional, Sequence, Tuple

import torch


# Stream capture in `thread_local` error mode: with a torch.distributed process group alive (bench.py --gpus N, DataParallel's RCCL broadcast) the
# RCCL watchdog thread polls its events (hipEventQuery) at any time; under the default `global` mode such a call from ANOTHER thread while this
# thread captures is flagged as illegal and the watchdog aborts the process ("operation not permitted when stream is capturing" -- caught in
# round 5 by the forced 1-rank RCCL run: 4 of 8 runs died).  Only calls made by the capturing thread itself can invalidate its capture.
CAPTURE_MODE = "thread_local"


class _Lane:
    __slots__ = ("slot", "stream", "graph", "inputs", "outputs", "done", "busy", "collected", "keep", "key")


class ClipPipeline:
    """`models = (generator, vae | None, mel | None)`: modules already on `device`, eval mode; shared by all lanes.

    A step takes audio [B, n_samples] (or a ready spectrogram [B,128,T] when `mel is None`), text [B,60] int64,
    pre_pose [B,P,D], and -- when a VAE is present -- label [B,8] and z [B,32]; it returns the generator's 5-tuple."""

    def __init__(self, models: Tuple, example_inputs: dict, device, lanes: int = 4, branch_streams: bool = False):
        if lanes < 1:
            raise ValueError("lanes must be >= 1")
        self.device = torch.device(device)
        if self.device.type != "cuda":
            raise RuntimeError("ClipPipeline needs a GPU (no CPU fallback)")
        self.gen, self.vae, self.mel = models
        if branch_streams and lanes > 1:
            raise ValueError("branch_streams forks use per-handle side streams: only with lanes == 1")
        self.concurrent = bool(branch_streams)     # text / prior branches forked onto side streams inside the step (one lane only)
        # other lanes fill the chip: the products' tile is chosen for CU time (EgGeneratorConfig.reserved[4]).  A property of THIS pipeline, passed to
        # engine() per call: the generator keeps one engine per tile-policy class, so a 1-lane and an N-lane pipeline can live on one generator.
        self.shared_chip = lanes > 1
        self.lanes: List[_Lane] = []
        self._next = 0
        for i in range(lanes):
            ln = _Lane()
            ln.slot = i
            ln.stream = torch.cuda.Stream(self.device)
            ln.inputs = {k: v.to(self.device).clone() for k, v in example_inputs.items()}
            ln.done = torch.cuda.Event()
            ln.collected = None
            ln.busy = False
            self._capture(ln)
            self.lanes.append(ln)

    # one step on the current stream, reading the lane's static buffers
    def _step(self, ln: _Lane):
        g = ln.inputs
        with torch.no_grad():
            eng = self._gen_engine()
            spec = self.mel(g["audio"], out_frames=eng.cfg.spec_len, slot=ln.slot) if self.mel is not None else g["spec"]
            sampled = self.vae.sample(g["label"], z=g["z"], slot=ln.slot) if self.vae is not None else g.get("sampled")
            return eng.forward(spec, g["text"], g["pre_pose"], sampled, slot=ln.slot)      # = Transformer.forward in eval mode, on this pipeline's engine

    def _gen_engine(self):
        if self.gen.training:
            raise RuntimeError("ClipPipeline runs eval-mode modules (call generator.eval())")
        return self.gen.engine(shared_chip=self.shared_chip, concurrent=self.concurrent)

    def _engine_state(self):
        """What a captured graph has baked in as raw pointers: the engines, their weight arenas and this lane's workspaces."""
        eng = self._gen_engine()
        veng = self.vae.engine() if self.vae is not None else None
        return eng, eng.arena, veng, (veng.arena if veng is not None else None)

    def _capture(self, ln: _Lane) -> None:
        cap = torch.cuda.Stream(self.device)
        cap.wait_stream(torch.cuda.current_stream(self.device))
        with torch.cuda.stream(cap):
            for _ in range(2):
                self._step(ln)            # workspaces, kernel attributes: outside the capture
        torch.cuda.current_stream(self.device).wait_stream(cap)
        torch.cuda.synchronize(self.device)
        ln.graph = torch.cuda.CUDAGraph()
        with torch.cuda.graph(ln.graph, capture_error_mode=CAPTURE_MODE):
            ln.outputs = self._step(ln)
        ln.graph.replay()                 # capture does not execute: run once so that the lane's outputs are defined from the start
        torch.cuda.synchronize(self.device)
        # The graph replays raw pointers into the engines' arenas and this lane's workspaces.  Keep those objects alive for as
        # long as the graph exists (a later repack / precision change builds NEW engines and arenas; the old ones must not be
        # freed under a graph that may still be replayed), and remember their identity so that a stale graph is detected.
        st = self._engine_state()
        ln.key = tuple(id(o) for o in st)
        ln.keep = st + (dict(st[0]._ws), dict(st[2]._ws) if st[2] is not None else None,
                        dict(self.mel._ws) if self.mel is not None else None)

    def stale(self) -> bool:
        """True when the models' engines / weight arenas are no longer the ones the lanes were captured with
        (weights reloaded, precision / keep_taps / concurrent changed)."""
        key = tuple(id(o) for o in self._engine_state())
        return any(ln.key != key for ln in self.lanes)

    def refresh(self) -> None:
        """Re-capture every lane against the models' current engines (after a weight update or a mode change)."""
        torch.cuda.synchronize(self.device)
        for ln in self.lanes:
            self._capture(ln)

    # ---- low level: the bench drives these directly (inputs already resident in the lanes' buffers) ----
    def launch_next(self) -> int:
        """Replay the next lane's step on its stream; returns the lane index."""
        i = self._next
        self._next = (self._next + 1) % len(self.lanes)
        ln = self.lanes[i]
        eng = self.gen.engine_peek(self.shared_chip, self.concurrent)
        if eng is None or id(eng) != ln.key[0] or id(eng.arena) != ln.key[1]:
            raise RuntimeError("ClipPipeline: the generator's engine / weight arena changed after capture (weights reloaded or "
                               "precision / keep_taps / concurrent flipped); call refresh() before launching")
        with torch.cuda.stream(ln.stream):
            if ln.collected is not None:      # a caller-stream copy of this lane's previous outputs must finish before they are overwritten
                ln.stream.wait_event(ln.collected)
                ln.collected = None
            ln.graph.replay()
            ln.done.record(ln.stream)
        ln.busy = True
        return i

    def outputs(self, lane: int):
        """The lane's output tensors (valid after `wait(lane)`; overwritten by that lane's next step)."""
        return self.lanes[lane].outputs

    def wait(self, lane: int) -> None:
        self.lanes[lane].done.synchronize()
        self.lanes[lane].busy = False

    # ---- high level: stream batches through the lanes, results in submission order ----
    def run(self, batches: Iterable[dict]) -> Iterator[Tuple[torch.Tensor, ...]]:
        """Every batch is a dict with the keys of `example_inputs` (same shapes).  Yields a copy of each step's 5-tuple in order;
        up to `lanes` batches are in flight."""
        if self.stale():                  # full check (parameter versions) once per call; launch_next() re-checks identity cheaply
            self.refresh()
        pending: List[int] = []
        for b in batches:
            if len(pending) == len(self.lanes):
                yield self._collect(pending.pop(0))
            i = self._next
            ln = self.lanes[i]
            ln.stream.wait_stream(torch.cuda.current_stream(self.device))      # the batch was produced on the caller's stream
            with torch.cuda.stream(ln.stream):
                for k, buf in ln.inputs.items():
                    src = b[k]
                    if tuple(src.shape) != tuple(buf.shape):
                        raise ValueError(f"ClipPipeline.run: {k} has shape {tuple(src.shape)}, the lanes were captured for {tuple(buf.shape)}")
                    buf.copy_(src, non_blocking=True)
                    if src.is_cuda:           # the caller may drop `src` right away: its block must not be reused before this copy ran
                        src.record_stream(ln.stream)
            pending.append(self.launch_next())
        while pending:
            yield self._collect(pending.pop(0))

    def _collect(self, lane: int):
        # The copies are allocated and run on the CALLER's stream (so the caching allocator orders their reuse with the caller's
        # later work) after that stream has waited for the lane's step; the lane's next replay waits for the copies (`collected`).
        ln = self.lanes[lane]
        cur = torch.cuda.current_stream(self.device)
        cur.wait_event(ln.done)
        out = tuple(None if t is None else t.clone() for t in ln.outputs)
        ln.collected = torch.cuda.Event()
        ln.collected.record(cur)
        ln.busy = False
        return out

    def synchronize(self) -> None:
        torch.cuda.synchronize(self.device)
