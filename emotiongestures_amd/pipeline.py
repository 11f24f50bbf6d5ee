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


class _Lane:
    __slots__ = ("slot", "stream", "graph", "inputs", "outputs", "done", "busy")


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
        self.gen.concurrent = bool(branch_streams)
        self.lanes: List[_Lane] = []
        self._next = 0
        for i in range(lanes):
            ln = _Lane()
            ln.slot = i
            ln.stream = torch.cuda.Stream(self.device)
            ln.inputs = {k: v.to(self.device).clone() for k, v in example_inputs.items()}
            ln.done = torch.cuda.Event()
            ln.busy = False
            self._capture(ln)
            self.lanes.append(ln)

    # one step on the current stream, reading the lane's static buffers
    def _step(self, ln: _Lane):
        g = ln.inputs
        with torch.no_grad():
            spec = self.mel(g["audio"], out_frames=self.gen.engine().cfg.spec_len, slot=ln.slot) if self.mel is not None else g["spec"]
            sampled = self.vae.sample(g["label"], z=g["z"], slot=ln.slot) if self.vae is not None else g.get("sampled")
            return self.gen(spec, g["text"], g["pre_pose"], sampled, slot=ln.slot)

    def _capture(self, ln: _Lane) -> None:
        cap = torch.cuda.Stream(self.device)
        cap.wait_stream(torch.cuda.current_stream(self.device))
        with torch.cuda.stream(cap):
            for _ in range(2):
                self._step(ln)            # workspaces, kernel attributes: outside the capture
        torch.cuda.current_stream(self.device).wait_stream(cap)
        torch.cuda.synchronize(self.device)
        ln.graph = torch.cuda.CUDAGraph()
        with torch.cuda.graph(ln.graph):
            ln.outputs = self._step(ln)
        ln.graph.replay()                 # capture does not execute: run once so that the lane's outputs are defined from the start
        torch.cuda.synchronize(self.device)

    # ---- low level: the bench drives these directly (inputs already resident in the lanes' buffers) ----
    def launch_next(self) -> int:
        """Replay the next lane's step on its stream; returns the lane index."""
        i = self._next
        self._next = (self._next + 1) % len(self.lanes)
        ln = self.lanes[i]
        with torch.cuda.stream(ln.stream):
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
            pending.append(self.launch_next())
        while pending:
            yield self._collect(pending.pop(0))

    def _collect(self, lane: int):
        # copy on the lane's own stream: ordered after its step and before its next replay; the caller's stream then waits
        ln = self.lanes[lane]
        with torch.cuda.stream(ln.stream):
            out = tuple(None if t is None else t.clone() for t in ln.outputs)
        torch.cuda.current_stream(self.device).wait_stream(ln.stream)
        ln.busy = False
        return out

    def synchronize(self) -> None:
        torch.cuda.synchronize(self.device)
