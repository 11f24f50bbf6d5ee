"""Whole training step as one captured hipGraph.

A training step is ~2 300 small launches issued through autograd; below ~64 clips per GPU the host (Python + launch calls, ~11 us each)
is slower than the GPU.  `GraphedStep` captures forward + backward + gradient collection + Adam once (torch.cuda.graph: stream capture,
allocations from a private pool) and replays it per step: HIP graphs instead of a tracing compiler, the MI355X-native answer to a
launch-bound loop.  Everything that varies between steps lives in device memory: the inputs (static tensors refreshed with `copy_`), the
Adam step count (`FlatAdam.use_device_step`), BatchNorm's running statistics.  Host scalars are frozen at capture, so stochastic dropout
(host-side mask counter) cannot be active inside a graphed step.
"""
from __future__ import annotations

from typing import Callable, Dict

import torch

from .. import _lib as L
from . import nets
from .optim import FlatAdam


class GraphedStep:
    def __init__(self, step_fn: Callable[[Dict[str, torch.Tensor]], torch.Tensor], inputs: Dict[str, torch.Tensor], optimizer: FlatAdam = None,
                 warmup: int = 3, device=None):
        """step_fn(inputs) runs zero_grad, forward, backward, the gradient collection and -- when `optimizer` is given -- optimizer.step(),
        and returns the loss tensor.  `inputs` are the static device tensors the captured graph reads; `run(new_inputs)` copies fresh
        values into them.  Data parallel: capture forward + backward + collection only (optimizer=None, GradBuckets.deferred = True) and run
        `GradBuckets.reduce_deferred()` and the optimiser after each replay -- the collective stays outside the captured graph."""
        dev = optimizer.fp.flat.device if optimizer is not None else torch.device(device)
        if dev.type != "cuda":
            raise L.EgError("GraphedStep: needs a GPU")
        if nets._P["on"]:
            raise L.EgError("GraphedStep: stochastic dropout keeps its mask counter on the host and cannot be captured")
        self.inputs, self.opt = inputs, optimizer
        if optimizer is not None:
            optimizer.use_device_step()
        side = torch.cuda.Stream(dev)
        side.wait_stream(torch.cuda.current_stream(dev))
        with torch.cuda.stream(side):               # warm-up off the capture: lazily sized scratch buffers, kernel attributes, autograd state
            for _ in range(max(1, warmup)):
                step_fn(inputs)
        torch.cuda.current_stream(dev).wait_stream(side)
        torch.cuda.synchronize(dev)
        self.graph = torch.cuda.CUDAGraph()
        with torch.cuda.graph(self.graph):
            self.loss = step_fn(inputs)
        if optimizer is not None:
            optimizer.t -= 1                        # the capture recorded a step without executing it

    def run(self, new_inputs: Dict[str, torch.Tensor] = None) -> torch.Tensor:
        if new_inputs:
            for k, v in new_inputs.items():
                self.inputs[k].copy_(v, non_blocking=True)
        self.graph.replay()
        if self.opt is not None:
            self.opt.t += 1
            self.opt.fp.bump_versions()             # the replay rewrote parameters and running statistics without passing through torch
        return self.loss
