"""HIP-graph capture of a whole training step (forward, loss, backward, fused optimizer step).

The reference wraps its models in ``torch.compile`` (``/root/reference/src/pytorch/CNN/main.py:158``) to cut
framework overhead.  Here the hot path is pre-compiled HIP, so what is left for small models (LeNet-5, ResNet-18 on
CIFAR, the reference's 209k-parameter DenseNet) is launch latency: hundreds of kernels of a few microseconds each.
Capturing the step once into a hipGraph and replaying it removes the Python / launch cost of every one of them.

What makes the step capturable:

* every kernel of this package launches on torch's current stream (``ops/_lib.py``), allocations come from torch's
  caching allocator (graph-private pool during capture), and nothing in the step reads a value back to the host;
* BN slabs are self-cleaning and persistent, gradients accumulate in the flat arena, the optimizer zeroes them in
  its own kernel: replaying the same launches on the same addresses IS the next step;
* hyper-parameters (lr, Adam bias corrections, ...) live in a device array that the host refreshes BEFORE each
  replay (``optimizer.host_prepare``), so schedulers keep working;
* the side-stream wgrad fork / join (``ops/streams.py``) happens inside the capture and becomes graph edges.

Gradient collectives can be part of the capture when they go through the native communicator
(``DistributedDataParallel(comm="native")``: RCCL calls are plain stream operations there, the comm stream joins the
capture through the usual event edges); collectives issued through ``torch.distributed`` are not captured — run
those steps eagerly.  Also unsupported: BN with ``momentum=None`` (its factor depends on a host counter) and
data-dependent control flow.
"""
from __future__ import annotations

from typing import Optional

import torch

from ..nn.modules import BatchNorm2d
from ..ops import streams


class GraphedTrainStep:
    """``step = GraphedTrainStep(model, criterion, optimizer, x0, y0); loss = step(x, y)``.

    ``x0`` / ``y0`` fix the batch shape and dtype; later batches are copied into the static buffers.  ``loss`` and
    ``correct`` are static device tensors that the next replay overwrites (clone them to keep a history).
    """

    def __init__(self, model, criterion, optimizer, example_x: torch.Tensor, example_y: torch.Tensor, warmup: int = 3,
                 ddp=None):
        if not example_x.is_cuda:
            raise ValueError("GraphedTrainStep needs CUDA/HIP tensors")
        if ddp is not None and ddp.dist and ddp.ncomm is None:
            raise NotImplementedError("graph capture of torch.distributed collectives is not supported: build the "
                                      "reducer with comm='native' or run the step eagerly")
        if (ddp is not None and ddp.dist and ddp.mode == "ps" and ddp.ncomm is None
                and getattr(optimizer, "norm_reduce", None) is not None and optimizer.param_groups[0].get("lars")):
            raise NotImplementedError("sharded LARS all-reduces its norms through torch.distributed: not capturable "
                                      "(the native communicator's norm all-reduce is)")
        self.ddp = ddp if (ddp is not None and ddp.dist) else None
        self.fwd = ddp if ddp is not None else model
        for m in model.modules():
            if isinstance(m, BatchNorm2d) and m.momentum is None:
                raise NotImplementedError("BatchNorm with momentum=None cannot be captured")
        self.model, self.criterion, self.optimizer = model, criterion, optimizer
        self.bns = [m for m in model.modules() if isinstance(m, BatchNorm2d)]
        self.x = example_x.clone()
        self.y = example_y.clone()
        self.loss: Optional[torch.Tensor] = None
        self.correct: Optional[torch.Tensor] = None
        self.graph = torch.cuda.CUDAGraph()
        self.replays = 0

        # Warm-up on a side stream (allocator pools, lazily-built slabs, autograd metadata), as eager steps.
        s = torch.cuda.Stream(device=example_x.device)
        s.wait_stream(torch.cuda.current_stream(example_x.device))
        with torch.cuda.stream(s):
            for _ in range(max(1, warmup)):
                self.warm_loss = self._eager_step().clone()
                wc = getattr(self.criterion, "last_correct", None)
                self.warm_correct = wc.clone() if wc is not None else None
        torch.cuda.current_stream(example_x.device).wait_stream(s)
        torch.cuda.synchronize(example_x.device)

        # Derived weight copies that are NOT arena-backed (bf16 packs of the stem / pad_out filters, LSTM packs, Linear
        # weights) are cached per weights epoch.  The warm-up refreshed them; a capture would bake in addresses of
        # copies that training makes stale, so force them to be re-derived INSIDE the graph ...
        from ..ops import state as _state
        _state.bump()
        self.optimizer.host_prepare()
        with torch.cuda.graph(self.graph):
            self.loss = self._forward_backward()
            if self.ddp is not None:
                d = self.ddp
                d.finish_grad_sync()                 # waits = event edges from the communicator's stream
                self.optimizer.launch(ranges=d.owned_ranges())
                if d.mode == "ps":
                    d.arena.zero_grad()
                    d.gather_parameters()
            else:
                streams.join()
                self.optimizer.launch()
        # ... and the copies cached DURING the capture live in the graph's private pool and were never executed (a
        # capture records launches without running them): an eager step or evaluation right after the capture - a
        # ragged last batch, --max-steps 1 - must not read them.  Invalidate them for every eager caller.
        _state.bump()
        self.correct = getattr(self.criterion, "last_correct", None)
        # Capturing records the launches without running them: undo the host-side bookkeeping of that "step".
        self.optimizer.step_count -= 1
        for bn in self.bns:
            bn._nbt_pending -= 1             # noqa: SLF001

    def _forward_backward(self) -> torch.Tensor:
        out = self.fwd(self.x)
        loss = self.criterion(out, self.y)
        loss.backward()
        return loss.detach()

    def _eager_step(self) -> torch.Tensor:
        loss = self._forward_backward()
        if self.ddp is not None:
            self.ddp.step(self.optimizer)
        else:
            self.optimizer.step()
        return loss

    def __call__(self, x: torch.Tensor, y: torch.Tensor) -> torch.Tensor:
        self.x.copy_(x, non_blocking=True)
        self.y.copy_(y, non_blocking=True)
        self.optimizer.host_prepare()
        self.graph.replay()
        for bn in self.bns:                  # host-side counters that the eager path advances inside forward()
            bn._count_batch()                # noqa: SLF001
        self.optimizer.host_finish()
        self.replays += 1
        return self.loss


class GraphedEvalStep:
    """Forward + loss of an evaluation batch (``model.eval()``, no gradients) as one hipGraph:
    ``ev = GraphedEvalStep(model, criterion, x0, y0); loss, correct = ev(x, y)``.

    The validation / test passes of the small workloads are as launch-bound as their training step (a third of an
    epoch's wall time for the reference's 70/10/20 split once the training step is graphed).  Eval-mode BatchNorm
    derives its affine from the running statistics INSIDE the graph, so a replay always sees the current buffers.
    The model must already be in eval mode; outputs are static tensors overwritten by the next replay."""

    def __init__(self, model, criterion, example_x: torch.Tensor, example_y: torch.Tensor):
        if not example_x.is_cuda:
            raise ValueError("GraphedEvalStep needs CUDA/HIP tensors")
        if model.training:
            raise ValueError("GraphedEvalStep captures the eval-mode forward: call model.eval() first")
        self.model, self.criterion = model, criterion
        self.x, self.y = example_x.clone(), example_y.clone()
        self.graph = torch.cuda.CUDAGraph()
        self.replays = 0
        s = torch.cuda.Stream(device=example_x.device)
        s.wait_stream(torch.cuda.current_stream(example_x.device))
        with torch.cuda.stream(s), torch.no_grad():
            self._forward()                                   # warm-up: allocator pools, lazily-built caches
        torch.cuda.current_stream(example_x.device).wait_stream(s)
        torch.cuda.synchronize(example_x.device)
        # Derived weight copies (bf16 packs of filters / Linear weights not backed by the arena shadow) are cached per
        # weights epoch; the warm-up just refreshed them, so a capture would bake in the addresses of copies that later
        # training makes stale.  Bumping the epoch forces every such copy to be re-derived INSIDE the graph.
        from ..ops import state as _state
        _state.bump()
        with torch.no_grad(), torch.cuda.graph(self.graph):
            self.loss = self._forward()
        _state.bump()       # ...and the copies cached DURING the capture live in the graph's pool (filled only by replays):
        #                     invalidate them for every eager caller
        self.correct = getattr(self.criterion, "last_correct", None)

    def _forward(self) -> torch.Tensor:
        return self.criterion(self.model(self.x), self.y).detach()

    def __call__(self, x: torch.Tensor, y: torch.Tensor):
        self.x.copy_(x, non_blocking=True)
        self.y.copy_(y, non_blocking=True)
        self.graph.replay()
        self.replays += 1
        return self.loss, self.correct
