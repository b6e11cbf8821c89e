"""Whole-step HIP graph for fine-tuning: capture once, replay every step.

The reference's step (`/root/reference/code/train.py:41-69`: autocast forward, three per-scale losses, backward,
optimizer step) is ~1,000 kernel launches here, many of them a few microseconds long; issued from Python one by
one the GPU idles between them. Every kernel of this package launches on the current stream, allocates nothing
behind PyTorch's back and — with :class:`FusedYOLOLoss` and the per-forward NaN guard off — never waits for the host,
so ``torch.cuda.graph`` can record the whole step into ONE HIP graph. Static shapes only: one graph per
(batch, image size); multi-scale training keeps one per size.
"""
from __future__ import annotations

import os

import torch

from .loss import FusedYOLOLoss


class GraphedTrainStep:
    """``step = GraphedTrainStep(model, optimizer, scaled_anchors, x, targets)`` then ``loss = step(x, targets)``.

    ``x``: (B,3,S,S) batch, ``targets``: the three (B,3,g,g,6) tensors of the loader, ``scaled_anchors``: (3,3,2) anchors
    in grid units (`train.py:195-197`). The example batch passed to the constructor fixes the shapes and is used for
    the warm-up steps on a side stream (so they DO update the model, like three ordinary steps). ``autocast_dtype``
    (``torch.bfloat16`` / ``torch.float16`` / None) selects what `train.py:53` selects. Returns the summed loss as a
    0-dim tensor that is overwritten by the next call.
    """

    def __init__(self, model, optimizer, scaled_anchors, x, targets, autocast_dtype=None, loss_fn=None, warmup=3,
                 zero_grad=True, allow_data_parallel=None):
        if not x.is_cuda:
            raise RuntimeError("GraphedTrainStep needs CUDA/HIP tensors (no CPU fallback)")
        if allow_data_parallel is None:
            allow_data_parallel = os.environ.get("YOLO_DP_GRAPH", "0") == "1"
        self._dp = getattr(model._engine, "ddp", None) is not None and model._engine.ddp[0] is not None
        if self._dp and not allow_data_parallel:
            raise NotImplementedError("GraphedTrainStep does not capture the data-parallel gradient all-reduce by default (RCCL inside a "
                                      "HIP graph capture has only been exercised with a 1-rank group here: DESIGN.md); pass "
                                      "allow_data_parallel=True / YOLO_DP_GRAPH=1 to capture it, or replay per rank on one GPU")
        self.model, self.opt = model, optimizer
        self.loss_fn = loss_fn if loss_fn is not None else FusedYOLOLoss()
        self.autocast_dtype = autocast_dtype
        self.anchors = [a.detach().to(x.device, torch.float32).contiguous() for a in scaled_anchors]
        self.x = x.detach().clone()
        self.targets = [t.detach().clone().float().contiguous() for t in targets]
        self.zero_grad = zero_grad
        self._nan_check = model._engine.nan_check
        model._engine.nan_check = False                 # the guard reads a flag on the host: a sync, not capturable
        try:
            side = torch.cuda.Stream(device=x.device)
            side.wait_stream(torch.cuda.current_stream(x.device))
            with torch.cuda.stream(side):
                for _ in range(max(1, warmup)):         # allocator warm-up, plan build, lazily created optimizer state
                    self._step_body()
            torch.cuda.current_stream(x.device).wait_stream(side)
            # The graph bakes in raw pointers to the train plan's buffers (activations, statistics, workspaces, packed
            # gradient weights). Own the plan: a strong reference keeps its memory alive, `pinned` exempts it from the
            # model's LRU of plans (multi-scale training builds other sizes in between replays).
            plans = [pl for k, pl in model._engine._plans.items() if k[0] == "train" and k[1] == x.shape[0] and k[2] == x.shape[2]]
            if not plans:
                raise RuntimeError("GraphedTrainStep: the warm-up steps left no train plan for this shape")
            self._plan = plans[-1]
            self._plan.pinned = True
            self.graph = torch.cuda.CUDAGraph()
            # Hyper-parameters across replays: `yt.SGD` keeps them in device memory and reads them when the kernel runs
            # (`sync_hyper` before each replay), so LR schedulers keep working. Any other optimizer bakes its Python floats
            # into the captured kernels: remember them and refuse a replay at different values rather than train silently
            # at the capture-time learning rate.
            self._baked = None if hasattr(self.opt, "sync_hyper") else self._hyper_snapshot()
            if self.zero_grad:
                self.opt.zero_grad(set_to_none=True)
            # data parallel (opt-in): the warm-up steps above have created the communicator and run every bucket's collective
            # once eagerly; inside the capture the buckets issue synchronous all-reduces on the capturing stream
            # (dist.GradBuckets.fire), and capture errors stay local to this thread
            kw = {"capture_error_mode": "thread_local"} if self._dp else {}
            with torch.cuda.graph(self.graph, **kw):
                self.loss = self._step_body(zero=False)
        finally:
            model._engine.nan_check = self._nan_check

    def _step_body(self, zero=True):
        if zero and self.zero_grad:
            self.opt.zero_grad(set_to_none=True)
        with torch.autocast("cuda", dtype=self.autocast_dtype or torch.bfloat16, enabled=self.autocast_dtype is not None):
            preds = self.model(self.x)
            loss = sum(sum(self.loss_fn(preds[i], self.targets[i], self.anchors[i])) for i in range(3))
        loss.backward()
        self.opt.step()
        return loss.detach()

    def _hyper_snapshot(self):
        keys = ("lr", "momentum", "dampening", "weight_decay", "nesterov", "maximize", "betas", "eps")
        return [tuple((k, g[k]) for k in keys if k in g and not isinstance(g[k], torch.Tensor)) for g in self.opt.param_groups]

    def __call__(self, x, targets):
        if tuple(x.shape) != tuple(self.x.shape):
            raise ValueError(f"this graph was captured for input shape {tuple(self.x.shape)}, got {tuple(x.shape)}")
        if self._plan.dropped:
            raise RuntimeError("GraphedTrainStep: the model dropped its plans after this graph was captured (model.to(...) / "
                               ".float() / .half() move the parameters the graph points at): capture a new GraphedTrainStep")
        if self._baked is None:
            self.opt.sync_hyper()                       # stream-ordered in front of the replay; no-op when nothing changed
        elif self._hyper_snapshot() != self._baked:
            raise RuntimeError("GraphedTrainStep: the optimizer's hyper-parameters changed since the capture "
                               f"({self._baked} -> {self._hyper_snapshot()}) and this optimizer bakes them into the captured "
                               "kernels; use yolo_for_turbines_amd.SGD (reads them from device memory at every replay) or "
                               "capture a new GraphedTrainStep")
        self.x.copy_(x, non_blocking=True)
        for dst, src in zip(self.targets, targets):
            dst.copy_(src, non_blocking=True)
        self.graph.replay()
        # the replay rewrote every parameter and BatchNorm statistic without dispatching a torch op: version counters did
        # not move, so every packed weight / BN fold cached for eval is stale by construction
        self.model._engine.invalidate()
        return self.loss

    def release(self):
        """Give the pinned plan back to the model's LRU (its memory is freed once evicted and the graph is dropped)."""
        self._plan.pinned = False
        self.graph = None
