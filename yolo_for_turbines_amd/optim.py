"""SGD step of the fine-tune loop as one HIP launch (`csrc/optim.hip`).

Drop-in for `torch.optim.SGD` as the reference constructs it (`code/train.py:171-172`:
`SGD(model.parameters(), lr, momentum, weight_decay)`, stepped at `:68` through the GradScaler): same constructor,
same `state` / `state_dict()` layout (`momentum_buffer` per parameter, so `utils.py:383-416` checkpoints load
either way), same bits after every step. Only fp32 parameters on the GPU (what the model holds); anything
else raises instead of falling back.
"""
import torch

from . import _lib as L


class _GroupTable:
    """What one parameter group needs for its launch: the chunk table (built once: tensor sizes never change) and the
    pointer table, which travels host -> device asynchronously every step."""

    def __init__(self, params):
        dev = params[0].device
        ce = L.lib().yolo_sgd_chunk_elems()
        chunks = []
        for i, p in enumerate(params):
            if not (p.is_cuda and p.dtype == torch.float32 and p.is_contiguous() and p.device == dev):
                raise TypeError("yolo_for_turbines_amd.optim.SGD: parameters must be contiguous fp32 tensors on one GPU")
            chunks += [(i, s) for s in range(0, p.numel(), ce)]
        self.params = list(params)
        self.chunks = torch.tensor(chunks, dtype=torch.int32).reshape(-1, 2).to(dev)
        self.items = torch.empty((len(params), 4), dtype=torch.int64, device=dev)
        # eager steps: a ring of pinned buffers, each guarded by the event of its last copy, so the host (which runs ahead of
        # the GPU) never rewrites a table the GPU has not read yet
        self.ring = [[self._pinned(), None] for _ in range(3)]
        self.next_slot = 0
        # captures: a captured copy reads its pinned source at every replay, so each capture gets a buffer nothing else will
        # ever write; pinned memory cannot be allocated inside a capture, so eager steps keep two spares ready
        self.spares = []
        self.captured = []

    def _pinned(self):
        return torch.zeros((len(self.params), 4), dtype=torch.int64).pin_memory()

    def matches(self, params):
        return len(self.params) == len(params) and all(a is b for a, b in zip(self.params, params))

    def host_buffer(self, capturing):
        """(pinned buffer to fill, ring slot or None)."""
        if capturing:
            if not self.spares:
                raise RuntimeError("yolo_for_turbines_amd.optim.SGD: take one eager step before capturing")
            host = self.spares.pop()
            self.captured.append(host)
            return host, None
        while len(self.spares) < 2:
            self.spares.append(self._pinned())
        slot = self.ring[self.next_slot]
        self.next_slot = (self.next_slot + 1) % len(self.ring)
        if slot[1] is not None:
            slot[1].synchronize()
        return slot[0], slot


class SGD(torch.optim.SGD):
    def __init__(self, params, lr=1e-3, momentum=0.0, dampening=0.0, weight_decay=0.0, nesterov=False, *, maximize=False,
                 foreach=None, differentiable=False, fused=None):
        # foreach / fused select among PyTorch's own implementations: accepted for signature compatibility, irrelevant here
        if differentiable:
            raise ValueError("yolo_for_turbines_amd.optim.SGD: differentiable=True is not supported")
        super().__init__(params, lr=lr, momentum=momentum, dampening=dampening, weight_decay=weight_decay, nesterov=nesterov,
                         maximize=maximize)
        self._tables = {}                                   # group index -> _GroupTable

    def _table(self, gi, params):
        t = self._tables.get(gi)
        if t is None or not t.matches(params):
            t = self._tables[gi] = _GroupTable(params)
        return t

    @torch.no_grad()
    def step(self, closure=None):
        loss = None
        if closure is not None:
            with torch.enable_grad():
                loss = closure()
        lib = L.lib()
        for gi, group in enumerate(self.param_groups):
            params = group["params"]
            if not params:
                continue
            tab = self._table(gi, params)
            mom = float(group["momentum"])
            capturing = torch.cuda.is_current_stream_capturing()
            host, slot = tab.host_buffer(capturing)
            rows = host.numpy()                              # [p, g, momentum buffer, n] per parameter (yolo_sgd_item)
            updated = []
            for i, p in enumerate(tab.params):
                g = p.grad
                if g is None:
                    rows[i, 1] = 0                           # skipped by the kernel, like PyTorch
                    continue
                if g.is_sparse or g.dtype != torch.float32 or not g.is_contiguous() or g.device != p.device:
                    raise TypeError("yolo_for_turbines_amd.optim.SGD: gradients must be dense contiguous fp32 on the parameter's GPU")
                n = p.numel()
                rows[i, 0] = p.data_ptr()
                rows[i, 1] = g.data_ptr()
                if mom != 0.0:
                    st = self.state[p]
                    buf = st.get("momentum_buffer")
                    if buf is None:
                        if capturing:
                            raise RuntimeError("yolo_for_turbines_amd.optim.SGD: take one eager step before capturing (the first step "
                                               "initialises the momentum buffers)")
                        buf = st["momentum_buffer"] = torch.empty_like(p, memory_format=torch.contiguous_format)
                        n = -n                               # first step: the kernel writes buf = g + wd * p
                    rows[i, 2] = buf.data_ptr()
                else:
                    rows[i, 2] = p.data_ptr()                # never touched
                rows[i, 3] = n
                updated.append(p)
            # the kernel writes through raw pointers: tell PyTorch's version counters (the engine re-packs a weight when its
            # version moves - without this the forward would keep using the weights of step 0)
            torch.autograd.graph.increment_version(updated)
            tab.items.copy_(host, non_blocking=True)
            if slot is not None:
                slot[1] = torch.cuda.Event()
                slot[1].record()
            L.check(lib.yolo_sgd_step(tab.items.data_ptr(), tab.chunks.data_ptr(), tab.chunks.shape[0], float(group["lr"]), mom,
                                      float(group["dampening"]), float(group["weight_decay"]), int(bool(group["nesterov"])),
                                      int(bool(group["maximize"])), L.current_stream()), "yolo_sgd_step")
        return loss
