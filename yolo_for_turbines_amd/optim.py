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
        self.uploaded = None                                # host copy of the row table the device holds (None: unknown)
        # captures: a captured copy reads its pinned source at every replay, so each capture gets a buffer nothing else will
        # ever write; pinned memory cannot be allocated inside a capture, so eager steps keep two spares ready
        self.spares = []
        self.captured = []
        # {lr, momentum, dampening, weight_decay} live in DEVICE memory and the kernel reads them when it runs: a step
        # captured in a HIP graph then follows the LR scheduler (train.py:71-74 steps LinearLR after every batch) instead of
        # replaying the capture-time values. Written stream-ordered, outside any capture, whenever the group's values change
        # (SGD.sync_hyper); pinned staging ring guarded by events like the row tables.
        self.hyper = torch.zeros(4, dtype=torch.float32, device=dev)
        self.hyper_host = None                              # the values last pushed
        self.hyper_ring = [[torch.zeros(4, dtype=torch.float32).pin_memory(), None] for _ in range(4)]
        self.hyper_slot = 0

    def _pinned(self):
        return torch.zeros((len(self.params), 4), dtype=torch.int64).pin_memory()

    def matches(self, params):
        return len(self.params) == len(params) and all(a is b for a, b in zip(self.params, params))

    def push_hyper(self, values):
        """Enqueue {lr, momentum, dampening, weight_decay} -> device on the current stream if they changed."""
        if values == self.hyper_host:
            return
        if torch.cuda.is_current_stream_capturing():
            raise RuntimeError("yolo_for_turbines_amd.optim.SGD: the hyper-parameters changed inside a stream capture; call "
                               "optimizer.sync_hyper() (GraphedTrainStep does) before the replay instead")
        slot = self.hyper_ring[self.hyper_slot]
        self.hyper_slot = (self.hyper_slot + 1) % len(self.hyper_ring)
        if slot[1] is not None:
            slot[1].synchronize()
        h = slot[0]
        h[0], h[1], h[2], h[3] = values
        self.hyper.copy_(h, non_blocking=True)
        slot[1] = torch.cuda.Event()
        slot[1].record()
        self.hyper_host = values

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

    @staticmethod
    def _hyper_of(group):
        mom, damp = float(group["momentum"]), float(group["dampening"])
        if group["nesterov"] and (mom <= 0.0 or damp != 0.0):
            raise ValueError("Nesterov momentum requires a momentum and zero dampening")
        return (float(group["lr"]), mom, damp, float(group["weight_decay"]))

    def sync_hyper(self):
        """Push every group's current {lr, momentum, dampening, weight_decay} to the device (stream-ordered, a no-op when
        nothing changed). `step()` does this itself in eager mode; call it before replaying a HIP graph that captured
        `step()` whenever an LR scheduler (train.py:71-74,187-189) or the user changed `param_groups` in between."""
        for gi, group in enumerate(self.param_groups):
            tab = self._tables.get(gi)
            if tab is not None:
                tab.push_hyper(self._hyper_of(group))

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
            hyper = self._hyper_of(group)
            if capturing and tab.hyper_host is None:
                raise RuntimeError("yolo_for_turbines_amd.optim.SGD: take one eager step before capturing")
            if not capturing or hyper != tab.hyper_host:      # inside a capture an unchanged value needs no node at all
                tab.push_hyper(hyper)
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
            # With gradient buckets / a captured graph every address is the same step after step: upload the table only when it
            # differs from the one the device already holds (an H2D copy costs the GPU ~85 us of idle queue in front of it)
            same = (not capturing) and tab.uploaded is not None and bool((rows == tab.uploaded).all())
            if not same:
                tab.items.copy_(host, non_blocking=True)
                tab.uploaded = None if capturing else rows.copy()
                if slot is not None:
                    slot[1] = torch.cuda.Event()
                    slot[1].record()
            L.check(lib.yolo_sgd_step_hp(tab.items.data_ptr(), tab.chunks.data_ptr(), tab.chunks.shape[0], tab.hyper.data_ptr(),
                                         int(bool(group["nesterov"])), int(bool(group["maximize"])), L.current_stream()),
                    "yolo_sgd_step_hp")
        return loss
