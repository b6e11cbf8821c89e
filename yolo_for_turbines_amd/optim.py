"""SGD step of the fine-tune loop as one HIP launch (`csrc/optim.hip`).

Drop-in for `torch.optim.SGD` as the reference constructs it (`code/train.py:171-172`:
`SGD(model.parameters(), lr, momentum, weight_decay)`, stepped at `:68`): same constructor, same
`state` / `state_dict()` layout (`momentum_buffer` per parameter, so `utils.py:383-416` checkpoints load
either way), same bits after every step. Only fp32 parameters on the GPU (what the model holds); anything
else raises instead of falling back.
"""
import torch

from . import _lib as L


class SGD(torch.optim.SGD):
    def __init__(self, params, lr=1e-3, momentum=0.0, dampening=0.0, weight_decay=0.0, nesterov=False, *, maximize=False,
                 foreach=None, differentiable=False, fused=None):
        # foreach / fused select among PyTorch's own implementations: accepted for signature compatibility, irrelevant here
        if differentiable:
            raise ValueError("yolo_for_turbines_amd.optim.SGD: differentiable=True is not supported")
        super().__init__(params, lr=lr, momentum=momentum, dampening=dampening, weight_decay=weight_decay, nesterov=nesterov,
                         maximize=maximize)
        self._tables = {}                                   # group index -> (params, chunk table, pinned items, device items)

    def _table(self, gi, params):
        t = self._tables.get(gi)
        if t is not None and len(t[0]) == len(params) and all(a is b for a, b in zip(t[0], params)):
            return t
        dev = params[0].device
        ce = L.lib().yolo_sgd_chunk_elems()
        chunks = []
        for i, p in enumerate(params):
            if not (p.is_cuda and p.dtype == torch.float32 and p.is_contiguous() and p.device == dev):
                raise TypeError("yolo_for_turbines_amd.optim.SGD: parameters must be contiguous fp32 tensors on one GPU")
            chunks += [(i, s) for s in range(0, p.numel(), ce)]
        ck = torch.tensor(chunks, dtype=torch.int32).reshape(-1, 2).to(dev)
        # the pointer table travels host -> device asynchronously every step: a ring of pinned buffers, each guarded by the
        # event of its last copy, so the host (which runs ahead of the GPU) never rewrites a table the GPU has not read yet
        ring = [[torch.zeros((len(params), 4), dtype=torch.int64).pin_memory(), None] for _ in range(3)]
        t = self._tables[gi] = [list(params), ck, ring, torch.empty((len(params), 4), dtype=torch.int64, device=dev), 0, [], []]
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
            plist, ck, ring, items = tab[0], tab[1], tab[2], tab[3]
            mom = float(group["momentum"])
            capturing = torch.cuda.is_current_stream_capturing()
            if capturing:
                # a captured copy reads its pinned source at every replay: give it a buffer nothing else will ever write
                # (pinned memory cannot be allocated inside a capture: spares are made by the eager steps before it)
                if not tab[6]:
                    raise RuntimeError("yolo_for_turbines_amd.optim.SGD: take one eager step before capturing")
                host = tab[6].pop()
                tab[5].append(host)
                slot = None
            else:
                while len(tab[6]) < 2:
                    tab[6].append(torch.zeros((len(plist), 4), dtype=torch.int64).pin_memory())
                slot = ring[tab[4]]
                tab[4] = (tab[4] + 1) % len(ring)
                if slot[1] is not None:
                    slot[1].synchronize()
                host = slot[0]
            rows = host.numpy()
            for i, p in enumerate(plist):
                g = p.grad
                if g is None:
                    rows[i, 1] = 0
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
            # the kernel writes through raw pointers: tell PyTorch's version counters (the engine re-packs a weight when its
            # version moves - without this the forward would keep using the weights of step 0)
            torch.autograd.graph.increment_version([p for p in plist if p.grad is not None])
            items.copy_(host, non_blocking=True)
            if slot is not None:
                slot[1] = torch.cuda.Event()
                slot[1].record()
            L.check(lib.yolo_sgd_step(items.data_ptr(), ck.data_ptr(), ck.shape[0], float(group["lr"]), mom, float(group["dampening"]),
                                      float(group["weight_decay"]), int(bool(group["nesterov"])), int(bool(group["maximize"])),
                                      L.current_stream()), "yolo_sgd_step")
        return loss
