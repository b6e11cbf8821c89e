"""Multi-GPU plumbing: one process per GPU (torchrun), RCCL through ``torch.distributed``.

Inference / decode / NMS shard by IMAGE with no data-path collective: each rank
owns a contiguous slice of the batch; only tiny results (kept counts / indices) are gathered. The
timing helper implements the bench contract: barrier + device sync on both sides of exactly K
steps, MAX over ranks.
"""
from __future__ import annotations

import os
import time

import torch


def env_world():
    """(rank, local_rank, world_size) from the torchrun environment (1-process default)."""
    return (int(os.environ.get("RANK", "0")), int(os.environ.get("LOCAL_RANK", "0")),
            int(os.environ.get("WORLD_SIZE", "1")))


def init(backend=None, device=None):
    """Initialise the default process group when WORLD_SIZE > 1. ``nccl`` IS RCCL on ROCm."""
    import torch.distributed as dist
    rank, local_rank, world = env_world()
    force = bool(os.environ.get("YOLO_FORCE_DIST"))          # exercise the collective path with one rank (tests)
    if (world == 1 and not force) or dist.is_initialized():
        return dist if (world > 1 or force) else None
    os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
    backend = backend or ("nccl" if torch.cuda.is_available() else "gloo")
    kw = {}
    if backend == "nccl" and device is not None:
        kw["device_id"] = device
    dist.init_process_group(backend, rank=rank, world_size=world, **kw)
    return dist


def shard_range(n_items: int, rank: int, world: int):
    """Contiguous [lo, hi) slice of ``n_items`` independent units for ``rank``; sizes differ by at
    most one, every item belongs to exactly one rank."""
    base, extra = divmod(n_items, world)
    lo = rank * base + min(rank, extra)
    return lo, lo + base + (1 if rank < extra else 0)


def _sync():
    if torch.cuda.is_available():
        torch.cuda.synchronize()


def timed_steps(step, steps: int, warmup: int, dist=None, device=None):
    """Run ``warmup`` untimed + exactly ``steps`` timed calls of ``step()``; returns the elapsed
    seconds, MAX over ranks (bench contract)."""
    for _ in range(warmup):
        step()

    def barrier():
        _sync()
        if dist is not None:
            dist.barrier()
        _sync()

    import gc
    gc.collect()                                         # no collector pause inside the timed region
    gc_was_on = gc.isenabled()
    gc.disable()
    barrier()
    t0 = time.perf_counter()
    for _ in range(steps):
        step()
    _sync()
    elapsed = time.perf_counter() - t0
    if gc_was_on:
        gc.enable()
    barrier()
    if dist is not None:
        t = torch.tensor([elapsed], dtype=torch.float64, device=device if device is not None else "cpu")
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        elapsed = float(t.item())
    return elapsed


def gather_counts(count: torch.Tensor, dist=None):
    """All-gather per-image kept counts of an image-sharded NMS (equal shard sizes)."""
    if dist is None:
        return count
    out = [torch.empty_like(count) for _ in range(dist.get_world_size())]
    dist.all_gather(out, count)
    return torch.cat(out)


class GradBuckets:
    """Data-parallel gradient exchange for the fused backward (no reference counterpart:
    the reference has no distributed code, the all-reduce sits where `train.py:67-68` has nothing).

    Gradients are not copied: every parameter's gradient tensor is a VIEW into a flat fp32 bucket
    (~``bucket_mb`` each, parameters in the order the backward produces them: heads first, stem last).
    When the backward has written the last gradient of a bucket, the bucket is all-reduced
    asynchronously (RCCL through ``torch.distributed``; the collective runs on RCCL's stream and waits
    for the compute stream at enqueue time), so the exchange of late layers overlaps with the
    dgrad / wgrad of earlier ones. ``finish()`` makes the compute stream wait for every bucket and
    turns sums into means (NCCL ``AVG`` when available). BatchNorm statistics stay per replica, as a
    non-distributed reference run would have them per batch.
    """

    def __init__(self, params_in_backward_order, dist, bucket_mb=25.0, device=None, only_trainable=True):
        self.dist = dist
        self.signature = None                 # ids of the member parameters, set by the train engine (rebuild when it changes)
        self.world = dist.get_world_size() if dist is not None else 1
        self.buckets = []                     # dict(buf, pending, total, work)
        self.slot = {}                        # id(param) -> (bucket index, offset, numel, shape)
        limit = int(bucket_mb * 1024 * 1024 / 4)
        cur, cur_n = [], 0
        groups = []
        for p in params_in_backward_order:
            if only_trainable and not p.requires_grad:
                continue
            if cur and cur_n + p.numel() > limit:
                groups.append(cur)
                cur, cur_n = [], 0
            cur.append(p)
            cur_n += p.numel()
        if cur:
            groups.append(cur)
        for bi, grp in enumerate(groups):
            n = sum(p.numel() for p in grp)
            dev = device if device is not None else grp[0].device
            buf = torch.zeros(n, dtype=torch.float32, device=dev)
            off = 0
            for p in grp:
                self.slot[id(p)] = (bi, off, p.numel(), tuple(p.shape))
                off += p.numel()
            self.buckets.append(dict(buf=buf, total=len(grp), pending=len(grp), work=None))
        self.use_avg = False
        if dist is not None and dist.get_backend() == "nccl":
            self.use_avg = True

    def begin(self, params=()):
        """Start of a backward. The gradient tensors handed to autograd are views into the buckets, and AccumulateGrad keeps
        such a view as ``p.grad`` when the parameter had none. If a ``p.grad`` from an earlier backward is still alive
        (``zero_grad(set_to_none=False)``, micro-batch accumulation) it IS the memory the kernels are about to overwrite, and
        autograd would then add the bucket to itself: give such a parameter its own copy first, so ``p.grad += new`` adds two
        different tensors like it does on one GPU."""
        for p in params:
            g = p.grad
            if g is not None and id(p) in self.slot:
                bi, off, n, _shape = self.slot[id(p)]
                base = self.buckets[bi]["buf"]
                lo = base.data_ptr() + off * 4
                if g.device == base.device and lo <= g.data_ptr() < lo + n * 4:
                    p.grad = g.clone()
        for b in self.buckets:
            b["pending"], b["work"] = b["total"], None

    def view(self, p):
        """Gradient storage of ``p`` inside its bucket."""
        bi, off, n, shape = self.slot[id(p)]
        return self.buckets[bi]["buf"].narrow(0, off, n).view(shape)

    def views_for(self, plist, flags):
        """Fresh gradient views for ``plist`` (None where ``flags`` is False or the parameter has no slot): one split per
        bucket + one reshape per parameter."""
        per_bucket = self.__dict__.get("_splits")
        if per_bucket is None:
            per_bucket = self._splits = []
            for bi in range(len(self.buckets)):
                members = sorted((off, n, pid, shape) for pid, (b, off, n, shape) in self.slot.items() if b == bi)
                per_bucket.append(([n for _o, n, _p, _s in members], [(pid, shape) for _o, _n, pid, shape in members]))
        got = {}
        for b, (sizes, members) in zip(self.buckets, per_bucket):
            for piece, (pid, shape) in zip(b["buf"].split_with_sizes(sizes), members):
                got[pid] = piece.view(shape)
        return [got.get(id(p)) if f else None for p, f in zip(plist, flags)]

    def ready(self, p):
        """The backward has finished writing the gradient of ``p``. Returns the bucket index when this completed a bucket whose
        all-reduce was started, else None."""
        bi = self.slot[id(p)][0]
        b = self.buckets[bi]
        b["pending"] -= 1
        if b["pending"] == 0 and self.dist is not None:
            self.fire(bi)
            return bi
        return None

    def fire(self, bi):
        """Start the all-reduce of bucket ``bi`` (every gradient in it has been enqueued on the current stream)."""
        b = self.buckets[bi]
        b["pending"] = 0
        if self.dist is not None:
            op = self.dist.ReduceOp.AVG if self.use_avg else self.dist.ReduceOp.SUM
            if torch.cuda.is_available() and torch.cuda.is_current_stream_capturing():
                # inside a HIP-graph capture: a synchronous collective (the process group joins its work back into the
                # capturing stream before returning), no work handle that would outlive the capture
                self.dist.all_reduce(b["buf"], op=op, async_op=False)
                b["work"] = None
            else:
                b["work"] = self.dist.all_reduce(b["buf"], op=op, async_op=True)

    def complete_all(self):
        """A replayed launch table has written every gradient: nothing is pending (buckets without a collective)."""
        for b in self.buckets:
            b["pending"] = 0

    def finish(self):
        for b in self.buckets:
            if b["pending"] != 0:
                raise RuntimeError("a gradient bucket was never completed")
            if b["work"] is not None:
                b["work"].wait()
                if not self.use_avg:
                    b["buf"].div_(self.world)

    def bytes(self):
        return sum(b["buf"].numel() for b in self.buckets) * 4


def data_parallel(model, dist, bucket_mb=25.0):
    """Enable data-parallel fine-tuning on ``model`` (one process per GPU): parameters are broadcast
    from rank 0 once, every backward all-reduces (averages) the gradients in buckets."""
    if dist is not None:
        for t in list(model.parameters()) + list(model.buffers()):
            dist.broadcast(t.data, src=0)
    model._engine.ddp = (dist, float(bucket_mb))
    model._engine.invalidate()
    return model
