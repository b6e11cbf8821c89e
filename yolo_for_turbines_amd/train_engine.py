"""Training path: train-mode forward (batch-statistics BatchNorm) and the whole backward of the
network as ONE ``torch.autograd.Function`` over the same symbolic launch list as inference.

Replaces, for the blocks of `/root/reference/code/model.py:47-225`, what PyTorch autograd does under
`grad_scaler.scale(loss).backward()` (`/root/reference/code/train.py:53-69`): conv dgrad / wgrad,
BatchNorm(train) forward + backward, LeakyReLU / Mish backward, skip-add, route/concat and
nn.Upsample gradient routing. The loss (`loss.py`) and the optimizer stay PyTorch, as in the reference.

Per block the forward keeps only z (raw conv output) and the block output y (the next block's input);
u = BN(z) is recomputed from z in backward. Gradients are accumulated per BUFFER: every tensor's
gradient lives in a buffer with the geometry (ld) of its forward buffer, so concat slices, skip
connections and multi-consumer tensors are sums written by the gradient kernels' residual-add
epilogues — no separate add / slice / concat kernels.
"""
from __future__ import annotations

import ctypes as C
import os

import torch
import torch.nn as nn

from . import _lib as L
from .engine import _DT, PackedBlock, Program, TView, _act_code, build_network_program, resolve_dtype
from .tape import CallTape, RecordingLib

_const_cache = {}


def _consts(device, n=2048):
    key = device.index
    c = _const_cache.get(key)
    if c is None:
        c = _const_cache[key] = (torch.ones(n, dtype=torch.float32, device=device),
                                 torch.zeros(n, dtype=torch.float32, device=device))
    return c


def _desc(B, x: TView, cin, cout, k, s, y_ld, y_off, r: TView = None, act=L.ACT_NONE, out_mode=L.OUT_NHWC, flags=0, dtype=L.F32):
    d = L.ConvDesc(n=B, h=x.H, w=x.W, cin=cin, cout=cout, ksize=k, stride=s, x_ld=x.ld, x_off=x.off, y_ld=y_ld,
                   y_off=y_off, act=act, out_mode=out_mode, dtype=dtype, flags=flags, tile=0)
    if r is not None:
        d.r_ld, d.r_off = r.ld, r.off
        d.flags |= L.FLAG_RESIDUAL
    return d


_WGRAD_OVERLAP = os.environ.get("YOLO_WGRAD_OVERLAP", "1") != "0"      # A/B switch: weight gradients on a side stream
# Launch tables (tape.py): the whole-network forward / backward record their ~900 launches once per plan and replay them
# through ONE C call (yolo_train_fwd_batch / yolo_train_bwd_batch). YOLO_TRAIN_TAPE=0 keeps the per-launch path (A/B).
_TAPE = os.environ.get("YOLO_TRAIN_TAPE", "1") != "0"
# BatchNorm batch statistics from the epilogue of the producing convolution (16-bit LDS-DMA kernels) instead of a separate pass
# over z. YOLO_BN_FUSED_STATS=0 keeps the separate pass (A/B).
_FUSED_STATS = os.environ.get("YOLO_BN_FUSED_STATS", "1") != "0"
# 16-bit backward: the reduction pass of a block's BatchNorm backward (sum du, sum du * zhat: one read of dy and z) rides on the
# input-gradient convolution that writes the last contribution to dy (yolo_conv_dgrad_bstats). YOLO_BN_FUSED_BSTATS=0: separate pass.
_FUSED_BSTATS = os.environ.get("YOLO_BN_FUSED_BSTATS", "1") != "0"


class TrainPlan:
    """Buffers for one (batch, size, dtype): activations y, raw conv outputs z (both in the compute
    dtype), per-block statistics (fp32)."""

    def __init__(self, prog: Program, device, dtype="fp32"):
        self.prog, self.device, self.B = prog, device, prog.B
        self.dtype = dtype
        self.code, self.tdtype = _DT[dtype]
        f32 = dict(dtype=torch.float32, device=device)
        act = dict(dtype=self.tdtype, device=device)
        self.ybuf = [torch.empty(n, **act) for n in prog.buf_numel]
        # 16-bit: the 3-channel first block has no matrix-core kernel; it runs on the stem kernel
        op0 = prog.ops[0] if prog.ops else None
        self.stem = (dtype != "fp32" and op0 is not None and op0["x"].buf == prog.input.buf and
                     bool(L.lib().yolo_stem_supported(op0["block"].conv.in_channels, op0["block"].conv.out_channels, op0["k"], op0["s"])))
        self.z, self.stats = [], []
        self.fused_stats = []             # per op: (rows, ld) of the partial sums its convolution's epilogue writes, or None
        max_bn, max_wg, max_st, max_cw = 256, 256, 0, 0
        lib = L.lib()
        for op in prog.ops:
            blk, cv = op["block"], op["block"].conv
            m = self.B * op["Ho"] * op["Wo"]
            if blk.batch_norm_act and m <= 1:                   # nn.BatchNorm2d in train mode refuses this too
                raise ValueError("Expected more than 1 value per channel when training, got input size "
                                 f"torch.Size([{self.B}, {cv.out_channels}, {op['Ho']}, {op['Wo']}])")
            fs = None
            if blk.batch_norm_act:
                self.z.append(torch.empty(m * cv.out_channels, **act))
                self.stats.append(torch.empty(4, cv.out_channels, **f32))        # mean, invstd, scale, shift
                max_bn = max(max_bn, lib.yolo_bn_workspace_bytes(m, cv.out_channels))
                if _FUSED_STATS and dtype != "fp32":
                    d = _desc(self.B, op["x"], cv.in_channels, cv.out_channels, op["k"], op["s"], cv.out_channels, 0, dtype=self.code)
                    ld = C.c_int(0)
                    rows = lib.yolo_conv_stats_rows(d, C.byref(ld))
                    if rows > 0:
                        fs = (rows, ld.value)
                        max_st = max(max_st, rows * 2 * ld.value * 4)
            else:
                self.z.append(None)
                self.stats.append(None)
                max_bn = max(max_bn, lib.yolo_bn_workspace_bytes(m, (cv.out_channels + 31) // 32 * 32))
            self.fused_stats.append(fs)
            # caller-owned workspace of yolo_conv_fwd_ws (fp32 3x3 stride 1 -> Winograd): the raw forward convolution and the
            # input-gradient convolution (cin' = cout rounded up to 32, cout' = cin) of this block
            coutp = (cv.out_channels + 31) // 32 * 32
            for dd in (_desc(self.B, op["x"], cv.in_channels, cv.out_channels, op["k"], op["s"], cv.out_channels, 0, dtype=self.code),
                       _desc(self.B, TView(-1, coutp, op["Ho"], op["Wo"], coutp, 0), coutp, cv.in_channels, op["k"], 1,
                             op["x"].ld, op["x"].off, dtype=self.code)):
                if op["s"] == 1:
                    max_cw = max(max_cw, lib.yolo_conv_workspace_bytes(dd))
            max_wg = max(max_wg, lib.yolo_wgrad_workspace_bytes(self.B, op["x"].H, op["x"].W, cv.in_channels, cv.out_channels,
                                                                 op["k"], op["s"], self.code))
        self.bn_ws = torch.empty(max_bn, dtype=torch.uint8, device=device)
        self.st_ws = torch.empty(max(max_st, 16), dtype=torch.uint8, device=device)
        self.wg_ws = torch.empty(max_wg, dtype=torch.uint8, device=device)
        self.conv_ws = torch.empty(max(max_cw, 16), dtype=torch.uint8, device=device)
        self.nan_flag = torch.zeros(1, dtype=torch.int32, device=device)
        self.blocks = [op["block"] for op in prog.ops]
        self.fwd_tape = None              # tape.CallTape of the train-mode forward
        self.bwd_tapes = {}               # (need pattern, upstream-gradient layout) -> (CallTape, gradient list)
        self.dgrad_w = {}                 # op index -> packed gradient-conv weights
        self.buckets = None               # dist.GradBuckets when data-parallel
        self.gen = 0                      # bumped by every train-mode forward: a backward must see the buffers of ITS forward
        self.pinned = False               # set by GraphedTrainStep: pointers into this plan are baked into a HIP graph
        self.dropped = False              # set when the model dropped its plans (model.to(), ...)

    def view_ptr(self, v: TView):
        return self.ybuf[v.buf].data_ptr()


def _forward(state, model, plan: TrainPlan, x, use_tape=False):
    """Train-mode forward of the launch list. use_tape (whole network): record the launches the first time, replay them
    through one C call afterwards."""
    prog, B, dev = plan.prog, plan.B, plan.device
    stream = L.current_stream()
    state.refresh_weights(plan.blocks, dev, stream, plan.dtype, fold_bn=False)
    xin = x.detach()
    if xin.dtype != torch.float32 or not xin.is_contiguous():
        xin = xin.float().contiguous()
    preds = [None] * prog.n_pred
    for op in prog.ops:
        if op["pred"] is not None:
            g = op["Ho"]
            preds[op["pred"]] = torch.empty((B, 3, g, g, op["block"].conv.out_channels // 3), dtype=torch.float32, device=dev)
    tape = plan.fwd_tape if use_tape else None
    if tape is not None and tape.fresh():
        slots = {"x": xin.data_ptr()}
        for k, t in enumerate(preds):
            slots[f"pred{k}"] = t.data_ptr()
        tape.run(slots, stream)
    else:
        lib = L.lib()
        if use_tape:
            tape = CallTape("fwd")
            tape.slot("x", xin.data_ptr(), xin.numel() * 4)
            for k, t in enumerate(preds):
                tape.slot(f"pred{k}", t.data_ptr(), t.numel() * 4)
            lib = RecordingLib(lib, tape)
        post = _forward_launches(lib, state, plan, x, xin, preds, stream)
        if use_tape:
            tape.post = post
            tape.guard(list(model.parameters()) + list(model.buffers()))
            plan.fwd_tape = tape.finish()
    tracked, stats_written, bn_blocks = (tape.post if tape is not None else post)
    if tracked:
        torch._foreach_add_(tracked, 1)
        # yolo_bn_stats updated running_mean / running_var in place behind PyTorch's back: bump their version counters
        # (what an in-place torch op would have done) and drop every BN fold made from them — including the folds of
        # blocks whose WEIGHTS were not stale (frozen backbone), which an eval forward would otherwise keep using
        torch.autograd.graph.increment_version(stats_written)
        state.mark_unfolded(bn_blocks)
    return preds


def _forward_launches(lib, state, plan: TrainPlan, x, xin, preds, stream):
    prog, B, dev = plan.prog, plan.B, plan.device
    ones, zeros = _consts(dev)
    code = plan.code
    L.check(lib.yolo_fill_zero(plan.nan_flag.data_ptr(), 4, stream), "nan flag")
    inp = prog.input
    L.check(lib.yolo_nchw_to_nhwc(xin.data_ptr(), plan.ybuf[inp.buf].data_ptr(), B, x.shape[1], x.shape[2], x.shape[3], inp.ld,
                                  code, plan.nan_flag.data_ptr(), stream), "yolo_nchw_to_nhwc")
    tracked = []                                         # num_batches_tracked counters: ONE foreach launch, not 72
    stats_written, bn_blocks = [], []                    # running statistics written through raw pointers below
    for i, op in enumerate(prog.ops):
        blk, cv = op["block"], op["block"].conv
        pk = state.packed(blk, dev, plan.dtype)
        xv, yv, rv = op["x"], op["y"], op["res"]
        cout = cv.out_channels
        if not blk.batch_norm_act:                       # bare conv + bias (heads): single fused launch
            out = preds[op["pred"]]
            d = _desc(B, xv, cv.in_channels, cout, op["k"], op["s"], 0, 0, act=L.ACT_NONE, out_mode=op["out_mode"], dtype=code)
            L.check(lib.yolo_conv_fwd(d, plan.view_ptr(xv), pk.w.data_ptr(), pk.scale.data_ptr(), pk.shift.data_ptr(), 0,
                                      out.data_ptr(), plan.nan_flag.data_ptr(), stream), "yolo_conv_fwd(head)")
            continue
        z = plan.z[i]
        fs = plan.fused_stats[i] if not (i == 0 and plan.stem) else None
        if i == 0 and plan.stem:
            L.check(lib.yolo_stem_fwd(xin.data_ptr(), pk.stem_w.data_ptr(), ones.data_ptr(), zeros.data_ptr(), z.data_ptr(), B,
                                      xv.H, xv.W, cout, cout, 0, L.ACT_NONE, code, plan.nan_flag.data_ptr(), stream), "yolo_stem_fwd(raw)")
        elif fs is not None:              # raw conv + the BatchNorm partial sums of z from its epilogue: no statistics pass over z
            d = _desc(B, xv, cv.in_channels, cout, op["k"], op["s"], cout, 0, dtype=code)
            L.check(lib.yolo_conv_fwd_stats(d, plan.view_ptr(xv), pk.w.data_ptr(), z.data_ptr(), plan.st_ws.data_ptr(), plan.st_ws.numel(),
                                            stream), "yolo_conv_fwd_stats")
        else:
            d = _desc(B, xv, cv.in_channels, cout, op["k"], op["s"], cout, 0, dtype=code)
            L.check(lib.yolo_conv_fwd_ws(d, plan.view_ptr(xv), pk.w.data_ptr(), ones.data_ptr(), zeros.data_ptr(), 0, z.data_ptr(),
                                         plan.conv_ws.data_ptr(), plan.conv_ws.numel(), plan.nan_flag.data_ptr(), stream), "yolo_conv_fwd_ws(raw)")
        bn = blk.batch_norm
        st = plan.stats[i]
        m = B * op["Ho"] * op["Wo"]
        track = bn.track_running_stats and bn.running_mean is not None
        mom = 0.1 if bn.momentum is None else float(bn.momentum)
        if fs is not None:
            L.check(lib.yolo_bn_stats_from_partials(plan.st_ws.data_ptr(), fs[0], fs[1], m, cout, bn.weight.data_ptr(), bn.bias.data_ptr(),
                                                    mom, float(bn.eps), bn.running_mean.data_ptr() if track else 0,
                                                    bn.running_var.data_ptr() if track else 0, st[0].data_ptr(), st[1].data_ptr(),
                                                    st[2].data_ptr(), st[3].data_ptr(), stream), "yolo_bn_stats_from_partials")
        else:
            L.check(lib.yolo_bn_stats(z.data_ptr(), m, cout, cout, 0, bn.weight.data_ptr(), bn.bias.data_ptr(), mom, float(bn.eps),
                                      bn.running_mean.data_ptr() if track else 0, bn.running_var.data_ptr() if track else 0,
                                      st[0].data_ptr(), st[1].data_ptr(), st[2].data_ptr(), st[3].data_ptr(), code,
                                      plan.bn_ws.data_ptr(), plan.bn_ws.numel(), stream), "yolo_bn_stats")
        if track:
            tracked.append(bn.num_batches_tracked)
            stats_written += [bn.running_mean, bn.running_var]
            bn_blocks.append(blk)
        flag_ptr = plan.nan_flag.data_ptr() if (op["flags"] & L.FLAG_NANCHECK) else 0
        L.check(lib.yolo_bn_act_fwd(z.data_ptr(), cout, 0, st[0].data_ptr(), st[2].data_ptr(), st[3].data_ptr(),
                                    plan.view_ptr(rv) if rv is not None else 0, rv.ld if rv is not None else 0,
                                    rv.off if rv is not None else 0, plan.view_ptr(yv), yv.ld, yv.off, B, op["Ho"], op["Wo"],
                                    cout, _act_code(blk), op["out_mode"], code, flag_ptr, stream), "yolo_bn_act_fwd")
    return tracked, stats_written, bn_blocks


class _Grads:
    """Per-buffer gradient accumulation (see module docstring)."""

    def __init__(self, plan: TrainPlan, lib=None, retain=None):
        self.plan = plan
        self.state = {}                                # symbolic buf -> ("alias", TView-like) | ("own", tensor)
        self.pool = {}
        self.lib = lib if lib is not None else L.lib()
        self.retain = retain                           # list that keeps every allocation alive (a recorded table points at them)

    def _new(self, numel):
        free = self.pool.get(numel)
        if free:
            return free.pop()
        t = torch.empty(numel, dtype=self.plan.tdtype, device=self.plan.device)
        if self.retain is not None:
            self.retain.append(t)
        return t

    def _zero(self, t):
        L.check(self.lib.yolo_fill_zero(t.data_ptr(), t.numel() * t.element_size(), L.current_stream()), "zero")

    def release(self, buf):
        st = self.state.pop(buf, None)
        if st is not None and st[0] == "own":
            self.pool.setdefault(st[1].numel(), []).append(st[1])

    def get(self, v: TView):
        """(ptr, ld, off) of the gradient of forward view ``v`` (must exist)."""
        st = self.state[v.buf]
        if st[0] == "alias":
            ptr, ld, off0 = st[1]
            return ptr, ld, off0 + (v.off - st[2])
        return st[1].data_ptr(), v.ld, v.off

    def add_alias(self, v: TView, ptr, ld, off):
        """Contribution that already exists in memory (skip connection): no launch if it is the first."""
        if v.buf not in self.state:
            self.state[v.buf] = ("alias", (ptr, ld, off), v.off)
            return
        own = self._own(v)
        ones, zeros = _consts(self.plan.device)
        B = self.plan.B
        L.check(self.lib.yolo_bn_act_fwd(ptr, ld, off, 0, ones.data_ptr(), zeros.data_ptr(), own.data_ptr(), v.ld, v.off,
                                         own.data_ptr(), v.ld, v.off, B, v.H, v.W, v.C, L.ACT_NONE, L.OUT_NHWC, self.plan.code, 0,
                                         L.current_stream()), "grad add")

    def _own(self, v: TView):
        st = self.state.get(v.buf)
        if st is not None and st[0] == "own":
            return st[1]
        raise RuntimeError("gradient buffer expected")

    def target(self, v: TView):
        """Where a gradient kernel should write its contribution to ``v``:
        returns (out_ptr, out_ld, out_off, res_ptr, res_ld, res_off)."""
        st = self.state.get(v.buf)
        numel = self.plan.prog.buf_numel[v.buf]
        if st is None:
            t = self._new(numel)
            if v.C != v.ld:                             # a slice is written first: the rest must read as zero
                self._zero(t)
            self.state[v.buf] = ("own", t)
            return t.data_ptr(), v.ld, v.off, 0, 0, 0
        if st[0] == "alias":
            ptr, ld, off0 = st[1]
            t = self._new(numel)
            if v.C != v.ld:
                self._zero(t)
            self.state[v.buf] = ("own", t)
            return t.data_ptr(), v.ld, v.off, ptr, ld, off0 + (v.off - st[2])
        t = st[1]
        return t.data_ptr(), v.ld, v.off, t.data_ptr(), v.ld, v.off


def _bstats_map(plan: TrainPlan):
    """op index i -> op index P for the input-gradient launches that can also take block P's BatchNorm-backward sums:
    P is the only producer of i's input tensor (same channel view, plain NHWC store, BatchNorm block) and i is the FIRST
    consumer of that tensor in forward order - the backward visits it last, so its dgrad writes the complete gradient."""
    got = plan.__dict__.get("_bmap")
    if got is not None:
        return got
    prog = plan.prog
    producers, first_use = {}, {}
    for j, op in enumerate(prog.ops):
        if op["y"] is not None:
            producers.setdefault(op["y"].buf, []).append(j)
        for v, kind in ((op["x"], "x"), (op["res"], "res")):
            if v is not None and v.buf not in first_use:
                first_use[v.buf] = (j, kind)
            elif v is not None and first_use[v.buf][0] == j:
                first_use[v.buf] = (j, "both")
    bmap = {}
    if plan.dtype != "fp32" and _FUSED_BSTATS:
        for i, op in enumerate(prog.ops):
            xv = op["x"]
            ps = producers.get(xv.buf, [])
            if op["s"] != 1 or len(ps) != 1 or first_use.get(xv.buf) != (i, "x"):
                continue
            P = ps[0]
            po = prog.ops[P]
            yv = po["y"]
            if (P < i and po["block"].batch_norm_act and po["out_mode"] == L.OUT_NHWC and yv.off == xv.off and yv.C == xv.C
                    and yv.ld == xv.ld and xv.C == op["block"].conv.in_channels and _act_code(po["block"]) in (L.ACT_LEAKY, L.ACT_MISH)):
                bmap[i] = P
    plan._bmap = bmap
    return bmap


def _backward(state, model, plan: TrainPlan, dpreds, need, seeds=None, want_input_grad=False, buckets=None, tape=None):
    """need: dict param-id -> bool; seeds: {symbolic buf: gradient tensor} for stand-alone blocks.
    tape (a CallTape being recorded): every launch goes through the recording proxy, every temporary stays alive in the
    tape, bucket completions become cut points, and everything runs on one stream.
    Returns (dict param-id -> grad tensor, gradient buffer of the input or None)."""
    lib = L.lib() if tape is None else RecordingLib(L.lib(), tape)
    retain = None if tape is None else tape.keep
    prog, B, dev = plan.prog, plan.B, plan.device
    stream = L.current_stream()
    ones, zeros = _consts(dev)
    code = plan.code
    grads = {}

    def new_grad(p, shape=None):
        if buckets is not None and id(p) in buckets.slot:
            return buckets.view(p)
        return torch.empty(tuple(p.shape) if shape is None else shape, dtype=torch.float32, device=dev)

    def done(p):
        if buckets is not None and id(p) in buckets.slot:
            fired = buckets.ready(p)
            if fired is not None and tape is not None:
                tape.cut(fired)                                        # the host fires this bucket's all-reduce here

    def keep(t):
        if retain is not None:
            retain.append(t)
        return t

    G = _Grads(plan, lib, retain)
    bmap = _bstats_map(plan)
    pending, bst_pool = {}, {}                                         # block index -> (rows buffer, rows, ld, dy view) left by a dgrad
    for b, t in (seeds or {}).items():
        G.state[b] = ("own", t)
    dz_scratch = {}
    # Weight gradients on a SIDE stream (single GPU): wgrad + its split-K reduce only feed the optimizer, while the chain
    # dz -> dgrad -> BatchNorm backward of the previous block is what the next block waits for. The BatchNorm passes and the
    # reduce are HBM-bound, the convolutions matrix-bound, so the two streams complement each other on the CUs. dz scratch
    # buffers then rotate in pairs: a buffer is rewritten only after the wgrad that read it has finished (event), and the
    # main stream joins the side stream before this function returns (allocator reuse stays stream-ordered). With gradient
    # buckets (data parallel) everything stays on one stream: the all-reduce hooks are ordered against it.
    # Measured (bf16, B=32, 416^2): eager 22.8 -> 21.0 ms (fp32 80.0 -> 76.2): the second stream mostly fills the launch gaps of
    # the first. Replayed as ONE HIP graph there are no gaps and the concurrent kernels only disturb each other (19.0 -> 19.7
    # ms), so a capture keeps everything on one stream.
    overlap = buckets is None and tape is None and _WGRAD_OVERLAP and not torch.cuda.is_current_stream_capturing()
    main_s = torch.cuda.current_stream()
    side_s = None
    if overlap:
        side_s = getattr(plan, "side_stream", None)
        if side_s is None:
            side_s = plan.side_stream = torch.cuda.Stream(device=dev)
    busy = {}                                                          # id(scratch tensor) -> event of its last side-stream reader
    rot = {}

    def scratch(numel):
        if not overlap:
            t = dz_scratch.get(numel)
            if t is None:
                t = dz_scratch[numel] = keep(torch.empty(numel, dtype=plan.tdtype, device=dev))
            return t
        k = rot.get(numel, 0)
        rot[numel] = k ^ 1
        t = dz_scratch.get((numel, k))
        if t is None:
            t = dz_scratch[(numel, k)] = torch.empty(numel, dtype=plan.tdtype, device=dev)
        ev = busy.pop(id(t), None)
        if ev is not None:
            main_s.wait_event(ev)
        return t

    # Frozen backbone (`freeze=True`, model.py:306-309,330-334): nothing below the first block that owns a trainable
    # parameter needs a gradient, so the backward stops there (ops are in topological order: every producer of a block's
    # input has a smaller index) and that block itself skips its input gradient.
    def _trainable(op):
        blk = op["block"]
        ps = [blk.conv.weight] + ([blk.batch_norm.weight, blk.batch_norm.bias] if blk.batch_norm_act else [blk.conv.bias])
        return any(need.get(id(q), False) for q in ps)
    first_needed = 0 if want_input_grad else next((j for j, o in enumerate(prog.ops) if _trainable(o)), len(prog.ops))

    def wants_dgrad(i, op):
        return (op["x"].buf != prog.input.buf or want_input_grad) and i > first_needed - (1 if want_input_grad else 0)

    # 16-bit: the input-gradient weights of all stride-1 layers in one launch per 48 layers (they only depend on the master
    # weights, which do not change between forward and backward)
    prepacked = set()
    if plan.dtype != "fp32":
        items = []
        for i in range(first_needed, len(prog.ops)):
            op = prog.ops[i]
            cv = op["block"].conv
            w = cv.weight.detach()
            if op["s"] == 1 and wants_dgrad(i, op) and w.dtype == torch.float32 and w.is_contiguous():
                wp = plan.dgrad_w.get(i)
                if wp is None:
                    wp = plan.dgrad_w[i] = torch.empty(L.lib().yolo_packed_dgrad_bytes(cv.out_channels, cv.in_channels, op["k"], 1, code),
                                                       dtype=torch.uint8, device=dev)
                items.append(L.PackItem(w.data_ptr(), wp.data_ptr(), cv.out_channels, cv.in_channels, op["k"], 0))
                prepacked.add(i)
        if items:
            arr = keep((L.PackItem * len(items))(*items))
            L.check(lib.yolo_pack_weights_batch(C.cast(arr, C.c_void_p), len(items), 1, code, stream), "yolo_pack_weights_batch(dgrad)")

    for i in range(len(prog.ops) - 1, first_needed - 1, -1):
        op = prog.ops[i]
        blk, cv = op["block"], op["block"].conv
        xv, yv, rv = op["x"], op["y"], op["res"]
        cin, cout, k, s = cv.in_channels, cv.out_channels, op["k"], op["s"]
        Ho, Wo = op["Ho"], op["Wo"]
        m = B * Ho * Wo
        # ---------------------------------------------------------------- dz of this block
        if not blk.batch_norm_act:
            dp = dpreds[op["pred"]]
            coutp = (cout + 31) // 32 * 32
            dz = scratch(m * coutp)
            dz_ld = coutp
            if dp is None:
                L.check(lib.yolo_fill_zero(dz.data_ptr(), dz.numel() * dz.element_size(), stream), "zero dz")
            else:
                dp = dp.float()
                strides = (C.c_int64 * 5)(*dp.stride())
                L.check(lib.yolo_head_grad_to_nhwc(dp.data_ptr(), strides, dz.data_ptr(), B, Ho, cout // 3, coutp, code, stream),
                        "yolo_head_grad_to_nhwc")
            if need.get(id(cv.bias), False):
                db = keep(torch.empty(coutp, dtype=torch.float32, device=dev))
                L.check(lib.yolo_bn_act_bwd(dz.data_ptr(), coutp, 0, 0, 0, 0, 0, 0, 0, 0, 0, m, coutp, L.ACT_NONE, 0, db.data_ptr(),
                                            0, 0, 0, code, plan.bn_ws.data_ptr(), plan.bn_ws.numel(), stream), "bias grad")
                gb = new_grad(cv.bias)
                L.check(lib.yolo_copy_d2d(gb.data_ptr(), db.data_ptr(), cout * 4, stream), "bias grad copy")   # drops the channel padding
                grads[id(cv.bias)] = gb
                done(cv.bias)
        else:
            if op["out_mode"] == L.OUT_UPSAMPLE2X:      # y lives upsampled inside the concat buffer
                gptr, gld, goff = G.get(yv)
                dy = scratch(m * cout + 1)              # distinct key from dz of the same size
                L.check(lib.yolo_upsample2x_bwd(gptr, gld, goff, dy.data_ptr(), cout, 0, B, Ho, Wo, cout, code, stream), "upsample2x_bwd")
                dy_ptr, dy_ld, dy_off = dy.data_ptr(), cout, 0
            else:
                dy_ptr, dy_ld, dy_off = G.get(yv)
            if rv is not None:                          # skip connection: d(skip input) += dy
                G.add_alias(rv, dy_ptr, dy_ld, dy_off + 0)
            bn, st = blk.batch_norm, plan.stats[i]
            dz = scratch(m * cout)
            dz_ld = cout
            dgamma = new_grad(bn.weight)
            dbeta = new_grad(bn.bias)
            pend = pending.pop(i, None)
            if pend is not None and (pend[3], pend[4], pend[5]) == (dy_ptr, dy_ld, dy_off):
                # the convolution that wrote dy left the reduction's partial sums behind: finalize + apply only
                L.check(lib.yolo_bn_act_bwd_rows(dy_ptr, dy_ld, dy_off, plan.z[i].data_ptr(), cout, 0, bn.weight.data_ptr(), st[0].data_ptr(),
                                                 st[1].data_ptr(), st[2].data_ptr(), st[3].data_ptr(), m, cout, _act_code(blk),
                                                 dgamma.data_ptr(), dbeta.data_ptr(), dz.data_ptr(), cout, 0, code, pend[0].data_ptr(),
                                                 pend[1], pend[2], stream), "yolo_bn_act_bwd_rows")
                bst_pool.setdefault(pend[0].numel(), []).append(pend[0])
            else:
                L.check(lib.yolo_bn_act_bwd(dy_ptr, dy_ld, dy_off, plan.z[i].data_ptr(), cout, 0, bn.weight.data_ptr(), st[0].data_ptr(),
                                            st[1].data_ptr(), st[2].data_ptr(), st[3].data_ptr(), m, cout, _act_code(blk),
                                            dgamma.data_ptr(), dbeta.data_ptr(), dz.data_ptr(), cout, 0, code, plan.bn_ws.data_ptr(),
                                            plan.bn_ws.numel(), stream), "yolo_bn_act_bwd")
            grads[id(bn.weight)] = dgamma
            grads[id(bn.bias)] = dbeta
            done(bn.weight)
            done(bn.bias)
        # ---------------------------------------------------------------- wgrad
        if need.get(id(cv.weight), False):
            dw = new_grad(cv.weight)
            wstream = stream
            if overlap:
                e_dz = torch.cuda.Event()
                e_dz.record(main_s)
                side_s.wait_event(e_dz)
                dw.record_stream(side_s)
                wstream = side_s.cuda_stream
            L.check(lib.yolo_conv_wgrad(dz.data_ptr(), dz_ld, 0, plan.view_ptr(xv), xv.ld, xv.off, dw.data_ptr(), B, xv.H, xv.W, cin,
                                        cout, k, s, code, plan.wg_ws.data_ptr(), plan.wg_ws.numel(), wstream), "yolo_conv_wgrad")
            if overlap:
                e_w = torch.cuda.Event()
                e_w.record(side_s)
                busy[id(dz)] = e_w
            grads[id(cv.weight)] = dw
            done(cv.weight)
        # ---------------------------------------------------------------- dgrad into the input's gradient
        if wants_dgrad(i, op):
            # (the image itself needs no gradient — train.py never asks — and neither does anything under a frozen prefix)
            w = cv.weight.detach()
            flip = 1 if s == 1 else 0
            wp = plan.dgrad_w.get(i)
            if wp is None:
                wp = plan.dgrad_w[i] = torch.empty(L.lib().yolo_packed_dgrad_bytes(cout, cin, k, flip, code), dtype=torch.uint8, device=dev)
            if i not in prepacked:
                L.check(lib.yolo_pack_weights_dgrad(w.data_ptr(), wp.data_ptr(), cout, cin, k, flip, code, stream),
                        "yolo_pack_weights_dgrad")
            optr, old, ooff, rptr, rld, roff = G.target(xv)
            if s == 1:
                coutp = (cout + 31) // 32 * 32
                src = TView(-1, coutp, Ho, Wo, dz_ld, 0)
                d = _desc(B, src, coutp, cin, k, 1, old, ooff, dtype=code)
                if rptr:
                    d.r_ld, d.r_off = rld, roff
                    d.flags |= L.FLAG_RESIDUAL
                P = bmap.get(i)
                rows = 0
                if P is not None and P >= first_needed and not overlap:
                    rl = plan.__dict__.setdefault("_brows", {}).get((i, d.flags))
                    if rl is None:
                        ldv = C.c_int(0)
                        rl = plan._brows[(i, d.flags)] = (L.lib().yolo_conv_bstats_rows(d, C.byref(ldv)), ldv.value)
                    rows, rld_ = rl
                if rows:
                    pst = plan.stats[P]
                    numel = rows * 2 * rld_ + 3 * cin
                    free = bst_pool.get(numel)
                    bst = free.pop() if free else keep(torch.empty(numel, dtype=torch.float32, device=dev))
                    L.check(lib.yolo_conv_dgrad_bstats(d, dz.data_ptr(), wp.data_ptr(), rptr, optr, plan.z[P].data_ptr(), cin, 0,
                                                       pst[0].data_ptr(), pst[2].data_ptr(), pst[3].data_ptr(),
                                                       _act_code(prog.ops[P]["block"]), bst.data_ptr(), numel * 4, stream),
                            "yolo_conv_dgrad_bstats")
                    pending[P] = (bst, rows, rld_, optr, old, ooff)
                else:
                    L.check(lib.yolo_conv_fwd_ws(d, dz.data_ptr(), wp.data_ptr(), ones.data_ptr(), zeros.data_ptr(), rptr, optr,
                                                 plan.conv_ws.data_ptr(), plan.conv_ws.numel(), 0, stream), "dgrad (stride 1)")
            else:
                L.check(lib.yolo_conv_dgrad_s2(dz.data_ptr(), dz_ld, 0, wp.data_ptr(), rptr, rld, roff, optr, old, ooff, B, Ho, Wo,
                                               cin, cout, code, stream), "yolo_conv_dgrad_s2")
        # the gradient of this block's output is consumed: recycle its buffer (unless shared with a concat slice
        # whose other producer has not been processed yet)
        if yv is not None and op["out_mode"] != L.OUT_UPSAMPLE2X and not _shared_later(prog, i, yv.buf):
            G.release(yv.buf)
        elif yv is not None and op["out_mode"] == L.OUT_UPSAMPLE2X and not _shared_later(prog, i, yv.buf):
            G.release(yv.buf)
    if overlap:
        main_s.wait_stream(side_s)                                     # join: every gradient is ready in main-stream order
    gin = G.state.get(prog.input.buf)
    return grads, (gin[1] if (want_input_grad and gin is not None and gin[0] == "own") else None)


def _shared_later(prog, i, buf):
    """True if an op processed LATER in backward (index < i) also writes this buffer (concat producers)."""
    return any(prog.ops[j]["y"] is not None and prog.ops[j]["y"].buf == buf for j in range(i))


def _bucket_order(plan, need):
    order = []
    for op in reversed(plan.prog.ops):               # the order in which _backward produces gradients
        blk = op["block"]
        if blk.batch_norm_act:
            order += [blk.batch_norm.weight, blk.batch_norm.bias, blk.conv.weight]
        else:
            order += [blk.conv.bias, blk.conv.weight]
    return [p for p in order if need.get(id(p), False)]


class YoloTrainFn(torch.autograd.Function):
    @staticmethod
    def forward(ctx, x, holder, *params):
        state, model, plan, _plist = holder
        plan.gen += 1
        ctx.gen = plan.gen
        preds = _forward(state, model, plan, x, use_tape=_TAPE)
        ctx.holder = holder
        return tuple(preds)

    @staticmethod
    def backward(ctx, *dpreds):
        state, model, plan, plist = ctx.holder
        if plan.gen != ctx.gen:
            raise RuntimeError("yolo_for_turbines_amd: a newer train-mode forward of the same (batch, size, dtype) has overwritten "
                               "the activations this backward needs (one set of buffers per shape): call backward() before the "
                               "next forward, or run the second forward under torch.no_grad() / model.eval()")
        flags = tuple(ctx.needs_input_grad[2:])
        need = {id(p): f for p, f in zip(plist, flags)}
        with torch.cuda.device(plan.device):
            # Gradient storage: flat fp32 buckets in backward-production order (dist.GradBuckets). Data parallel: each bucket is
            # all-reduced as soon as its last gradient has been enqueued. Single GPU with launch tables: the same buckets
            # without a collective - they give every gradient a fixed address, which is what lets the table be replayed.
            buckets = None
            if state.ddp is not None or _TAPE:
                cached = plan.__dict__.setdefault("_orders", {}).get(flags)
                if cached is None or cached[2] is not plist:
                    order = _bucket_order(plan, need)
                    cached = plan._orders[flags] = (order, tuple(id(p) for p in order), plist)
                order, sig = cached[0], cached[1]
                buckets = plan.buckets
                if buckets is None or buckets.signature != sig:      # first backward, or the trainable set changed (unfreeze)
                    from .dist import GradBuckets
                    dd = state.ddp if state.ddp is not None else (None, 25.0)
                    buckets = plan.buckets = GradBuckets(order, dd[0], dd[1], plan.device, only_trainable=False)
                    buckets.signature = sig
                    plan.bwd_tapes.clear()                           # the tables point into the old buckets
                buckets.begin(order)
            if not _TAPE:
                grads, _ = _backward(state, model, plan, list(dpreds), need, buckets=buckets)
                out = [grads.get(id(p)) if need[id(p)] else None for p in plist]
            else:
                out = _backward_taped(state, model, plan, dpreds, need, flags, plist, buckets)
            if buckets is not None:
                buckets.finish()
        return (None, None, *out)


def _backward_taped(state, model, plan, dpreds, need, flags, plist, buckets):
    dps = [None if d is None else (d if d.dtype == torch.float32 else d.float()) for d in dpreds]
    key = (flags, tuple(None if d is None else tuple(d.stride()) for d in dps))
    stream = L.current_stream()
    got = plan.bwd_tapes.get(key)
    if got is not None and got[0].fresh():
        tape = got[0]
        slots = {f"dp{k}": d.data_ptr() for k, d in enumerate(dps) if d is not None}
        lo = 0
        for ci, bi in tape.cuts:                                     # data parallel: an all-reduce starts behind its last producer
            if ci > lo:
                tape.run(slots, stream, lo, ci)
                lo = ci
            buckets.fire(bi)
        tape.run(slots, stream, lo, None)
        buckets.complete_all()
        # FRESH view objects every step: AccumulateGrad keeps an incoming gradient as p.grad without copying only when nobody
        # else holds a reference to it; a cached list of views made it clone all 222 gradients (222 memcpys, ~3 ms per step)
        return buckets.views_for(plist, flags)
    tape = CallTape("bwd")
    for k, d in enumerate(dps):
        if d is not None:
            lo_addr = d.data_ptr()
            span = (sum((n - 1) * st for n, st in zip(d.shape, d.stride())) + 1) * 4
            tape.slot(f"dp{k}", lo_addr, span)
    grads, _ = _backward(state, model, plan, dps, need, buckets=buckets, tape=tape)
    out = [grads.get(id(p)) if need[id(p)] else None for p in plist]
    tape.guard(list(model.parameters()) + list(model.buffers()))
    if len(plan.bwd_tapes) >= 2:                                     # each table keeps its temporaries alive: bound them
        plan.bwd_tapes.pop(next(iter(plan.bwd_tapes)))
    plan.bwd_tapes[key] = (tape.finish(),)
    return out


def _same_params(plan, plist):
    """Cheap identity check of a cached ``list(model.parameters())``: the plan's blocks are in module order, so their own
    parameter dictionaries (conv.weight, [conv.bias], [bn.weight, bn.bias]) enumerate the same objects in the same order."""
    i = 0
    n = len(plist)
    for blk in plan.blocks:
        mods = blk._modules
        cp = mods["conv"]._parameters
        if blk.batch_norm_act:
            bp = mods["batch_norm"]._parameters
            own = (cp["weight"], bp["weight"], bp["bias"])
        else:
            own = (cp["weight"], cp["bias"])
        for prm in own:
            if i >= n or plist[i] is not prm:
                return False
            i += 1
    return i == n


def forward_train(state, model, x):
    B, Cc, H, W = x.shape
    if Cc != model.in_channels or H != W or H % 32:
        raise ValueError(f"input must be (B,{model.in_channels},S,S) with S a multiple of 32, got {tuple(x.shape)}")
    with torch.cuda.device(x.device):
        if state._train_nan_pending is not None:
            state.poll_nan()                              # guards left by earlier train-mode forwards (nan_check = "deferred"): never waits
        dt = resolve_dtype(state.compute_dtype)           # autocast (train.py:53) selects the 16-bit kernels
        key = ("train", B, H, x.device.index, dt)
        plan = state._plans.get(key)
        if plan is None:
            prog = build_network_program(model, B, H, ch_align=8 if dt != "fp32" else 4)   # 16-bit kernels read 8-channel pieces
            plan = TrainPlan(prog, x.device, dt)
            if dt != "fp32" and not plan.stem:
                raise NotImplementedError("the 16-bit path needs the 3->32 stem block as the first layer")
        state._remember(key, plan)
        # (walking the module tree for model.parameters() costs 1.3 ms per step: the list is cached per plan and checked
        #  against the parameter count and the identity of its ends - replacing a Parameter object drops the plans anyway
        #  through the address guards of the launch tables)
        plist = plan.__dict__.get("_plist")
        if plist is None or not _same_params(plan, plist):
            plist = plan._plist = [p for p in model.parameters()]
        holder = (state, model, plan, plist)
        preds = YoloTrainFn.apply(x, holder, *plist)
        if state.nan_check == "deferred":
            # The guard's flag read is a host sync that waits for this forward's own kernels, and the loss / backward can only be
            # enqueued after it: ~1 ms of idle queue per eager step. Deferred: the flag goes to pinned host memory behind the
            # forward and is looked at by later forwards once its event has completed (or by flush_nan()) - same exception, a
            # step or two later, and the host never waits for the device.
            state.defer_nan(plan.nan_flag)
        elif state.nan_check:
            state.raise_on_nan(plan.nan_flag)
        hd = state.head_dtype()
        if hd != torch.float32:                       # what autocast hands the reference's loss (train.py:53-65); the cast is an
            preds = [t.to(hd) for t in preds]         # autograd op, so the incoming gradient is widened back to fp32
    return list(preds)


def run_module_train(module, x):
    """Stand-alone CNNBlock / ResidualBlock in training mode under autograd (block-level parity tests)."""
    from .engine import module_state
    from .model import CNNBlock, ResidualBlock
    B, Cc, H, W = x.shape
    with torch.cuda.device(x.device):
        mst = module_state(module)
        dt = resolve_dtype(mst.compute_dtype)
        prog = Program(B)
        al = 8 if dt != "fp32" else 4
        cpad = (Cc + al - 1) // al * al
        cur = TView(prog.new_buf(H, W, cpad), Cc, H, W, cpad, 0)
        prog.input = cur
        if isinstance(module, CNNBlock):
            out = prog.emit_cnn(module, cur)
        elif isinstance(module, ResidualBlock):
            out = prog.emit_res(module, cur, nancheck=False)
        else:
            raise NotImplementedError("stand-alone training is provided for CNNBlock and ResidualBlock")
        plan = TrainPlan(prog, x.device, dt)
        return _ModuleTrainFn.apply(x, (mst, module, plan, out), *list(module.parameters()))


class _ModuleTrainFn(torch.autograd.Function):
    """Block-level variant: returns the block output as an NCHW tensor and also produces dx."""

    @staticmethod
    def forward(ctx, x, holder, *params):
        state, module, plan, out = holder
        _forward(state, module, plan, x)
        lib = L.lib()
        y = torch.empty((plan.B, out.C, out.H, out.W), dtype=torch.float32, device=x.device)
        L.check(lib.yolo_nhwc_to_nchw(plan.ybuf[out.buf].data_ptr(), y.data_ptr(), plan.B, out.C, out.H, out.W, out.ld, out.off,
                                      plan.code, L.current_stream()), "yolo_nhwc_to_nchw")
        ctx.holder = holder
        ctx.plist = list(module.parameters())
        ctx.xshape = tuple(x.shape)
        return y

    @staticmethod
    def backward(ctx, dy):
        state, module, plan, out = ctx.holder
        lib = L.lib()
        stream = L.current_stream()
        dev = plan.device
        B = plan.B
        need = {id(p): ctx.needs_input_grad[2 + j] for j, p in enumerate(ctx.plist)}
        # seed the gradient of the block output (NCHW -> NHWC) and make the input differentiable
        dyc = dy.float().contiguous()
        g_out = torch.empty(plan.prog.buf_numel[out.buf], dtype=plan.tdtype, device=dev)
        L.check(lib.yolo_nchw_to_nhwc(dyc.data_ptr(), g_out.data_ptr(), B, out.C, out.H, out.W, out.ld, plan.code, 0, stream), "seed")
        grads, gx = _backward(state, module, plan, [], need, seeds={out.buf: g_out}, want_input_grad=ctx.needs_input_grad[0])
        dx = None
        if gx is not None:
            Bc, Cc, H, W = ctx.xshape
            dx = torch.empty(ctx.xshape, dtype=torch.float32, device=dev)
            inp = plan.prog.input
            L.check(lib.yolo_nhwc_to_nchw(gx.data_ptr(), dx.data_ptr(), Bc, Cc, H, W, inp.ld, 0, plan.code, stream), "dx")
        return (dx, None, *[grads.get(id(p)) if need[id(p)] else None for p in ctx.plist])
