"""Host-side engine: turns the module tree into a flat table of fused convolution launches
over NHWC buffers in HBM and replays it through ``libyolo_mi355x.so``.

Data layout in HBM (fp32 path)
  * activations: NHWC, one allocation per live tensor, recycled by exact size once dead
    (liveness pooling);
  * route / concat tensors (model.py:186-191): ONE allocation of Cu+Cr channels per concat; the
    route-producing residual stage writes its slice [Cu, Cu+Cr) directly, the 1x1 conv in front
    of nn.Upsample writes slice [0, Cu) with a 2x-upsampling store — no copy kernels;
  * weights: packed [Cout_pad][K_pad] (K = (kh,kw,ci)) + folded BatchNorm scale/shift, cached per
    block and refreshed when the parameter tensors change;
  * head outputs: (B,3,g,g,5+nc) contiguous, freshly allocated per call (callers mutate them).
"""
from __future__ import annotations

import ctypes as C
import os
import weakref
from dataclasses import dataclass
from typing import Optional

import torch
import torch.nn as nn

from . import _lib as L

_DT = {"fp32": (L.F32, torch.float32), "fp16": (L.F16, torch.float16), "bf16": (L.BF16, torch.bfloat16)}


def resolve_dtype(requested=None):
    """Compute dtype of a forward: an explicit request ("fp32" | "fp16" | "bf16"), else the active
    ``torch.autocast`` dtype (the reference wraps its training forward in autocast, train.py:53), else fp32."""
    if requested is not None:
        if requested not in _DT:
            raise ValueError(f"compute dtype must be one of {sorted(_DT)}, got {requested!r}")
        return requested
    if torch.is_autocast_enabled():
        dt = torch.get_autocast_dtype("cuda")
        return "bf16" if dt == torch.bfloat16 else "fp16"
    return "fp32"


BN_NOTE = "nn.BatchNorm2d eval semantics: scale = gamma / sqrt(var + eps), shift = beta - mean * scale"


@dataclass
class TView:
    """A (B,H,W,C) slice of an NHWC buffer."""
    buf: int
    C: int
    H: int
    W: int
    ld: int
    off: int


def _act_code(block):
    a = block.activation
    if a is None:
        return L.ACT_NONE
    if isinstance(a, nn.LeakyReLU):
        if abs(a.negative_slope - 0.1) > 1e-12:
            raise NotImplementedError("only LeakyReLU(0.1) is built into the epilogue")
        return L.ACT_LEAKY
    if isinstance(a, nn.Mish):
        return L.ACT_MISH
    raise NotImplementedError(f"activation {type(a).__name__}")


def _conv_geom(block):
    cv = block.conv
    k, s, p = cv.kernel_size[0], cv.stride[0], cv.padding[0]
    if cv.kernel_size[0] != cv.kernel_size[1] or cv.stride[0] != cv.stride[1] or k not in (1, 3) or s not in (1, 2) \
            or p != k // 2 or cv.groups != 1 or cv.dilation != (1, 1):
        raise NotImplementedError(f"conv config {cv} is outside the YOLOv3 set")
    return k, s


class Program:
    """Symbolic launch list for one input shape."""

    def __init__(self, batch):
        self.B = batch
        self.buf_numel = []          # per symbolic buffer
        self.ops = []                # dict(block, x, y, res, out_mode, flags, pred)
        self.n_pred = 0

    def new_buf(self, H, W, ld):
        self.buf_numel.append(self.B * H * W * ld)
        return len(self.buf_numel) - 1

    def emit_cnn(self, block, x: TView, res: Optional[TView] = None, dst: Optional[TView] = None,
                 out_mode=L.OUT_NHWC, nancheck=False, pred=None):
        k, s = _conv_geom(block)
        cout = block.conv.out_channels
        if block.conv.in_channels != x.C:
            raise ValueError(f"channel mismatch: conv expects {block.conv.in_channels}, tensor has {x.C}")
        Ho, Wo = (x.H + 2 * (k // 2) - k) // s + 1, (x.W + 2 * (k // 2) - k) // s + 1
        if out_mode == L.OUT_HEAD:
            y = None
        elif dst is not None:
            y = dst
        else:
            y = TView(self.new_buf(Ho, Wo, cout), cout, Ho, Wo, cout, 0)
        flags = (L.FLAG_RESIDUAL if res is not None else 0) | (L.FLAG_NANCHECK if nancheck else 0)
        self.ops.append(dict(block=block, x=x, y=y, res=res, out_mode=out_mode, flags=flags, pred=pred,
                             k=k, s=s, Ho=Ho, Wo=Wo))
        return y

    def emit_res(self, rb, x: TView, final_dst: Optional[TView] = None, nancheck=True):
        n = len(rb.layers)
        for j, seq in enumerate(rb.layers):
            a, b = seq[0], seq[1]
            t = self.emit_cnn(a, x)
            last = j == n - 1
            x = self.emit_cnn(b, t, res=x if rb.use_residual else None,
                              dst=final_dst if last else None, nancheck=nancheck and last)
        return x

    def emit_head(self, sp, x: TView):
        t = self.emit_cnn(sp.pred_block[0], x)
        self.emit_cnn(sp.pred_block[1], t, out_mode=L.OUT_HEAD, pred=self.n_pred)
        self.n_pred += 1


def build_network_program(model, B, S, ch_align=4):
    """Walk ``model.layers`` the way the reference forward does (model.py:172-193)."""
    from .model import CNNBlock, ResidualBlock, ScalePredictionBlock
    layers = list(model.layers)
    # pre-pass: pair each route (8-unit residual stage) with the nn.Upsample that pops it (LIFO)
    c, stack, pair = model.in_channels, [], {}
    for i, m in enumerate(layers):
        if isinstance(m, CNNBlock):
            c = m.conv.out_channels
        elif isinstance(m, ResidualBlock) and m.num_blocks == 8:
            stack.append((i, c))
        elif isinstance(m, nn.Upsample):
            if not stack:
                raise ValueError("nn.Upsample without a pending route")
            ri, cr = stack.pop()
            pair[ri] = pair[i] = dict(cu=c, cr=cr)
            c = c + cr
    prog = Program(B)
    cin_pad = (model.in_channels + ch_align - 1) // ch_align * ch_align
    cur = TView(prog.new_buf(S, S, cin_pad), model.in_channels, S, S, cin_pad, 0)
    prog.input = cur
    concat_view = {}                                  # upsample layer index -> TView of the whole concat
    for i, m in enumerate(layers):
        if isinstance(m, ScalePredictionBlock):
            prog.emit_head(m, cur)
            continue
        if isinstance(m, CNNBlock):
            nxt = layers[i + 1] if i + 1 < len(layers) else None
            if isinstance(nxt, nn.Upsample):
                info = pair[i + 1]
                if info["cu"] != m.conv.out_channels:
                    raise ValueError("unexpected channel count in front of nn.Upsample")
                cat = info["view"]
                dst = TView(cat.buf, info["cu"], cat.H, cat.W, cat.ld, 0)
                prog.emit_cnn(m, cur, dst=dst, out_mode=L.OUT_UPSAMPLE2X, nancheck=True)
                cur = TView(cat.buf, info["cu"], cat.H // 2, cat.W // 2, cat.ld, 0)   # pre-upsample logical view
            else:
                cur = prog.emit_cnn(m, cur, nancheck=True)
        elif isinstance(m, ResidualBlock):
            if i in pair:                              # route: write straight into the concat slice
                info = pair[i]
                ld = info["cu"] + info["cr"]
                buf = prog.new_buf(cur.H, cur.W, ld)
                info["view"] = TView(buf, ld, cur.H, cur.W, ld, 0)
                dst = TView(buf, info["cr"], cur.H, cur.W, ld, info["cu"])
                cur = prog.emit_res(m, cur, final_dst=dst)
            else:
                cur = prog.emit_res(m, cur)
        elif isinstance(m, nn.Upsample):
            cur = pair[i]["view"]                      # upsampled first, route second (model.py:190)
        else:
            raise NotImplementedError(type(m).__name__)
    return prog


# ----------------------------------------------------------------------------------------
class PackedBlock:
    """Device-side packed weights + folded BN of one CNNBlock."""

    def __init__(self, block, device, dtype="fp32"):
        cv = block.conv
        self.dtype = dtype
        self.code = _DT[dtype][0]
        lib = L.lib()
        stem_ok = bool(lib.yolo_stem_supported(cv.in_channels, cv.out_channels, cv.kernel_size[0], cv.stride[0]))
        n = lib.yolo_packed_weight_bytes(cv.out_channels, cv.in_channels, cv.kernel_size[0], self.code)
        if n == 0 and not (stem_ok and dtype != "fp32"):
            raise NotImplementedError(f"conv {cv.in_channels}->{cv.out_channels} k{cv.kernel_size[0]} has no {dtype} kernel "
                                      "(16-bit needs cin % 32 == 0)")
        self.w = torch.empty(max(n, 16), dtype=torch.uint8, device=device)
        self.scale = torch.empty(cv.out_channels, dtype=torch.float32, device=device)
        self.shift = torch.empty(cv.out_channels, dtype=torch.float32, device=device)
        self.stem_w = None
        self.packs_conv = n != 0
        if stem_ok:
            self.stem_w = torch.empty(27 * cv.out_channels, dtype=torch.float32, device=device)   # [27][cout]
        self.stamp = None                 # conv weight (data_ptr, version) the packed copy was made from
        self.fold_stamp = None            # same for the tensors behind scale / shift; None = not folded
        self.folded = False

    @staticmethod
    def stamp_of(block):
        """(weight stamp, fold stamp): identity + version of the conv weight, and of the tensors the folded scale / shift
        are made of. Versions only see writes PyTorch dispatched; tensors this library writes through raw pointers (the
        BatchNorm running statistics in a train-mode forward, every parameter in a HIP-graph replay) are covered by
        ``ModelState.mark_unfolded`` / ``invalidate`` instead."""
        w, rest = PackedBlock.stamp_tensors(block)
        gen = block.__dict__.get("_pack_gen", 0)       # CNNBlock.set_layers (the reference loader's hand-back) counts here
        return ((w.data_ptr(), w._version, gen),), tuple((t.data_ptr(), t._version, gen) for t in rest)

    @staticmethod
    def stamp_tensors(block):
        """(conv weight, the tensors behind scale / shift). Straight from the modules' dictionaries: nn.Module.__getattr__
        costs ~1 us per hop, and 75 blocks x 10 hops per forward was 0.5 ms of host time in front of the first launch -
        the GPU sat idle for it after every forward's NaN-flag sync (a 320 us gap per 3.5 ms bf16 forward in the trace)."""
        mods = block._modules
        conv = mods["conv"]
        w = conv._parameters["weight"]
        if block.batch_norm_act:
            bn = mods["batch_norm"]
            pr, bf = bn._parameters, bn._buffers
            return w, (pr["weight"], pr["bias"], bf["running_mean"], bf["running_var"])
        return w, (conv._parameters["bias"],)

    @staticmethod
    def stamp_slots(block):
        """(owner dictionary, key) of every tensor `stamp_tensors` returns, for the fast freshness walk."""
        mods = block._modules
        conv = mods["conv"]
        if block.batch_norm_act:
            bn = mods["batch_norm"]
            pr, bf = bn._parameters, bn._buffers
            return ((conv._parameters, "weight"), (pr, "weight"), (pr, "bias"), (bf, "running_mean"), (bf, "running_var"))
        return ((conv._parameters, "weight"), (conv._parameters, "bias"))

    def refresh(self, block, stream, fold_bn=True, conv_packed=False, pack=True):
        """fold_bn=False (training: batch statistics are used, not the running ones) skips the BN fold.
        conv_packed: the conv weights were already written by a batched pack (ModelState.refresh_weights).
        pack=False: the packed conv weights are current, only the folded scale / shift are refreshed."""
        lib = L.lib()
        cv = block.conv
        w = cv.weight.detach()
        if w.dtype != torch.float32 or not w.is_contiguous():
            w = w.float().contiguous()
        if pack:
            if self.packs_conv and not conv_packed:
                L.check(lib.yolo_pack_weights(w.data_ptr(), self.w.data_ptr(), cv.out_channels, cv.in_channels,
                                              cv.kernel_size[0], self.code, stream), "yolo_pack_weights")
            if self.stem_w is not None:
                L.check(lib.yolo_stem_pack(w.data_ptr(), self.stem_w.data_ptr(), cv.out_channels, stream), "yolo_stem_pack")
        self.folded = True
        if block.batch_norm_act and not fold_bn:
            self.folded = False
        elif block.batch_norm_act:
            bn = block.batch_norm
            g, b, m, v = (t.detach().float().contiguous() for t in (bn.weight, bn.bias, bn.running_mean, bn.running_var))
            L.check(lib.yolo_bn_fold(g.data_ptr(), b.data_ptr(), m.data_ptr(), v.data_ptr(), float(bn.eps),
                                     self.scale.data_ptr(), self.shift.data_ptr(), cv.out_channels, stream), "yolo_bn_fold")
        else:
            b = cv.bias.detach().float().contiguous()
            L.check(lib.yolo_bn_fold(0, b.data_ptr(), 0, 0, 0.0, self.scale.data_ptr(), self.shift.data_ptr(),
                                     cv.out_channels, stream), "yolo_bn_fold")
        self.stamp, fstamp = self.stamp_of(block)
        self.fold_stamp = fstamp if self.folded else None


class Plan:
    """Physical buffers + the ctypes launch table for one Program on one device."""

    def __init__(self, prog: Program, state: "ModelState", device, tile_override=None, use_stem=True, dtype="fp32"):
        self.prog = prog
        self.device = device
        self.dtype = dtype
        self.code, self.tdtype = _DT[dtype]
        B = prog.B
        n_ops = len(prog.ops)
        # ---- liveness pooling of activation buffers
        last_use = [-1] * len(prog.buf_numel)
        first_def = [n_ops] * len(prog.buf_numel)
        first_def[prog.input.buf] = -1
        for i, op in enumerate(prog.ops):
            for v in (op["x"], op["res"]):
                if v is not None:
                    last_use[v.buf] = max(last_use[v.buf], i)
            if op["y"] is not None:
                first_def[op["y"].buf] = min(first_def[op["y"].buf], i)
                last_use[op["y"].buf] = max(last_use[op["y"].buf], i)
        pool = {}
        self.phys = [None] * len(prog.buf_numel)
        self.total_bytes = 0

        def alloc(bid):
            numel = prog.buf_numel[bid]
            free = pool.get(numel)
            if free:
                self.phys[bid] = free.pop()
            else:
                self.phys[bid] = torch.empty(numel, dtype=self.tdtype, device=device)
                self.total_bytes += numel * self.phys[bid].element_size()
        # stem: the first block reads the caller's NCHW tensor directly (no NHWC copy of the input)
        op0 = prog.ops[0] if prog.ops else None
        self.stem = None
        if (use_stem and op0 is not None and op0["x"].buf == prog.input.buf and op0["res"] is None
                and op0["out_mode"] == L.OUT_NHWC and last_use[prog.input.buf] == 0
                and L.lib().yolo_stem_supported(op0["block"].conv.in_channels, op0["block"].conv.out_channels, op0["k"], op0["s"])):
            self.stem = op0
        else:
            alloc(prog.input.buf)
        for i in range(n_ops):
            for bid in range(len(prog.buf_numel)):
                if first_def[bid] == i and self.phys[bid] is None:
                    alloc(bid)
            for bid in range(len(prog.buf_numel)):
                if last_use[bid] == i and self.phys[bid] is not None and (bid != prog.input.buf or self.stem is not None):
                    pool.setdefault(prog.buf_numel[bid], []).append(self.phys[bid])
        # ---- launch table
        self.first = 1 if self.stem is not None else 0      # ops[first:] go through the conv launch table
        self.table = (L.ConvOp * n_ops)()
        self.pred_ops = {}
        self.blocks = []
        for i, op in enumerate(prog.ops):
            blk = op["block"]
            pk = state.packed(blk, device, dtype)
            self.blocks.append(blk)
            x, y, r = op["x"], op["y"], op["res"]
            e = self.table[i]
            d = e.d
            d.n, d.h, d.w = B, x.H, x.W
            d.cin, d.cout = blk.conv.in_channels, blk.conv.out_channels
            d.ksize, d.stride = op["k"], op["s"]
            d.x_ld, d.x_off = x.ld, x.off
            if y is not None:
                d.y_ld, d.y_off = y.ld, y.off
                e.y = self.phys[y.buf].data_ptr()
            if r is not None:
                d.r_ld, d.r_off = r.ld, r.off
                e.residual = self.phys[r.buf].data_ptr()
            d.act = _act_code(blk)
            d.out_mode = op["out_mode"]
            d.dtype = self.code
            d.flags = op["flags"]
            d.tile = tile_override or 0
            if i >= self.first:
                e.x = self.phys[x.buf].data_ptr()
            e.w_packed, e.scale, e.shift = pk.w.data_ptr(), pk.scale.data_ptr(), pk.shift.data_ptr()
            if i == 0 and self.stem is not None:
                self.stem_pk = pk
            if op["pred"] is not None:
                self.pred_ops[op["pred"]] = (i, op["Ho"], blk.conv.out_channels // 3)
        # one workspace for the launches that take one (fp32 3x3 stride 1 -> Winograd, yolo_conv_fwd_ws): the table runs on one
        # stream, so the largest request serves them all
        lib = L.lib()
        need = [lib.yolo_conv_workspace_bytes(C.byref(self.table[i].d)) if i >= self.first else 0 for i in range(n_ops)]
        self.workspace = torch.empty(max(need), dtype=torch.uint8, device=device) if n_ops and max(need) else None
        if self.workspace is not None:
            self.total_bytes += self.workspace.numel()
            for i, nb in enumerate(need):
                if nb:
                    self.table[i].workspace, self.table[i].workspace_bytes = self.workspace.data_ptr(), self.workspace.numel()
        self.blocks0 = self.blocks[:1]    # the stem block alone (its own freshness check, see ModelState.forward)
        self.nan_flag = torch.zeros(1, dtype=torch.int32, device=device)
        self.dropped = False

    def load_input(self, xin, stream):
        """NCHW fp32 input -> first activation: stem kernel, or the NHWC boundary copy."""
        lib = L.lib()
        B, Cc, H, W = xin.shape
        if self.stem is not None:
            y, pk = self.stem["y"], self.stem_pk
            L.check(lib.yolo_stem_fwd(xin.data_ptr(), pk.stem_w.data_ptr(), pk.scale.data_ptr(), pk.shift.data_ptr(),
                                      self.phys[y.buf].data_ptr(), B, H, W, y.C, y.ld, y.off, _act_code(self.stem["block"]),
                                      self.code, self.nan_flag.data_ptr(), stream), "yolo_stem_fwd")
        else:
            inp = self.prog.input
            L.check(lib.yolo_nchw_to_nhwc(xin.data_ptr(), self.phys[inp.buf].data_ptr(), B, Cc, H, W, inp.ld, self.code,
                                          self.nan_flag.data_ptr(), stream), "yolo_nchw_to_nhwc")

    def launch(self, stream):
        n = len(self.table) - self.first
        if n > 0:
            first = C.cast(C.byref(self.table, self.first * C.sizeof(L.ConvOp)), C.POINTER(L.ConvOp))
            L.check(L.lib().yolo_conv_fwd_batch(first, n, self.nan_flag.data_ptr(), stream), "yolo_conv_fwd_batch")


class ModelState:
    """Per-model caches (packed weights, plans). Holds no reference to the model so that the
    model stays picklable / deep-copyable."""

    def __init__(self, model=None):
        # keyed by the block MODULE (weakly): an id()-keyed cache could hand a new block the packed
        # weights of a dead one that happened to reuse its address and parameter storage
        self._packed = weakref.WeakKeyDictionary()
        self._defer_nan = False           # detect_images(): the forward leaves its NaN flag for the caller to read
        self._pending_flag = None
        self._gen = 0                     # bumped whenever packed state is declared out of date by hand (invalidate, mark_unfolded)
        self._fast = {}                   # (id(blocks), device, dtype) -> (gen, blocks, [(owner dict, key, tensor, version, address)])
        self._plans = {}
        # True: the NaN guards of model.py:175,183-184 raise inside the forward that saw the NaN (one host sync per forward).
        # "deferred" (train mode only): they raise at the start of the next forward, or from flush_nan() - the eager fine-tune
        # step then never waits for its own forward before enqueueing the backward. False: no guard.
        self.nan_check = True
        self._train_nan_pending = None   # not None: deferred guards are queued (see defer_nan)
        self._nan_host, self._nan_queue, self._nan_next = None, [], 0
        self.tile_override = None
        self.compute_dtype = None        # None: follow torch.autocast (fp32 outside it); or "fp32" / "fp16" / "bf16"
        self.ddp = None                  # (torch.distributed module, bucket MB) when data-parallel (dist.data_parallel)
        # Under an active torch.autocast the reference's forward returns its predictions in the autocast dtype (the head
        # conv runs in fp16 / bf16: model.py:145-148 under train.py:53). True = do the same (the heads are computed with fp32
        # accumulation and rounded ONCE); False = always hand out the fp32 heads. Outside autocast the heads are fp32 either way
        # (also with an explicit ``compute_dtype``): detect() / decode read them at full precision.
        self.autocast_heads = True
        # Plans own their activation buffers (a batch-64 608x608 training plan is ~60 GB): keep the most recently
        # used few, so multi-scale training (train.py:45-46 switches S every 10 batches) does not accumulate one
        # full set of buffers per size.
        self.max_train_plans = 2
        self.max_eval_plans = 4

    def _remember(self, key, plan):
        """Insert / refresh ``plan`` as most recently used and evict the oldest plans of its kind beyond the cap."""
        self._plans.pop(key, None)
        self._plans[key] = plan
        is_train = key[0] == "train"
        cap = self.max_train_plans if is_train else self.max_eval_plans
        # plans pinned by a captured HIP graph (GraphedTrainStep) hold pointers baked into the graph: never evicted,
        # and they do not count against the cap
        same = [k for k in self._plans if (k[0] == "train") == is_train and not getattr(self._plans[k], "pinned", False)]
        for k in same[:max(0, len(same) - cap)]:
            del self._plans[k]
        return plan

    def __getstate__(self):
        return {"nan_check": self.nan_check, "autocast_heads": self.autocast_heads}

    def __setstate__(self, st):
        self.__init__()
        self.nan_check = st.get("nan_check", True)
        self.autocast_heads = st.get("autocast_heads", True)

    def __deepcopy__(self, memo):
        new = ModelState()
        new.nan_check = self.nan_check
        new.autocast_heads = self.autocast_heads
        return new

    def head_dtype(self):
        """dtype the prediction tensors are handed out in (see ``autocast_heads``)."""
        if self.autocast_heads and torch.is_autocast_enabled():
            return torch.get_autocast_dtype("cuda")
        return torch.float32

    def invalidate(self, drop_plans=False):
        self._gen += 1
        self._fast.clear()
        for per_dev in self._packed.values():
            for pk in per_dev.values():
                pk.stamp = None
                pk.fold_stamp = None
                pk.folded = False
        if drop_plans:
            self._packed = weakref.WeakKeyDictionary()
            for plan in self._plans.values():
                plan.dropped = True              # a HIP graph holding raw pointers into this plan must not replay
            self._plans.clear()

    def packed(self, block, device, dtype="fp32"):
        per_dev = self._packed.get(block)
        if per_dev is None:
            per_dev = self._packed[block] = {}
        pk = per_dev.get((device.index, dtype))
        if pk is None:
            pk = per_dev[(device.index, dtype)] = PackedBlock(block, device, dtype)
        return pk

    def refresh_weights(self, blocks, device, stream, dtype="fp32", fold_bn=True):
        # fast path (eval plans, whose block list lives as long as the plan): nothing this library or PyTorch wrote since
        # the last full check of this very list of blocks - one flat walk over (owner dict, key, tensor, version, address)
        # instead of rebuilding 75 pairs of stamps. The walk re-reads the modules' CURRENT entries: a replaced Parameter
        # object (pruning, parametrize, `conv.weight = nn.Parameter(...)`) fails `is` and takes the slow path.
        # Training (fold_bn=False) never uses it: after an optimizer step every weight is stale anyway, and
        # `mark_unfolded` moves the generation on every train-mode forward.
        if not fold_bn:
            self._refresh_weights_slow(blocks, device, stream, dtype, fold_bn)
            return
        fkey = (id(blocks), device.index, dtype)
        fast = self._fast.get(fkey)
        if fast is not None and fast[0] == self._gen and fast[1] is blocks:
            for owner, name, t, ver, ptr in fast[2]:
                if owner[name] is not t or t._version != ver or t.data_ptr() != ptr:
                    break
            else:
                if all(blk.__dict__.get("_pack_gen", 0) == gen for blk, gen in fast[3]):
                    return
        self._refresh_weights_slow(blocks, device, stream, dtype, fold_bn)
        flat = []
        for blk in blocks:
            for owner, name in PackedBlock.stamp_slots(blk):
                t = owner[name]
                flat.append((owner, name, t, t._version, t.data_ptr()))
        if len(self._fast) > 64:
            self._fast.clear()
        self._fast[fkey] = (self._gen, blocks, flat, [(blk, blk.__dict__.get("_pack_gen", 0)) for blk in blocks])

    def _refresh_weights_slow(self, blocks, device, stream, dtype="fp32", fold_bn=True):
        stale, refold = [], []
        for blk in blocks:
            pk = self.packed(blk, device, dtype)
            wst, fst = PackedBlock.stamp_of(blk)
            if pk.stamp is None or pk.stamp != wst:
                stale.append((blk, pk))
            elif (fold_bn or not blk.batch_norm_act) and (not pk.folded or pk.fold_stamp != fst):
                # weights current, scale / shift not: a frozen conv with live running statistics (eval), or - in training
                # too, where the heads have no BatchNorm and feed scale / shift = (1, conv bias) to the kernel - a head
                # bias that changed on its own (prior-probability init after a first forward, partial load_state_dict)
                refold.append((blk, pk))
        # 16-bit: all stale conv weights in one launch per 48 layers (after an optimizer step that is every layer)
        batched = set()
        if dtype != "fp32" and len(stale) > 1:
            items, keep = [], []
            for blk, pk in stale:
                cv, w = blk.conv, blk.conv.weight.detach()
                if pk.packs_conv and w.dtype == torch.float32 and w.is_contiguous():
                    items.append(L.PackItem(w.data_ptr(), pk.w.data_ptr(), cv.out_channels, cv.in_channels, cv.kernel_size[0], 0))
                    keep.append(id(pk))
            if items:
                arr = (L.PackItem * len(items))(*items)
                L.check(L.lib().yolo_pack_weights_batch(C.cast(arr, C.c_void_p), len(items), 0, _DT[dtype][0], stream), "yolo_pack_weights_batch")
                batched = set(keep)
        for blk, pk in stale:
            pk.refresh(blk, stream, fold_bn, conv_packed=id(pk) in batched)
        for blk, pk in refold:
            pk.refresh(blk, stream, True, pack=False)

    def mark_unfolded(self, blocks):
        """The BatchNorm running statistics of ``blocks`` were just written by a kernel (train-mode forward): every folded
        scale / shift made from them, in any dtype, is out of date — whether or not the block was stale before."""
        self._gen += 1
        for blk in blocks:
            per_dev = self._packed.get(blk)
            if per_dev:
                for pk in per_dev.values():
                    pk.folded = False
                    pk.fold_stamp = None

    # ------------------------------------------------------------------ inference forward
    def forward(self, model, x):
        if not isinstance(x, torch.Tensor) or x.dim() != 4:
            raise ValueError("expected a (B,C,S,S) tensor")
        if not x.is_cuda:
            raise RuntimeError("yolo_for_turbines_amd runs on MI355X only: move the model and the input to the GPU "
                               "(there is no CPU fallback)")
        L.lib()                                                   # fail loudly if the HIP library is missing
        if self._train_nan_pending is not None and not (model.training and torch.is_grad_enabled()):
            self.flush_nan()                                      # (forward_train does it under the right device)
        if model.training and torch.is_grad_enabled():
            from . import train_engine
            return train_engine.forward_train(self, model, x)
        B, Cc, H, W = x.shape
        if Cc != model.in_channels or H != W or H % 32:
            raise ValueError(f"input must be (B,{model.in_channels},S,S) with S a multiple of 32, got {tuple(x.shape)}")
        if model.training:
            raise NotImplementedError("train-mode forward without autograd (batch statistics) is not supported; "
                                      "call model.eval() for inference")
        with torch.cuda.device(x.device):
            stream = L.current_stream()
            dt = resolve_dtype(self.compute_dtype)
            key = ("eval", B, H, x.device.index, self.tile_override, dt)
            plan = self._plans.get(key)
            if plan is None:
                prog = build_network_program(model, B, H, ch_align=8 if dt != "fp32" else 4)
                plan = Plan(prog, self, x.device, self.tile_override, dtype=dt)
                if dt != "fp32" and plan.stem is None:
                    raise NotImplementedError("the 16-bit path needs the 3->32 stem block as the first layer")
            self._remember(key, plan)
            xin = x.detach()
            if xin.dtype != torch.float32 or not xin.is_contiguous():
                xin = xin.float().contiguous()
            plan.nan_flag.zero_()
            if plan.stem is not None:
                # the first block only needs ITS weights: check those, get the stem kernel going, and walk the other 74
                # blocks while it runs (the GPU is idle from the previous forward's flag sync until this launch)
                self.refresh_weights(plan.blocks0, x.device, stream, dt)
                plan.load_input(xin, stream)
                self.refresh_weights(plan.blocks, x.device, stream, dt)
            else:
                self.refresh_weights(plan.blocks, x.device, stream, dt)
                plan.load_input(xin, stream)
            preds = []
            for k in range(plan.prog.n_pred):
                i, g, c3 = plan.pred_ops[k]
                out = torch.empty((B, 3, g, g, c3), dtype=torch.float32, device=x.device)
                plan.table[i].y = out.data_ptr()
                preds.append(out)
            plan.launch(stream)
            if self.nan_check and not self._defer_nan:
                self.raise_on_nan(plan.nan_flag)                  # the one host sync of a forward
            elif self.nan_check:
                self._pending_flag = plan.nan_flag                # detect_images(): checked after the post-processing is enqueued
            hd = self.head_dtype()
            if hd != torch.float32:
                preds = [t.to(hd) for t in preds]
        return preds

    # ---- deferred NaN guard (train mode, ``nan_check = "deferred"``)
    # The flag of every forward is copied (by an ordinary kernel: no blit path, no host sync) into a slot of pinned host memory,
    # followed by an event. Later forwards look at the slots whose event has completed - never waiting for the GPU - so the host
    # keeps running ahead of the device; flush_nan() waits for all of them.
    _NAN_SLOTS = 8

    def defer_nan(self, flag_tensor):
        if self._nan_host is None:
            self._nan_host = torch.zeros(self._NAN_SLOTS, dtype=torch.int32).pin_memory()
            self._nan_queue = []
        if len(self._nan_queue) >= self._NAN_SLOTS:
            self.poll_nan(block_oldest=True)
        k = self._nan_next % self._NAN_SLOTS
        self._nan_next += 1
        L.check(L.lib().yolo_copy_d2d(self._nan_host.data_ptr() + 4 * k, flag_tensor.data_ptr(), 4, L.current_stream()), "nan flag copy")
        ev = torch.cuda.Event()
        ev.record()
        self._nan_queue.append((ev, k))
        self._train_nan_pending = True

    def poll_nan(self, block_oldest=False, block_all=False):
        q = self._nan_queue
        while q:
            ev, k = q[0]
            if block_all or block_oldest:
                ev.synchronize()
                block_oldest = False
            elif not ev.query():
                break
            q.pop(0)
            flag = int(self._nan_host[k])
            if flag:
                for e2, _ in q:                                    # one exception per poisoned run: drop the younger reports
                    e2.synchronize()
                q.clear()
                self._train_nan_pending = None
                self._raise_flag(flag)
        if not q:
            self._train_nan_pending = None

    def flush_nan(self):
        """Wait for and read the NaN guards that train-mode forwards left pending (``nan_check = "deferred"``); raises like the
        forward would have."""
        if self._train_nan_pending is not None:
            self.poll_nan(block_all=True)

    def raise_on_nan(self, flag_tensor):
        # (against the 0.2 ms gap after an fp32 forward, tried and measured equal on the same box: a copy into pinned memory +
        # event wait, the same with a Python busy-poll on the event, ROC_ACTIVE_WAIT_TIMEOUT and HSA_ENABLE_INTERRUPT=0 -
        # 1,673-1,690 images/s either way, so the gap is not the host's wake-up)
        self._raise_flag(int(flag_tensor.item()))

    @staticmethod
    def _raise_flag(flag):
        assert not (flag & 1), "NaN in the input tensor"          # model.py:175
        if flag & 2:
            raise ValueError("Nan in layer")                      # model.py:183-184


# ------------------------------------------------------------------ stand-alone sub-modules
def module_state(module) -> ModelState:
    """The engine state of a block that is called on its own (model_tests.py:16-45 call CNNBlock / ResidualBlock /
    ScalePredictionBlock directly): created on first use and kept ON THE MODULE (`module._engine`, like `YOLOv3._engine`), so
    knobs such as `tile_override` / `compute_dtype` are per object - the library keeps no process-wide mutable state besides
    its handle caches."""
    st = module.__dict__.get("_engine")
    if st is None:
        st = module.__dict__["_engine"] = ModelState()
    return st


def run_module_nchw(module, x):
    """Run a CNNBlock / ResidualBlock / ScalePredictionBlock by itself on an NCHW tensor
    (the reference's block-level tests call them directly: model_tests.py:16-45)."""
    from .model import CNNBlock, ResidualBlock, ScalePredictionBlock
    if not x.is_cuda:
        raise RuntimeError("yolo_for_turbines_amd runs on MI355X only (no CPU fallback)")
    if module.training and any(isinstance(m, nn.BatchNorm2d) for m in module.modules()) and torch.is_grad_enabled():
        from . import train_engine
        return train_engine.run_module_train(module, x)
    lib = L.lib()
    B, Cc, H, W = x.shape
    st = module_state(module)
    dt = resolve_dtype(st.compute_dtype)
    with torch.cuda.device(x.device):
        stream = L.current_stream()
        prog = Program(B)
        al = 8 if dt != "fp32" else 4
        cpad = (Cc + al - 1) // al * al
        cur = TView(prog.new_buf(H, W, cpad), Cc, H, W, cpad, 0)
        prog.input = cur
        if isinstance(module, CNNBlock):
            out = prog.emit_cnn(module, cur)
        elif isinstance(module, ResidualBlock):
            out = prog.emit_res(module, cur, nancheck=False)
        elif isinstance(module, ScalePredictionBlock):
            prog.emit_head(module, cur)
            out = None
        else:
            raise NotImplementedError(type(module).__name__)
        try:
            return _run_module_plan(st, module, x, prog, cur, out, stream, dt)
        finally:
            st.invalidate(drop_plans=True)


def _run_module_plan(st, module, x, prog, cur, out, stream, dt="fp32"):
    lib = L.lib()
    B, Cc, H, W = x.shape
    cpad = cur.ld
    plan = Plan(prog, st, x.device, st.tile_override, use_stem=not st.tile_override, dtype=dt)
    st.refresh_weights(plan.blocks, x.device, stream, dt)
    xin = x.detach().float().contiguous()
    plan.load_input(xin, stream)
    result = None
    if out is None:
        i, g, c3 = plan.pred_ops[0]
        result = torch.empty((B, 3, g, g, c3), dtype=torch.float32, device=x.device)
        plan.table[i].y = result.data_ptr()
    plan.launch(stream)
    if out is not None:
        result = torch.empty((B, out.C, out.H, out.W), dtype=torch.float32, device=x.device)
        L.check(lib.yolo_nhwc_to_nchw(plan.phys[out.buf].data_ptr(), result.data_ptr(), B, out.C, out.H, out.W,
                                      out.ld, out.off, plan.code, stream), "yolo_nhwc_to_nchw")
    torch.cuda.current_stream().synchronize()      # plan buffers die with this call
    return result
