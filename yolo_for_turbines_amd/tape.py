"""Record-and-replay of the train step's launches through ONE C call (`yolo_train_fwd_batch` / `yolo_train_bwd_batch`).

The reference's loop is eager (`/root/reference/code/train.py:41-82`), and so is a drop-in user's: forward, losses,
backward, optimizer step, statement by statement. The train-mode forward and the backward are ~900 launches here; issued from
Python through ctypes one at a time the host needs ~17.7 ms to enqueue what the GPU runs in ~17.3 ms. `train_engine` therefore
runs the ordinary per-launch path ONCE per (batch, size, dtype) plan with a recording proxy in front of the library, which
writes every call into a `yolo_call` table (function id + arguments), and replays the table with one FFI call afterwards.
Pointers that change from step to step (input batch, fresh prediction tensors, upstream gradients) are registered as SLOTS
before recording; every recorded argument that falls into a slot's byte range becomes a relocation.

Everything else a table refers to is owned by the plan or by the model and does not move: activation / statistics /
workspace buffers, packed weights, gradient buckets (`dist.GradBuckets`), parameters and BatchNorm buffers (a table checks
their addresses before it replays and is re-recorded when one has moved).
"""
from __future__ import annotations

import ctypes as C
import struct

from . import _lib as L


class CallTape:
    def __init__(self, kind):
        self.kind = kind                      # "fwd" | "bwd": which named entry point replays it
        self.calls = []                       # (fn id, [64-bit words])
        self.relocs = []                      # (call, arg, slot, offset)
        self.slot_names = []                  # slot index -> name
        self.slot_range = {}                  # name -> (index, base, nbytes) at record time
        self.keep = []                        # ctypes objects / tensors the table points at
        self.cuts = []                        # [(call index, payload)]: the host does something after call index - 1
        self.guards = []                      # (tensor, data_ptr) that must not have moved
        self._c = None

    # ---- recording
    def slot(self, name, base, nbytes):
        if name not in self.slot_range:
            self.slot_names.append(name)
        self.slot_range[name] = (self.slot_names.index(name), int(base), int(nbytes))

    def guard(self, tensors):
        self.guards = [(t, t.data_ptr()) for t in tensors]

    def cut(self, payload):
        self.cuts.append((len(self.calls), payload))

    def record(self, name, argtypes, args):
        words = []
        ci = len(self.calls)
        for k, (tp, v) in enumerate(zip(argtypes[:-1], args[:-1])):          # the trailing stream is supplied at replay
            if tp in (C.c_float, C.c_double):
                words.append(struct.unpack("<Q", struct.pack("<d", float(v)))[0])
                continue
            if tp in (C.c_int, C.c_int32, C.c_size_t, C.c_int64, C.c_uint64):
                words.append(int(v) & 0xFFFFFFFFFFFFFFFF)
                continue
            # pointers: raw integers / None, or ctypes objects (descriptors, stride arrays, item tables) passed by reference
            if v is None:
                ptr = 0
            elif isinstance(v, int):
                ptr = v
            elif isinstance(v, C.c_void_p):
                ptr = v.value or 0
                self.keep.append(v)                   # a cast object keeps the array it was made from alive
            else:
                ptr = C.addressof(v.contents) if hasattr(v, "contents") else C.addressof(v)
                self.keep.append(v)
            for sname, (si, base, nb) in self.slot_range.items():
                if ptr and base <= ptr < base + nb:
                    self.relocs.append((ci, k, si, ptr - base))
                    break
            words.append(ptr)
        if len(words) > L.CALL_MAX_ARGS:
            raise RuntimeError(f"{name}: {len(words)} arguments do not fit a yolo_call")
        self.calls.append((L.FN_IDS[name], words))

    def finish(self):
        n = len(self.calls)
        arr = (L.Call * max(n, 1))()
        for i, (fn, words) in enumerate(self.calls):
            arr[i].fn = fn
            for k, w in enumerate(words):
                arr[i].a[k] = w
        rel = (L.Reloc * max(len(self.relocs), 1))()
        for i, (ci, k, si, off) in enumerate(sorted(self.relocs)):
            rel[i].call, rel[i].arg, rel[i].slot, rel[i].offset = ci, k, si, off
        self._c = (arr, n, rel, len(self.relocs), (C.c_uint64 * max(len(self.slot_names), 1))())
        return self

    # ---- replay
    def fresh(self):
        for t, ptr in self.guards:
            if t.data_ptr() != ptr:
                return False
        return True

    def run(self, slots, stream, lo=0, hi=None):
        """Replay calls[lo:hi] with the given slot values ({name: address})."""
        arr, n, rel, nrel, sl = self._c
        hi = n if hi is None else hi
        for name, v in slots.items():
            sl[self.slot_range[name][0]] = int(v)
        fn = L.lib().yolo_train_fwd_batch if self.kind == "fwd" else L.lib().yolo_train_bwd_batch
        if lo == 0 and hi == n:
            L.check(fn(arr, n, rel, nrel, sl, len(self.slot_names), stream), "yolo_train_%s_batch" % self.kind)
            return
        # a segment (data parallel: the host fires an all-reduce between segments): its calls and their relocations
        first = C.cast(C.byref(arr, lo * C.sizeof(L.Call)), C.POINTER(L.Call))
        seg = self._segment_relocs(lo, hi)
        L.check(fn(first, hi - lo, seg[0], seg[1], sl, len(self.slot_names), stream), "yolo_train_%s_batch" % self.kind)

    def _segment_relocs(self, lo, hi):
        cache = self.__dict__.setdefault("_seg", {})
        got = cache.get((lo, hi))
        if got is None:
            items = [(ci - lo, k, si, off) for ci, k, si, off in sorted(self.relocs) if lo <= ci < hi]
            rel = (L.Reloc * max(len(items), 1))()
            for i, (ci, k, si, off) in enumerate(items):
                rel[i].call, rel[i].arg, rel[i].slot, rel[i].offset = ci, k, si, off
            got = cache[(lo, hi)] = (rel, len(items))
        return got


class RecordingLib:
    """Stands in front of the loaded library while a table is recorded: every call is executed AND written down."""

    def __init__(self, lib, tape: CallTape):
        self._lib, self._tape = lib, tape

    def __getattr__(self, name):
        fn = getattr(self._lib, name)
        if name not in L.FN_IDS:
            return fn                                  # size queries etc.: not launches
        argtypes = L._SIGS[name][1]
        tape = self._tape

        def call(*args):
            rc = fn(*args)
            if rc == 0:
                tape.record(name, argtypes, args)
            return rc
        self.__dict__[name] = call
        return call
