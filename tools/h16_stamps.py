#!/usr/bin/env python3
"""Per-block phase stamps of conv_patch_h16 (diagnostic build: `make -C yolo_for_turbines_amd/csrc stamps`, then YOLO_MI355X_LIB=yolo_for_turbines_amd/libyolo_mi355x_stamps.so).
Prints phase lengths (prologue / main loop / epilogue), per-CU concurrency and the gap between consecutive blocks
of one workgroup slot."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import torch
from yolo_for_turbines_amd import _lib as L
from tools.conv_bench import LAYERS

name = sys.argv[1] if len(sys.argv) > 1 else "c52_3x3"
tile = int(sys.argv[2]) if len(sys.argv) > 2 else 0
H, cin, cout, k, s = LAYERS[name]
B = 32
lib = L.lib()
dev = torch.device("cuda:0")
Ho = (H + 2 * (k // 2) - k) // s + 1
x = torch.randn(B * H * H * cin, device=dev).to(torch.bfloat16)
w = torch.randn(cout, cin, k, k, device=dev) * (1.0 / (cin * k * k)) ** 0.5
wp = torch.empty(lib.yolo_packed_weight_bytes(cout, cin, k, L.BF16), dtype=torch.uint8, device=dev)
st = L.current_stream()
L.check(lib.yolo_pack_weights(w.data_ptr(), wp.data_ptr(), cout, cin, k, L.BF16, st))
scale = torch.ones(cout, device=dev); shift = torch.zeros(cout, device=dev)
y = torch.empty(B * Ho * Ho * cout, device=dev, dtype=torch.bfloat16)
stamps = torch.zeros(8192 * 6, dtype=torch.int64, device=dev)
d = L.ConvDesc(n=B, h=H, w=H, cin=cin, cout=cout, ksize=k, stride=s, x_ld=cin, x_off=0, y_ld=cout, y_off=0, r_ld=cout, r_off=0,
               act=L.ACT_LEAKY, out_mode=L.OUT_NHWC, dtype=L.BF16, flags=0, tile=tile)
for _ in range(3):
    stamps.zero_()
    L.check(lib.yolo_conv_fwd(d, x.data_ptr(), wp.data_ptr(), scale.data_ptr(), shift.data_ptr(), 0, y.data_ptr(), stamps.data_ptr(), st))
torch.cuda.synchronize()
a = stamps.cpu().numpy().reshape(-1, 6)
a = a[a[:, 3] != 0]
nb = len(a)
t0, t1, t2, t3, hw, xcc = a.T
drain, sleep = (xcc >> 8) & 0xfffffff, xcc >> 36          # conv3_dma_h16 packs: cycles from last store issue to all stores done; stagger sleep
xcc = xcc & 0xff
base = t0.min()
print(f"{name} tile {tile}: {nb} blocks; kernel span {(t3.max() - base)} cycles")
if drain.max() > 0:
    print(f"  store drain after the last issue: mean {drain.mean():.0f} p90 {np.percentile(drain, 90):.0f}; stagger sleep mean {sleep.mean():.0f}")
for nm, v in (("prologue", t1 - t0), ("main loop", t2 - t1), ("epilogue", t3 - t2), ("total", t3 - t0)):
    print(f"  {nm:10s} mean {v.mean():9.0f}  p10 {np.percentile(v, 10):9.0f}  p90 {np.percentile(v, 90):9.0f}")
cu = ((xcc & 0xf) << 16) | (((hw >> 13) & 7) << 8) | ((hw >> 8) & 0xf)          # xcc, se, cu
ids = np.unique(cu)
print(f"  distinct (xcc,se,cu): {len(ids)}; blocks per CU mean {nb / len(ids):.2f}")
gaps, util = [], []
for c in ids[:256]:
    m = cu == c
    order = np.argsort(t0[m])
    s0, e0 = t0[m][order], t3[m][order]
    span = e0.max() - s0.min()
    busy = np.zeros(int(span) + 1, dtype=np.int8)
    util.append(sum(e - s for s, e in zip(s0, e0)) / span)
    # gap: for each block start (after the first two), time since the most recent block end on this CU
    for i in range(2, len(s0)):
        prev_end = e0[:i][e0[:i] <= s0[i] + 50]
        if len(prev_end):
            gaps.append(s0[i] - prev_end.max())
print(f"  avg resident blocks per CU over its span: {np.mean(util):.2f}; start-after-previous-end gap: median {np.median(gaps):.0f}, mean {np.mean(gaps):.0f} cycles")
first = np.sort(t0 - base)
print(f"  block start times: first wave by {first[min(511, nb - 1)]} cycles; last start {first[-1]}; last end {(t3 - base).max()}")
if os.environ.get("STAMPS_TIMELINE"):
    for c in ids[:int(os.environ["STAMPS_TIMELINE"])]:
        m = cu == c
        order = np.argsort(t0[m])
        b0 = t0[m].min()
        print(f"  CU {c:#x}: start / loop / epilogue / end (k cycles since the CU's first block)")
        for s_, a_, b_, e_ in zip(t0[m][order], t1[m][order], t2[m][order], t3[m][order]):
            print(f"    {(s_ - b0) / 1e3:7.1f} {(a_ - b0) / 1e3:7.1f} {(b_ - b0) / 1e3:7.1f} {(e_ - b0) / 1e3:7.1f}   loop {(b_ - a_) / 1e3:5.1f}")
