#!/usr/bin/env python3
"""Per-workgroup phase stamps of conv_wino_f32 (diagnostic build: `make -C yolo_for_turbines_amd/csrc wstamps`, then
YOLO_MI355X_LIB=yolo_for_turbines_amd/libyolo_mi355x_wstamps.so python tools/wino_stamps.py c52_3x3).
Prints prologue / main loop / epilogue lengths in s_memtime ticks, the per-CU round structure and the idle time between
consecutive workgroups of one CU."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import torch
from yolo_for_turbines_amd import _lib as L
from tools.conv_bench import LAYERS

name = sys.argv[1] if len(sys.argv) > 1 else "c52_3x3"
residual = len(sys.argv) > 2 and sys.argv[2] == "res"
tile = int(sys.argv[3]) if len(sys.argv) > 3 else 13          # 14: two passes, two workgroups per CU; 16 + bits: timing probes (1 no barrier, 2 no DMA, 4 no fragment reads)
H, cin, cout, k, s = LAYERS[name]
B = 32
lib = L.lib()
dev = torch.device("cuda:0")
x = torch.randn(B * H * H * cin, device=dev)
w = torch.randn(cout, cin, k, k, device=dev) * (1.0 / (cin * k * k)) ** 0.5
wp = torch.empty(lib.yolo_packed_weight_bytes(cout, cin, k, L.F32), dtype=torch.uint8, device=dev)
st = L.current_stream()
L.check(lib.yolo_pack_weights(w.data_ptr(), wp.data_ptr(), cout, cin, k, L.F32, st))
scale = torch.ones(cout, device=dev); shift = torch.zeros(cout, device=dev)
y = torch.empty(B * H * H * cout, device=dev)
r = torch.randn(B * H * H * cout, device=dev) if residual else None
stamps = torch.zeros(16384 * 6, dtype=torch.int64, device=dev)
d = L.ConvDesc(n=B, h=H, w=H, cin=cin, cout=cout, ksize=k, stride=s, x_ld=cin, x_off=0, y_ld=cout, y_off=0, r_ld=cout, r_off=0,
               act=L.ACT_LEAKY, out_mode=L.OUT_NHWC, dtype=L.F32, flags=L.FLAG_RESIDUAL if residual else 0, tile=tile)
need = lib.yolo_conv_workspace_bytes(d)
ws = torch.empty(need, dtype=torch.uint8, device=dev)
e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
for _ in range(3):
    stamps.zero_()
    e0.record()
    L.check(lib.yolo_conv_fwd_ws(d, x.data_ptr(), wp.data_ptr(), scale.data_ptr(), shift.data_ptr(), L.ptr(r), y.data_ptr(),
                                 ws.data_ptr(), need, stamps.data_ptr(), st))
    e1.record()
torch.cuda.synchronize()
a = stamps.cpu().numpy().reshape(-1, 6)
a = a[a[:, 3] != 0]
nb = len(a)
t0, t1, t2, t3, hw, xcc = a.T
base = t0.min()
span = t3.max() - base
print(f"{name} tile {tile}: {nb} workgroups; transform + GEMM launch pair {e0.elapsed_time(e1) * 1e3:.1f} us; GEMM kernel span {span} ticks")
for nm, v in (("prologue", t1 - t0), ("main loop", t2 - t1), ("epilogue", t3 - t2), ("total", t3 - t0)):
    print(f"  {nm:10s} mean {v.mean():9.0f}  p10 {np.percentile(v, 10):9.0f}  p90 {np.percentile(v, 90):9.0f}   ({100.0 * v.mean() / (t3 - t0).mean():.1f} % of a workgroup)")
print(f"  main loop per stage: {(t2 - t1).mean() / (cin // 4):.1f} ticks")
cu = ((xcc & 0xf) << 16) | (((hw >> 13) & 7) << 8) | ((hw >> 8) & 0xf)          # xcc, se, cu
ids = np.unique(cu)
print(f"  distinct (xcc,se,cu): {len(ids)}; workgroups per CU mean {nb / len(ids):.2f} max {max((cu == c).sum() for c in ids)}")
gaps, busy = [], []
for c in ids:
    m = cu == c
    order = np.argsort(t0[m])
    s0, e0_ = t0[m][order], t3[m][order]
    busy.append((e0_ - s0).sum() / span)
    gaps += list(s0[1:] - e0_[:-1])
print(f"  CU busy fraction of the kernel span: mean {np.mean(busy):.3f} min {np.min(busy):.3f}; gap between consecutive workgroups: mean {np.mean(gaps):.0f} p90 {np.percentile(gaps, 90):.0f}")
print(f"  last workgroup start / kernel span: {(t0.max() - base) / span:.3f}")
