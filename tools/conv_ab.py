#!/usr/bin/env python3
"""Interleaved A/B timing of conv variants in ONE process (cdna_hip_programming.md rule 24: clocks drift between
processes and devices differ by ~10 %): every round times each variant once (reps launches), rounds alternate the order;
prints median and min per variant.

  python tools/conv_ab.py --dtype bf16 --layer c52_3x3 --tiles 6,8,9 [--residual] [--rounds 7] [--reps 10]
"""
import argparse, os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import torch
from yolo_for_turbines_amd import _lib as L
from tools.conv_bench import LAYERS

ap = argparse.ArgumentParser()
ap.add_argument("--layer", default="c52_3x3")
ap.add_argument("--tiles", default="6,8")
ap.add_argument("--dtype", default="bf16")
ap.add_argument("--batch", type=int, default=32)
ap.add_argument("--rounds", type=int, default=7)
ap.add_argument("--reps", type=int, default=10)
ap.add_argument("--residual", action="store_true")
ap.add_argument("--zeros", action="store_true", help="all-zero activations and weights (same instruction stream, minimal switching power)")
a = ap.parse_args()
dev = torch.device("cuda:0")
lib = L.lib()
code, tdt = {"fp32": (L.F32, torch.float32), "fp16": (L.F16, torch.float16), "bf16": (L.BF16, torch.bfloat16)}[a.dtype]
for name in a.layer.split(","):
    H, cin, cout, k, s = LAYERS[name]
    cpad = (cin + 3) // 4 * 4
    Ho = (H + 2 * (k // 2) - k) // s + 1
    x = torch.randn(a.batch * H * H * cpad, device=dev).to(tdt)
    w = torch.randn(cout, cin, k, k, device=dev) * (1.0 / (cin * k * k)) ** 0.5
    if a.zeros:
        x.zero_()
        w.zero_()
    wp = torch.empty(lib.yolo_packed_weight_bytes(cout, cin, k, code), dtype=torch.uint8, device=dev)
    st = L.current_stream()
    L.check(lib.yolo_pack_weights(w.data_ptr(), wp.data_ptr(), cout, cin, k, code, st))
    scale = torch.rand(cout, device=dev) + 0.5
    shift = torch.randn(cout, device=dev) * 0.1
    y = torch.empty(a.batch * Ho * Ho * cout, device=dev, dtype=tdt)
    r = torch.randn(a.batch * Ho * Ho * cout, device=dev).to(tdt) if a.residual else None
    flag = torch.zeros(1, dtype=torch.int32, device=dev)
    tiles = [int(t) for t in a.tiles.split(",")]
    descs = {t: L.ConvDesc(n=a.batch, h=H, w=H, cin=cin, cout=cout, ksize=k, stride=s, x_ld=cpad, x_off=0, y_ld=cout, y_off=0, r_ld=cout, r_off=0,
                           act=L.ACT_LEAKY, out_mode=L.OUT_NHWC, dtype=code, flags=(L.FLAG_RESIDUAL if a.residual else 0) | L.FLAG_NANCHECK, tile=t)
             for t in tiles}

    def launch(t):
        L.check(lib.yolo_conv_fwd(descs[t], x.data_ptr(), wp.data_ptr(), scale.data_ptr(), shift.data_ptr(), L.ptr(r), y.data_ptr(), flag.data_ptr(), st), "conv")
    for t in tiles:
        for _ in range(3):
            launch(t)
    torch.cuda.synchronize()
    res = {t: [] for t in tiles}
    for rd in range(a.rounds):
        order = tiles if rd % 2 == 0 else tiles[::-1]
        for t in order:
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            e0.record()
            for _ in range(a.reps):
                launch(t)
            e1.record()
            torch.cuda.synchronize()
            res[t].append(e0.elapsed_time(e1) / a.reps * 1e3)
    gflop = 2.0 * a.batch * Ho * Ho * cout * cin * k * k / 1e9
    for t in tiles:
        v = np.array(res[t])
        print(f"{name:10s} {a.dtype} res={int(a.residual)} tile={t}: median {np.median(v):7.1f} us  min {v.min():7.1f} us  ({gflop / np.median(v) * 1e3:6.1f} TF median)", flush=True)
