#!/usr/bin/env python3
"""Random-shape stress of yolo_conv_fwd against torch's own convolution on the same GPU (fp64 reference):
python tools/conv_stress.py [n_shapes] [dtype=bf16|fp16|fp32] [seed]. Odd spatial sizes, batch 1-5, strides 1/2, k 1/3,
channels in multiples of 32 (16-bit) or 4 (fp32), residual / activation variants. Prints the worst relative error."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import torch
import torch.nn.functional as F
from yolo_for_turbines_amd import _lib as L

n_shapes = int(sys.argv[1]) if len(sys.argv) > 1 else 100
dtype = sys.argv[2] if len(sys.argv) > 2 else "bf16"
seed = int(sys.argv[3]) if len(sys.argv) > 3 else 0
code, tdt, tol = {"fp32": (L.F32, torch.float32, 2e-4), "fp16": (L.F16, torch.float16, 4e-3), "bf16": (L.BF16, torch.bfloat16, 3e-2)}[dtype]
rng = np.random.default_rng(seed)
lib, dev = L.lib(), torch.device("cuda:0")
worst, fails = 0.0, 0
for it in range(n_shapes):
    k = int(rng.choice([1, 3, 3]))
    s = int(rng.choice([1, 2])) if k == 3 else 1
    step = 32 if dtype != "fp32" else 4
    cin = int(rng.integers(1, 9)) * 32                      # the kernels take cin <= 4 (stem) or a multiple of 32
    cout = int(rng.integers(1, 300))
    if dtype != "fp32":
        cout = (cout + 7) // 8 * 8
    H, W = int(rng.integers(3, 60)), int(rng.integers(3, 60))
    if s == 2:
        H += H & 1                                          # stride 2 needs even H, W (the network's sizes are multiples of 32)
    W = H
    B = int(rng.integers(1, 6))
    act = int(rng.choice([L.ACT_NONE, L.ACT_LEAKY, L.ACT_MISH]))
    residual = bool(rng.integers(0, 2)) and s == 1 and cin == cout and k == 3
    Ho = (H + 2 * (k // 2) - k) // s + 1
    x = torch.randn(B, cin, H, W, device=dev)
    w = torch.randn(cout, cin, k, k, device=dev) * (1.0 / (cin * k * k)) ** 0.5
    scale = torch.rand(cout, device=dev) + 0.5
    shift = torch.randn(cout, device=dev) * 0.1
    xh = x.permute(0, 2, 3, 1).contiguous().to(tdt)
    r = torch.randn(B, Ho, Ho, cout, device=dev).to(tdt) if residual else None
    nbytes = lib.yolo_packed_weight_bytes(cout, cin, k, code)
    if nbytes == 0:
        continue
    wp = torch.empty(nbytes, dtype=torch.uint8, device=dev)
    st = L.current_stream()
    L.check(lib.yolo_pack_weights(w.data_ptr(), wp.data_ptr(), cout, cin, k, code, st))
    y = torch.full((B, Ho, Ho, cout), float("nan"), device=dev, dtype=tdt)
    flag = torch.zeros(1, dtype=torch.int32, device=dev)
    d = L.ConvDesc(n=B, h=H, w=W, cin=cin, cout=cout, ksize=k, stride=s, x_ld=cin, x_off=0, y_ld=cout, y_off=0, r_ld=cout, r_off=0,
                   act=act, out_mode=L.OUT_NHWC, dtype=code, flags=(L.FLAG_RESIDUAL if residual else 0) | L.FLAG_NANCHECK, tile=0)
    rc = lib.yolo_conv_fwd(d, xh.data_ptr(), wp.data_ptr(), scale.data_ptr(), shift.data_ptr(), r.data_ptr() if residual else 0,
                           y.data_ptr(), flag.data_ptr(), st)
    if rc == -2:                                            # YOLO_ERR_UNSUPPORTED: a documented shape restriction, reported loudly
        continue
    L.check(rc, "yolo_conv_fwd")
    ref = F.conv2d(xh.double().permute(0, 3, 1, 2), w.to(tdt).double() if dtype != "fp32" else w.double(), stride=s, padding=k // 2)
    ref = ref * scale.double().view(1, -1, 1, 1) + shift.double().view(1, -1, 1, 1)
    if act == L.ACT_LEAKY:
        ref = F.leaky_relu(ref, 0.1)
    elif act == L.ACT_MISH:
        ref = F.mish(ref)
    ref = ref.permute(0, 2, 3, 1)
    if residual:
        ref = ref + r.double()
    err = float((y.double() - ref).abs().max() / (ref.abs().max() + 1e-9))
    bad = not (err <= tol) or int(flag.item()) != 0
    worst = max(worst, err if err == err else float("inf"))
    if bad:
        fails += 1
        print(f"FAIL B={B} {cin}->{cout} k{k} s{s} H={H} act={act} res={residual}: rel err {err:.3e} flag {int(flag.item())}")
print(f"{dtype}: {n_shapes} shapes, {fails} failures, worst relative error {worst:.3e} (tolerance {tol:g})")
sys.exit(1 if fails else 0)
