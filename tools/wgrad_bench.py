#!/usr/bin/env python3
"""Time yolo_conv_wgrad (16-bit) on the 3x3 / 1x1 layer shapes of the network at batch 32, 416x416.
YOLO_NO_WGRAD_DMA=1 selects round 2's kernel for the 3x3 stride-1 layers (A/B in two processes on one box)."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from yolo_for_turbines_amd import _lib as L

lib = L.lib()
B = int(os.environ.get("B", 32))
shapes = [(32, 64, 3, 1, 208), (64, 128, 3, 1, 104), (128, 256, 3, 1, 52), (256, 512, 3, 1, 26), (512, 1024, 3, 1, 13),
          (64, 32, 1, 1, 208), (128, 64, 1, 1, 104), (256, 128, 1, 1, 52), (512, 256, 1, 1, 26), (1024, 512, 1, 1, 13),
          (32, 64, 3, 2, 416), (128, 256, 3, 2, 104), (512, 1024, 3, 2, 26), (3, 32, 3, 1, 416)]
if len(sys.argv) > 1:
    shapes = [s for s in shapes if (str(s[2]) in sys.argv[1].split(",")) or ("stem" in sys.argv[1] and s[0] <= 3)]
code = L.BF16
for cin, cout, k, s, H in shapes:
    Ho = H // s
    x_ld = 8 if cin <= 3 else cin                   # the first block reads the 8-channel 16-bit input buffer
    x = torch.zeros(B, H, H, x_ld, device="cuda").bfloat16()
    x[..., :cin] = torch.randn(B, H, H, cin, device="cuda").bfloat16()
    dz = torch.randn(B, Ho, Ho, cout, device="cuda").bfloat16()
    dw = torch.empty(cout, cin, k, k, device="cuda")
    ws = torch.empty(lib.yolo_wgrad_workspace_bytes(B, H, H, cin, cout, k, s, code), dtype=torch.uint8, device="cuda")
    st = L.current_stream()
    def run():
        L.check(lib.yolo_conv_wgrad(dz.data_ptr(), cout, 0, x.data_ptr(), x_ld, 0, dw.data_ptr(), B, H, H, cin, cout, k, s, code,
                                    ws.data_ptr(), ws.numel(), st), "wgrad")
    for _ in range(3):
        run()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    reps = 20
    e0.record()
    for _ in range(reps):
        run()
    e1.record()
    torch.cuda.synchronize()
    us = e0.elapsed_time(e1) * 1e3 / reps
    fl = 2.0 * B * Ho * Ho * cout * cin * k * k
    print(f"wgrad {cin:4d}->{cout:4d} k{k} s{s} {H:3d}: {us:7.1f} us  {fl / us / 1e6:7.1f} TF  (kernel + reduce)", flush=True)
