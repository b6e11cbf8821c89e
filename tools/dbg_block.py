import sys, os
sys.path.insert(0, os.getcwd())
import numpy as np, torch, torch.nn as nn, torch.nn.functional as F
import yolo_for_turbines_amd as yt
torch.manual_seed(0)
def check(cin, cout, k, s, H, B, act="leaky_relu"):
    blk = yt.CNNBlock(cin, cout, activation=act, kernel_size=k, stride=s, padding=1 if k == 3 else 0)
    with torch.no_grad():
        blk.conv.weight.copy_(torch.randn_like(blk.conv.weight) * (1.0 / (cin * k * k)) ** 0.5)
        blk.batch_norm.weight.copy_(torch.rand(cout) + 0.5); blk.batch_norm.bias.copy_(torch.randn(cout) * 0.1)
    x = torch.randn(B, cin, H, H)
    ref = yt.CNNBlock.__new__(yt.CNNBlock)  # placeholder
    # fp64 truth with plain torch
    w = blk.conv.weight.detach().double().requires_grad_(True)
    g = blk.batch_norm.weight.detach().double().requires_grad_(True)
    b = blk.batch_norm.bias.detach().double().requires_grad_(True)
    xd = x.double().requires_grad_(True)
    z = F.conv2d(xd, w, None, stride=s, padding=1 if k == 3 else 0)
    u = F.batch_norm(z, None, None, g, b, True, 0.1, 1e-5)
    y = F.leaky_relu(u, 0.1) if act == "leaky_relu" else F.mish(u)
    gy = torch.randn_like(y)
    y.backward(gy)
    blk = blk.cuda().train()
    xg = x.cuda().requires_grad_(True)
    yo = blk(xg)
    yo.backward(gy.float().cuda())
    def rel(a, t): return float((a.cpu().double() - t).abs().max() / t.abs().max())
    print(f"{cin}->{cout} k{k} s{s} H{H} B{B}: y {rel(yo.detach(), y.detach()):.1e} dx {rel(xg.grad, xd.grad):.1e} dw {rel(blk.conv.weight.grad, w.grad):.1e} "
          f"dgamma {rel(blk.batch_norm.weight.grad, g.grad):.1e} dbeta {rel(blk.batch_norm.bias.grad, b.grad):.1e}", flush=True)
for args in [(512, 1024, 3, 1, 3, 4), (512, 1024, 3, 1, 7, 2), (256, 512, 3, 1, 6, 4), (768, 256, 1, 1, 6, 4), (1024, 512, 1, 1, 3, 4),
             (512, 1024, 3, 2, 6, 4), (128, 256, 3, 1, 12, 4), (32, 1024, 1, 1, 3, 4), (64, 128, 3, 1, 24, 4)]:
    check(*args)
