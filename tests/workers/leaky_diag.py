#!/usr/bin/env python3
"""(Test infrastructure: lives under tests/ because it uses the oracle's seeded input generators.)
Diagnostic: the golden fine-tune step on the GPU; per sampled parameter |ours - fp64| against |reference fp32 - fp64|
(tests/golden/train_step.npz = imported reference in fp32, train_step_fp64.npz = the same step in float64)."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import numpy as np
import torch
import yolo_for_turbines_amd as yt
from oracle import net as onet
from tests import golden_inputs as gi

g32 = np.load("tests/golden/train_step.npz")
g64 = np.load("tests/golden/train_step_fp64.npz")
c = gi.TRAIN_CASE
for tag, act in (("leaky", "leaky_relu"), ("mish", "mish")):
    sd = onet.synth_state_dict(c["wseed"], 3, c["nc"], gain=gi.NET_GAIN)
    m = yt.YOLOv3(num_classes=c["nc"], activation=act)
    m.load_state_dict(sd)
    m = m.cuda().train()
    x = onet.synth_input(c["xseed"], c["batch"], c["size"]).cuda()
    tg = [torch.from_numpy(t).cuda() for t in gi.synth_targets(c["batch"], c["size"], c["nc"], c["anchors"], c["tseed"])]
    grids = [c["size"] // 32, c["size"] // 16, c["size"] // 8]
    sa = (torch.tensor(c["anchors"]) * torch.tensor(grids).view(3, 1, 1)).cuda()
    lf = yt.YOLOLoss()
    preds = m(x)
    parts = torch.stack([torch.stack(lf(preds[i], tg[i].clone(), sa[i])) for i in range(3)])
    parts.sum().backward()
    named = dict(m.named_parameters())
    for key in [k[len(tag) + 6:] for k in g64.files if k.startswith(f"{tag}/grad/")]:
        got = named[key].grad.cpu().double()
        got = got.reshape(-1)[::gi.TRAIN_GRAD_STRIDE].numpy() if got.numel() > 4096 else got.numpy()
        w64, w32 = g64[f"{tag}/grad/{key}"], g32[f"{tag}/grad/{key}"]
        sc = np.abs(w64).max()
        e_ours, e_ref = np.abs(got - w64).max(), np.abs(w32 - w64).max()
        print(f"{tag:5s} {key:42s} max|g| {sc:9.3e}  ours-fp64 {e_ours:9.3e} ({e_ours / sc:8.2e})  ref32-fp64 {e_ref:9.3e} ({e_ref / sc:8.2e})  ratio {e_ours / max(e_ref, 1e-30):6.2f}"
              f"  rms ours {np.sqrt(np.mean((got - w64) ** 2)):9.3e} ref {np.sqrt(np.mean((w32 - w64) ** 2)):9.3e}")
    norms = np.array([float(p.grad.double().norm()) for p in m.parameters()])
    print(tag, "gradnorm rel max: ours", np.abs(norms / g64[f"{tag}/gradnorm_all"] - 1).max(), "ref32", np.abs(g32[f"{tag}/gradnorm_all"] / g64[f"{tag}/gradnorm_all"] - 1).max())
