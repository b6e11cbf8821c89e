#!/usr/bin/env python3
"""(Test infrastructure: lives under tests/ because it uses the oracle's seeded input generators.)
Diagnostic: walk the 3-step reference trajectory (tests/golden/train_traj.npz) and print per-step deviations."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import numpy as np, torch
import yolo_for_turbines_amd as yt
from oracle import net as onet
from tests import golden_inputs as gi

g = np.load(os.path.join(os.path.dirname(__file__), "..", "tests", "golden", "train_traj.npz"))
c = gi.TRAIN_CASE
for tag, act in (("mish", "mish"), ("leaky", "leaky_relu")):
    for kind in ("yt", "torch"):
        for lossk in ("mirror", "fused"):
            sd = onet.synth_state_dict(c["wseed"], 3, c["nc"], gain=gi.NET_GAIN)
            m = yt.YOLOv3(num_classes=c["nc"], activation=act); m.load_state_dict(sd); m = m.cuda().train()
            x = onet.synth_input(c["xseed"], c["batch"], c["size"]).cuda()
            tg = [torch.from_numpy(t).cuda() for t in gi.synth_targets(c["batch"], c["size"], c["nc"], c["anchors"], c["tseed"])]
            grids = [c["size"] // 32, c["size"] // 16, c["size"] // 8]
            sa = (torch.tensor(c["anchors"]) * torch.tensor(grids).view(3, 1, 1)).cuda()
            lf = yt.YOLOLoss() if lossk == "mirror" else yt.FusedYOLOLoss()
            opt = (yt.SGD if kind == "yt" else torch.optim.SGD)(m.parameters(), **gi.TRAJ_OPT)
            sched = torch.optim.lr_scheduler.LinearLR(opt, **gi.TRAJ_SCHED)
            for step in range(gi.TRAJ_STEPS):
                opt.zero_grad()
                preds = m(x)
                parts = torch.stack([torch.stack(lf(preds[i], tg[i].clone(), sa[i])) for i in range(3)])
                ref = g[f"{tag}/loss_parts"][step]
                rel = np.abs(parts.detach().cpu().numpy() - ref) / (np.abs(ref) + 1e-6)
                parts.sum().backward()
                print(tag, kind, lossk, "step", step, "lr", opt.param_groups[0]["lr"], "loss", float(parts.sum()), "ref", float(ref.sum()),
                      "max rel", float(rel.max()))
                opt.step(); sched.step()
            norms = np.array([float(p.grad.double().norm()) for p in m.parameters()])
            r = g[f"{tag}/gradnorm_step3"]
            print("   gradnorm step3 max rel", float((np.abs(norms - r) / (r + 1e-6 * r.max())).max()))
