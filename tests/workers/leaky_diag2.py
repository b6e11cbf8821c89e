#!/usr/bin/env python3
"""(Test infrastructure: lives under tests/ because it uses the oracle's seeded input generators.)
Diagnostic: per block of the golden fine-tune case, how far the train-mode forward activations are from a float64
run of the CPU oracle — ours (GPU fp32) next to the oracle's own fp32 run — and how many LeakyReLU branches differ."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import numpy as np
import torch
import yolo_for_turbines_amd as yt
from oracle import net as onet
from tests import golden_inputs as gi

c = gi.TRAIN_CASE
act = sys.argv[1] if len(sys.argv) > 1 else "leaky_relu"
sd = onet.synth_state_dict(c["wseed"], 3, c["nc"], gain=gi.NET_GAIN)
x = onet.synth_input(c["xseed"], c["batch"], c["size"])
t64, t32 = {}, {}
onet.forward({k: (v.double() if v.dtype.is_floating_point else v) for k, v in sd.items()}, x.double(), c["nc"], act, training=True, new_stats={}, taps=t64)
onet.forward(sd, x, c["nc"], act, training=True, new_stats={}, taps=t32)
m = yt.YOLOv3(num_classes=c["nc"], activation=act)
m.load_state_dict(sd)
m = m.cuda().train()
preds = m(x.cuda())
plan = [p for k, p in m._engine._plans.items() if k[0] == "train"][-1]
names = {id(mod): name for name, mod in m.named_modules()}
B = c["batch"]
print(f"{'block':34s} {'rms|y|':>9s} {'ours-64 rms':>11s} {'o32-64 rms':>11s} {'ratio':>6s} {'flips ours':>10s} {'flips o32':>9s} {'elements':>9s}")
for op in plan.prog.ops:
    blk, yv = op["block"], op["y"]
    name = names[id(blk)]
    if yv is None or op["out_mode"] != 0 or name not in t64:
        continue
    buf = plan.ybuf[yv.buf].view(B, yv.H, yv.W, yv.ld)[..., yv.off:yv.off + yv.C].permute(0, 3, 1, 2).double().cpu()
    r64, r32 = t64[name], t32[name].double()
    if op["res"] is not None:      # the engine's buffer holds x + y for residual units; the tap is the block output y
        continue
    sc = float(r64.pow(2).mean().sqrt())
    e1, e2 = float((buf - r64).pow(2).mean().sqrt()), float((r32 - r64).pow(2).mean().sqrt())
    f1 = int(((buf > 0) != (r64 > 0)).sum())
    f2 = int(((r32 > 0) != (r64 > 0)).sum())
    print(f"{name:34s} {sc:9.3e} {e1:11.3e} {e2:11.3e} {e1 / max(e2, 1e-30):6.2f} {f1:10d} {f2:9d} {r64.numel():9d}")
