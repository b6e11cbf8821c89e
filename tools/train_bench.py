#!/usr/bin/env python3
"""Fine-tune step timing (tuning aid): forward(train) + 3 x YOLOLoss + backward + SGD."""
import argparse, os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import yolo_for_turbines_amd as yt
from tests import golden_inputs as gi

ap = argparse.ArgumentParser()
ap.add_argument("--batch", type=int, default=32)
ap.add_argument("--size", type=int, default=416)
ap.add_argument("--classes", type=int, default=2)
ap.add_argument("--steps", type=int, default=5)
ap.add_argument("--dtype", default="fp32", choices=["fp32", "bf16", "fp16"])
ap.add_argument("--fused-loss", action="store_true", help="FusedYOLOLoss (3 HIP kernels per scale) instead of the PyTorch loss")
ap.add_argument("--freeze", type=int, default=0, help="freeze the parameters of the first N top-level modules (freeze=True backbone)")
ap.add_argument("--graph", action="store_true", help="capture the whole step in a HIP graph and replay it")
a = ap.parse_args()
dev = torch.device("cuda:0")
from bench import seeded_model
m = seeded_model(yt, a.classes, dev).train()
m._engine.compute_dtype = a.dtype
for layer in list(m.layers)[:a.freeze]:
    for p in layer.parameters():
        p.requires_grad_(False)
anchors = gi.TRAIN_CASE["anchors"]
grids = [a.size // 32, a.size // 16, a.size // 8]
sa = (torch.tensor(anchors) * torch.tensor(grids).view(3, 1, 1)).to(dev)
x = torch.rand(a.batch, 3, a.size, a.size, device=dev)
tg = [torch.from_numpy(t).to(dev) for t in gi.synth_targets(a.batch, a.size, a.classes, anchors, 3)]
lf = yt.FusedYOLOLoss() if (a.fused_loss or a.graph) else yt.YOLOLoss()
fresh = (lambda t: t.clone()) if isinstance(lf, yt.YOLOLoss) else (lambda t: t)      # the reference's loss overwrites its targets (loss.py:70)
opt = (torch.optim.SGD if os.environ.get('TORCH_SGD') else yt.SGD)([p for p in m.parameters() if p.requires_grad], lr=1e-4, momentum=0.9, weight_decay=5e-4)

def step(timing=None):
    ev = [torch.cuda.Event(enable_timing=True) for _ in range(5)]
    opt.zero_grad(set_to_none=True)
    ev[0].record()
    preds = m(x)
    ev[1].record()
    loss = sum(sum(lf(preds[i], fresh(tg[i]), sa[i])) for i in range(3))
    ev[2].record()
    loss.backward()
    ev[3].record()
    opt.step()
    ev[4].record()
    torch.cuda.synchronize()
    if timing is not None:
        timing.append([ev[i].elapsed_time(ev[i + 1]) for i in range(4)])
    return float(loss)

for _ in range(2):
    step()
if a.graph:
    # whole-step capture: forward(train) + loss + backward + SGD as ONE graph launch (no per-kernel launch cost).
    # Needs the per-forward NaN guard's host sync off and capturable optimizer state.
    m._engine.nan_check = False
    opt = (torch.optim.SGD if os.environ.get('TORCH_SGD') else yt.SGD)([p for p in m.parameters() if p.requires_grad], lr=1e-4, momentum=0.9, weight_decay=5e-4)
    side = torch.cuda.Stream()
    side.wait_stream(torch.cuda.current_stream())
    with torch.cuda.stream(side):
        for _ in range(3):
            opt.zero_grad(set_to_none=True)
            preds = m(x)
            loss = sum(sum(lf(preds[i], fresh(tg[i]), sa[i])) for i in range(3))
            loss.backward()
            opt.step()
    torch.cuda.current_stream().wait_stream(side)
    g = torch.cuda.CUDAGraph()
    opt.zero_grad(set_to_none=True)
    with torch.cuda.graph(g):
        preds = m(x)
        gloss = sum(sum(lf(preds[i], fresh(tg[i]), sa[i])) for i in range(3))
        gloss.backward()
        opt.step()
    for _ in range(2):
        g.replay()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(a.steps):
        g.replay()
    torch.cuda.synchronize()
    dt = (time.perf_counter() - t0) / a.steps
    print(f"{a.dtype} B={a.batch} S={a.size} nc={a.classes} [graph]: {dt*1e3:.1f} ms/step = {a.batch/dt:.1f} img/s | loss {float(gloss):.3f}")
    sys.exit(0)
t = []
t0 = time.perf_counter()
for _ in range(a.steps):
    l = step(t)
dt = (time.perf_counter() - t0) / a.steps
import numpy as np
t = np.mean(np.array(t), 0)
print(f"{a.dtype} B={a.batch} S={a.size} nc={a.classes}: {dt*1e3:.1f} ms/step = {a.batch/dt:.1f} img/s | fwd {t[0]:.1f} loss {t[1]:.1f} bwd {t[2]:.1f} sgd {t[3]:.1f} ms | loss {l:.3f}")
