#!/usr/bin/env python3
"""How much host time does one eager fine-tune step cost (python tools/host_overhead.py [bf16|fp32] [steps])?

Three numbers per configuration:
  * idle-GPU enqueue: synchronise, then time ONE step's Python (zero_grad, forward, loss, backward, optimizer step) until the
    last launch has been handed to the runtime - the host cost proper, with an empty queue in front of it;
  * steady state: N steps without synchronising - host time to enqueue them (once the host is faster than the GPU this reads
    the GPU's time: the runtime's queue pushes back) and time until the GPU has finished them;
YOLO_TRAIN_TAPE=0 selects the per-launch path (no launch tables) for A/B; NAN_CHECK=deferred|off sets the model's NaN guard mode
(default: the reference's immediate guard, one host sync inside every forward); GRAPH=1 adds the whole step as one HIP graph."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import yolo_for_turbines_amd as yt
from tests import golden_inputs as gi

dtype = sys.argv[1] if len(sys.argv) > 1 else "bf16"
steps = int(sys.argv[2]) if len(sys.argv) > 2 else 6
B, S, nc = 32, 416, 2
dev = torch.device("cuda:0")
m = yt.YOLOv3(num_classes=nc).to(dev).train()
_mode = os.environ.get("NAN_CHECK", "immediate")
m._engine.nan_check = {"immediate": True, "deferred": "deferred", "off": False}[_mode]
opt = (torch.optim.SGD if os.environ.get("TORCH_SGD") else yt.SGD)(m.parameters(), lr=1e-4, momentum=0.9, weight_decay=5e-4)
anchors = gi.TRAIN_CASE["anchors"]
sa = (torch.tensor(anchors) * torch.tensor([S // 32, S // 16, S // 8]).view(3, 1, 1)).to(dev)
tg = [torch.from_numpy(t).to(dev) for t in gi.synth_targets(B, S, nc, anchors, 3)]
x = torch.rand(B, 3, S, S, device=dev)
lf = yt.FusedYOLOLoss()
ac = None if dtype == "fp32" else torch.bfloat16
parts = {}


def step(timed=False):
    t = [time.perf_counter()]
    opt.zero_grad(set_to_none=True)
    t.append(time.perf_counter())
    with torch.autocast("cuda", dtype=ac or torch.bfloat16, enabled=ac is not None):
        preds = m(x)
        t.append(time.perf_counter())
        loss = sum(sum(lf(preds[i], tg[i], sa[i])) for i in range(3))
    t.append(time.perf_counter())
    loss.backward()
    t.append(time.perf_counter())
    opt.step()
    t.append(time.perf_counter())
    if timed:
        for k, name in enumerate(("zero_grad", "forward", "loss", "backward", "optimizer")):
            parts[name] = parts.get(name, 0.0) + (t[k + 1] - t[k])


for _ in range(3):
    step()
torch.cuda.synchronize()
idle = []
for _ in range(steps):
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    step(timed=True)
    idle.append(time.perf_counter() - t0)
torch.cuda.synchronize()
t0 = time.perf_counter()
for _ in range(steps):
    step()
t1 = time.perf_counter()
torch.cuda.synchronize()
t2 = time.perf_counter()
idle.sort()
print(f"{dtype} nan_check={_mode}: idle-GPU enqueue {1e3 * idle[len(idle) // 2]:.2f} ms/step (median of {steps}; "
      + ", ".join(f"{k} {1e3 * v / steps:.2f}" for k, v in parts.items()) + f") | steady state: host enqueue "
      f"{1e3 * (t1 - t0) / steps:.2f} ms/step, until finished {1e3 * (t2 - t0) / steps:.2f} ms/step")
m._engine.flush_nan()
if os.environ.get("GRAPH"):
    gs = yt.GraphedTrainStep(m, opt, list(sa), x, tg, autocast_dtype=ac)
    for _ in range(3):
        gs(x, tg)
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(steps):
        gs(x, tg)
    torch.cuda.synchronize()
    print(f"{dtype}: whole step as one HIP graph {1e3 * (time.perf_counter() - t0) / steps:.2f} ms/step")
if os.environ.get("HOST_PROFILE"):
    import cProfile, pstats
    pr = cProfile.Profile()
    torch.cuda.synchronize()
    pr.enable()
    for _ in range(3):
        step()
        torch.cuda.synchronize()
    pr.disable()
    pstats.Stats(pr).sort_stats("cumulative").print_stats(32)
