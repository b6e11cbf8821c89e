#!/usr/bin/env python3
"""How much of the eager fine-tune step is host time? Enqueue N steps without synchronising and compare the time the host
needed to ENQUEUE them with the time until the GPU has FINISHED them (python tools/host_overhead.py [bf16|fp32] [steps])."""
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
opt = (torch.optim.SGD if os.environ.get("TORCH_SGD") else yt.SGD)(m.parameters(), lr=1e-4, momentum=0.9, weight_decay=5e-4)
anchors = gi.TRAIN_CASE["anchors"]
sa = (torch.tensor(anchors) * torch.tensor([S // 32, S // 16, S // 8]).view(3, 1, 1)).to(dev)
tg = [torch.from_numpy(t).to(dev) for t in gi.synth_targets(B, S, nc, anchors, 3)]
x = torch.rand(B, 3, S, S, device=dev)
lf = yt.FusedYOLOLoss()
ac = None if dtype == "fp32" else torch.bfloat16


def step():
    opt.zero_grad(set_to_none=True)
    with torch.autocast("cuda", dtype=ac or torch.bfloat16, enabled=ac is not None):
        preds = m(x)
        loss = sum(sum(lf(preds[i], tg[i], sa[i])) for i in range(3))
    loss.backward()
    opt.step()


for _ in range(3):
    step()
torch.cuda.synchronize()
t0 = time.perf_counter()
for _ in range(steps):
    step()
t1 = time.perf_counter()
torch.cuda.synchronize()
t2 = time.perf_counter()
print(f"{dtype}: host enqueue {1e3 * (t1 - t0) / steps:.2f} ms/step, until finished {1e3 * (t2 - t0) / steps:.2f} ms/step")
if os.environ.get("HOST_PROFILE"):
    import cProfile, pstats
    pr = cProfile.Profile()
    pr.enable()
    for _ in range(3):
        step()
    pr.disable()
    torch.cuda.synchronize()
    pstats.Stats(pr).sort_stats("cumulative").print_stats(28)
