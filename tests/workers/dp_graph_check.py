#!/usr/bin/env python3
"""(Test infrastructure: lives under tests/ because it uses the oracle's seeded input generators.)
One-shot check of RCCL collectives inside a HIP-graph capture (run as a child process with YOLO_FORCE_DIST=1 and the torchrun
environment of ONE rank): a data-parallel GraphedTrainStep (bucketed all-reduce captured) must walk the same parameter
trajectory as eager data-parallel steps. Prints DP_GRAPH_OK on success."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import torch
import yolo_for_turbines_amd as yt
from yolo_for_turbines_amd import dist as ydist
from oracle import net as onet
from tests import golden_inputs as gi

dev = torch.device("cuda:0")
torch.cuda.set_device(dev)
dist = ydist.init("nccl", dev)
assert dist is not None, "needs YOLO_FORCE_DIST=1"
NC, S, B = 2, 96, 2
anchors = gi.TRAIN_CASE["anchors"]
sd = onet.synth_state_dict(311, 3, NC, gain=gi.NET_GAIN)
x = onet.synth_input(312, B, S).to(dev)
tg = [torch.from_numpy(t).to(dev) for t in gi.synth_targets(B, S, NC, anchors, 313)]
sa = (torch.tensor(anchors) * torch.tensor([S // 32, S // 16, S // 8]).view(3, 1, 1)).to(dev)
lf = yt.FusedYOLOLoss()


def make():
    m = yt.YOLOv3(num_classes=NC, activation="mish")
    m.load_state_dict({k: v.clone() for k, v in sd.items()})
    m = m.to(dev).train()
    ydist.data_parallel(m, dist, bucket_mb=8.0)
    return m, yt.SGD(m.parameters(), lr=1e-3, momentum=0.9, weight_decay=5e-4)


def eager(m, o):
    o.zero_grad(set_to_none=True)
    po = m(x)
    sum(sum(lf(po[i], tg[i], sa[i])) for i in range(3)).backward()
    o.step()


m1, o1 = make()
m2, o2 = make()
step = yt.GraphedTrainStep(m2, o2, sa, x, tg, allow_data_parallel=True)     # 3 eager warm-up steps, then the capture
for _ in range(3):
    eager(m1, o1)
for _ in range(4):
    eager(m1, o1)
    step(x, tg)
torch.cuda.synchronize()
for (k, a), (_, b) in zip(m1.state_dict().items(), m2.state_dict().items()):
    assert torch.equal(a, b), k
dist.barrier()
dist.destroy_process_group()
print("DP_GRAPH_OK")
