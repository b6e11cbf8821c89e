#!/usr/bin/env python3
"""(Test infrastructure: lives under tests/ because it uses the oracle's seeded input generators.)
One bf16 fine-tune step on seeded data; prints a JSON line with the loss, every per-scale loss part and the norms / a few
entries of selected gradients. Run it twice with different A/B environment switches (they are read once, at library load) and
compare (AB_FP32=1: the same steps without autocast, the yardstick): tests/test_gpu_parity.py::test_round3_kernels_agree_with_round2_paths."""
import json, os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import torch
import yolo_for_turbines_amd as yt
from tests import golden_inputs as gi
from oracle import net as onet

B, S, nc = int(os.environ.get("AB_B", 4)), int(os.environ.get("AB_S", 224)), 2
sd = onet.synth_state_dict(61, 3, nc, gain=gi.NET_GAIN)
m = yt.YOLOv3(num_classes=nc, activation=os.environ.get("AB_ACT", "mish"))
m.load_state_dict(sd)
m = m.cuda().train()
x = onet.synth_input(S, B, S).cuda()
anchors = gi.TRAIN_CASE["anchors"]
tg = [torch.from_numpy(t).cuda() for t in gi.synth_targets(B, S, nc, anchors, 5)]
sa = (torch.tensor(anchors) * torch.tensor([S // 32, S // 16, S // 8]).view(3, 1, 1)).cuda()
lf = yt.FusedYOLOLoss()
out = {}
for step in range(2):
    m.zero_grad(set_to_none=True)
    with torch.autocast("cuda", dtype=torch.bfloat16, enabled=not os.environ.get("AB_FP32")):
        preds = m(x)
        parts = [lf(preds[i], tg[i], sa[i]) for i in range(3)]
        loss = sum(sum(p) for p in parts)
    loss.backward()
    out[f"loss{step}"] = float(loss)
    out[f"parts{step}"] = [float(v) for p in parts for v in p]
names = ["layers.0.conv.weight", "layers.1.conv.weight", "layers.2.layers.0.0.conv.weight", "layers.2.layers.0.1.conv.weight",
         "layers.3.conv.weight", "layers.0.batch_norm.weight", "layers.2.layers.0.1.batch_norm.bias", "layers.10.layers.3.1.conv.weight",
         "layers.29.pred_block.1.conv.bias"]
prm = dict(m.named_parameters())
out["grads"] = {n: {"norm": float(prm[n].grad.double().norm()), "head": [float(v) for v in prm[n].grad.flatten()[:6]]} for n in names}
out["rm0"] = [float(v) for v in m.layers[0].batch_norm.running_mean[:8]]
print(json.dumps(out))
