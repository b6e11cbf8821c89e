import sys, os
sys.path.insert(0, os.getcwd())
import numpy as np, torch
from oracle import net as onet
from tests import golden_inputs as gi
import yolo_for_turbines_amd as yt
g = np.load("tests/golden/train_step.npz")
tag, act = "leaky", "leaky_relu"
c = gi.TRAIN_CASE
sd = onet.synth_state_dict(c["wseed"], 3, c["nc"], gain=gi.NET_GAIN)
m = yt.YOLOv3(num_classes=c["nc"], activation=act); m.load_state_dict(sd); m = m.cuda().train()
x = onet.synth_input(c["xseed"], c["batch"], c["size"]).cuda()
tg = [torch.from_numpy(t).cuda() for t in gi.synth_targets(c["batch"], c["size"], c["nc"], c["anchors"], c["tseed"])]
grids = [c["size"] // 32, c["size"] // 16, c["size"] // 8]
sa = (torch.tensor(c["anchors"]) * torch.tensor(grids).view(3, 1, 1)).cuda()
lf = yt.YOLOLoss()
preds = m(x)
parts = torch.stack([torch.stack(lf(preds[i], tg[i].clone(), sa[i])) for i in range(3)])
parts.sum().backward()
norms = np.array([float(p.grad.double().norm()) for p in m.parameters()])
ref = g[f"{tag}/gradnorm_all"]
names = [n for n, _ in m.named_parameters()]
for n, a, b in zip(names, norms, ref):
    print(f"{n:48s} ours {a:12.5g} ref {b:12.5g} rel {abs(a-b)/max(b,1e-12):.2e}")
g64 = np.load("tests/golden/train_step_fp64.npz")
named = dict(m.named_parameters())
for key in [k[len("leaky/grad/"):] for k in g64.files]:
    t = g64["leaky/grad/" + key]; r = g["leaky/grad/" + key]
    got = named[key].grad.cpu()
    got = got.reshape(-1)[::gi.TRAIN_GRAD_STRIDE].numpy() if got.numel() > 4096 else got.numpy()
    sc = np.abs(t).max()
    print(f"{key:44s} ours-vs-fp64 {np.abs(got-t).max()/sc:.2e}  ref32-vs-fp64 {np.abs(r-t).max()/sc:.2e}  ours-vs-ref32 {np.abs(got-r).max()/sc:.2e}")
