"""Worker of test_two_rank_data_parallel_on_gpu: one fine-tune step of a 2-rank data-parallel group whose ranks share
cuda:0 (RCCL refuses two ranks on one device, so the exchange runs over gloo — the bucketing / ordering / averaging code
is the same). Writes the averaged gradients' checksums; rank 0 also writes what a single process gets for the mean of
the two shards."""
import json
import os
import sys

import numpy as np
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import yolo_for_turbines_amd as yt                     # noqa: E402
from yolo_for_turbines_amd import dist as ydist        # noqa: E402
from oracle import net as onet                         # noqa: E402  (inputs / weights only)
from tests import golden_inputs as gi                  # noqa: E402


def run_shard(model, shard, nc, S, autocast, zero="none"):
    """zero: "none" = zero_grad(set_to_none=True) first, "zero" = zero_grad(set_to_none=False) (p.grad tensors survive),
    "keep" = no zero_grad at all (micro-batch accumulation)."""
    anchors = gi.TRAIN_CASE["anchors"]
    x = onet.synth_input(500 + shard, 2, S).cuda()
    tg = [torch.from_numpy(t).cuda() for t in gi.synth_targets(2, S, nc, anchors, 600 + shard)]
    sa = (torch.tensor(anchors) * torch.tensor([S // 32, S // 16, S // 8]).view(3, 1, 1)).cuda()
    lf = yt.FusedYOLOLoss()
    if zero == "none":
        model.zero_grad(set_to_none=True)
    elif zero == "zero":
        model.zero_grad(set_to_none=False)
    with torch.autocast("cuda", dtype=torch.bfloat16, enabled=autocast):
        po = model(x)
    sum(sum(lf(po[i], tg[i], sa[i])) for i in range(3)).backward()
    return {k: (None if p.grad is None else p.grad.detach().double().clone()) for k, p in model.named_parameters()}


def rel(a, b):
    return float((a - b).norm() / (b.norm() + 1e-30))


def main():
    out_dir, autocast = sys.argv[1], sys.argv[2] == "bf16"
    nc, S = 2, 96
    rank = int(os.environ["RANK"])
    dist = ydist.init("gloo", torch.device("cuda:0"))
    sd = onet.synth_state_dict(7, 3, nc, gain=gi.NET_GAIN)
    m = yt.YOLOv3(num_classes=nc, activation="mish")
    m.load_state_dict(sd)
    m = m.cuda().train()
    ydist.data_parallel(m, dist, bucket_mb=8.0)         # several buckets
    g = run_shard(m, rank, nc, S, autocast)
    res = {"rank": rank, "norms": {k: float(v.norm()) for k, v in g.items()}}
    # second step with the old p.grad tensors alive (they alias the buckets), third step accumulating on top of them
    g_zero = run_shard(m, rank + 2, nc, S, autocast, zero="zero")
    g_acc = run_shard(m, rank + 4, nc, S, autocast, zero="keep")
    # the trainable set changes: freeze the first 9 top-level modules, then unfreeze them again (buckets are rebuilt)
    frozen = [p for layer in list(m.layers)[:9] for p in layer.parameters()]
    for p in frozen:
        p.requires_grad_(False)
    g_frozen = run_shard(m, rank, nc, S, autocast)
    for p in frozen:
        p.requires_grad_(True)
    g_again = run_shard(m, rank, nc, S, autocast)
    res["norms_again"] = {k: float(v.norm()) for k, v in g_again.items()}
    if rank == 0:                                       # single-process reference: mean of the two shards' gradients
        m2 = yt.YOLOv3(num_classes=nc, activation="mish")
        m2.load_state_dict(sd)
        m2 = m2.cuda().train()
        gs = [run_shard(m2, k, nc, S, autocast) for k in range(6)]
        worst = {"first": 0.0, "zero_keep": 0.0, "accumulate": 0.0, "frozen": 0.0, "unfrozen": 0.0}
        n_frozen = 0
        for k in g:
            m01, m23, m45 = (gs[0][k] + gs[1][k]) / 2, (gs[2][k] + gs[3][k]) / 2, (gs[4][k] + gs[5][k]) / 2
            worst["first"] = max(worst["first"], rel(g[k], m01))
            worst["zero_keep"] = max(worst["zero_keep"], rel(g_zero[k], m23))
            worst["accumulate"] = max(worst["accumulate"], rel(g_acc[k], m23 + m45))
            worst["unfrozen"] = max(worst["unfrozen"], rel(g_again[k], m01))
            if int(k.split(".")[1]) < 9:
                n_frozen += g_frozen[k] is None
            else:
                worst["frozen"] = max(worst["frozen"], rel(g_frozen[k], m01))
        res["worst_rel_err_vs_mean_of_shards"] = worst["first"]
        res["worst"] = worst
        res["n_frozen_without_grad"] = n_frozen
    with open(os.path.join(out_dir, f"rank{rank}.json"), "w") as f:
        json.dump(res, f)
    dist.barrier()
    dist.destroy_process_group()


if __name__ == "__main__":
    main()
