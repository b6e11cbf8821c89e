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


def run_shard(model, shard, nc, S, autocast):
    anchors = gi.TRAIN_CASE["anchors"]
    x = onet.synth_input(500 + shard, 2, S).cuda()
    tg = [torch.from_numpy(t).cuda() for t in gi.synth_targets(2, S, nc, anchors, 600 + shard)]
    sa = (torch.tensor(anchors) * torch.tensor([S // 32, S // 16, S // 8]).view(3, 1, 1)).cuda()
    lf = yt.FusedYOLOLoss()
    model.zero_grad(set_to_none=True)
    with torch.autocast("cuda", dtype=torch.bfloat16, enabled=autocast):
        po = model(x)
    sum(sum(lf(po[i], tg[i], sa[i])) for i in range(3)).backward()
    return {k: p.grad.detach().double().clone() for k, p in model.named_parameters()}


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
    if rank == 0:                                       # single-process reference: mean of the two shards' gradients
        m2 = yt.YOLOv3(num_classes=nc, activation="mish")
        m2.load_state_dict(sd)
        m2 = m2.cuda().train()
        g0, g1 = run_shard(m2, 0, nc, S, autocast), run_shard(m2, 1, nc, S, autocast)
        worst = 0.0
        for k in g:
            mean = (g0[k] + g1[k]) / 2
            worst = max(worst, float((g[k] - mean).norm() / (mean.norm() + 1e-30)))
        res["worst_rel_err_vs_mean_of_shards"] = worst
    with open(os.path.join(out_dir, f"rank{rank}.json"), "w") as f:
        json.dump(res, f)
    dist.barrier()
    dist.destroy_process_group()


if __name__ == "__main__":
    main()
