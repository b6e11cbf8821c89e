"""ORACLE — TEST INFRASTRUCTURE ONLY. Never imported by the product package.

CPU restatement of the reference's ``calc_mAP`` (`/root/reference/code/utils.py:193-274`): per class with ground
truth, detections in descending-objectness order (stable list sort), greedy matching to the ground truths of the
same image in list order (first maximum IoU, strict ``>`` against a running best starting at 0; true positive iff
that IoU > threshold and the ground truth is unassigned), fp32 cumulative precision / recall with (0, 1) prepended,
``torch.trapz``; mean over those classes. Pinned by tests/golden/kat.npz (``map_*`` entries, produced by the
reference function itself).
"""
import torch

from .postprocess import calc_iou


def calc_map(pred_boxes, true_boxes, iou_threshold=0.5, box_format="center", num_classes=20):
    aps = []
    for c in range(num_classes):
        dets = [d for d in pred_boxes if d[-1] == c]
        gts = [g for g in true_boxes if g[-1] == c]
        if not gts:
            continue
        per_img = {}
        for g in gts:
            per_img.setdefault(g[0], []).append(g)
        taken = {k: [False] * len(v) for k, v in per_img.items()}
        dets = sorted(dets, key=lambda d: d[5], reverse=True)
        tp = torch.zeros(len(dets))
        fp = torch.zeros(len(dets))
        for i, d in enumerate(dets):
            cand = per_img.get(d[0], [])
            best, best_j = 0, 0
            for j, g in enumerate(cand):
                iou = calc_iou(torch.tensor(d[1:5]), torch.tensor(g[1:5]), box_format)
                if iou > best:
                    best, best_j = iou, j
            if best > iou_threshold and not taken[d[0]][best_j]:
                tp[i] = 1
                taken[d[0]][best_j] = True
            else:
                fp[i] = 1
        ctp, cfp = torch.cumsum(tp, 0), torch.cumsum(fp, 0)
        prec = torch.cat((torch.tensor([1]), ctp / (ctp + cfp)))
        rec = torch.cat((torch.tensor([0]), ctp / len(gts)))
        aps.append(torch.trapz(prec, rec))
    return sum(aps) / len(aps)
