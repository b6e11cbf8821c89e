"""ORACLE — TEST INFRASTRUCTURE ONLY. Never imported by the product package.

Functional restatement of the reference's per-scale YOLO loss
(`/root/reference/code/loss.py:29-81`), including its in-place side effects:
``predictions[..., 1:3]`` is overwritten with its sigmoid (indices 1:3, not 0:2 — the
reference's behaviour, loss.py:71) and ``targets[..., 2:4]`` with log(1e-16 + wh/anchor).
Returns [5*box, 1*obj, 0.5*noobj, 1*class]; every term is a mean over the selected
elements of this call. Pinned by tests/golden/train_step.npz.
"""
import torch
import torch.nn.functional as F

from .postprocess import calc_iou

LAMBDA_BOX, LAMBDA_OBJ, LAMBDA_NOOBJ, LAMBDA_CLASS = 5, 1, 0.5, 1     # loss.py:24-27


def yolo_loss(predictions, targets, anchors):
    obj = targets[..., 4] == 1
    noobj = targets[..., 4] == 0                     # -1 cells are ignored by both masks
    anchors = anchors.reshape(1, 3, 1, 1, 2)
    zero = torch.tensor(0.0, device=predictions.device)
    box_loss = object_loss = class_loss = zero
    no_obj_loss = F.binary_cross_entropy_with_logits(predictions[..., 4][noobj], targets[..., 4][noobj])
    if obj.any():
        xy = torch.sigmoid(predictions[..., :2])
        wh = torch.exp(predictions[..., 2:4]) * anchors
        iou = calc_iou(torch.cat([xy, wh], -1)[obj], targets[..., :4][obj]).unsqueeze(1).detach()
        object_loss = F.mse_loss(predictions[..., 4:5][obj], iou * targets[..., 4:5][obj])
        predictions[..., 1:3] = torch.sigmoid(predictions[..., 1:3])
        targets[..., 2:4] = torch.log(1e-16 + targets[..., 2:4] / anchors)
        box_loss = F.mse_loss(predictions[..., :4][obj], targets[..., :4][obj])
        class_loss = F.cross_entropy(predictions[..., 5:][obj], targets[..., 5][obj].long())
    return [LAMBDA_BOX * box_loss, LAMBDA_OBJ * object_loss, LAMBDA_NOOBJ * no_obj_loss, LAMBDA_CLASS * class_loss]
