"""Drop-in mirror of the reference's per-scale loss (`/root/reference/code/loss.py:6-81`).

The loss stays ordinary PyTorch in the reference and it does here: it is
boolean-mask gathers plus four tiny reductions, and it defines the gradient that enters the HIP
backward kernels. Same constructor, same `forward(predictions, targets, anchors)` returning
``[5*box, 1*obj, 0.5*noobj, 1*class]``, same in-place side effects on its arguments.
"""
import torch
import torch.nn as nn

from .utils import calc_iou


class YOLOLoss(nn.Module):
    def __init__(self):
        super().__init__()
        self.mse = nn.MSELoss()
        self.bce_logits = nn.BCEWithLogitsLoss()
        self.cross_entropy = nn.CrossEntropyLoss()
        self.sigmoid = nn.Sigmoid()
        self.lambda_box, self.lambda_obj, self.lambda_noobj, self.lambda_class = 5, 1, 0.5, 1

    def forward(self, predictions, targets, anchors):
        has_obj = targets[..., 4] == 1
        no_obj = targets[..., 4] == 0                      # cells marked -1 are ignored by both masks
        anchors = anchors.reshape(1, 3, 1, 1, 2)
        zero = torch.tensor(0.0, device=predictions.device)
        box_loss, object_loss, class_loss = zero, zero, zero
        no_obj_loss = self.bce_logits(predictions[..., 4][no_obj], targets[..., 4][no_obj])
        if has_obj.any():
            decoded = torch.cat([self.sigmoid(predictions[..., :2]), torch.exp(predictions[..., 2:4]) * anchors], dim=-1)
            ious = calc_iou(decoded[has_obj], targets[..., :4][has_obj]).unsqueeze(1).detach()
            object_loss = self.mse(predictions[..., 4:5][has_obj], ious * targets[..., 4:5][has_obj])
            predictions[..., 1:3] = self.sigmoid(predictions[..., 1:3])          # indices 1:3 as in loss.py:71
            targets[..., 2:4] = torch.log(1e-16 + targets[..., 2:4] / anchors)
            box_loss = self.mse(predictions[..., :4][has_obj], targets[..., :4][has_obj])
            class_loss = self.cross_entropy(predictions[..., 5:][has_obj], targets[..., 5][has_obj].long())
        return [self.lambda_box * box_loss, self.lambda_obj * object_loss, self.lambda_noobj * no_obj_loss,
                self.lambda_class * class_loss]
