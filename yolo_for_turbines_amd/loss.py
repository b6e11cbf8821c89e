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


class _FusedLossFn(torch.autograd.Function):
    @staticmethod
    def forward(ctx, pred, target, anchors):
        import ctypes as C
        from . import _lib as L
        lib = L.lib()
        B, A, g, _, D = pred.shape
        dev = pred.device
        with torch.cuda.device(dev):
            out = torch.empty(4, dtype=torch.float32, device=dev)
            counts = torch.empty(2, dtype=torch.float32, device=dev)
            ws = torch.empty(lib.yolo_loss_workspace_bytes(B, g), dtype=torch.uint8, device=dev)
            strides = (C.c_int64 * 5)(*pred.stride())
            L.check(lib.yolo_loss_fwd(pred.data_ptr(), strides, target.data_ptr(), anchors.data_ptr(), B, g, D - 5, out.data_ptr(),
                                      counts.data_ptr(), ws.data_ptr(), ws.numel(), L.current_stream()), "yolo_loss_fwd")
        ctx.save_for_backward(pred, target, anchors, counts)
        return out

    @staticmethod
    def backward(ctx, gout):
        import ctypes as C
        from . import _lib as L
        pred, target, anchors, counts = ctx.saved_tensors
        B, A, g, _, D = pred.shape
        with torch.cuda.device(pred.device):
            dpred = torch.empty((B, A, g, g, D), dtype=torch.float32, device=pred.device)
            gout = gout.float().contiguous()
            strides = (C.c_int64 * 5)(*pred.stride())
            L.check(L.lib().yolo_loss_bwd(pred.data_ptr(), strides, target.data_ptr(), anchors.data_ptr(), B, g, D - 5,
                                          counts.data_ptr(), gout.data_ptr(), dpred.data_ptr(), L.current_stream()), "yolo_loss_bwd")
        return dpred, None, None


class FusedYOLOLoss(nn.Module):
    """Same values and gradients as :class:`YOLOLoss` (reference `loss.py:29-81`) from three fused HIP
    kernels per scale instead of ~35 boolean-mask / reduction launches: no data-dependent shapes, no host
    sync, deterministic sums — so a whole fine-tune step can be captured in a HIP graph
    (``tools/train_bench.py --graph``). Same call signature and return value; it does NOT mutate
    ``predictions`` / ``targets`` (the reference overwrites ``predictions[...,1:3]`` and ``targets[...,2:4]``)."""

    def forward(self, predictions, targets, anchors):
        if not predictions.is_cuda:
            raise RuntimeError("FusedYOLOLoss runs on MI355X only (no CPU fallback); use YOLOLoss on the CPU")
        if predictions.dim() != 5 or predictions.shape[1] != 3:
            raise ValueError("predictions must be a (B,3,g,g,5+nc) tensor")
        if predictions.dtype in (torch.float16, torch.bfloat16):
            predictions = predictions.float()          # autocast heads (train.py:53): the loss terms are evaluated in fp32, as autocast does
        elif predictions.dtype != torch.float32:
            raise ValueError("predictions must be a floating-point (B,3,g,g,5+nc) tensor")
        t = targets.detach()
        if t.dtype != torch.float32 or not t.is_contiguous():
            t = t.float().contiguous()
        a = anchors.detach().reshape(3, 2).to(device=predictions.device, dtype=torch.float32).contiguous()
        out = _FusedLossFn.apply(predictions, t, a)
        # ONE backward node for the four parts (unbind -> stack): indexing out[0] .. out[3] gave four SelectBackward nodes, each a
        # zero-fill + a 4-byte blit copy + an accumulate - 36 tiny launches per step over the three scales
        return list(out.unbind(0))
