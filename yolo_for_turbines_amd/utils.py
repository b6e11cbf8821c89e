"""Drop-in mirror of the reference's post-processing functions
(`/root/reference/code/utils.py:22-191`) on top of the HIP kernels.

Two levels:
  * list-returning wrappers with the reference's names and signatures (``cells_to_boxes``,
    ``non_max_suppression``, ``calc_iou``, ``iou_aligned``) so ``demo.predict`` /
    ``get_eval_boxes`` keep working unchanged (they pay the reference's ``.tolist()`` cost);
  * device-resident entries (``decode_boxes``, ``nms_indices``, ``detect``) that keep every
    intermediate in HBM and return tensors — this is the path that is benchmarked.
"""
from __future__ import annotations

import ctypes as C

import torch

from . import _lib as L

__all__ = ["iou_aligned", "calc_iou", "cells_to_boxes", "non_max_suppression", "decode_boxes", "nms_indices",
           "detect", "build_targets", "calc_mAP", "accuracy_counts", "check_model_accuracy", "eval_boxes", "get_eval_boxes", "letterbox", "unletterbox_boxes",
           "save_checkpoint", "load_checkpoint"]


# -------------------------------------------------------------------------------- IoU
def iou_aligned(box1, box2):
    """Width/height IoU of centre-aligned boxes (utils.py:22-36). Elementwise torch ops; not on
    the accelerated path (used by the dataset's anchor matching)."""
    inter = torch.min(box1[..., 0], box2[..., 0]) * torch.min(box1[..., 1], box2[..., 1])
    return inter / (box1[..., 0] * box1[..., 1] + box2[..., 0] * box2[..., 1] - inter)


def calc_iou(boxes1, boxes2, box_format="center"):
    """IoU of cxcywh ("center") or x1y1wh (anything else) boxes, +1e-6 in the denominator
    (utils.py:38-84). Elementwise torch ops on whatever device the inputs live on: it is the
    objectness-loss helper (loss.py:64); inside NMS the same arithmetic runs in the HIP kernel."""
    if boxes1.dim() == 1:
        boxes1 = boxes1.unsqueeze(0)
    if boxes2.dim() == 1:
        boxes2 = boxes2.unsqueeze(0)
    if box_format == "center":
        x1, y1 = boxes1[..., 0] - boxes1[..., 2] / 2, boxes1[..., 1] - boxes1[..., 3] / 2
        x2, y2 = boxes2[..., 0] - boxes2[..., 2] / 2, boxes2[..., 1] - boxes2[..., 3] / 2
    else:
        x1, y1, x2, y2 = boxes1[..., 0], boxes1[..., 1], boxes2[..., 0], boxes2[..., 1]
    w1, h1, w2, h2 = boxes1[..., 2], boxes1[..., 3], boxes2[..., 2], boxes2[..., 3]
    iw = torch.clamp(torch.min(x1 + w1, x2 + w2) - torch.max(x1, x2), min=0)
    ih = torch.clamp(torch.min(y1 + h1, y2 + h2) - torch.max(y1, y2), min=0)
    inter = iw * ih
    return inter / (w1 * h1 + w2 * h2 - inter + 1e-6)


# ------------------------------------------------------------------------------ decode
def decode_boxes(predictions, anchors, grid_size=None, is_pred=True, out=None, box_offset=0, mutate=True):
    """Device decode of one scale: (B,3,g,g,5+nc) -> (B, 3*g*g, 6) fp32 tensor
    [cx,cy,w,h,obj,cls] normalised to [0,1]; mutates ``predictions[...,0:4]`` in place when
    ``is_pred`` exactly like the reference (utils.py:106-110) unless ``mutate=False``. ``out`` / ``box_offset`` let
    several scales share one (B, N_total, 6) buffer."""
    if not predictions.is_cuda:
        raise RuntimeError("decode_boxes runs on MI355X only (no CPU fallback)")
    if predictions.dtype in (torch.float16, torch.bfloat16):
        # heads handed out under autocast: decode an fp32 copy and write the in-place side effect (utils.py:106-110) back
        p32 = predictions.float()
        res = decode_boxes(p32, anchors, grid_size, is_pred, out, box_offset, mutate)
        if is_pred and mutate:
            predictions[..., 0:4] = p32[..., 0:4].to(predictions.dtype)
        return res
    if predictions.dtype != torch.float32:
        raise NotImplementedError("decode_boxes: floating-point predictions only")
    B, A, g, g2, D = predictions.shape
    if A != 3 or g != g2 or (grid_size is not None and int(grid_size) != g):
        raise ValueError(f"bad prediction shape {tuple(predictions.shape)} for grid {grid_size}")
    n = 3 * g * g
    if out is None:
        out = torch.empty((B, n, 6), dtype=torch.float32, device=predictions.device)
        box_offset = 0
    if out.dtype != torch.float32 or not out.is_contiguous() or out.shape[0] != B or out.shape[2] != 6:
        raise ValueError("out must be a contiguous (B, N, 6) fp32 tensor")
    anc = torch.as_tensor(anchors, dtype=torch.float32, device=predictions.device).reshape(3, 2).contiguous() \
        if is_pred else None
    strides = (C.c_int64 * 5)(*predictions.stride())
    with torch.cuda.device(predictions.device):
        L.check(L.lib().yolo_decode(predictions.data_ptr(), strides, L.ptr(anc), B, g, D - 5, (1 if mutate else 2) if is_pred else 0,
                                    out.data_ptr(), out.shape[1], int(box_offset), L.current_stream()), "yolo_decode")
    return out


def cells_to_boxes(predictions, anchors, grid_size, is_pred=True):
    """Reference signature (utils.py:86-148): returns ``list[B][3*g*g][6]``."""
    return decode_boxes(predictions, anchors, grid_size, is_pred).tolist()


# --------------------------------------------------------------------------------- NMS
_ws_cache = {}


def _workspace(nbytes, device):
    key = device.index
    ws = _ws_cache.get(key)
    if ws is None or ws.numel() < nbytes:
        ws = _ws_cache[key] = torch.empty(nbytes, dtype=torch.uint8, device=device)
    return ws


def nms_indices(boxes, iou_threshold, obj_threshold, box_format="corners"):
    """Batched device NMS. ``boxes``: (B, N, 6) or (N, 6) fp32 CUDA tensor of
    [x,y,w,h,obj,cls]. Returns (keep_idx (B,N) int32, keep_count (B,) int32): for image b the
    first keep_count[b] entries index its input rows in the reference's output order."""
    if not boxes.is_cuda:
        raise RuntimeError("nms_indices runs on MI355X only (no CPU fallback)")
    single = boxes.dim() == 2
    if single:
        boxes = boxes.unsqueeze(0)
    if boxes.dim() != 3 or boxes.shape[-1] != 6:
        raise ValueError("boxes must be (B, N, 6)")
    boxes = boxes.float().contiguous()
    B, N, _ = boxes.shape
    keep = torch.empty((B, max(N, 1)), dtype=torch.int32, device=boxes.device)
    count = torch.empty((B,), dtype=torch.int32, device=boxes.device)
    lib = L.lib()
    with torch.cuda.device(boxes.device):
        nbytes = lib.yolo_nms_workspace_bytes(B, N)
        ws = _workspace(max(nbytes, 256), boxes.device)
        L.check(lib.yolo_nms(boxes.data_ptr(), B, N, float(iou_threshold), float(obj_threshold),
                             int(box_format == "center"), keep.data_ptr(), count.data_ptr(), ws.data_ptr(), ws.numel(),
                             L.current_stream()), "yolo_nms")
    if single:
        return keep[0], count[0]
    return keep, count


def non_max_suppression(boxes, iou_threshold, obj_threshold, box_format="corners"):
    """Reference signature (utils.py:150-191): list of [x,y,w,h,obj,cls] lists in, kept boxes
    out (objectness-descending). The list is shipped to the GPU, suppressed there, and the kept
    rows are returned as Python lists of the same fp32 values the reference would produce."""
    if len(boxes) == 0:
        return []
    dev = torch.device("cuda", torch.cuda.current_device())
    t = torch.tensor(boxes, dtype=torch.float32).reshape(-1, 6).to(dev)
    keep, count = nms_indices(t, iou_threshold, obj_threshold, box_format)
    k = int(count.item())
    return t[keep[:k].long()].tolist() if k else []


def detect(predictions, scaled_anchors, iou_threshold=0.45, obj_threshold=0.5, box_format="center", mutate=False):
    """Fused post-processing of a forward pass (the sequence of demo.py:44-55 /
    utils.py:300-321, all images at once, nothing leaves HBM):
    decode three scales into one (B, N, 6) buffer in the reference's concatenation order
    (scale 0, 1, 2), then per-image NMS. Returns (boxes (B,N,6), keep_idx (B,N), keep_count (B,)).
    ``mutate=True`` also performs the in-place sigmoid / exp write-back ``cells_to_boxes`` does to the prediction tensors
    (utils.py:106-110); the reference's callers of this sequence never read them again, so the default leaves them alone."""
    B = predictions[0].shape[0]
    n_per = [3 * p.shape[2] * p.shape[2] for p in predictions]
    total = sum(n_per)
    dev = predictions[0].device
    boxes = torch.empty((B, total, 6), dtype=torch.float32, device=dev)
    if len(predictions) == 3 and all(p.is_cuda and p.dtype == torch.float32 and p.dim() == 5 for p in predictions) \
            and len({p.shape[4] for p in predictions}) == 1:
        with torch.cuda.device(dev):                      # the three scales in one launch
            anc = [torch.as_tensor(a, dtype=torch.float32).reshape(3, 2).to(dev).contiguous() for a in scaled_anchors]
            pp = (C.c_void_p * 3)(*[p.data_ptr() for p in predictions])
            st = (C.c_int64 * 15)(*[v for p in predictions for v in p.stride()])
            ap = (C.c_void_p * 3)(*[a.data_ptr() for a in anc])
            gg = (C.c_int * 3)(*[p.shape[2] for p in predictions])
            L.check(L.lib().yolo_decode3_ex(pp, st, ap, gg, B, predictions[0].shape[4] - 5, int(bool(mutate)), boxes.data_ptr(), total,
                                            L.current_stream()), "yolo_decode3")
    else:
        off = 0
        for p, a, n in zip(predictions, scaled_anchors, n_per):
            decode_boxes(p, a, p.shape[2], True, out=boxes, box_offset=off, mutate=mutate)
            off += n
    keep, count = nms_indices(boxes, iou_threshold, obj_threshold, box_format)
    return boxes, keep, count


def detect_images(model, x, scaled_anchors, iou_threshold=0.45, obj_threshold=0.5, box_format="center"):
    """``detect(model(x), ...)`` with ONE host synchronisation instead of two back to back: the forward's NaN guards
    (model.py:175,183-184) are read after decode and NMS have been enqueued, so the GPU does not sit idle between the
    forward and the post-processing while the host wakes up and launches them (~0.1-0.25 ms per batch). Same results, same
    exceptions (``AssertionError`` for a NaN input, ``ValueError("Nan in layer")``), raised before anything is returned."""
    eng = model._engine
    eng._defer_nan, eng._pending_flag = True, None
    try:
        with torch.no_grad():
            preds = model(x)
        flag = eng._pending_flag
    finally:
        eng._defer_nan, eng._pending_flag = False, None
    out = detect(preds, scaled_anchors, iou_threshold, obj_threshold, box_format)
    if flag is not None:
        eng.raise_on_nan(flag)
    return out


# ------------------------------------------------------------------------------ targets
def build_targets(boxes, anchors, image_size, counts=None, ignore_iou_threshold=0.5, device=None):
    """Batched device version of the target loop of ``YOLODataset.__getitem__`` (dataset.py:119-161).

    ``boxes``: per-image lists ``[[x, y, w, h, class], ...]`` (normalised, the dataset's order) or a padded
    ``(B, max_boxes, 5)`` tensor with ``counts`` (B). ``anchors``: the 3x3x2 normalised anchor table
    (``config.ANCHORS``). Returns the tuple of three ``(B, 3, g, g, 6)`` fp32 tensors the loss consumes."""
    if isinstance(boxes, torch.Tensor):
        if counts is None:
            raise ValueError("a padded box tensor needs `counts`")
        bt = boxes.to(dtype=torch.float32)
        dev = bt.device if device is None else torch.device(device)
        ct = torch.as_tensor(counts, dtype=torch.int32)
    else:
        B = len(boxes)
        mb = max(1, max((len(b) for b in boxes), default=1))
        host = torch.zeros((B, mb, 5), dtype=torch.float32)
        ct = torch.tensor([len(b) for b in boxes], dtype=torch.int32)
        for i, bl in enumerate(boxes):
            if len(bl):
                host[i, :len(bl)] = torch.as_tensor(bl, dtype=torch.float32).reshape(-1, 5)
        bt = host
        dev = torch.device("cuda" if device is None else device)
    if dev.type != "cuda":
        raise RuntimeError("build_targets runs on MI355X only (no CPU fallback)")
    bt = bt.to(dev).contiguous()
    ct = ct.to(dev).contiguous()
    anc = torch.as_tensor(anchors, dtype=torch.float32).reshape(9, 2).to(dev).contiguous()
    B, mb = bt.shape[0], bt.shape[1]
    S = int(image_size)
    with torch.cuda.device(dev):
        outs = [torch.empty((B, 3, g, g, 6), dtype=torch.float32, device=dev) for g in (S // 32, S // 16, S // 8)]
        L.check(L.lib().yolo_build_targets(bt.data_ptr(), ct.data_ptr(), mb, anc.data_ptr(), B, S, float(ignore_iou_threshold),
                                           outs[0].data_ptr(), outs[1].data_ptr(), outs[2].data_ptr(), L.current_stream()),
                "yolo_build_targets")
    return tuple(outs)


def _stable_order(major, minor, descending_minor):
    """Indices that order rows by (``major`` ascending, ``minor`` ascending / descending, original position): what two stable
    Python list sorts produce (utils.py:206,232). ``major`` holds small non-negative integers (class ids), ``minor`` any
    floats. One sort of unique 64-bit keys  major (11 bits) | orderable(minor) (32) | index (20)  by ``yolo_sort_u64`` (the
    NMS ordering kernels); above 262,144 rows or 2,047 classes the same two stable sorts run through ``torch.sort``."""
    n = int(major.shape[0])
    if n == 0:
        return torch.empty(0, dtype=torch.long, device=major.device)
    if n > 262144 or n >= (1 << 20) or float(major.max()) > 2046:
        o = torch.sort(minor, descending=descending_minor, stable=True).indices
        return o[torch.sort(major[o], stable=True).indices]
    bits = (minor.float() + 0.0).contiguous().view(torch.int32).to(torch.int64) & 0xffffffff      # -0.0 -> +0.0: equal in Python
    u = torch.where(bits >= 0x80000000, bits ^ 0xffffffff, bits | 0x80000000)                     # ascending-orderable
    if descending_minor:
        u = 0xffffffff - u
    keys = (major.to(torch.int64) << 52) | (u << 20) | torch.arange(n, dtype=torch.int64, device=major.device)
    out = torch.empty_like(keys)
    with torch.cuda.device(major.device):
        ws = torch.empty(L.lib().yolo_sort_u64_workspace_bytes(n), dtype=torch.uint8, device=major.device)
        L.check(L.lib().yolo_sort_u64(keys.data_ptr(), out.data_ptr(), n, ws.data_ptr(), ws.numel(), L.current_stream()), "yolo_sort_u64")
    return out & 0xfffff


# ------------------------------------------------------------------------------ mAP
def calc_mAP(pred_boxes, true_boxes, iou_threshold=0.5, box_format="center", num_classes=20):
    """Drop-in for the reference's ``calc_mAP`` (utils.py:193-274): rows ``[image_id, cx, cy, w, h, obj, class]``
    (lists or tensors), returns the mean over the classes that have ground truth of the trapezoid area under the
    precision/recall curve, as a 0-dim CPU tensor. The two stable list sorts become one device sort each
    (``_stable_order``), the per-pair Python loop two kernels (sequential matching per class, AP integration)."""
    dev = torch.device("cuda")
    dets = torch.as_tensor(pred_boxes, dtype=torch.float32).reshape(-1, 7).to(dev)
    gts = torch.as_tensor(true_boxes, dtype=torch.float32).reshape(-1, 7).to(dev)
    nc = int(num_classes)

    def class_rows(t):                                     # `box[-1] == c` for c in range(num_classes)
        c = t[:, 6]
        ok = (c >= 0) & (c < nc) & (c == torch.floor(c))
        return t[ok]
    dets, gts = class_rows(dets), class_rows(gts)
    # detections: objectness descending, then class ascending — both stable, so ties keep list order (list.sort)
    dets = dets[_stable_order(dets[:, 6], dets[:, 5], descending_minor=True)].contiguous()
    # ground truths: image ascending, then class ascending (stable): per (class, image) the list order survives
    gts = gts[_stable_order(gts[:, 6], gts[:, 0], descending_minor=False)].contiguous()

    def offsets(t):
        cnt = torch.bincount(t[:, 6].long(), minlength=nc)[:nc]
        return torch.cat([torch.zeros(1, dtype=torch.long, device=dev), torch.cumsum(cnt, 0)]).to(torch.int32).contiguous()
    d_off, g_off = offsets(dets), offsets(gts)
    with torch.cuda.device(dev):
        assigned = torch.empty(max(1, gts.shape[0]), dtype=torch.int32, device=dev)
        tp = torch.zeros(max(1, dets.shape[0]), dtype=torch.float32, device=dev)
        ap = torch.empty(nc, dtype=torch.float32, device=dev)
        L.check(L.lib().yolo_map_match(dets.data_ptr(), d_off.data_ptr(), gts.data_ptr(), g_off.data_ptr(), nc, gts.shape[0],
                                       float(iou_threshold), 1 if box_format == "center" else 0, assigned.data_ptr(), tp.data_ptr(),
                                       ap.data_ptr(), L.current_stream()), "yolo_map_match")
    ap = ap.cpu()
    valid = ap >= 0
    if not bool(valid.any()):
        raise ZeroDivisionError("division by zero")        # the reference divides by len([]) here
    return ap[valid].sum() / int(valid.sum())


# ------------------------------------------------------------------------------ accuracy
def accuracy_counts(predictions, targets, object_threshold, counts=None):
    """Accumulate the five counters of ``check_model_accuracy`` (utils.py:355-372) for one batch: ``predictions`` /
    ``targets`` are the three per-scale tensors; returns an int64 tensor [class correct, n_obj, obj correct,
    noobj correct, n_noobj] on the device (pass it back in as ``counts`` to keep accumulating)."""
    dev = predictions[0].device
    if dev.type != "cuda":
        raise RuntimeError("accuracy_counts runs on MI355X only (no CPU fallback)")
    if counts is None:
        counts = torch.zeros(5, dtype=torch.int64, device=dev)
    with torch.cuda.device(dev):
        for p, t in zip(predictions, targets):
            if p.dtype != torch.float32:
                raise ValueError("fp32 predictions expected")
            t = t.to(dev, torch.float32).contiguous()
            B, _, g, _, D = p.shape
            strides = (C.c_int64 * 5)(*p.stride())
            L.check(L.lib().yolo_accuracy_counts(p.data_ptr(), strides, t.data_ptr(), B, g, D - 5, float(object_threshold),
                                                 counts.data_ptr(), L.current_stream()), "yolo_accuracy_counts")
    return counts


def check_model_accuracy(model, loader, object_threshold):
    """Drop-in for the reference's ``check_model_accuracy`` (utils.py:334-381): class / no-object / object accuracy over
    a loader; one fused counting kernel per scale instead of ~15 mask / gather / reduce launches, one host sync at the end."""
    was_training = model.training
    model.eval()
    counts = None
    dev = next(model.parameters()).device
    for x, target in loader:
        with torch.no_grad():
            out = model(x.to(dev))
        counts = accuracy_counts(out, list(target), object_threshold, counts)
    c = [0] * 5 if counts is None else [int(v) for v in counts.tolist()]
    ct = [torch.tensor(v) for v in c]                      # int64 / (int64 + 1e-16) -> fp32 division, as in the reference
    class_accuracy = ct[0] / (ct[1] + 1e-16)
    noobj_accuracy = ct[3] / (ct[4] + 1e-16)
    obj_accuracy = ct[2] / (ct[1] + 1e-16)
    print(f"Class accuracy is: {(class_accuracy) * 100:2f}%")
    print(f"No obj accuracy is: {(noobj_accuracy) * 100:2f}%")
    print(f"Obj accuracy is: {(obj_accuracy) * 100:2f}%")
    if was_training:
        model.train()
    return class_accuracy, noobj_accuracy, obj_accuracy


# ------------------------------------------------------------------------------ evaluation boxes
def eval_boxes(loader, model, iou_threshold, anchors, obj_threshold, box_format="center"):
    """Device-resident ``get_eval_boxes`` (utils.py:276-332): per batch one forward, one fused decode + batched NMS
    (:func:`detect`) and one decode of the last-scale targets; returns two tensors of rows
    ``[image_id, cx, cy, w, h, obj, class]`` — kept predictions in the reference's order (image by image,
    objectness-descending) and ground-truth boxes (cells of ``targets[2]`` with objectness > ``obj_threshold``).
    Feed them straight to :func:`calc_mAP`."""
    was_training = model.training
    model.eval()
    dev = next(model.parameters()).device
    preds_out, trues_out = [], []
    data_idx = 0
    for x, targets in loader:
        with torch.no_grad():
            predictions = model(x.to(dev))
        B = x.shape[0]
        sa = [torch.as_tensor(anchors[i], dtype=torch.float32, device=dev).reshape(3, 2) * predictions[i].shape[2] for i in range(3)]
        boxes, keep, count = detect(predictions, sa, iou_threshold, obj_threshold, box_format)
        g = predictions[2].shape[2]
        true_boxes = decode_boxes(targets[2].to(dev, torch.float32).contiguous(), sa[2], g, is_pred=False)     # (B, 3 g^2, 6)
        cnt = count.tolist()
        for b in range(B):
            kb = boxes[b, keep[b, :cnt[b]].long()]
            ids = torch.full((kb.shape[0], 1), float(data_idx + b), device=dev)
            preds_out.append(torch.cat([ids, kb], 1))
            tb = true_boxes[b][true_boxes[b][:, 4] > obj_threshold]
            trues_out.append(torch.cat([torch.full((tb.shape[0], 1), float(data_idx + b), device=dev), tb], 1))
        data_idx += B
    if was_training:
        model.train()
    empty = torch.zeros((0, 7), device=dev)
    return (torch.cat(preds_out) if preds_out else empty), (torch.cat(trues_out) if trues_out else empty)


def get_eval_boxes(loader, model, iou_threshold, anchors, obj_threshold, box_format="center", device=None):
    """Drop-in for the reference's ``get_eval_boxes`` (utils.py:276-332): same arguments, same two Python lists of
    ``[image_id, cx, cy, w, h, obj, class]`` rows. (``device`` is accepted for signature compatibility; the model's
    device is used.) It leaves the model in train mode like the reference (utils.py:331)."""
    p, t = eval_boxes(loader, model, iou_threshold, anchors, obj_threshold, box_format)
    model.train()
    return p.tolist(), t.tolist()


# ------------------------------------------------------------------------------ letterbox
def letterbox(images, image_size=416, device=None):
    """``config.set_only_image_transforms`` (config.py:101-113) on the device: each uint8 (H, W, 3) image (tensor or
    array) is resized so that its longer side is ``image_size`` (bilinear), centred on a zero canvas, scaled to [0,1]
    and laid out CHW. Returns ``(batch (B,3,S,S) fp32, meta)`` with ``meta[i] = (orig_h, orig_w, new_h, new_w,
    pad_top, pad_left)`` for :func:`unletterbox_boxes`. Parity with cv2 is unpinned (see csrc/preprocess.hip)."""
    if isinstance(images, (torch.Tensor,)) and images.dim() == 3 or not isinstance(images, (list, tuple)):
        images = [images]
    dev = torch.device("cuda" if device is None else device)
    if dev.type != "cuda":
        raise RuntimeError("letterbox runs on MI355X only (no CPU fallback)")
    S = int(image_size)
    with torch.cuda.device(dev):
        out = torch.empty((len(images), 3, S, S), dtype=torch.float32, device=dev)
        meta = []
        for i, im in enumerate(images):
            t = torch.as_tensor(im)
            if t.dtype != torch.uint8 or t.dim() != 3 or t.shape[2] != 3:
                raise ValueError("images must be uint8 (H, W, 3)")
            t = t.to(dev).contiguous()
            nhw, pad = (C.c_int * 2)(), (C.c_int * 2)()
            L.check(L.lib().yolo_letterbox(t.data_ptr(), t.shape[0], t.shape[1], S, out[i].data_ptr(), nhw, pad, L.current_stream()),
                    "yolo_letterbox")
            t.record_stream(torch.cuda.current_stream())
            meta.append((int(t.shape[0]), int(t.shape[1]), nhw[0], nhw[1], pad[0], pad[1]))
    return out, meta


def unletterbox_boxes(boxes, original_hw, resized_hw):
    """Box mapping of ``plot_original`` (utils.py:475-501): normalised letterboxed ``[cx, cy, w, h, obj, cls]`` rows ->
    coordinates normalised to the ORIGINAL image. Same arithmetic (incl. its ``int(o * scale)`` size and ``// 2`` padding)."""
    o_h, o_w = original_hw
    r_h, r_w = resized_hw
    scale = min(r_w / o_w, r_h / o_h)
    new_width, new_height = int(o_w * scale), int(o_h * scale)
    pad_width, pad_height = (r_w - new_width) // 2, (r_h - new_height) // 2
    return [[(b[0] * r_w - pad_width) / new_width, (b[1] * r_h - pad_height) / new_height, (b[2] * r_w) / new_width,
             (b[3] * r_h) / new_height, b[4], b[5]] for b in boxes]


# ------------------------------------------------------------------------ checkpoints
def save_checkpoint(model, optimizer, filename="YOLOv3TurbineCheckpoint.pth.tar"):
    """`utils.py:383-396`: ``{"state_dict": model.state_dict(), "optimizer": optimizer.state_dict()}`` through ``torch.save``.
    The model's 438 ``state_dict`` entries and its parameter order are the reference's, so a file written here loads with
    the reference's ``load_checkpoint`` and the other way round (pinned by ``tests/golden/checkpoint.npz``)."""
    torch.save({"state_dict": model.state_dict(), "optimizer": optimizer.state_dict()}, filename)


def load_checkpoint(model, optimizer, lr, filename="", model_folder=None, map_location=None):
    """`utils.py:398-416`: restore model and optimizer, then force every param group's ``lr``. The reference prefixes
    ``config.MODEL_FOLDER`` and maps to ``config.DEVICE``; there is no global config here, so both are arguments
    (``model_folder=None``: ``filename`` is the path; ``map_location=None``: the model's own device)."""
    path = filename if model_folder is None else f"{model_folder}/{filename}"
    if map_location is None:
        map_location = next(model.parameters()).device
    checkpoint = torch.load(path, map_location=map_location)
    model.load_state_dict(checkpoint["state_dict"])          # invalidates the packed-weight cache (post-hook in model.py)
    optimizer.load_state_dict(checkpoint["optimizer"])
    for param_group in optimizer.param_groups:
        param_group["lr"] = lr
    print(f"Checkpoint loaded from {filename}")
