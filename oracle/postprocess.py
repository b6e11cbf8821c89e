"""ORACLE — TEST INFRASTRUCTURE ONLY. Never imported by the product package.

CPU restatement of the reference post-processing:
  * ``calc_iou``            `/root/reference/code/utils.py:38-84`
  * ``cells_to_boxes``      `/root/reference/code/utils.py:86-148`
  * ``non_max_suppression`` `/root/reference/code/utils.py:150-191`

Three forms of NMS are kept on purpose:
  ``nms_list``     — same data flow as the reference (Python ``sorted`` on lists, greedy
                     loop of small torch ops). This is the "port" that bench.py times as
                     ``cpu_baseline`` for the NMS metric.
  ``nms_indices``  — numpy fp32, one rounding per operation, returns kept *indices into
                     the input list* (what the native entry point returns).
  ``nms_ref.c``    — the same in plain C (``-ffp-contract=off``), loaded by ``load_c()``;
                     used where Python would be too slow (N = 10,000+).
All three are pinned by ``tests/golden/nms_*.npz`` generated from the imported reference.
"""
from __future__ import annotations

import ctypes
import os
import subprocess

import numpy as np
import torch

_HERE = os.path.dirname(os.path.abspath(__file__))
F32 = np.float32


# ---------------------------------------------------------------- calc_iou (utils.py:38-84)
def calc_iou(boxes1: torch.Tensor, boxes2: torch.Tensor, box_format: str = "center") -> torch.Tensor:
    if boxes1.dim() == 1:
        boxes1 = boxes1[None]
    if boxes2.dim() == 1:
        boxes2 = boxes2[None]
    if box_format == "center":                      # only this string converts (utils.py:57)
        a = torch.cat([boxes1[..., :2] - boxes1[..., 2:4] / 2, boxes1[..., 2:4]], -1)
        b = torch.cat([boxes2[..., :2] - boxes2[..., 2:4] / 2, boxes2[..., 2:4]], -1)
    else:                                           # taken as (x1, y1, w, h) as is
        a, b = boxes1, boxes2
    xa = torch.max(a[..., 0], b[..., 0])
    ya = torch.max(a[..., 1], b[..., 1])
    xb = torch.min(a[..., 0] + a[..., 2], b[..., 0] + b[..., 2])
    yb = torch.min(a[..., 1] + a[..., 3], b[..., 1] + b[..., 3])
    inter = torch.clamp(xb - xa, min=0) * torch.clamp(yb - ya, min=0)
    union = a[..., 2] * a[..., 3] + b[..., 2] * b[..., 3] - inter
    return inter / (union + 1e-6)


def iou_np(top, rest, center: bool):
    """fp32 numpy, same op order as above. ``top`` (4,), ``rest`` (M,4)."""
    top = np.asarray(top, F32)
    rest = np.asarray(rest, F32)
    if center:
        ax, ay = top[0] - top[2] / F32(2), top[1] - top[3] / F32(2)
        bx, by = rest[:, 0] - rest[:, 2] / F32(2), rest[:, 1] - rest[:, 3] / F32(2)
    else:
        ax, ay = top[0], top[1]
        bx, by = rest[:, 0], rest[:, 1]
    aw, ah, bw, bh = top[2], top[3], rest[:, 2], rest[:, 3]
    xa = np.maximum(ax, bx)
    ya = np.maximum(ay, by)
    xb = np.minimum(ax + aw, bx + bw)
    yb = np.minimum(ay + ah, by + bh)
    iw = np.maximum(xb - xa, F32(0))
    ih = np.maximum(yb - ya, F32(0))
    inter = iw * ih
    union = (aw * ah + bw * bh) - inter
    return inter / (union + F32(1e-6))


# ------------------------------------------------------- cells_to_boxes (utils.py:86-148)
def cells_to_boxes(predictions: torch.Tensor, anchors: torch.Tensor, grid_size: int, is_pred: bool = True):
    """Returns the (B, 3*g*g, 6) tensor (the reference returns ``.tolist()`` of it) and,
    like the reference, overwrites ``predictions[..., 0:4]`` when ``is_pred``."""
    b = predictions.shape[0]
    na = len(anchors)
    box = predictions[..., :4]
    if is_pred:
        box[..., 0:2] = torch.sigmoid(box[..., 0:2])
        box[..., 2:] = torch.exp(box[..., 2:]) * anchors.reshape(1, na, 1, 1, 2)
        obj = torch.sigmoid(predictions[..., 4:5])
        cls = torch.argmax(predictions[..., 5:], dim=-1).unsqueeze(-1)
    else:
        obj = predictions[..., 4:5]
        cls = predictions[..., 5:]
    col = torch.arange(grid_size, device=predictions.device).view(1, 1, 1, grid_size, 1)
    row = torch.arange(grid_size, device=predictions.device).view(1, 1, grid_size, 1, 1)
    inv = 1 / grid_size
    cx = inv * (box[..., 0:1] + col)
    cy = inv * (box[..., 1:2] + row)
    wh = inv * box[..., 2:]
    out = torch.cat((cx, cy, wh, obj, cls), dim=-1)
    return out.reshape(b, na * grid_size * grid_size, 6)


# ------------------------------------------------ non_max_suppression (utils.py:150-191)
def nms_list(boxes, iou_threshold, obj_threshold, box_format="corners"):
    """Same flow as the reference: list filter, stable sort, greedy loop of torch ops."""
    cand = [bx for bx in boxes if bx[4] > obj_threshold]
    cand.sort(key=lambda bx: bx[4], reverse=True)          # stable, ties keep list order
    rest = torch.tensor(cand)
    kept = []
    while rest.size(0) > 0:
        top, rest = rest[0], rest[1:]
        ious = calc_iou(top[:4].unsqueeze(0), rest[:, :4], box_format)
        rest = rest[(rest[:, 5] != top[5]) | (ious < iou_threshold)]
        kept.append(top)
    return torch.stack(kept).tolist() if kept else []


def nms_indices(boxes, iou_threshold, obj_threshold, box_format="corners"):
    """Kept indices into ``boxes`` (score-descending, ties by input order)."""
    arr = np.asarray(boxes, dtype=np.float64).reshape(-1, 6)
    if arr.shape[0] == 0:
        return np.zeros(0, np.int64)
    cand = np.nonzero(arr[:, 4] > float(obj_threshold))[0]       # Python-float compare (utils.py:165)
    order = cand[np.argsort(-arr[cand, 4], kind="stable")]
    b = arr[order].astype(F32)                                  # torch.tensor(list) -> fp32 (utils.py:166)
    thr = F32(iou_threshold)                                    # compared in fp32
    center = box_format == "center"
    alive = np.ones(len(order), bool)
    keep = []
    for i in range(len(order)):
        if not alive[i]:
            continue
        keep.append(order[i])
        j = np.nonzero(alive[i + 1:])[0] + i + 1
        if j.size == 0:
            continue
        iou = iou_np(b[i, :4], b[j, :4], center)
        survive = (b[j, 5] != b[i, 5]) | (iou < thr)
        alive[j[~survive]] = False
    return np.asarray(keep, np.int64)


# ---------------------------------------------------------------------------- C checker
_C = None


def build_c(force=False):
    so = os.path.join(_HERE, "libnms_ref.so")
    src = os.path.join(_HERE, "nms_ref.c")
    if force or not os.path.exists(so) or os.path.getmtime(so) < os.path.getmtime(src):
        subprocess.check_call(["gcc", "-O2", "-ffp-contract=off", "-fno-fast-math", "-shared", "-fPIC",
                               "-o", so, src])
    return so


def load_c():
    global _C
    if _C is None:
        lib = ctypes.CDLL(build_c())
        lib.nms_ref.restype = ctypes.c_int
        lib.nms_ref.argtypes = [ctypes.c_void_p, ctypes.c_int, ctypes.c_double, ctypes.c_double,
                                ctypes.c_int, ctypes.c_void_p]
        lib.decode_ref.restype = None
        lib.decode_ref.argtypes = [ctypes.c_void_p, ctypes.c_void_p, ctypes.c_int, ctypes.c_int,
                                   ctypes.c_int, ctypes.c_void_p]
        _C = lib
    return _C


def nms_indices_c(boxes_f32: np.ndarray, iou_threshold, obj_threshold, box_format="corners"):
    """``boxes_f32``: (N,6) float32 (values as the reference would hold them after .tolist())."""
    lib = load_c()
    b = np.ascontiguousarray(boxes_f32, dtype=F32).reshape(-1, 6)
    keep = np.empty(max(len(b), 1), np.int32)
    k = lib.nms_ref(b.ctypes.data, len(b), float(iou_threshold), float(obj_threshold),
                    int(box_format == "center"), keep.ctypes.data)
    return keep[:k].astype(np.int64)


def decode_c(pred_f32: np.ndarray, anchors_f32: np.ndarray, grid: int):
    """(B,3,g,g,5+nc) fp32 contiguous -> (B,3*g*g,6) fp32; libm expf-based (tolerance check only)."""
    lib = load_c()
    p = np.ascontiguousarray(pred_f32, dtype=F32)
    b, a, g, _, d = p.shape
    assert a == 3 and g == grid
    out = np.empty((b, 3 * g * g, 6), F32)
    anc = np.ascontiguousarray(anchors_f32, dtype=F32)
    lib.decode_ref(p.ctypes.data, anc.ctypes.data, b, g, d - 5, out.ctypes.data)
    return out
