"""Seeded input generators shared by ``tests/gen_golden.py`` (which runs the imported
reference in the build container) and by the parity tests (which regenerate the same
inputs on the GPU box, where the reference does not exist).  Only OUTPUTS are stored in
``tests/golden/``; everything here is numpy ``Generator(PCG64(seed))``.
"""
from __future__ import annotations

import numpy as np

F32 = np.float32

# ------------------------------------------------------------------ whole-network cases
# name -> (num_classes, size, batch, activation, weight seed, input seed, mode)
NET_CASES = {
    "nc80_s96_b2_leaky": dict(nc=80, size=96, batch=2, act="leaky_relu", wseed=0, xseed=1, full=True),
    "nc2_s128_b1_leaky": dict(nc=2, size=128, batch=1, act="leaky_relu", wseed=3, xseed=4, full=True),
    "nc80_s96_b1_mish": dict(nc=80, size=96, batch=1, act="mish", wseed=5, xseed=6, full=True),
    "nc80_s416_b1_leaky": dict(nc=80, size=416, batch=1, act="leaky_relu", wseed=0, xseed=7, full=False),
    "nc80_s608_b1_leaky": dict(nc=80, size=608, batch=1, act="leaky_relu", wseed=0, xseed=8, full=False),
}
NET_GAIN = 0.8
SAMPLE_STRIDE = 97          # strided sample of big tensors
TAP_KEYS = ("layers.0", "layers.1", "layers.6.layers.7.1", "layers.11", "layers.16", "layers.18", "layers.25")
TAP_STRIDE = 53

# --------------------------------------------------------- the 23 conv configs
# (cin, cout, k, stride, bn) with a reduced spatial size H for the block-level goldens
BLOCK_CONFIGS = [
    (3, 32, 3, 1, True, 20), (32, 64, 3, 2, True, 20),
    (64, 32, 1, 1, True, 14), (32, 64, 3, 1, True, 14), (64, 128, 3, 2, True, 14),
    (128, 64, 1, 1, True, 12), (64, 128, 3, 1, True, 12), (128, 256, 3, 2, True, 12),
    (256, 128, 1, 1, True, 10), (128, 256, 3, 1, True, 10), (384, 128, 1, 1, True, 10),
    (256, 255, 1, 1, False, 10), (256, 21, 1, 1, False, 10), (256, 512, 3, 2, True, 10),
    (512, 256, 1, 1, True, 9), (256, 512, 3, 1, True, 9), (768, 256, 1, 1, True, 9),
    (256, 128, 1, 1, True, 9), (512, 255, 1, 1, False, 9), (512, 21, 1, 1, False, 9),
    (512, 1024, 3, 2, True, 8),
    (1024, 512, 1, 1, True, 7), (512, 1024, 3, 1, True, 7), (512, 256, 1, 1, True, 7),
    (1024, 255, 1, 1, False, 7), (1024, 21, 1, 1, False, 7),
]
BLOCK_BATCH = 2
BLOCK_STRIDE = 11
BLOCK_DW_STRIDE = 41
TRAIN_GRAD_STRIDE = 101


def block_params(i, cin, cout, k, bn):
    """Parameters + input for block config ``i`` (dict of fp32 numpy arrays)."""
    rng = np.random.Generator(np.random.PCG64(1000 + i))
    p = {"w": (rng.standard_normal((cout, cin, k, k), dtype=F32) * F32(np.sqrt(1.0 / (cin * k * k))))}
    if bn:
        p["gamma"] = rng.uniform(0.5, 1.5, cout).astype(F32)
        p["beta"] = (0.2 * rng.standard_normal(cout)).astype(F32)
        p["mean"] = (0.2 * rng.standard_normal(cout)).astype(F32)
        p["var"] = rng.uniform(0.5, 1.5, cout).astype(F32)
    else:
        p["bias"] = (0.2 * rng.standard_normal(cout)).astype(F32)
    return p


def block_input(i, cin, h):
    rng = np.random.Generator(np.random.PCG64(2000 + i))
    return rng.standard_normal((BLOCK_BATCH, cin, h, h), dtype=F32)


# -------------------------------------------------------------------------- decode cases
DECODE_CASES = {
    "g13_nc80": dict(batch=2, g=13, nc=80, seed=31, anchors=[(0.28, 0.22), (0.38, 0.48), (0.9, 0.78)]),
    "g7_nc2": dict(batch=3, g=7, nc=2, seed=32, anchors=[(0.06, 0.143), (0.143, 0.189), (0.408, 0.181)]),
    "g19_nc80": dict(batch=1, g=19, nc=80, seed=33, anchors=[(0.02, 0.03), (0.04, 0.07), (0.08, 0.06)]),
}


def decode_input(case):
    c = DECODE_CASES[case]
    rng = np.random.Generator(np.random.PCG64(c["seed"]))
    pred = rng.standard_normal((c["batch"], 3, c["g"], c["g"], 5 + c["nc"]), dtype=F32) * F32(1.5)
    anchors = (np.asarray(c["anchors"], np.float64) * c["g"]).astype(F32)   # anchors * grid (demo.py:33-35)
    return pred, anchors


# ----------------------------------------------------------------------------- NMS cases
def boxes_uniform(n, nc, seed):
    """SURVEY §8d Config 5 (a): every box above the 0.5 objectness threshold."""
    rng = np.random.Generator(np.random.PCG64(seed))
    cx, cy = rng.random(n), rng.random(n)
    w = 0.02 + 0.3 * rng.random(n) ** 2
    h = 0.02 + 0.3 * rng.random(n) ** 2
    obj = 0.5 + 0.5 * rng.random(n)
    obj = np.where(obj <= 0.5, 0.75, obj)
    cls = rng.integers(0, nc, n).astype(np.float64)
    return np.stack([cx, cy, w, h, obj, cls], 1).astype(F32)


def boxes_clustered(n, nc, seed, n_gt=60, jitter=0.05):
    """SURVEY §8d Config 5 (b): n_gt objects, each spawning jittered same-class boxes."""
    rng = np.random.Generator(np.random.PCG64(seed))
    gcx, gcy = rng.random(n_gt), rng.random(n_gt)
    gw, gh = 0.05 + 0.3 * rng.random(n_gt), 0.05 + 0.3 * rng.random(n_gt)
    gcls = rng.integers(0, nc, n_gt)
    owner = rng.integers(0, n_gt, n)
    jit = lambda s: 1.0 + 0.05 * rng.standard_normal(n)
    cx = gcx[owner] + jitter * gw[owner] * rng.standard_normal(n)
    cy = gcy[owner] + jitter * gh[owner] * rng.standard_normal(n)
    w = gw[owner] * jit(0)
    h = gh[owner] * jit(1)
    obj = 0.5 + 0.5 * rng.beta(2, 2, n)
    obj = np.where(obj <= 0.5, 0.75, obj)
    return np.stack([cx, cy, w, h, obj, gcls[owner].astype(np.float64)], 1).astype(F32)


def boxes_mixed(n, nc, seed):
    """Half of the boxes below the objectness threshold; scores quantised so ties are common."""
    b = boxes_clustered(n, nc, seed, n_gt=12)
    rng = np.random.Generator(np.random.PCG64(seed + 7))
    b[:, 4] = (np.round(rng.random(n) * 20) / 20).astype(F32)       # 21 distinct scores
    return b


def boxes_adversarial():
    """Equal scores, exact duplicates, IoU exactly at / next to the fp32 threshold, zero-area
    boxes, score exactly at the objectness threshold."""
    rows = []
    # two identical boxes, same class, same score -> second suppressed (IoU = a/(a+1e-6) > 0.45)
    rows += [[0.5, 0.5, 0.2, 0.2, 0.9, 1.0], [0.5, 0.5, 0.2, 0.2, 0.9, 1.0]]
    # same box, different class -> both kept
    rows += [[0.5, 0.5, 0.2, 0.2, 0.9, 2.0]]
    # score exactly at threshold 0.5 -> filtered (strict >)
    rows += [[0.2, 0.2, 0.1, 0.1, 0.5, 0.0]]
    # next float above 0.5 -> kept
    rows += [[0.2, 0.2, 0.1, 0.1, float(np.nextafter(F32(0.5), F32(1.0))), 0.0]]
    # horizontally shifted pairs sweeping IoU through 0.45 in tiny steps
    for k in range(40):
        d = 0.0758 + 0.00002 * k          # IoU of two 0.2x0.2 boxes shifted by d: (0.2-d)/(0.2+d)
        rows += [[0.30, 0.80, 0.2, 0.2, 0.8 - 0.001 * k, 5.0 + k], [0.30 + d, 0.80, 0.2, 0.2, 0.7 - 0.001 * k, 5.0 + k]]
    # zero-area and negative-size boxes
    rows += [[0.7, 0.7, 0.0, 0.0, 0.95, 3.0], [0.7, 0.7, 0.0, 0.0, 0.94, 3.0], [0.7, 0.7, -0.1, 0.1, 0.93, 3.0]]
    # many equal scores, overlapping chain (tie order = input order)
    for k in range(30):
        rows += [[0.1 + 0.01 * k, 0.4, 0.05, 0.05, 0.6, 7.0]]
    return np.asarray(rows, np.float64).astype(F32)


NMS_CASES = {
    # name: (generator, kwargs, iou_thr, obj_thr, box_format)
    "empty": ("uniform", dict(n=0, nc=80, seed=1000), 0.45, 0.5, "center"),
    "one": ("uniform", dict(n=1, nc=80, seed=1001), 0.45, 0.5, "center"),
    "u64_nc80": ("uniform", dict(n=64, nc=80, seed=1002), 0.45, 0.5, "center"),
    "u65_nc2": ("uniform", dict(n=65, nc=2, seed=1003), 0.45, 0.5, "center"),
    "u1000_nc80": ("uniform", dict(n=1000, nc=80, seed=1004), 0.45, 0.5, "center"),
    "u1000_nc2": ("uniform", dict(n=1000, nc=2, seed=1005), 0.45, 0.5, "center"),
    "u1000_nc1_corners": ("uniform", dict(n=1000, nc=1, seed=1006), 0.3, 0.6, "corners"),
    "c2000_nc80": ("clustered", dict(n=2000, nc=80, seed=1007, jitter=0.3), 0.45, 0.5, "center"),
    "c3000_nc2_midpoint": ("clustered", dict(n=3000, nc=2, seed=1008, jitter=0.2), 0.5, 0.55, "midpoint"),
    "mixed1500_nc3": ("mixed", dict(n=1500, nc=3, seed=1009), 0.45, 0.5, "center"),
    "adversarial": ("adversarial", dict(), 0.45, 0.5, "center"),
    "u10000_nc80": ("uniform", dict(n=10000, nc=80, seed=1010), 0.45, 0.5, "center"),
    "u10000_nc2": ("uniform", dict(n=10000, nc=2, seed=1011), 0.45, 0.5, "center"),
    "c10000_nc80": ("clustered", dict(n=10000, nc=80, seed=1012, jitter=0.15), 0.45, 0.5, "center"),
}


def nms_boxes(case):
    gen, kw, iou_thr, obj_thr, fmt = NMS_CASES[case]
    if gen == "uniform":
        b = boxes_uniform(**kw)
    elif gen == "clustered":
        b = boxes_clustered(**kw)
    elif gen == "mixed":
        b = boxes_mixed(**kw)
    else:
        b = boxes_adversarial()
    return b, iou_thr, obj_thr, fmt


# -------------------------------------------------------------------- train-step case (G7)
TRAIN_CASE = dict(nc=2, size=96, batch=4, wseed=11, xseed=12, tseed=13, act="leaky_relu",
                  anchors=[[(0.215, 0.461), (0.992, 0.349), (0.436, 0.952)],
                           [(0.06, 0.143), (0.143, 0.189), (0.408, 0.181)],
                           [(0.016, 0.0349), (0.0408, 0.0598), (0.110, 0.0777)]])

# multi-step trajectory on TRAIN_CASE (gen_golden.py train_traj): the reference's loop body, TRAJ_STEPS iterations
TRAJ_STEPS = 3
TRAJ_OPT = dict(lr=1e-2, momentum=0.9, weight_decay=5e-4)
TRAJ_SCHED = dict(start_factor=0.1, total_iters=8)       # train.py:187-189: LinearLR warm-up, stepped after every batch
TRAJ_WEIGHT_KEYS = ("layers.0.conv.weight", "layers.29.pred_block.1.conv.weight")
TRAJ_PERTURB = 1e-6        # relative input perturbation of the reference's own conditioning runs (train_traj.npz */perturbed_totals)
# whole network with in_channels = 1 (gen_golden.py net_in1)
NET_IN1 = dict(nc=2, size=96, batch=2, act="leaky_relu", wseed=41, xseed=42, in_channels=1)


def synth_targets(batch, size, nc, anchors, seed, mean_boxes=7):
    """COCO-shaped synthetic targets in the dataset's tensor format
    (`/root/reference/code/dataset.py:119-167`): per scale (B,3,g,g,6) with
    [x_cell, y_cell, w_cells, h_cells, obj in {1,0,-1}, class]. Own restatement of the
    assignment rule: anchors ranked by width/height IoU (`utils.py:22-36`), the best free
    anchor of each scale takes the box, other anchors of that scale with IoU > 0.5 are
    marked ignore (-1)."""
    rng = np.random.Generator(np.random.PCG64(seed))
    grids = [size // 32, size // 16, size // 8]
    anc = np.asarray(anchors, np.float64).reshape(9, 2)
    out = [np.zeros((batch, 3, g, g, 6), F32) for g in grids]
    for b in range(batch):
        n = max(1, rng.poisson(mean_boxes))
        for _ in range(n):
            a = rng.integers(0, 9)
            w, h = np.clip(anc[a] * np.exp(0.25 * rng.standard_normal(2)), 0.01, 0.99)
            x, y = rng.uniform(0.02, 0.98, 2)
            c = rng.integers(0, nc)
            inter = np.minimum(anc[:, 0], w) * np.minimum(anc[:, 1], h)
            iou = inter / (anc[:, 0] * anc[:, 1] + w * h - inter)
            has = [False, False, False]
            for ai in np.argsort(-iou, kind="stable"):
                s, k = divmod(int(ai), 3)
                g = grids[s]
                i, j = int(g * y), int(g * x)
                taken = out[s][b, k, i, j, 0]      # dataset.py:141 tests element 0 (x), not obj
                if not taken and not has[s]:
                    out[s][b, k, i, j] = [g * x - j, g * y - i, w * g, h * g, 1.0, c]
                    has[s] = True
                elif not taken and iou[ai] > 0.5:
                    out[s][b, k, i, j, 4] = -1.0
    return out


# -------------------------------------------------------------------- target-builder cases (dataset.py:119-161)
COCO_ANCHORS = [[(0.28, 0.22), (0.38, 0.48), (0.9, 0.78)], [(0.07, 0.15), (0.15, 0.11), (0.14, 0.29)],
                [(0.02, 0.03), (0.04, 0.07), (0.08, 0.06)]]                       # config.ANCHORS
TARGET_CASES = {
    "s416_coco": dict(size=416, nc=80, batch=6, seed=401, anchors=COCO_ANCHORS, mean_boxes=9),
    "s320_turbine": dict(size=320, nc=2, batch=5, seed=402, anchors=TRAIN_CASE["anchors"], mean_boxes=4),
    "s608_crowded": dict(size=608, nc=80, batch=3, seed=403, anchors=COCO_ANCHORS, mean_boxes=60),
    "s96_collisions": dict(size=96, nc=3, batch=4, seed=404, anchors=COCO_ANCHORS, mean_boxes=25),
}


def target_boxes(case):
    """Seeded per-image box lists [[x, y, w, h, class], ...] with fp32-representable coordinates (the reference
    gets Python floats, the kernel fp32: identical values). Image 0 has no boxes; the last image carries boxes on exact
    cell corners (x offset 0: the reference's "taken" test, dataset.py:141, then sees a free cell) and duplicates."""
    c = TARGET_CASES[case]
    rng = np.random.Generator(np.random.PCG64(c["seed"]))
    anc = np.asarray(c["anchors"], np.float64).reshape(9, 2)
    out = []
    for b in range(c["batch"]):
        n = 0 if b == 0 else max(1, rng.poisson(c["mean_boxes"]))
        boxes = []
        for _ in range(n):
            a = rng.integers(0, 9)
            w, h = np.clip(anc[a] * np.exp(0.3 * rng.standard_normal(2)), 0.01, 0.95)
            x, y = rng.uniform(0.01, 0.99, 2)
            boxes.append([float(F32(x)), float(F32(y)), float(F32(w)), float(F32(h)), float(rng.integers(0, c["nc"]))])
        if b == c["batch"] - 1:
            extra = [[0.5, 0.5, 0.2, 0.3, 1.0], [0.5, 0.5, 0.21, 0.29, 0.0], [0.25, 0.75, 0.05, 0.06, 1.0], [0.25, 0.75, 0.05, 0.06, 1.0]]
            boxes += [[float(F32(v)) for v in r] for r in extra]
        out.append(boxes)
    return out


# -------------------------------------------------------------------- mAP cases (utils.py:193-274)
MAP_CASES = {
    "small": dict(images=4, nc=3, gt_mean=3, seed=501, dup=1, noise=4),
    "coco_like": dict(images=12, nc=20, gt_mean=6, seed=502, dup=2, noise=10),
    "two_class_dense": dict(images=6, nc=2, gt_mean=20, seed=503, dup=3, noise=30),
    "ties": dict(images=3, nc=2, gt_mean=4, seed=504, dup=2, noise=3, quantise=True),
}


def map_boxes(case):
    """Seeded (pred_boxes, true_boxes) row lists [img, cx, cy, w, h, obj, cls] with fp32-representable values:
    jittered copies of every ground truth (some above, some below IoU 0.5), duplicates, pure-noise detections,
    a class with ground truth but no detection and detections of a class without ground truth; `quantise` makes
    objectness ties (stable sort order matters)."""
    c = MAP_CASES[case]
    rng = np.random.Generator(np.random.PCG64(c["seed"]))
    f = lambda v: float(F32(v))
    preds, trues = [], []
    for img in range(c["images"]):
        n = max(1, rng.poisson(c["gt_mean"]))
        for _ in range(n):
            cls = int(rng.integers(0, c["nc"] - 1))                       # the last class never has ground truth
            cx, cy = rng.uniform(0.15, 0.85, 2)
            w, h = rng.uniform(0.05, 0.3, 2)
            trues.append([img, f(cx), f(cy), f(w), f(h), 1.0, cls])
            if cls == 0 and img == 0:
                continue                                                  # ground truth that nobody detects
            for _ in range(c["dup"]):
                jit = rng.choice([0.02, 0.08, 0.3])
                s = rng.uniform(0.3, 1.0)
                if c.get("quantise"):
                    s = round(s * 5) / 5
                preds.append([img, f(cx + jit * w * rng.standard_normal()), f(cy + jit * h * rng.standard_normal()),
                              f(w * np.exp(jit * rng.standard_normal())), f(h * np.exp(jit * rng.standard_normal())), f(s), cls])
        for _ in range(c["noise"]):
            preds.append([img, f(rng.uniform(0.1, 0.9)), f(rng.uniform(0.1, 0.9)), f(rng.uniform(0.05, 0.4)), f(rng.uniform(0.05, 0.4)),
                          f(rng.uniform(0.1, 0.9)), int(rng.integers(0, c["nc"]))])
    return preds, trues


# -------------------------------------------------------------------- check_model_accuracy cases (utils.py:334-381)
ACC_CASE = dict(nc=5, size=96, batches=3, batch=4, seed=601, thr=0.6)


def accuracy_batches():
    """Seeded loader content [(x, [t0,t1,t2]), ...] and the predictions a stub model returns for batch k. Objectness
    logits within 1e-4 of logit(thr) are pushed away so that no implementation's last-bit sigmoid decides a count."""
    c = ACC_CASE
    rng = np.random.Generator(np.random.PCG64(c["seed"]))
    S, nc = c["size"], c["nc"]
    edge = float(np.log(c["thr"] / (1 - c["thr"])))
    out = []
    for k in range(c["batches"]):
        tg = synth_targets(c["batch"], S, nc, TRAIN_CASE["anchors"], c["seed"] + 10 + k, mean_boxes=6)
        preds = []
        for t in tg:
            p = rng.standard_normal(t.shape[:4] + (5 + nc,), dtype=F32) * 1.5
            near = np.abs(p[..., 4] - edge) < 1e-4
            p[..., 4][near] += 1e-2
            obj = t[..., 4] == 1                      # make about 60 % of the object cells predict the right class
            hit = obj & (rng.random(obj.shape) < 0.6)
            cls = t[..., 5].astype(np.int64)
            idx = np.nonzero(hit)
            p[idx + (5 + cls[idx],)] += 6.0
            preds.append(p)
        out.append((np.zeros((c["batch"], 3, S, S), F32), tg, preds))
    return out


# -------------------------------------------------------------------- get_eval_boxes case (utils.py:276-332)
EVAL_CASE = dict(nc=4, size=64, batches=2, batch=2, seed=701, iou_thr=0.45, obj_thr=0.5, anchors=COCO_ANCHORS)


def eval_batches():
    """Loader content and the predictions of a stub model for get_eval_boxes: small grids (2, 4, 8) so that the reference's
    Python NMS stays fast; objectness logits pushed away from the threshold's logit."""
    c = EVAL_CASE
    rng = np.random.Generator(np.random.PCG64(c["seed"]))
    S, nc = c["size"], c["nc"]
    out = []
    for k in range(c["batches"]):
        tg = synth_targets(c["batch"], S, nc, c["anchors"], c["seed"] + 10 + k, mean_boxes=3)
        preds = []
        for t in tg:
            p = (rng.standard_normal(t.shape[:4] + (5 + nc,), dtype=F32) * 0.8).astype(F32)
            p[..., 4] -= 1.0                                             # ~25 % of the cells pass the objectness threshold
            near = np.abs(p[..., 4]) < 1e-3
            p[..., 4][near] += 1e-2
            preds.append(p)
        out.append((np.zeros((c["batch"], 3, S, S), F32), tg, preds))
    return out


# ------------------------------------------------------------------ checkpoint format
def checkpoint_setup(model_cls, nc=2):
    """Deterministic model + SGD state shared by the generator (reference model) and the tests (this package's model):
    synthetic weights, grad = 0.01 * param, one SGD step, so every parameter has a momentum buffer."""
    import torch
    from oracle import net as onet
    sd = onet.synth_state_dict(31, 3, nc, gain=1.0)
    m = model_cls(num_classes=nc)
    m.load_state_dict(sd)
    opt = torch.optim.SGD(m.parameters(), lr=1e-3, momentum=0.9, weight_decay=5e-4)     # train.py:171-172
    for p in m.parameters():
        p.grad = 0.01 * p.detach()
    opt.step()
    return m, opt
