"""ORACLE — TEST INFRASTRUCTURE ONLY. Never imported by the product package.  **PARITY UNPINNED.**

numpy restatement of the reference's inference pre-processing (`/root/reference/code/config.py:101-113`:
``A.LongestMaxSize`` -> ``A.PadIfNeeded(border_mode=BORDER_CONSTANT, value=0)`` -> ``A.Normalize(mean 0, std 1,
max_pixel_value 255)`` -> ``ToTensorV2``) and of the box mapping in ``plot_original`` (`utils.py:475-501`).
The arithmetic lives in two third-party packages that are NOT installed here and not vendored under
/root/reference: albumentations (requirements.txt pins 1.4.x) and opencv-python. This file restates their
published algorithms — albumentations: ``scale = max_size / max(h, w)``, new sizes by banker's rounding, pad
``floor(diff / 2)`` on top / left; OpenCV ``resize`` INTER_LINEAR for uint8: source coordinate
``(d + 0.5) * scale - 0.5``, coefficients ``cvRound(f * 2048)``, int32 horizontal pass, vertical pass
``(((b0 * (r0 >> 4)) >> 16) + ((b1 * (r1 >> 4)) >> 16) + 2) >> 2`` — but it could not be checked against them, so
the parity claim for this step is "matches this restatement bit for bit", nothing more.
"""
import numpy as np


def _coef(n_dst, n_src):
    scale = n_src / n_dst
    s0 = np.zeros(n_dst, np.int64); s1 = np.zeros(n_dst, np.int64)
    c0 = np.zeros(n_dst, np.int64); c1 = np.zeros(n_dst, np.int64)
    for d in range(n_dst):
        f = (d + 0.5) * scale - 0.5
        s = int(np.floor(f))
        f -= s
        if s < 0:
            s, f = 0, 0.0
        if s >= n_src - 1:
            s, f = n_src - 1, 0.0
        ff = np.float32(f)
        s0[d], s1[d] = s, min(s + 1, n_src - 1)
        c1[d] = min(int(np.rint(np.float64(ff * np.float32(2048)))), 32767)
        c0[d] = min(int(np.rint(np.float64((np.float32(1) - ff) * np.float32(2048)))), 32767)
    return s0, s1, c0, c1


def resize_linear_u8(img, nh, nw):
    h, w, _ = img.shape
    if (nh, nw) == (h, w):
        return img.copy()
    x0, x1, a0, a1 = _coef(nw, w)
    y0, y1, b0, b1 = _coef(nh, h)
    src = img.astype(np.int64)
    rows = src[:, x0, :] * a0[None, :, None] + src[:, x1, :] * a1[None, :, None]          # (h, nw, 3)
    r0, r1 = rows[y0], rows[y1]
    t = (((b0[:, None, None] * (r0 >> 4)) >> 16) + ((b1[:, None, None] * (r1 >> 4)) >> 16) + 2) >> 2
    return np.clip(t, 0, 255).astype(np.uint8)


def py3round(v):
    return int(np.rint(v))                                   # half to even


def letterbox(img, size):
    h, w, _ = img.shape
    scale = size / float(max(h, w))
    nh, nw = (py3round(h * scale), py3round(w * scale)) if scale != 1.0 else (h, w)
    nh, nw = max(nh, 1), max(nw, 1)
    res = resize_linear_u8(img, nh, nw)
    top, left = (size - nh) // 2, (size - nw) // 2
    canvas = np.zeros((size, size, 3), np.uint8)
    canvas[top:top + nh, left:left + nw] = res
    out = canvas.astype(np.float32) * np.float32(1.0 / 255.0)
    return np.ascontiguousarray(out.transpose(2, 0, 1)), (h, w, nh, nw, top, left)


def unletterbox_boxes(boxes, original_hw, resized_hw):
    o_h, o_w = original_hw
    r_h, r_w = resized_hw
    scale = min(r_w / o_w, r_h / o_h)
    nw, nh = int(o_w * scale), int(o_h * scale)
    pw, ph = (r_w - nw) // 2, (r_h - nh) // 2
    return [[(b[0] * r_w - pw) / nw, (b[1] * r_h - ph) / nh, (b[2] * r_w) / nw, (b[3] * r_h) / nh, b[4], b[5]] for b in boxes]
