// metrics.hip — mean average precision on the device (the validation step the reference names as its
// bottleneck, README.md:32; next to the hot path: it consumes the kept boxes of NMS).
//
// Replaces calc_mAP (reference: code/utils.py:193-274), an O(D*G) Python loop that builds two fresh tensors per
// (detection, ground truth) pair. Same semantics:
//   per class with >= 1 ground truth: detections in descending-objectness order (stable); each detection looks at
//   the ground truths of ITS image and class in list order, takes the FIRST one with the largest IoU (> 0), and is a
//   true positive iff that IoU > threshold and the ground truth is still unassigned; precision/recall from fp32
//   cumulative sums, the point (recall 0, precision 1) prepended, trapezoid area; mAP = mean over those classes.
// Ordering (two stable sorts of 64-bit keys) is done by the caller; this file does the sequential-per-class matching
// (one wave per class: lanes evaluate the IoUs of one detection in parallel, a wave-wide first-argmax picks the
// ground truth) and the AP integration. IoU is calc_iou's arithmetic (utils.py:38-84), fp32, no FMA contraction
// (this file is built with -ffp-contract=off like postprocess.hip).
#include "common.h"

namespace yolo {

__device__ __forceinline__ float iou_boxes(const float* a, const float* b, int center) {
    float ax = a[0], ay = a[1], bx = b[0], by = b[1];
    const float aw = a[2], ah = a[3], bw = b[2], bh = b[3];
    if (center) { ax = ax - aw / 2; ay = ay - ah / 2; bx = bx - bw / 2; by = by - bh / 2; }
    const float xa = fmaxf(ax, bx), ya = fmaxf(ay, by);
    const float xb = fminf(ax + aw, bx + bw), yb = fminf(ay + ah, by + bh);
    float iw = xb - xa, ih = yb - ya;
    iw = iw > 0.f ? iw : 0.f;
    ih = ih > 0.f ? ih : 0.f;
    const float inter = iw * ih;
    const float uni = (aw * ah + bw * bh) - inter;
    return inter / (uni + 1e-6f);
}

// dets / gts: rows [img, x, y, w, h, obj, cls] already ordered (dets: class asc, objectness desc, stable;
// gts: class asc, image asc, stable). det_off / gt_off: [nc + 1] class boundaries. One 64-lane block per class.
__global__ __launch_bounds__(64) void map_match_kernel(const float* __restrict__ dets, const int* __restrict__ det_off,
                                                       const float* __restrict__ gts, const int* __restrict__ gt_off,
                                                       int* __restrict__ assigned, float* __restrict__ tp, float iou_thr, int center) {
    const int c = blockIdx.x, lane = threadIdx.x;
    const int d0 = det_off[c], d1 = det_off[c + 1], g0 = gt_off[c], g1 = gt_off[c + 1];
    if (g1 == g0) return;                                  // class without ground truth: skipped by the reference
    for (int d = d0; d < d1; ++d) {
        const float* det = dets + (size_t)d * 7;
        const float img = det[0];
        // ground truths of this image inside the class range (sorted by image): lower / upper bound
        int lo = g0, hi = g1;
        while (lo < hi) { const int mid = (lo + hi) >> 1; if (gts[(size_t)mid * 7] < img) lo = mid + 1; else hi = mid; }
        const int s = lo;
        hi = g1;
        while (lo < hi) { const int mid = (lo + hi) >> 1; if (gts[(size_t)mid * 7] <= img) lo = mid + 1; else hi = mid; }
        const int e = lo;
        float best = 0.f;                                  // `best_iou = 0`, updated on strict `>`
        int best_idx = 0;
        for (int base = s; base < e; base += 64) {
            const int gi = base + lane;
            float v = -1.f;
            if (gi < e) v = iou_boxes(det + 1, gts + (size_t)gi * 7 + 1, center);
            // wave max, then the lowest index holding it (first maximum)
            float m = v;
            for (int sft = 32; sft > 0; sft >>= 1) m = fmaxf(m, __shfl_xor(m, sft));
            if (m > best) {
                const unsigned long long who = __ballot(v == m);
                best = m;
                best_idx = base - s + (__ffsll((long long)who) - 1);
            }
        }
        if (lane == 0) {
            float flag = 0.f;
            if (best > iou_thr) {
                if (assigned[s + best_idx] == 0) { flag = 1.f; assigned[s + best_idx] = 1; }
            }
            tp[d] = flag;
        }
        __syncthreads();                                   // the next detection must see the assignment
    }
}

// one block per class: AP = trapz(precisions, recalls) with (0, 1) prepended; ap[c] = -1 for classes without GT
__global__ __launch_bounds__(256) void map_ap_kernel(const float* __restrict__ tp, const int* __restrict__ det_off,
                                                     const int* __restrict__ gt_off, float* __restrict__ ap) {
    __shared__ float s_tp[256];
    __shared__ double s_area[256];
    __shared__ float carry_tp;
    __shared__ float prev_p, prev_r;
    const int c = blockIdx.x, t = threadIdx.x;
    const int d0 = det_off[c], d1 = det_off[c + 1];
    const int ng = gt_off[c + 1] - gt_off[c];
    if (ng == 0) { if (t == 0) ap[c] = -1.f; return; }
    if (t == 0) { carry_tp = 0.f; prev_p = 1.f; prev_r = 0.f; }
    double area = 0.0;
    __syncthreads();
    for (int base = d0; base < d1; base += 256) {
        const int d = base + t;
        s_tp[t] = d < d1 ? tp[d] : 0.f;
        __syncthreads();
        for (int off = 1; off < 256; off <<= 1) {          // inclusive scan (exact: small integers in fp32)
            const float v = t >= off ? s_tp[t - off] : 0.f;
            __syncthreads();
            s_tp[t] += v;
            __syncthreads();
        }
        const float ctp = carry_tp + s_tp[t];
        const float n_seen = (float)(d - d0 + 1);          // cum_TP + cum_FP
        const float p = ctp / n_seen, r = ctp / (float)ng;
        // trapezoid with the previous point
        float pp, pr;
        if (t == 0) { pp = prev_p; pr = prev_r; }
        else {
            const float ctp_prev = carry_tp + s_tp[t - 1];
            pp = ctp_prev / (n_seen - 1.f);
            pr = ctp_prev / (float)ng;
        }
        s_area[t] = d < d1 ? (double)((r - pr) * (p + pp)) : 0.0;
        __syncthreads();
        for (int s = 128; s > 0; s >>= 1) {
            if (t < s) s_area[t] += s_area[t + s];
            __syncthreads();
        }
        if (t == 0) area += s_area[0];
        __syncthreads();
        if (t == 255 || d == d1 - 1) {
            if (d < d1) { prev_p = p; prev_r = r; }
        }
        __syncthreads();
        if (t == 0) carry_tp += s_tp[255];
        __syncthreads();
    }
    if (t == 0) ap[c] = (float)(area / 2.0);
}

// check_model_accuracy (utils.py:334-381) for one scale of one batch: five integer counters, accumulated with integer
// atomics (order-independent, so still deterministic): [class correct, n_obj, obj correct, noobj correct, n_noobj].
__global__ __launch_bounds__(256) void accuracy_kernel(const float* __restrict__ pred, long long sb, long long sa, long long sy, long long sx,
                                                       long long sk, const float* __restrict__ tgt, long long cells, int g, int nc,
                                                       float thr, unsigned long long* __restrict__ counts) {
    unsigned c_cls = 0, c_nobj = 0, c_obj = 0, c_noobj = 0, c_nnoobj = 0;
    for (long long cell = blockIdx.x * 256LL + threadIdx.x; cell < cells; cell += (long long)gridDim.x * 256) {
        const float t4 = tgt[cell * 6 + 4];
        if (t4 != 1.f && t4 != 0.f) continue;
        const int x = (int)(cell % g);
        const long long r1 = cell / g;
        const int y = (int)(r1 % g);
        const long long r2 = r1 / g;
        const float* q = pred + (r2 / 3) * sb + (r2 % 3) * sa + y * sy + x * sx;
        const bool obj_pred = 1.f / (1.f + expf(-q[4 * sk])) > thr;
        if (t4 == 1.f) {
            int best = 0;
            float bv = q[5 * sk];
            for (int k = 1; k < nc; ++k) {                   // first maximum; NaN counts as maximum (torch.argmax)
                const float v = q[(5 + k) * sk];
                if (v > bv || (v != v && bv == bv)) { bv = v; best = k; }
            }
            c_cls += (float)best == tgt[cell * 6 + 5];
            c_nobj += 1;
            c_obj += obj_pred;
        } else {
            c_noobj += !obj_pred;
            c_nnoobj += 1;
        }
    }
    unsigned v[5] = {c_cls, c_nobj, c_obj, c_noobj, c_nnoobj};
#pragma unroll
    for (int k = 0; k < 5; ++k) {
        unsigned s = v[k];
        for (int sft = 32; sft > 0; sft >>= 1) s += __shfl_xor(s, sft);
        if ((threadIdx.x & 63) == 0 && s) atomicAdd(counts + k, (unsigned long long)s);
    }
}

}  // namespace yolo

using namespace yolo;

extern "C" {

int yolo_map_match(const float* dets_sorted, const int32_t* det_class_offsets, const float* gts_sorted, const int32_t* gt_class_offsets,
                   int num_classes, int n_gt, float iou_threshold, int center, int32_t* assigned, float* tp_flags, float* ap_per_class,
                   void* stream) {
    if (!det_class_offsets || !gt_class_offsets || !ap_per_class || num_classes <= 0) return fail(YOLO_ERR_ARG, "map: bad arguments");
    hipStream_t s = (hipStream_t)stream;
    if (n_gt > 0) {
        if (!assigned || !gts_sorted) return fail(YOLO_ERR_ARG, "map: null pointer");
        if (hipMemsetAsync(assigned, 0, sizeof(int32_t) * (size_t)n_gt, s) != hipSuccess) return fail(YOLO_ERR_LAUNCH, "map: memset");
    }
    hipLaunchKernelGGL(map_match_kernel, dim3(num_classes), dim3(64), 0, s, dets_sorted, det_class_offsets, gts_sorted, gt_class_offsets,
                       assigned, tp_flags, iou_threshold, center);
    int rc = check_launch("map_match");
    if (rc) return rc;
    hipLaunchKernelGGL(map_ap_kernel, dim3(num_classes), dim3(256), 0, s, tp_flags, det_class_offsets, gt_class_offsets, ap_per_class);
    return check_launch("map_ap");
}

int yolo_accuracy_counts(const float* pred, const int64_t* strides5, const float* target, int b, int g, int nc, float obj_threshold,
                         unsigned long long* counts5, void* stream) {
    if (!pred || !strides5 || !target || !counts5 || b <= 0 || g <= 0 || nc <= 0) return fail(YOLO_ERR_ARG, "accuracy: bad arguments");
    const long long cells = (long long)b * 3 * g * g;
    long long nb = (cells + 255) / 256;
    if (nb > 2048) nb = 2048;
    hipLaunchKernelGGL(accuracy_kernel, dim3((unsigned)nb), dim3(256), 0, (hipStream_t)stream, pred, (long long)strides5[0],
                       (long long)strides5[1], (long long)strides5[2], (long long)strides5[3], (long long)strides5[4], target, cells, g, nc,
                       obj_threshold, counts5);
    return check_launch("accuracy");
}

}  // extern "C"
