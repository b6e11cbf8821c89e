// targets.hip — ground-truth tensor builder for the loss (one of the steps next to the hot path).
//
// Replaces the per-sample Python loop of YOLODataset.__getitem__ (reference: code/dataset.py:119-161) for a whole
// batch: for every ground-truth box (x, y, w, h, class; normalised), rank the 9 anchors by width/height IoU
// (utils.py:22-36), give the box to the best FREE anchor of each scale —
//     target[scale][anchor, i, j] = [g x - j, g y - i, w g, h g, 1, class],  i = int(g y), j = int(g x)
// — and mark the scale's other free anchors with IoU > 0.5 as ignored (objectness -1). Reference behaviours kept:
// "free" is tested on element 0 of the cell (x offset, dataset.py:141), so a cell whose stored x offset is exactly
// 0 still looks free; boxes are processed in list order; grid arithmetic is done in double like the Python floats
// and rounded to fp32 at the store; IoU and its ranking are fp32 (torch tensors).
// The assignment is sequential per image (every box sees the cells written by the previous ones) and independent
// across images: one thread per image; the data is a few KB per image, so this is latency-, not bandwidth-bound.
#include "common.h"

namespace yolo {

__global__ void build_targets_kernel(const float* __restrict__ boxes, const int* __restrict__ counts, int max_boxes,
                                     const float* __restrict__ anchors, int B, int S, float ignore_thr,
                                     float* __restrict__ t0, float* __restrict__ t1, float* __restrict__ t2) {
    const int b = blockIdx.x * blockDim.x + threadIdx.x;
    if (b >= B) return;
    const int g[3] = {S / 32, S / 16, S / 8};
    float* T[3] = {t0 + (size_t)b * 3 * g[0] * g[0] * 6, t1 + (size_t)b * 3 * g[1] * g[1] * 6, t2 + (size_t)b * 3 * g[2] * g[2] * 6};
    float aw[9], ah[9];
    for (int a = 0; a < 9; ++a) { aw[a] = anchors[2 * a]; ah[a] = anchors[2 * a + 1]; }
    int n = counts[b];
    if (n > max_boxes) n = max_boxes;
    for (int q = 0; q < n; ++q) {
        const float* bx = boxes + ((size_t)b * max_boxes + q) * 5;
        const float x = bx[0], y = bx[1], w = bx[2], h = bx[3];
        const int cls = (int)bx[4];
        float iou[9];
        int order[9];
        for (int a = 0; a < 9; ++a) {
            const float inter = fminf(w, aw[a]) * fminf(h, ah[a]);
            iou[a] = inter / (w * h + aw[a] * ah[a] - inter);
            int pos = a;                                        // stable insertion: descending IoU, ties keep anchor order
            while (pos > 0 && iou[order[pos - 1]] < iou[a]) { order[pos] = order[pos - 1]; --pos; }
            order[pos] = a;
        }
        bool has[3] = {false, false, false};
        for (int r = 0; r < 9; ++r) {
            const int ai = order[r], s = ai / 3, k = ai - 3 * s, gg = g[s];
            const double gx = (double)gg * (double)x, gy = (double)gg * (double)y;
            const int i = (int)gy, j = (int)gx;
            if ((unsigned)i >= (unsigned)gg || (unsigned)j >= (unsigned)gg) continue;     // x or y == 1.0: the reference would raise
            float* cell = T[s] + (((size_t)k * gg + i) * gg + j) * 6;
            const bool taken = cell[0] != 0.f;
            if (!taken && !has[s]) {
                cell[4] = 1.f;
                cell[5] = (float)cls;
                cell[0] = (float)(gx - j);
                cell[1] = (float)(gy - i);
                cell[2] = (float)((double)w * gg);
                cell[3] = (float)((double)h * gg);
                has[s] = true;
            } else if (!taken && iou[ai] > ignore_thr) {
                cell[4] = -1.f;
            }
        }
    }
}

}  // namespace yolo

using namespace yolo;

extern "C" {

int yolo_build_targets(const float* boxes, const int32_t* counts, int max_boxes, const float* anchors_9x2, int b, int image_size,
                       float ignore_iou, float* t0, float* t1, float* t2, void* stream) {
    if (!boxes || !counts || !anchors_9x2 || !t0 || !t1 || !t2 || b <= 0 || max_boxes <= 0 || image_size < 32 || image_size % 32)
        return fail(YOLO_ERR_ARG, "build_targets: bad arguments");
    hipStream_t s = (hipStream_t)stream;
    const int g[3] = {image_size / 32, image_size / 16, image_size / 8};
    float* t[3] = {t0, t1, t2};
    for (int k = 0; k < 3; ++k)
        if (hipMemsetAsync(t[k], 0, (size_t)b * 3 * g[k] * g[k] * 6 * sizeof(float), s) != hipSuccess) return fail(YOLO_ERR_LAUNCH, "build_targets: memset");
    hipLaunchKernelGGL(build_targets_kernel, dim3(ceil_div(b, 64)), dim3(64), 0, s, boxes, counts, max_boxes, anchors_9x2, b, image_size,
                       ignore_iou, t0, t1, t2);
    return check_launch("build_targets");
}

}  // extern "C"
