// loss_fused.hip — per-scale YOLO loss, forward and gradient, as three small fused kernels.
//
// Same arithmetic as the reference's YOLOLoss.forward (code/loss.py:29-81) for one scale:
//   noobj (t4 == 0):  BCEWithLogits(p4, 0)                                   mean over the no-object cells
//   obj   (t4 == 1):  iou = calc_iou([sig(p0), sig(p1), exp(p2) aw, exp(p3) ah], t0..3)  ("center", detached)
//                     object = (p4 - iou)^2                                  mean over the object cells
//                     box    = MSE([p0, sig(p1), sig(p2), p3], [t0, t1, log(1e-16 + t2/aw), log(1e-16 + t3/ah)])
//                              (the reference overwrites indices 1:3 with their sigmoid, loss.py:71 — kept)
//                     class  = CrossEntropy(p5.., t5)
//   returns [5 box, 1 object, 0.5 noobj, 1 class]; cells with t4 == -1 are ignored.
// The reference implements this with boolean-mask gathers (data-dependent shapes, a host sync per mask and
// ~35 tiny launches per scale); here every cell is visited once with masks as predicates, sums go through
// a fixed-order fp64 tree (deterministic), and nothing depends on the host — the whole fine-tune step can be
// captured in a HIP graph. Unlike the reference it does NOT mutate its arguments.
#include "common.h"

namespace yolo {

struct LossArgs {
    const float* pred;          // (B,3,g,g,5+nc) through element strides
    const float* tgt;           // (B,3,g,g,6) contiguous
    const float* anchors;       // (3,2) in grid units
    long long sb, sa, sy, sx, sk;
    int B, g, nc;
    long long cells;
};

__device__ __forceinline__ float sigmoidf_(float x) { return 1.f / (1.f + expf(-x)); }

// IoU of two cxcywh boxes, utils.py:38-84 with box_format "center"
__device__ __forceinline__ float iou_center(float ax, float ay, float aw, float ah, float bx, float by, float bw, float bh) {
    const float ax1 = ax - aw / 2, ay1 = ay - ah / 2, bx1 = bx - bw / 2, by1 = by - bh / 2;
    float iw = fminf(ax1 + aw, bx1 + bw) - fmaxf(ax1, bx1);
    float ih = fminf(ay1 + ah, by1 + bh) - fmaxf(ay1, by1);
    iw = iw > 0.f ? iw : 0.f;
    ih = ih > 0.f ? ih : 0.f;
    const float inter = iw * ih;
    return inter / (aw * ah + bw * bh - inter + 1e-6f);
}

__device__ __forceinline__ const float* cell_ptr(const LossArgs& p, long long cell, int* a_out) {
    const int x = (int)(cell % p.g);
    const long long r1 = cell / p.g;
    const int y = (int)(r1 % p.g);
    const long long r2 = r1 / p.g;
    const int a = (int)(r2 % 3);
    const long long b = r2 / 3;
    *a_out = a;
    return p.pred + b * p.sb + a * p.sa + y * p.sy + x * p.sx;
}

// sums[blk][6] (double): box, obj, noobj, class, n_obj, n_noobj
__global__ __launch_bounds__(256) void loss_partial(const LossArgs p, double* __restrict__ partial) {
    double acc[6] = {0, 0, 0, 0, 0, 0};
    for (long long cell = blockIdx.x * 256LL + threadIdx.x; cell < p.cells; cell += (long long)gridDim.x * 256) {
        const float* t = p.tgt + cell * 6;
        const float t4 = t[4];
        if (t4 == 0.f) {
            int a;
            const float x = cell_ptr(p, cell, &a)[4 * p.sk];
            acc[2] += (double)(fmaxf(x, 0.f) + log1pf(expf(-fabsf(x))));
            acc[5] += 1.0;
        } else if (t4 == 1.f) {
            int a;
            const float* q = cell_ptr(p, cell, &a);
            const float aw = p.anchors[a * 2], ah = p.anchors[a * 2 + 1];
            const float p0 = q[0], p1 = q[p.sk], p2 = q[2 * p.sk], p3 = q[3 * p.sk], p4 = q[4 * p.sk];
            const float iou = iou_center(sigmoidf_(p0), sigmoidf_(p1), expf(p2) * aw, expf(p3) * ah, t[0], t[1], t[2], t[3]);
            const float d4 = p4 - iou * t4;
            acc[1] += (double)(d4 * d4);
            const float e0 = p0 - t[0], e1 = sigmoidf_(p1) - t[1];
            const float e2 = sigmoidf_(p2) - logf(1e-16f + t[2] / aw), e3 = p3 - logf(1e-16f + t[3] / ah);
            acc[0] += (double)(e0 * e0) + (double)(e1 * e1) + (double)(e2 * e2) + (double)(e3 * e3);
            float mx = -INFINITY;
            for (int k = 0; k < p.nc; ++k) mx = fmaxf(mx, q[(5 + k) * p.sk]);
            float se = 0.f;
            for (int k = 0; k < p.nc; ++k) se += expf(q[(5 + k) * p.sk] - mx);
            const int cls = (int)t[5];
            acc[3] += (double)(mx + logf(se) - q[(5 + cls) * p.sk]);
            acc[4] += 1.0;
        }
    }
    __shared__ double red[256];
#pragma unroll
    for (int k = 0; k < 6; ++k) {
        __syncthreads();
        red[threadIdx.x] = acc[k];
        __syncthreads();
        for (int s = 128; s > 0; s >>= 1) {                 // fixed tree: deterministic
            if ((int)threadIdx.x < s) red[threadIdx.x] += red[threadIdx.x + s];
            __syncthreads();
        }
        if (threadIdx.x == 0) partial[blockIdx.x * 6 + k] = red[0];
    }
}

__global__ __launch_bounds__(64) void loss_finalize(const double* __restrict__ partial, int nblk, float* __restrict__ losses4,
                                                    float* __restrict__ counts2) {
    // lane l adds blocks l, l+64, ... (independent loads), then the 64 lanes are added in a fixed tree: deterministic.
    // (One thread walking 1024 x 6 dependent loads took 85 us per scale.)
    __shared__ double red[6][64];
    const int l = threadIdx.x;
    double s[6] = {0, 0, 0, 0, 0, 0};
    for (int b = l; b < nblk; b += 64)
#pragma unroll
        for (int k = 0; k < 6; ++k) s[k] += partial[b * 6 + k];
#pragma unroll
    for (int k = 0; k < 6; ++k) red[k][l] = s[k];
    __syncthreads();
    for (int st = 32; st > 0; st >>= 1) {
        if (l < st)
#pragma unroll
            for (int k = 0; k < 6; ++k) red[k][l] += red[k][l + st];
        __syncthreads();
    }
    if (l != 0) return;
#pragma unroll
    for (int k = 0; k < 6; ++k) s[k] = red[k][0];
    const double n_obj = s[4], n_noobj = s[5];
    losses4[0] = n_obj > 0 ? (float)(5.0 * s[0] / (4.0 * n_obj)) : 0.f;      // loss.py:24-27,78-81
    losses4[1] = n_obj > 0 ? (float)(s[1] / n_obj) : 0.f;
    losses4[2] = (float)(0.5 * s[2] / n_noobj);                               // empty mean = NaN, like torch
    losses4[3] = n_obj > 0 ? (float)(s[3] / n_obj) : 0.f;
    counts2[0] = (float)n_obj;
    counts2[1] = (float)n_noobj;
}

// dpred (B,3,g,g,5+nc) contiguous = sum_k gout[k] * d loss_k / d pred
__global__ __launch_bounds__(256) void loss_grad(const LossArgs p, const float* __restrict__ counts2, const float* __restrict__ gout4,
                                                 float* __restrict__ dpred) {
    const float n_obj = counts2[0], n_noobj = counts2[1];
    const float g_box = gout4[0] * 5.f, g_obj = gout4[1], g_noobj = gout4[2] * 0.5f, g_cls = gout4[3];
    const int D = 5 + p.nc;
    for (long long cell = blockIdx.x * 256LL + threadIdx.x; cell < p.cells; cell += (long long)gridDim.x * 256) {
        const float* t = p.tgt + cell * 6;
        float* d = dpred + cell * D;
        const float t4 = t[4];
        if (t4 == 1.f) {
            int a;
            const float* q = cell_ptr(p, cell, &a);
            const float aw = p.anchors[a * 2], ah = p.anchors[a * 2 + 1];
            const float p0 = q[0], p1 = q[p.sk], p2 = q[2 * p.sk], p3 = q[3 * p.sk], p4 = q[4 * p.sk];
            const float s1 = sigmoidf_(p1), s2 = sigmoidf_(p2);
            const float iou = iou_center(sigmoidf_(p0), s1, expf(p2) * aw, expf(p3) * ah, t[0], t[1], t[2], t[3]);
            const float kb = g_box * 2.f / (4.f * n_obj);
            d[0] = kb * (p0 - t[0]);
            d[1] = kb * (s1 - t[1]) * s1 * (1.f - s1);
            d[2] = kb * (s2 - logf(1e-16f + t[2] / aw)) * s2 * (1.f - s2);
            d[3] = kb * (p3 - logf(1e-16f + t[3] / ah));
            d[4] = g_obj * 2.f * (p4 - iou * t4) / n_obj;
            float mx = -INFINITY;
            for (int k = 0; k < p.nc; ++k) mx = fmaxf(mx, q[(5 + k) * p.sk]);
            float se = 0.f;
            for (int k = 0; k < p.nc; ++k) se += expf(q[(5 + k) * p.sk] - mx);
            const int cls = (int)t[5];
            const float kc = g_cls / n_obj;
            for (int k = 0; k < p.nc; ++k) d[5 + k] = kc * (expf(q[(5 + k) * p.sk] - mx) / se - (k == cls ? 1.f : 0.f));
        } else {
            for (int k = 0; k < D; ++k) d[k] = 0.f;
            if (t4 == 0.f) {
                int a;
                const float x = cell_ptr(p, cell, &a)[4 * p.sk];
                d[4] = g_noobj * sigmoidf_(x) / n_noobj;
            }
        }
    }
}

static int loss_blocks(long long cells) {
    long long b = (cells + 255) / 256;
    return (int)(b < 1 ? 1 : (b > 1024 ? 1024 : b));
}

}  // namespace yolo

using namespace yolo;

extern "C" {

size_t yolo_loss_workspace_bytes(int b, int g) {
    if (b <= 0 || g <= 0) return 0;
    return (size_t)loss_blocks((long long)b * 3 * g * g) * 6 * sizeof(double);
}

static int fill_loss_args(LossArgs* a, const float* pred, const int64_t* s5, const float* target, const float* anchors, int b, int g, int nc) {
    if (!pred || !s5 || !target || !anchors || b <= 0 || g <= 0 || nc <= 0) return fail(YOLO_ERR_ARG, "loss: bad arguments");
    a->pred = pred; a->tgt = target; a->anchors = anchors;
    a->sb = s5[0]; a->sa = s5[1]; a->sy = s5[2]; a->sx = s5[3]; a->sk = s5[4];
    a->B = b; a->g = g; a->nc = nc; a->cells = (long long)b * 3 * g * g;
    return YOLO_OK;
}

int yolo_loss_fwd(const float* pred, const int64_t* strides5, const float* target, const float* anchors_3x2, int b, int g, int nc,
                  float* losses4, float* counts2, void* workspace, size_t workspace_bytes, void* stream) {
    LossArgs a;
    int rc = fill_loss_args(&a, pred, strides5, target, anchors_3x2, b, g, nc);
    if (rc) return rc;
    if (!losses4 || !counts2 || !workspace || workspace_bytes < yolo_loss_workspace_bytes(b, g)) return fail(YOLO_ERR_WORKSPACE, "loss: workspace");
    const int nblk = loss_blocks(a.cells);
    hipStream_t s = (hipStream_t)stream;
    hipLaunchKernelGGL(loss_partial, dim3(nblk), dim3(256), 0, s, a, (double*)workspace);
    rc = check_launch("loss_partial");
    if (rc) return rc;
    hipLaunchKernelGGL(loss_finalize, dim3(1), dim3(64), 0, s, (const double*)workspace, nblk, losses4, counts2);
    return check_launch("loss_finalize");
}

int yolo_loss_bwd(const float* pred, const int64_t* strides5, const float* target, const float* anchors_3x2, int b, int g, int nc,
                  const float* counts2, const float* grad_losses4, float* dpred, void* stream) {
    LossArgs a;
    int rc = fill_loss_args(&a, pred, strides5, target, anchors_3x2, b, g, nc);
    if (rc) return rc;
    if (!counts2 || !grad_losses4 || !dpred) return fail(YOLO_ERR_ARG, "loss_bwd: null pointer");
    hipLaunchKernelGGL(loss_grad, dim3(loss_blocks(a.cells)), dim3(256), 0, (hipStream_t)stream, a, counts2, grad_losses4, dpred);
    return check_launch("loss_grad");
}

}  // extern "C"
