// optim.hip — the SGD update of the fine-tune step as ONE launch over every parameter.
//
// Replaces `optimizer.step()` of `torch.optim.SGD(model.parameters(), lr, momentum, weight_decay)` (reference:
// code/train.py:171-172 constructs it, :68 steps it through the GradScaler). PyTorch's default (foreach) implementation
// walks the 222 parameter tensors of the network in four multi-tensor passes (weight decay, momentum scale, momentum add,
// parameter update): ~19 launches and 0.66 ms of the 18.7 ms bf16 step at batch 32. Here a block owns a 4,096-element chunk
// of one tensor and does the whole update in registers: p, g and the momentum buffer are read once, p and the buffer written
// once (62 M parameters x 20 bytes = 1.2 GB per step).
//
// Same bits as PyTorch: each of its passes computes `a + alpha * b` with ONE rounding (a fused multiply-add in its
// elementwise functor), so the chain below is fmaf / mul in exactly that order:
//   g' = fma(wd, p, g)                     grad.add(param, alpha=weight_decay)
//   b  = first ? g' : fma(1 - dampening, g', b * momentum)      buf.mul_(momentum).add_(grad, alpha=1 - dampening)
//   g" = nesterov ? fma(momentum, b, g') : b
//   p  = fma(-lr, g", p)                   param.add_(grad, alpha=-lr)
#include "common.h"

namespace yolo {

struct SgdItem { float* p; const float* g; float* buf; long long n; };   // n < 0: the momentum buffer is new (first step): buf = g'
static_assert(sizeof(SgdItem) == 32, "matches yolo_sgd_item in the header");

constexpr int SGD_CHUNK = 4096;          // elements per block: 256 threads x 4 x float4

template <bool VEC>
__device__ __forceinline__ void sgd_update(float* __restrict__ p, const float* __restrict__ g, float* __restrict__ b, long long i0,
                                           int cnt, bool first, float neg_lr, float momentum, float omd, float wd, int nesterov,
                                           int maximize) {
    auto one = [&](float pv, float gv, float bv, float& pn, float& bn) {
        if (maximize) gv = -gv;
        if (wd != 0.f) gv = __builtin_fmaf(wd, pv, gv);
        if (momentum != 0.f) {
            bn = first ? gv : __builtin_fmaf(omd, gv, bv * momentum);
            gv = nesterov ? __builtin_fmaf(momentum, bn, gv) : bn;
        } else {
            bn = bv;
        }
        pn = __builtin_fmaf(neg_lr, gv, pv);
    };
    if (VEC) {
        const f32x4 pv = *reinterpret_cast<const f32x4*>(p + i0), gv = *reinterpret_cast<const f32x4*>(g + i0);
        f32x4 bv = {0.f, 0.f, 0.f, 0.f};
        if (momentum != 0.f && !first) bv = *reinterpret_cast<const f32x4*>(b + i0);
        f32x4 pn, bn;
#pragma unroll
        for (int e = 0; e < 4; ++e) { float a, c; one(pv[e], gv[e], bv[e], a, c); pn[e] = a; bn[e] = c; }
        *reinterpret_cast<f32x4*>(p + i0) = pn;
        if (momentum != 0.f) *reinterpret_cast<f32x4*>(b + i0) = bn;
    } else {
        for (int e = 0; e < cnt; ++e) {
            float pn, bn;
            one(p[i0 + e], g[i0 + e], (momentum != 0.f && !first) ? b[i0 + e] : 0.f, pn, bn);
            p[i0 + e] = pn;
            if (momentum != 0.f) b[i0 + e] = bn;
        }
    }
}

// chunks[c] = (item index, first element): built once per optimizer (sizes never change), the items every step.
// hyper != nullptr: {lr, momentum, dampening, weight_decay} are read from device memory at EXECUTION time, so a captured
// launch follows a learning-rate schedule (train.py:71-74 steps a LinearLR warm-up after every batch) instead of replaying
// the values of the capture; the by-value arguments are then ignored.
__global__ __launch_bounds__(256) void sgd_step_kernel(const SgdItem* __restrict__ items, const int2* __restrict__ chunks,
                                                       const float* __restrict__ hyper, float neg_lr, float momentum, float omd, float wd,
                                                       int nesterov, int maximize) {
    if (hyper != nullptr) {
        const f32x4 h = *reinterpret_cast<const f32x4*>(hyper);
        neg_lr = -h[0]; momentum = h[1]; omd = 1.0f - h[2]; wd = h[3];
    }
    const int2 ck = chunks[blockIdx.x];
    const SgdItem it = items[ck.x];
    if (it.g == nullptr) return;                         // parameter without a gradient this step: skipped, like PyTorch
    const bool first = it.n < 0;
    const long long n = first ? -it.n : it.n;
    const bool aligned = ((((size_t)it.p) | ((size_t)it.g) | ((size_t)it.buf)) & 15) == 0;
#pragma unroll
    for (int r = 0; r < SGD_CHUNK / 1024; ++r) {
        const long long i0 = (long long)ck.y + r * 1024 + threadIdx.x * 4;
        if (i0 >= n) break;
        const int cnt = n - i0 < 4 ? (int)(n - i0) : 4;
        if (aligned && cnt == 4) sgd_update<true>(it.p, it.g, it.buf, i0, 4, first, neg_lr, momentum, omd, wd, nesterov, maximize);
        else sgd_update<false>(it.p, it.g, it.buf, i0, cnt, first, neg_lr, momentum, omd, wd, nesterov, maximize);
    }
}

}  // namespace yolo

using namespace yolo;

extern "C" {

int yolo_sgd_chunk_elems(void) { return SGD_CHUNK; }

int yolo_sgd_step(const void* items_dev, const int32_t* chunks_dev, int n_chunks, float lr, float momentum, float dampening,
                  float weight_decay, int nesterov, int maximize, void* stream) {
    if (n_chunks == 0) return YOLO_OK;
    if (!items_dev || !chunks_dev || n_chunks < 0) return fail(YOLO_ERR_ARG, "sgd_step: bad arguments");
    if (nesterov && (momentum <= 0.f || dampening != 0.f)) return fail(YOLO_ERR_ARG, "sgd_step: nesterov needs momentum > 0 and dampening = 0");
    hipLaunchKernelGGL(sgd_step_kernel, dim3((unsigned)n_chunks), dim3(256), 0, (hipStream_t)stream, (const SgdItem*)items_dev,
                       (const int2*)chunks_dev, (const float*)nullptr, -lr, momentum, 1.0f - dampening, weight_decay, nesterov, maximize);
    return check_launch("sgd_step");
}

int yolo_sgd_step_hp(const void* items_dev, const int32_t* chunks_dev, int n_chunks, const float* hyper4_dev, int nesterov, int maximize,
                     void* stream) {
    if (n_chunks == 0) return YOLO_OK;
    if (!items_dev || !chunks_dev || !hyper4_dev || n_chunks < 0 || ((size_t)hyper4_dev & 15)) return fail(YOLO_ERR_ARG, "sgd_step_hp: bad arguments");
    hipLaunchKernelGGL(sgd_step_kernel, dim3((unsigned)n_chunks), dim3(256), 0, (hipStream_t)stream, (const SgdItem*)items_dev,
                       (const int2*)chunks_dev, hyper4_dev, 0.f, 0.f, 1.f, 0.f, nesterov, maximize);
    return check_launch("sgd_step_hp");
}

}  // extern "C"
