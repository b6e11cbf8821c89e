// calls.hip — launch tables for the train-mode forward and the backward.
//
// The reference's loop is eager (code/train.py:41-82): forward, three losses, backward, optimizer step, one Python statement at
// a time. Here that is ~900 launches per step, and issued from Python through ctypes one by one the host needs as long to
// enqueue them (17.7 ms) as the GPU needs to run them (17.3 ms): the eager step was host-bound. The inference forward has had a
// C-side table since round 1 (yolo_conv_fwd_batch); this is the same idea for the training entry points, in the most general
// form: a table of recorded calls (function id + its arguments as 64-bit words) that ONE call replays in order on a stream,
// with relocations for the handful of pointers that change from step to step (the input batch, the freshly allocated
// prediction tensors, the upstream gradients). The host side (train_engine.CallTape) records a table the first time a
// (batch, size, dtype) plan runs by executing the ordinary per-launch path once, and replays it afterwards.
// No device work of its own: every entry dispatches to the exported function of the same name.
#include "common.h"

using namespace yolo;

namespace {

// memset(0) / device-to-device copy as KERNELS: hipMemsetAsync / hipMemcpyAsync go through the runtime's blit path, which in
// a rocprofv3 trace of the eager step started 85-90 us after the previous kernel had ended, every time (11 of them per step
// = 0.9 ms of idle GPU); an ordinary launch starts back to back.
typedef unsigned int u32x4k __attribute__((ext_vector_type(4)));
__global__ __launch_bounds__(256) void fill_zero_kernel(unsigned char* __restrict__ p, size_t bytes) {
    const size_t n16 = bytes >> 4;                           // p is 16-byte aligned (checked on the host), tail by bytes
    const u32x4k z = {0u, 0u, 0u, 0u};
    for (size_t i = blockIdx.x * 256ull + threadIdx.x; i < n16; i += (size_t)gridDim.x * 256) reinterpret_cast<u32x4k*>(p)[i] = z;
    if (blockIdx.x == 0 && threadIdx.x < (bytes & 15)) p[(n16 << 4) + threadIdx.x] = 0;
}
__global__ __launch_bounds__(256) void copy_kernel(unsigned char* __restrict__ d, const unsigned char* __restrict__ s, size_t bytes, int vec) {
    if (vec) {
        const size_t n16 = bytes >> 4;
        for (size_t i = blockIdx.x * 256ull + threadIdx.x; i < n16; i += (size_t)gridDim.x * 256)
            reinterpret_cast<u32x4k*>(d)[i] = reinterpret_cast<const u32x4k*>(s)[i];
        if (blockIdx.x == 0 && threadIdx.x < (bytes & 15)) d[(n16 << 4) + threadIdx.x] = s[(n16 << 4) + threadIdx.x];
    } else {
        for (size_t i = blockIdx.x * 256ull + threadIdx.x; i < bytes; i += (size_t)gridDim.x * 256) d[i] = s[i];
    }
}
inline unsigned blocks_for(size_t bytes) {
    const size_t b = (bytes / 16 + 255) / 256;
    return (unsigned)(b < 1 ? 1 : (b > 2048 ? 2048 : b));
}

inline float f_of(uint64_t bits) { double d; memcpy(&d, &bits, 8); return (float)d; }   // floats travel as the bits of a double
template <typename T> inline T* p_of(uint64_t v) { return reinterpret_cast<T*>(v); }

int run_one(int fn, const uint64_t* a, void* s) {
    switch (fn) {
    case YOLO_FN_FILL_ZERO:
        return yolo_fill_zero(p_of<void>(a[0]), (size_t)a[1], s);
    case YOLO_FN_COPY_D2D:
        return yolo_copy_d2d(p_of<void>(a[0]), p_of<const void>(a[1]), (size_t)a[2], s);
    case YOLO_FN_NCHW_TO_NHWC:
        return yolo_nchw_to_nhwc(p_of<const float>(a[0]), p_of<void>(a[1]), (int)a[2], (int)a[3], (int)a[4], (int)a[5], (int)a[6],
                                 (int)a[7], p_of<int32_t>(a[8]), s);
    case YOLO_FN_STEM_FWD:
        return yolo_stem_fwd(p_of<const float>(a[0]), p_of<const float>(a[1]), p_of<const float>(a[2]), p_of<const float>(a[3]),
                             p_of<void>(a[4]), (int)a[5], (int)a[6], (int)a[7], (int)a[8], (int)a[9], (int)a[10], (int)a[11], (int)a[12],
                             p_of<int32_t>(a[13]), s);
    case YOLO_FN_CONV_FWD:
        return yolo_conv_fwd(p_of<const yolo_conv_desc>(a[0]), p_of<const void>(a[1]), p_of<const void>(a[2]), p_of<const float>(a[3]),
                             p_of<const float>(a[4]), p_of<const void>(a[5]), p_of<void>(a[6]), p_of<int32_t>(a[7]), s);
    case YOLO_FN_CONV_FWD_WS:
        return yolo_conv_fwd_ws(p_of<const yolo_conv_desc>(a[0]), p_of<const void>(a[1]), p_of<const void>(a[2]), p_of<const float>(a[3]),
                                p_of<const float>(a[4]), p_of<const void>(a[5]), p_of<void>(a[6]), p_of<void>(a[7]), (size_t)a[8],
                                p_of<int32_t>(a[9]), s);
    case YOLO_FN_BN_STATS:
        return yolo_bn_stats(p_of<const void>(a[0]), (int)a[1], (int)a[2], (int)a[3], (int)a[4], p_of<const float>(a[5]),
                             p_of<const float>(a[6]), f_of(a[7]), f_of(a[8]), p_of<float>(a[9]), p_of<float>(a[10]), p_of<float>(a[11]),
                             p_of<float>(a[12]), p_of<float>(a[13]), p_of<float>(a[14]), (int)a[15], p_of<void>(a[16]), (size_t)a[17], s);
    case YOLO_FN_BN_ACT_FWD:
        return yolo_bn_act_fwd(p_of<const void>(a[0]), (int)a[1], (int)a[2], p_of<const float>(a[3]), p_of<const float>(a[4]),
                               p_of<const float>(a[5]), p_of<const void>(a[6]), (int)a[7], (int)a[8], p_of<void>(a[9]), (int)a[10],
                               (int)a[11], (int)a[12], (int)a[13], (int)a[14], (int)a[15], (int)a[16], (int)a[17], (int)a[18],
                               p_of<int32_t>(a[19]), s);
    case YOLO_FN_BN_ACT_BWD:
        return yolo_bn_act_bwd(p_of<const void>(a[0]), (int)a[1], (int)a[2], p_of<const void>(a[3]), (int)a[4], (int)a[5],
                               p_of<const float>(a[6]), p_of<const float>(a[7]), p_of<const float>(a[8]), p_of<const float>(a[9]),
                               p_of<const float>(a[10]), (int)a[11], (int)a[12], (int)a[13], p_of<float>(a[14]), p_of<float>(a[15]),
                               p_of<void>(a[16]), (int)a[17], (int)a[18], (int)a[19], p_of<void>(a[20]), (size_t)a[21], s);
    case YOLO_FN_UPSAMPLE2X_BWD:
        return yolo_upsample2x_bwd(p_of<const void>(a[0]), (int)a[1], (int)a[2], p_of<void>(a[3]), (int)a[4], (int)a[5], (int)a[6],
                                   (int)a[7], (int)a[8], (int)a[9], (int)a[10], s);
    case YOLO_FN_CONV_WGRAD:
        return yolo_conv_wgrad(p_of<const void>(a[0]), (int)a[1], (int)a[2], p_of<const void>(a[3]), (int)a[4], (int)a[5],
                               p_of<float>(a[6]), (int)a[7], (int)a[8], (int)a[9], (int)a[10], (int)a[11], (int)a[12], (int)a[13],
                               (int)a[14], p_of<void>(a[15]), (size_t)a[16], s);
    case YOLO_FN_PACK_WEIGHTS_DGRAD:
        return yolo_pack_weights_dgrad(p_of<const float>(a[0]), p_of<void>(a[1]), (int)a[2], (int)a[3], (int)a[4], (int)a[5], (int)a[6], s);
    case YOLO_FN_PACK_WEIGHTS_BATCH:
        return yolo_pack_weights_batch(p_of<const yolo_pack_item>(a[0]), (int)a[1], (int)a[2], (int)a[3], s);
    case YOLO_FN_CONV_DGRAD_S2:
        return yolo_conv_dgrad_s2(p_of<const void>(a[0]), (int)a[1], (int)a[2], p_of<const void>(a[3]), p_of<const void>(a[4]), (int)a[5],
                                  (int)a[6], p_of<void>(a[7]), (int)a[8], (int)a[9], (int)a[10], (int)a[11], (int)a[12], (int)a[13],
                                  (int)a[14], (int)a[15], s);
    case YOLO_FN_HEAD_GRAD_TO_NHWC:
        return yolo_head_grad_to_nhwc(p_of<const float>(a[0]), p_of<const int64_t>(a[1]), p_of<void>(a[2]), (int)a[3], (int)a[4],
                                      (int)a[5], (int)a[6], (int)a[7], s);
    case YOLO_FN_CONV_FWD_STATS:
        return yolo_conv_fwd_stats(p_of<const yolo_conv_desc>(a[0]), p_of<const void>(a[1]), p_of<const void>(a[2]), p_of<void>(a[3]),
                                   p_of<float>(a[4]), (size_t)a[5], s);
    case YOLO_FN_BN_STATS_FROM_PARTIALS:
        return yolo_bn_stats_from_partials(p_of<const float>(a[0]), (int)a[1], (int)a[2], (int)a[3], (int)a[4], p_of<const float>(a[5]),
                                           p_of<const float>(a[6]), f_of(a[7]), f_of(a[8]), p_of<float>(a[9]), p_of<float>(a[10]),
                                           p_of<float>(a[11]), p_of<float>(a[12]), p_of<float>(a[13]), p_of<float>(a[14]), s);
    case YOLO_FN_CONV_DGRAD_BSTATS:
        return yolo_conv_dgrad_bstats(p_of<const yolo_conv_desc>(a[0]), p_of<const void>(a[1]), p_of<const void>(a[2]), p_of<const void>(a[3]),
                                      p_of<void>(a[4]), p_of<const void>(a[5]), (int)a[6], (int)a[7], p_of<const float>(a[8]),
                                      p_of<const float>(a[9]), p_of<const float>(a[10]), (int)a[11], p_of<float>(a[12]), (size_t)a[13], s);
    case YOLO_FN_BN_ACT_BWD_ROWS:
        return yolo_bn_act_bwd_rows(p_of<const void>(a[0]), (int)a[1], (int)a[2], p_of<const void>(a[3]), (int)a[4], (int)a[5],
                                    p_of<const float>(a[6]), p_of<const float>(a[7]), p_of<const float>(a[8]), p_of<const float>(a[9]),
                                    p_of<const float>(a[10]), (int)a[11], (int)a[12], (int)a[13], p_of<float>(a[14]), p_of<float>(a[15]),
                                    p_of<void>(a[16]), (int)a[17], (int)a[18], (int)a[19], p_of<float>(a[20]), (int)a[21], (int)a[22], s);
    default:
        return fail(YOLO_ERR_ARG, "run_calls: unknown function id %d", fn);
    }
}

}  // namespace

extern "C" {

int yolo_fill_zero(void* p, size_t bytes, void* stream) {
    if (bytes == 0) return YOLO_OK;
    if (!p) return fail(YOLO_ERR_ARG, "fill_zero: null pointer");
    if ((size_t)p & 15) {                                   // rare: let the runtime deal with an unaligned start
        if (hipMemsetAsync(p, 0, bytes, (hipStream_t)stream) != hipSuccess) return check_launch("fill_zero");
        return YOLO_OK;
    }
    hipLaunchKernelGGL(fill_zero_kernel, dim3(blocks_for(bytes)), dim3(256), 0, (hipStream_t)stream, (unsigned char*)p, bytes);
    return check_launch("fill_zero");
}

int yolo_copy_d2d(void* dst, const void* src, size_t bytes, void* stream) {
    if (bytes == 0) return YOLO_OK;
    if (!dst || !src) return fail(YOLO_ERR_ARG, "copy_d2d: null pointer");
    const int vec = ((((size_t)dst) | ((size_t)src)) & 15) == 0;
    hipLaunchKernelGGL(copy_kernel, dim3(blocks_for(vec ? bytes : bytes * 16)), dim3(256), 0, (hipStream_t)stream, (unsigned char*)dst,
                       (const unsigned char*)src, bytes, vec);
    return check_launch("copy_d2d");
}

int yolo_run_calls(const yolo_call* calls, int n_calls, const yolo_reloc* relocs, int n_relocs, const uint64_t* slots, int n_slots,
                   void* stream) {
    if (n_calls < 0 || n_relocs < 0 || (n_calls && !calls) || (n_relocs && (!relocs || !slots)))
        return fail(YOLO_ERR_ARG, "run_calls: bad arguments");
    int r = 0;                                              // relocations are sorted by call index
    for (int i = 0; i < n_calls; ++i) {
        const yolo_call& c = calls[i];
        if (r < n_relocs && relocs[r].call == i) {
            uint64_t a[YOLO_CALL_MAX_ARGS];
            memcpy(a, c.a, sizeof(a));
            for (; r < n_relocs && relocs[r].call == i; ++r) {
                const yolo_reloc& q = relocs[r];
                if (q.arg < 0 || q.arg >= YOLO_CALL_MAX_ARGS || q.slot < 0 || q.slot >= n_slots)
                    return fail(YOLO_ERR_ARG, "run_calls: relocation %d out of range", r);
                a[q.arg] = slots[q.slot] + (uint64_t)q.offset;
            }
            if (int rc = run_one(c.fn, a, stream)) return rc;
        } else {
            if (r < n_relocs && relocs[r].call < i) return fail(YOLO_ERR_ARG, "run_calls: relocations are not sorted by call");
            if (int rc = run_one(c.fn, c.a, stream)) return rc;
        }
    }
    return YOLO_OK;
}

/* the two tables of a fine-tune step have the same format: named entry points for the two halves of train.py:54 / :67 */
int yolo_train_fwd_batch(const yolo_call* calls, int n_calls, const yolo_reloc* relocs, int n_relocs, const uint64_t* slots, int n_slots,
                         void* stream) {
    return yolo_run_calls(calls, n_calls, relocs, n_relocs, slots, n_slots, stream);
}
int yolo_train_bwd_batch(const yolo_call* calls, int n_calls, const yolo_reloc* relocs, int n_relocs, const uint64_t* slots, int n_slots,
                         void* stream) {
    return yolo_run_calls(calls, n_calls, relocs, n_relocs, slots, n_slots, stream);
}

}  // extern "C"
