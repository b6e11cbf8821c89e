// stem_f32.hip — the network's first block (3 -> C0 channels, 3x3, stride 1, pad 1) as a direct
// convolution on the vector ALUs.
//
// Replaces CNNBlock.forward for layers[0] (reference: code/model.py:20-21,80-86) AND the NCHW -> NHWC
// boundary conversion, including the `assert torch.sum(torch.isnan(x)) == 0` input guard of
// model.py:175 (every input element is the centre tap of exactly one output pixel).
//
// Why not MFMA: K = 27. Padded to a 32/64-wide GEMM K the matrix kernel spends 0.89 ms here; the
// layer is bound by its 32-channel fp32 OUTPUT (B*S*S*128 bytes = 709 MB at B=32, S=416, ~0.18 ms
// at 4 TB/s) and needs only 9.6 GFLOP, which the VALU does in about the same time. One thread = one
// output pixel x all C0 channels: 27 coalesced plane reads (L1/L2 absorb the 9x overlap), weights
// broadcast from LDS, 128 contiguous output bytes per thread.
#include "common.h"

namespace yolo {

typedef float f32x4 __attribute__((ext_vector_type(4)));

__device__ __forceinline__ unsigned short stem_cvt(float f, int dtype) {
    if (dtype == YOLO_BF16) { __bf16 h = (__bf16)f; return *reinterpret_cast<unsigned short*>(&h); }
    _Float16 h = (_Float16)f;
    return *reinterpret_cast<unsigned short*>(&h);
}

// DT: YOLO_F32 writes fp32; YOLO_F16 / YOLO_BF16 write 16-bit activations for conv_h16.hip
template <int COUT>
__global__ __launch_bounds__(256) void stem3x3_f32(const float* __restrict__ x, const float* __restrict__ wt,
                                                   const float* __restrict__ scale, const float* __restrict__ shift,
                                                   void* __restrict__ yv, int N, int H, int W, int y_ld, int y_off, int act,
                                                   int dtype, int* nan_flag) {
    float* y = reinterpret_cast<float*>(yv);
    // weights [27][COUT] in LDS: every lane reads the same address (broadcast, conflict-free).
    // (Reading them through the scalar cache was tried first: hipcc hoists all 864 s_loads and
    // spills 800 SGPRs.)
    __shared__ __attribute__((aligned(16))) float ws[27 * COUT];
    for (int i = threadIdx.x; i < 27 * COUT; i += 256) ws[i] = wt[i];
    __syncthreads();
    const long long total = (long long)N * H * W;
    const long long p = (long long)blockIdx.x * 256 + threadIdx.x;
    if (p >= total) return;
    const int HW = H * W;
    const int n = (int)(p / HW);
    const int rem = (int)(p - (long long)n * HW);
    const int h = rem / W, w = rem - h * W;
    const float* xb = x + (size_t)n * 3 * HW;
    float acc[COUT];
#pragma unroll
    for (int co = 0; co < COUT; ++co) acc[co] = 0.f;
    bool bad_in = false;
    // a real (not unrolled) loop over the 9 (channel, row) pairs: with everything unrolled hipcc
    // hoists all 216 weight reads to the top and spills ~400 VGPRs
#pragma unroll 1
    for (int t = 0; t < 9; ++t) {
        const int c = t / 3, dh = t - 3 * c;
        const int hh = h + dh - 1;
        const bool rok = (unsigned)hh < (unsigned)H;
        const float* row = xb + (size_t)c * HW + (size_t)(rok ? hh : 0) * W;
        float in[3];
        in[0] = (rok && w > 0) ? row[w - 1] : 0.f;
        in[1] = rok ? row[w] : 0.f;
        in[2] = (rok && w + 1 < W) ? row[w + 1] : 0.f;
        if (dh == 1) bad_in |= (in[1] != in[1]);          // centre tap: every input element exactly once
        const float* wk = ws + t * 3 * COUT;
#pragma unroll
        for (int dw = 0; dw < 3; ++dw)
#pragma unroll
            for (int q = 0; q < COUT / 4; ++q) {
                const f32x4 wv = *reinterpret_cast<const f32x4*>(wk + dw * COUT + q * 4);
#pragma unroll
                for (int e = 0; e < 4; ++e) acc[q * 4 + e] = fmaf(in[dw], wv[e], acc[q * 4 + e]);
            }
    }
    if (bad_in) atomicOr(nan_flag, 1);                    // NaN in the INPUT tensor (model.py:175)
    bool bad = false;
    YOLO_SWITCH_ACT(act,
        _Pragma("unroll") for (int co = 0; co < COUT; ++co) {
            const float t = act_c<ACT>(acc[co] * scale[co] + shift[co]);
            bad |= (t != t);
            acc[co] = t;
        })
    if (dtype == YOLO_F32) {
        float* dst = y + (size_t)p * y_ld + y_off;
#pragma unroll
        for (int co = 0; co < COUT; co += 4) {
            f32x4 v = {acc[co], acc[co + 1], acc[co + 2], acc[co + 3]};
            *reinterpret_cast<f32x4*>(dst + co) = v;
        }
    } else {
        typedef unsigned int u32x4 __attribute__((ext_vector_type(4)));
        unsigned short* dst = reinterpret_cast<unsigned short*>(yv) + (size_t)p * y_ld + y_off;
#pragma unroll
        for (int co = 0; co < COUT; co += 8) {
            u32x4 o;
#pragma unroll
            for (int e = 0; e < 4; ++e)
                o[e] = (unsigned)stem_cvt(acc[co + 2 * e], dtype) | ((unsigned)stem_cvt(acc[co + 2 * e + 1], dtype) << 16);
            *reinterpret_cast<u32x4*>(dst + co) = o;
        }
    }
    if (bad) atomicOr(nan_flag, 2);
}

// OIHW (COUT,3,3,3) -> [27][COUT]
__global__ void stem_pack_kernel(const float* __restrict__ w, float* __restrict__ wt, int cout) {
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= 27 * cout) return;
    const int k = i / cout, co = i - k * cout;
    wt[i] = w[co * 27 + k];
}

}  // namespace yolo

using namespace yolo;

extern "C" {

int yolo_stem_supported(int cin, int cout, int ksize, int stride) { return cin == 3 && cout == 32 && ksize == 3 && stride == 1; }

int yolo_stem_pack(const float* w_oihw, float* w_k_major, int cout, void* stream) {
    if (!w_oihw || !w_k_major || cout != 32) return fail(YOLO_ERR_ARG, "stem_pack: bad arguments");
    hipLaunchKernelGGL(stem_pack_kernel, dim3(ceil_div(27 * cout, 256)), dim3(256), 0, (hipStream_t)stream, w_oihw, w_k_major, cout);
    return check_launch("stem_pack");
}

int yolo_stem_fwd(const float* x_nchw, const float* w_k_major, const float* scale, const float* shift, void* y, int n, int h,
                  int w, int cout, int y_ld, int y_off, int act, int dtype, int32_t* nan_flag, void* stream) {
    if (!x_nchw || !w_k_major || !scale || !shift || !y || !nan_flag) return fail(YOLO_ERR_ARG, "stem: null pointer");
    const int al = dtype == YOLO_F32 ? 3 : 7;
    if (cout != 32 || n <= 0 || h <= 0 || w <= 0 || (y_ld & al) || (y_off & al) || y_ld < cout || dtype < 0 || dtype > 2)
        return fail(YOLO_ERR_UNSUPPORTED, "stem: only 3 -> 32 channels, y_ld/y_off multiples of 16 bytes");
    const long long total = (long long)n * h * w;
    if (total > 0x7fffffffLL) return fail(YOLO_ERR_UNSUPPORTED, "stem: too many pixels");
    hipLaunchKernelGGL((stem3x3_f32<32>), dim3((unsigned)((total + 255) / 256)), dim3(256), 0, (hipStream_t)stream, x_nchw,
                       w_k_major, scale, shift, y, n, h, w, y_ld, y_off, act, dtype, nan_flag);
    return check_launch("stem3x3_f32");
}

}  // extern "C"
