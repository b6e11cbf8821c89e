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
    // Weights [27][COUT] come through the SCALAR cache as SGPR operands of the FMAs: 16 at a time (one s_load_dwordx16),
    // inside loops that are NOT unrolled, so at most 48 of them are live (unrolled, hipcc hoists all 864 s_loads and spills
    // 800 SGPRs). The earlier version broadcast them from LDS: one ds_read_b128 per four FMAs = 216 LDS reads per pixel made
    // the kernel LDS-issue-bound (864 LDS cycles per wave against ~1,700 FMA cycles: 2.3 TB/s of output instead of ~5).
    const long long total = (long long)N * H * W;
    const long long p_raw = (long long)blockIdx.x * 256 + threadIdx.x;
    const bool live = p_raw < total;
    const long long p = live ? p_raw : total - 1;          // tail lanes recompute the last pixel (they take part in the store transpose)
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
        const float* wk = wt + t * 3 * COUT;              // wave-uniform address: scalar loads
        // explicit channel pairs (co, co + 1): one v_pk_fma_f32 per pair with the two weights as an aligned SGPR pair (left to
        // the vectoriser, odd pairings needed two s_mov per FMA and the scalar unit became the bottleneck)
        typedef float f32x2 __attribute__((ext_vector_type(2)));
        const f32x2* wk2 = reinterpret_cast<const f32x2*>(wk);
#pragma unroll
        for (int dw = 0; dw < 3; ++dw) {
            const f32x2 inb = {in[dw], in[dw]};
#pragma unroll
            for (int k = 0; k < COUT / 2; ++k) {
                const f32x2 w2 = wk2[dw * (COUT / 2) + k];
                f32x2 a2 = {acc[2 * k], acc[2 * k + 1]};
                a2 = __builtin_elementwise_fma(inb, w2, a2);
                acc[2 * k] = a2[0];
                acc[2 * k + 1] = a2[1];
            }
        }
    }
    if (bad_in && live) atomicOr(nan_flag, 1);            // NaN in the INPUT tensor (model.py:175)
    bool bad = false;
    YOLO_SWITCH_ACT(act,
        _Pragma("unroll") for (int co = 0; co < COUT; ++co) {
            const float t = act_c<ACT>(acc[co] * scale[co] + shift[co]);
            bad |= (t != t);
            acc[co] = t;
        })
    if (y_ld == COUT) {
        // A wave's 64 pixels are ONE contiguous run of 64 * COUT elements in NHWC memory, but a lane owns a whole pixel
        // (64 or 128 bytes): stored directly, every instruction writes 64 separate 16-byte pieces (measured 1.6 - 2.3 TB/s).
        // Exchange through LDS so that instruction j writes bytes [1024 j, 1024 (j + 1)) of the run, 16 per lane.
        __shared__ __attribute__((aligned(16))) char tr[4][64 * COUT * 4];
        typedef unsigned int u32x4 __attribute__((ext_vector_type(4)));
        const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
        char* my = tr[wave];
        const int esz = dtype == YOLO_F32 ? 4 : 2;
        const int row = COUT * esz, slots = row / 16;                 // bytes and 16-byte slots per pixel (8 or 4)
        const long long wave_p0 = (long long)blockIdx.x * 256 + wave * 64;
        const long long nvalid = total - wave_p0 < 64 ? total - wave_p0 : 64;
        // slot q of pixel `lane` goes to slot q ^ ((lane >> 1) & (slots - 1)): spreads the 64 lanes' equal-q writes over the banks
#pragma unroll
        for (int q = 0; q < COUT * 4 / 16; ++q) {
            if (q < slots) {
                u32x4 v;
                if (dtype == YOLO_F32) {
#pragma unroll
                    for (int e = 0; e < 4; ++e) v[e] = __float_as_uint(acc[q * 4 + e]);
                } else {
#pragma unroll
                    for (int e = 0; e < 4; ++e)
                        v[e] = (unsigned)stem_cvt(acc[q * 8 + 2 * e], dtype) | ((unsigned)stem_cvt(acc[q * 8 + 2 * e + 1], dtype) << 16);
                }
                *reinterpret_cast<u32x4*>(my + lane * row + ((q ^ ((lane >> 1) & (slots - 1))) << 4)) = v;
            }
        }
        __builtin_amdgcn_wave_barrier();
        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");           // the wave's own LDS writes (no other wave touches this region)
        char* gbase = reinterpret_cast<char*>(yv) + ((size_t)wave_p0 * COUT + y_off) * esz;
#pragma unroll
        for (int j = 0; j < COUT * 4 / 16; ++j) {
            if (j < slots) {
                const int o = j * 1024 + lane * 16;                   // byte offset inside the wave's run
                const int px = o / row, sl = (o % row) >> 4;
                const u32x4 v = *reinterpret_cast<const u32x4*>(my + px * row + ((sl ^ ((px >> 1) & (slots - 1))) << 4));
                if (px < nvalid) *reinterpret_cast<u32x4*>(gbase + o) = v;
            }
        }
    } else if (dtype == YOLO_F32) {
        if (!live) return;
        float* dst = y + (size_t)p * y_ld + y_off;
#pragma unroll
        for (int co = 0; co < COUT; co += 4) {
            f32x4 v = {acc[co], acc[co + 1], acc[co + 2], acc[co + 3]};
            *reinterpret_cast<f32x4*>(dst + co) = v;
        }
    } else {
        if (!live) return;
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
    if (bad && live) atomicOr(nan_flag, 2);
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
    // 16-bit output: the matrix-core kernel of conv_h16.hip (input and weights rounded to the 16-bit type, as autocast does)
    static const bool valu_stem = getenv("YOLO_STEM_VALU") != nullptr;      // A/B switch: keep the vector kernel
    if (dtype != YOLO_F32 && !valu_stem)
        return stem_h16_launch(x_nchw, w_k_major, scale, shift, y, n, h, w, y_ld, y_off, act, dtype, nan_flag, (hipStream_t)stream);
    hipLaunchKernelGGL((stem3x3_f32<32>), dim3((unsigned)((total + 255) / 256)), dim3(256), 0, (hipStream_t)stream, x_nchw,
                       w_k_major, scale, shift, y, n, h, w, y_ld, y_off, act, dtype, nan_flag);
    return check_launch("stem3x3_f32");
}

}  // extern "C"
