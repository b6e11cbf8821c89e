// Shared host-side helpers for libyolo_mi355x.so (gfx950 only).
#pragma once
#include <hip/hip_runtime.h>
#include <atomic>
#include <cstdarg>
#include <cstdio>
#include <cstring>
#include "../../include/yolo_mi355x.h"

namespace yolo {

// thread-local message returned by yolo_last_error()
char* err_buf();
int fail(int code, const char* fmt, ...);

inline int check_launch(const char* what) {
    hipError_t e = hipGetLastError();
    if (e != hipSuccess) return fail(YOLO_ERR_LAUNCH, "%s: %s", what, hipGetErrorString(e));
    return YOLO_OK;
}

// Kernels that need more than 64 KiB of dynamic LDS have to ask for it once per kernel AND per device (the attribute is
// per device; plans are keyed by device index, so one process driving several GPUs is a supported case). One LdsOnce per
// call site (i.e. per kernel instantiation); bit d = "device d has been configured".
struct LdsOnce { std::atomic<unsigned long long> done{0}; };
inline int reserve_lds(LdsOnce& st, const void* fn, size_t bytes, const char* what) {
    int dev = 0;
    if (hipGetDevice(&dev) != hipSuccess) { (void)hipGetLastError(); return fail(YOLO_ERR_LAUNCH, "%s: no current device", what); }
    const unsigned long long bit = 1ull << (dev & 63);
    if (st.done.load(std::memory_order_relaxed) & bit) return YOLO_OK;
    if (hipFuncSetAttribute(fn, hipFuncAttributeMaxDynamicSharedMemorySize, (int)bytes) != hipSuccess) {
        (void)hipGetLastError();
        return fail(YOLO_ERR_LAUNCH, "%s: cannot reserve %zu bytes of LDS", what, bytes);
    }
    st.done.fetch_or(bit, std::memory_order_relaxed);
    return YOLO_OK;
}

inline int ceil_div(int a, int b) { return (a + b - 1) / b; }
inline int round_up(int a, int b) { return ceil_div(a, b) * b; }

// packed-weight geometry (see yolo_packed_weight_elems in the header)
inline int cin_pad_of(int cin) { return round_up(cin, 4); }
inline int kpad_of(int cin, int ks) { return round_up(ks * ks * cin_pad_of(cin), 32); }
inline int coutpad_of(int cout) { return round_up(cout, 128); }


// ---- element types of activation / gradient tensors: fp32, or 16-bit storage with fp32 arithmetic.
// Kernels are templated on T in {float, __bf16, _Float16}; S is the storage type behind the void* of the ABI.
typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef unsigned int u32x2 __attribute__((ext_vector_type(2)));

template <typename T> struct Elt;
template <> struct Elt<float> {
    typedef float S;
    static __device__ __forceinline__ f32x4 ld4(const S* p) { return *reinterpret_cast<const f32x4*>(p); }
    static __device__ __forceinline__ void st4(S* p, f32x4 v) { *reinterpret_cast<f32x4*>(p) = v; }
    static __device__ __forceinline__ float ld(const S* p) { return *p; }
    static __device__ __forceinline__ void st(S* p, float v) { *p = v; }
};
template <> struct Elt<__bf16> {
    typedef unsigned short S;
    static __device__ __forceinline__ f32x4 ld4(const S* p) {
        const u32x2 r = *reinterpret_cast<const u32x2*>(p);
        f32x4 v = {__uint_as_float(r[0] << 16), __uint_as_float(r[0] & 0xffff0000u), __uint_as_float(r[1] << 16),
                   __uint_as_float(r[1] & 0xffff0000u)};
        return v;
    }
    static __device__ __forceinline__ unsigned short cvt(float f) { __bf16 h = (__bf16)f; return *reinterpret_cast<unsigned short*>(&h); }
    static __device__ __forceinline__ void st4(S* p, f32x4 v) {
        u32x2 r = {(unsigned)cvt(v[0]) | ((unsigned)cvt(v[1]) << 16), (unsigned)cvt(v[2]) | ((unsigned)cvt(v[3]) << 16)};
        *reinterpret_cast<u32x2*>(p) = r;
    }
    static __device__ __forceinline__ float ld(const S* p) { return __uint_as_float((unsigned)*p << 16); }
    static __device__ __forceinline__ void st(S* p, float v) { *p = cvt(v); }
};
template <> struct Elt<_Float16> {
    typedef unsigned short S;
    typedef _Float16 h4 __attribute__((ext_vector_type(4)));
    static __device__ __forceinline__ f32x4 ld4(const S* p) {
        const h4 h = *reinterpret_cast<const h4*>(p);
        f32x4 v = {(float)h[0], (float)h[1], (float)h[2], (float)h[3]};
        return v;
    }
    static __device__ __forceinline__ void st4(S* p, f32x4 v) {
        h4 h = {(_Float16)v[0], (_Float16)v[1], (_Float16)v[2], (_Float16)v[3]};
        *reinterpret_cast<h4*>(p) = h;
    }
    static __device__ __forceinline__ float ld(const S* p) { return (float)*reinterpret_cast<const _Float16*>(p); }
    static __device__ __forceinline__ void st(S* p, float v) { *reinterpret_cast<_Float16*>(p) = (_Float16)v; }
};

// ---- activations: compile-time selected (a run-time `act` inside an unrolled epilogue loop made the compiler
// inline libm's tanhf/log1pf/expf once per element and branch per element: 16k-instruction epilogues that
// thrashed the instruction cache and cost more than the matrix work of a 16-bit block).
// Mish(x) = x * tanh(softplus(x)) = x * n / (n + 2) with n = e^x (e^x + 2): one v_exp_f32 + one v_rcp_f32,
// no cancellation for x -> -inf (n ~ 2 e^x), identity for x > 20. Derivative: with t = n/(n+2),
// s = sigmoid(x) = e/(1+e):  d/dx = t + x (1 - t^2) s.
template <int ACT>
__device__ __forceinline__ float act_c(float v) {
    if constexpr (ACT == YOLO_ACT_LEAKY) {
        return __builtin_fmaxf(v, v * 0.1f);                 // == v > 0 ? v : 0.1 v for every input (NaN stays NaN, -0 stays -0): 2 ops, not 3
    } else if constexpr (ACT == YOLO_ACT_MISH) {
        const float e = __expf(v < 20.f ? v : 20.f);
        const float n = e * (e + 2.f);
        const float m = v * (n * __builtin_amdgcn_rcpf(n + 2.f));
        return v > 20.f ? v : m;
    } else {
        return v;
    }
}
template <int ACT>
__device__ __forceinline__ float act_grad_c(float u) {
    if constexpr (ACT == YOLO_ACT_LEAKY) {
        return u > 0.f ? 1.f : 0.1f;
    } else if constexpr (ACT == YOLO_ACT_MISH) {
        const float e = __expf(u < 20.f ? u : 20.f);
        const float n = e * (e + 2.f);
        const float t = n * __builtin_amdgcn_rcpf(n + 2.f);
        const float sg = e * __builtin_amdgcn_rcpf(1.f + e);
        const float g = t + u * (1.f - t * t) * sg;
        return u > 20.f ? 1.f : g;
    } else {
        return 1.f;
    }
}
// run `body` (device or host code) with the constant ACT bound to the run-time activation code
#define YOLO_SWITCH_ACT(act, ...)                                                          \
    switch (act) {                                                                          \
    case YOLO_ACT_LEAKY: { constexpr int ACT = YOLO_ACT_LEAKY; __VA_ARGS__; } break;       \
    case YOLO_ACT_MISH: { constexpr int ACT = YOLO_ACT_MISH; __VA_ARGS__; } break;         \
    default: { constexpr int ACT = YOLO_ACT_NONE; __VA_ARGS__; } break;                    \
    }

// 16-byte vectors for the HBM-bound passes: VN elements (4 fp32 / 8 halfs) per load
template <typename T> struct Vec16;
template <> struct Vec16<float> {
    static constexpr int VN = 4;
    static __device__ __forceinline__ void ld(const float* p, float (&v)[4]) {
        const f32x4 r = *reinterpret_cast<const f32x4*>(p);
        v[0] = r[0]; v[1] = r[1]; v[2] = r[2]; v[3] = r[3];
    }
    static __device__ __forceinline__ void st(float* p, const float (&v)[4]) {
        const f32x4 r = {v[0], v[1], v[2], v[3]};
        *reinterpret_cast<f32x4*>(p) = r;
    }
};
template <> struct Vec16<__bf16> {
    static constexpr int VN = 8;
    typedef unsigned int u4 __attribute__((ext_vector_type(4)));
    static __device__ __forceinline__ void ld(const unsigned short* p, float (&v)[8]) {
        const u4 r = *reinterpret_cast<const u4*>(p);
#pragma unroll
        for (int e = 0; e < 4; ++e) { v[2 * e] = __uint_as_float(r[e] << 16); v[2 * e + 1] = __uint_as_float(r[e] & 0xffff0000u); }
    }
    static __device__ __forceinline__ void st(unsigned short* p, const float (&v)[8]) {
        u4 r;
#pragma unroll
        for (int e = 0; e < 4; ++e) r[e] = (unsigned)Elt<__bf16>::cvt(v[2 * e]) | ((unsigned)Elt<__bf16>::cvt(v[2 * e + 1]) << 16);
        *reinterpret_cast<u4*>(p) = r;
    }
};
template <> struct Vec16<_Float16> {
    static constexpr int VN = 8;
    typedef _Float16 h8 __attribute__((ext_vector_type(8)));
    static __device__ __forceinline__ void ld(const unsigned short* p, float (&v)[8]) {
        const h8 r = *reinterpret_cast<const h8*>(p);
#pragma unroll
        for (int e = 0; e < 8; ++e) v[e] = (float)r[e];
    }
    static __device__ __forceinline__ void st(unsigned short* p, const float (&v)[8]) {
        h8 r;
#pragma unroll
        for (int e = 0; e < 8; ++e) r[e] = (_Float16)v[e];
        *reinterpret_cast<h8*>(p) = r;
    }
};

// run `body` with T bound to the element type of `dtype` (host side)
#define YOLO_DISPATCH_DTYPE(dtype, what, ...)                                             \
    switch (dtype) {                                                                      \
    case YOLO_F32: { typedef float T; __VA_ARGS__; } break;                               \
    case YOLO_BF16: { typedef __bf16 T; __VA_ARGS__; } break;                             \
    case YOLO_F16: { typedef _Float16 T; __VA_ARGS__; } break;                            \
    default: return fail(YOLO_ERR_ARG, "%s: unknown dtype %d", what, (int)(dtype));       \
    }

// conv_f32_v2.hip ("patch + fragment stream" kernel, stride 1, cin % 32 == 0)
bool v2_eligible(const yolo_conv_desc* d);
size_t v2_frag_elems(int cout, int cin, int ks);
int v2_blocks(const yolo_conv_desc* d, int bn);          // grid size the patch kernel would launch
int v2_pack(const float* w_oihw, float* wf, int cout, int cin, int ks, hipStream_t s);
int conv_v2_launch(const yolo_conv_desc* d, const void* x, const float* wf, const float* scale, const float* shift,
                   const void* residual, void* y, int32_t* nan_flag, int bn, bool single_buffer, hipStream_t s);
// conv1_rs_f32.hip (fp32 1x1, weights stationary in registers, persistent workgroups)
bool conv1_rs_eligible(const yolo_conv_desc* d, const void* residual);
int conv1_rs_launch(const yolo_conv_desc* d, const void* x, const void* w, const float* scale, const float* shift, const void* residual,
                    void* y, int32_t* nan_flag, hipStream_t s);
// conv_wino_f32.hip (fp32 3x3 stride 1 by Winograd F(2x2, 3x3): transform pass into a caller-owned workspace + 16 GEMMs with
// the output transform in the epilogue). U sits behind the fragment-order copy in the packed fp32 buffer.
size_t wino_weight_elems(int cout, int cin, int ks);
int wino_pack(const float* w_oihw, float* U, int cout, int cin, hipStream_t s);
int wino_pack_dgrad(const float* w_oihw, float* U, int cout, int cin, hipStream_t s);
bool wino_supported(const yolo_conv_desc* d);
bool wino_eligible(const yolo_conv_desc* d);
size_t wino_workspace_bytes(const yolo_conv_desc* d);
int conv_wino_launch(const yolo_conv_desc* d, const void* x, const float* U, const float* scale, const float* shift,
                     const void* residual, void* y, void* workspace, size_t workspace_bytes, int32_t* nan_flag, hipStream_t s);
// conv_h16.hip (bf16 / fp16 patch kernel)
size_t h16_frag_elems(int cout, int cin, int ks);
int h16_pack(const float* w_oihw, void* wf, int cout, int cin, int ks, int dtype, hipStream_t s);
int h16_pack_dgrad(const float* w_oihw, void* wf, int cout, int cin, int ks, int dtype, hipStream_t s);
int h16_pack_batch(const float* const* w, void* const* wf, const int* cout, const int* cin, const int* ks, int n, int dgrad, int dtype,
                   hipStream_t s);
size_t h16_dgrad_s2_elems(int cout, int cin);
int h16_pack_dgrad_s2(const float* w_oihw, void* wf, int cout, int cin, int dtype, hipStream_t s);
int dgrad_s2_h16_launch(const void* dz, int dz_ld, int dz_off, const void* wf, const void* residual, int r_ld, int r_off, void* dx,
                        int dx_ld, int dx_off, int n, int ho, int wo, int cin, int cout, int dtype, hipStream_t s);
int conv_h16_launch(const yolo_conv_desc* d, const void* x, const void* wf, const float* scale, const float* shift,
                    const void* residual, void* y, int32_t* nan_flag, hipStream_t s);
// wgrad_stem_h16.hip: weight gradient of the first block (3x3 stride 1, <= 3 input channels, <= 32 output channels)
bool wgrad_stem_eligible(int n, int h, int w, int cin, int cout, int ksize, int stride, int dz_ld, int dz_off, int x_ld, int x_off);
size_t wgrad_stem_workspace(int n, int h, int w);
int wgrad_stem_launch(const void* dz, int dz_ld, int dz_off, const void* x, int x_off, float* workspace, float* dw_oihw, int n, int h,
                      int w, int cin, int cout, int dtype, hipStream_t s);
// backward statistics of an input-gradient launch: the block that produced the convolution's input (see ConvHArgs::bz)
struct ConvBStats { const void* z; int z_ld, z_off; const float* mean; const float* scale; const float* shift; int act; };
int conv_h16_launch_stats(const yolo_conv_desc* d, const void* x, const void* wf, const float* scale, const float* shift,
                          const void* residual, void* y, int32_t* nan_flag, float* stats, int* rows_ld, const size_t* stats_bytes,
                          hipStream_t s, const ConvBStats* bs);
// the first block (3 -> 32 channels, 3x3) with a 16-bit output on the matrix cores; same arguments as yolo_stem_fwd
int stem_h16_launch(const float* x, const float* wt, const float* scale, const float* shift, void* y, int n, int h, int w, int y_ld,
                    int y_off, int act, int dtype, int* nan_flag, hipStream_t s);
// wgrad_h16.hip (bf16 / fp16 weight gradient, transposing LDS reads)
bool wgrad_h16_eligible(int cin, int cout, int ks, int stride, int dz_ld, int dz_off, int x_ld, int x_off);
size_t wgrad_h16_workspace(int n, int h, int w, int cin, int cout, int ks, int stride);
int wgrad_h16_launch(const void* dz, int dz_ld, int dz_off, const void* x, int x_ld, int x_off, float* partial, int n, int h, int w,
                     int cin, int cout, int ks, int stride, int dtype, int* cout_pad, hipStream_t s);
int tr_probe_launch(const void* in, void* out, int ld, hipStream_t s);
// wgrad_dma_h16.hip (3x3 stride-1 weight gradient: LDS-DMA operands, one partial per CU, accumulator-order partials + own reduce)
bool wgrad_dma_eligible(int cin, int cout, int ks, int stride, int dz_ld, int dz_off, int x_ld, int x_off);
size_t wgrad_dma_workspace(int n, int h, int w, int cin, int cout);
int wgrad_dma_launch(const void* dz, int dz_ld, int dz_off, const void* x, int x_ld, int x_off, float* partial, float* dw, int n, int h,
                     int w, int cin, int cout, int dtype, hipStream_t s);
// offset (elements) of the fragment-order copy inside a packed weight buffer
inline size_t v0_packed_elems(int cout, int cin, int ks) { return (size_t)coutpad_of(cout) * kpad_of(cin, ks); }

}  // namespace yolo
