// layout.hip — boundary kernels: weight packing, BatchNorm folding, NCHW <-> NHWC.
// All are HBM-bound byte movers (no reuse): one pass, coalesced on the wider side.
#include "common.h"

namespace yolo {

static thread_local char g_err[512] = "";
char* err_buf() { return g_err; }
int fail(int code, const char* fmt, ...) {
    va_list ap;
    va_start(ap, fmt);
    vsnprintf(g_err, sizeof(g_err), fmt, ap);
    va_end(ap);
    return code;
}

// OIHW fp32 (nn.Conv2d.weight as the Darknet loader fills it, model.py:301-305) ->
// [Cout_pad][K_pad], K index = (kh*ks + kw)*cin_pad + ci. One thread per destination element;
// the destination is written coalesced, sources are ks*ks-strided gathers of a small tensor.
__global__ void pack_weights_f32(const float* __restrict__ w, float* __restrict__ wp, int cout, int cin, int ks,
                                 int cin_pad, int kpad, long long total) {
    for (long long i = blockIdx.x * (long long)blockDim.x + threadIdx.x; i < total; i += (long long)gridDim.x * blockDim.x) {
        const int o = (int)(i / kpad);
        const int k = (int)(i - (long long)o * kpad);
        const int tap = k / cin_pad;
        const int ci = k - tap * cin_pad;
        float v = 0.f;
        if (o < cout && tap < ks * ks && ci < cin) v = w[((size_t)o * cin + ci) * ks * ks + tap];
        wp[i] = v;
    }
}

__global__ void unpack_weights_f32(const float* __restrict__ wp, float* __restrict__ w, int cout, int cin, int ks,
                                   int cin_pad, int kpad, long long total) {
    const int kk = ks * ks;
    for (long long i = blockIdx.x * (long long)blockDim.x + threadIdx.x; i < total; i += (long long)gridDim.x * blockDim.x) {
        const int tap = (int)(i % kk);
        const long long oc = i / kk;
        const int ci = (int)(oc % cin);
        const int o = (int)(oc / cin);
        w[i] = wp[(size_t)o * kpad + tap * cin_pad + ci];
    }
}

__global__ void bn_fold_kernel(const float* gamma, const float* beta, const float* mean, const float* var, float eps,
                               float* scale, float* shift, int c) {
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= c) return;
    if (gamma) {
        const float s = gamma[i] / sqrtf(var[i] + eps);
        scale[i] = s;
        shift[i] = beta[i] - mean[i] * s;
    } else {
        scale[i] = 1.f;
        shift[i] = beta ? beta[i] : 0.f;
    }
}

// (N,C,H,W) -> (N,H,W,c_pad). C is tiny (3) at the network input: each thread handles one pixel,
// reads C planes (coalesced along W) and writes one 16-byte pixel.
__global__ void nchw_to_nhwc_small(const float* __restrict__ x, float* __restrict__ y, int n, int c, int hw, int c_pad,
                                   int* nan_flag) {
    const long long total = (long long)n * hw;
    bool bad = false;
    for (long long i = blockIdx.x * (long long)blockDim.x + threadIdx.x; i < total; i += (long long)gridDim.x * blockDim.x) {
        const long long img = i / hw;
        const long long pix = i - img * hw;
        const float* src = x + img * c * hw + pix;
        float* dst = y + i * c_pad;
        for (int k = 0; k < c_pad; ++k) {
            const float v = k < c ? src[(long long)k * hw] : 0.f;
            bad |= (v != v);
            dst[k] = v;
        }
    }
    if (bad && nan_flag) atomicOr(nan_flag, 1);
}

// the same for 16-bit outputs with c_pad == 8: one pixel = one 16-byte store (the fine-tune step converts the network input
// this way for the stem's weight gradient; through the generic 32x32 tiles below it took 176 us for 66 MB in / 88 MB out)
__global__ void nchw_to_nhwc_small_h16(const float* __restrict__ x, unsigned short* __restrict__ y, int n, int c, int hw, int dtype,
                                       int* nan_flag) {
    typedef unsigned int u32x4v __attribute__((ext_vector_type(4)));
    const long long total = (long long)n * hw;
    bool bad = false;
    for (long long i = blockIdx.x * (long long)blockDim.x + threadIdx.x; i < total; i += (long long)gridDim.x * blockDim.x) {
        const long long img = i / hw;
        const long long pix = i - img * hw;
        const float* src = x + img * c * hw + pix;
        unsigned short h[8];
#pragma unroll
        for (int k = 0; k < 8; ++k) {
            const float v = k < c ? src[(long long)k * hw] : 0.f;
            bad |= (v != v);
            if (dtype == YOLO_BF16) { __bf16 q = (__bf16)v; h[k] = *reinterpret_cast<unsigned short*>(&q); }
            else { _Float16 q = (_Float16)v; h[k] = *reinterpret_cast<unsigned short*>(&q); }
        }
        u32x4v o;
#pragma unroll
        for (int k = 0; k < 4; ++k) o[k] = (unsigned)h[2 * k] | ((unsigned)h[2 * k + 1] << 16);
        *reinterpret_cast<u32x4v*>(y + i * 8) = o;
    }
    if (bad && nan_flag) atomicOr(nan_flag, 1);
}

// generic tiled transpose for larger C (32x32 tiles through LDS), used by tests / debug taps
__device__ __forceinline__ unsigned short cvt16(float f, int dtype) {
    if (dtype == YOLO_BF16) { __bf16 h = (__bf16)f; return *reinterpret_cast<unsigned short*>(&h); }
    _Float16 h = (_Float16)f;
    return *reinterpret_cast<unsigned short*>(&h);
}
__device__ __forceinline__ float cvt32(unsigned short v, int dtype) {
    if (dtype == YOLO_BF16) return __uint_as_float((unsigned)v << 16);
    _Float16 h = *reinterpret_cast<_Float16*>(&v);
    return (float)h;
}

__global__ void nchw_to_nhwc_tiled(const float* __restrict__ x, void* __restrict__ yv, int c, int hw, int c_pad, int dtype, int* nan_flag) {
    float* y = reinterpret_cast<float*>(yv);
    __shared__ float t[32][33];
    const int img = blockIdx.z;
    const int p0 = blockIdx.x * 32, c0 = blockIdx.y * 32;
    bool bad = false;
    for (int r = threadIdx.y; r < 32; r += blockDim.y) {
        const int cc = c0 + r, pp = p0 + threadIdx.x;
        float v = (cc < c && pp < hw) ? x[((size_t)img * c + cc) * hw + pp] : 0.f;
        bad |= (v != v);
        t[r][threadIdx.x] = v;
    }
    __syncthreads();
    for (int r = threadIdx.y; r < 32; r += blockDim.y) {
        const int pp = p0 + r, cc = c0 + threadIdx.x;
        if (pp < hw && cc < c_pad) {
            if (dtype == YOLO_F32) y[((size_t)img * hw + pp) * c_pad + cc] = t[threadIdx.x][r];
            else reinterpret_cast<unsigned short*>(yv)[((size_t)img * hw + pp) * c_pad + cc] = cvt16(t[threadIdx.x][r], dtype);
        }
    }
    if (bad && nan_flag) atomicOr(nan_flag, 1);
}

__global__ void nhwc_to_nchw_tiled(const void* __restrict__ xv, float* __restrict__ y, int c, int hw, int x_ld, int x_off, int dtype) {
    const float* x = reinterpret_cast<const float*>(xv);
    __shared__ float t[32][33];
    const int img = blockIdx.z;
    const int p0 = blockIdx.x * 32, c0 = blockIdx.y * 32;
    for (int r = threadIdx.y; r < 32; r += blockDim.y) {
        const int pp = p0 + r, cc = c0 + threadIdx.x;
        float v = 0.f;
        if (pp < hw && cc < c) {
            const size_t idx = ((size_t)img * hw + pp) * x_ld + x_off + cc;
            v = dtype == YOLO_F32 ? x[idx] : cvt32(reinterpret_cast<const unsigned short*>(xv)[idx], dtype);
        }
        t[r][threadIdx.x] = v;
    }
    __syncthreads();
    for (int r = threadIdx.y; r < 32; r += blockDim.y) {
        const int cc = c0 + r, pp = p0 + threadIdx.x;
        if (cc < c && pp < hw) y[((size_t)img * c + cc) * hw + pp] = t[threadIdx.x][r];
    }
}

}  // namespace yolo

using namespace yolo;

extern "C" {

const char* yolo_last_error(void) { return yolo::err_buf(); }
int yolo_version(void) { return 100; }

size_t yolo_packed_weight_bytes(int cout, int cin, int ksize, int dtype) {
    if (cout <= 0 || cin <= 0 || (ksize != 1 && ksize != 3)) return 0;
    if (dtype == YOLO_F32) return yolo_packed_weight_elems(cout, cin, ksize) * sizeof(float);
    if (dtype == YOLO_F16 || dtype == YOLO_BF16) return cin % 32 ? 0 : h16_frag_elems(cout, cin, ksize) * 2;
    return 0;
}

size_t yolo_packed_weight_elems(int cout, int cin, int ksize) {
    if (cout <= 0 || cin <= 0 || (ksize != 1 && ksize != 3)) return 0;
    return v0_packed_elems(cout, cin, ksize) + v2_frag_elems(cout, cin, ksize) + wino_weight_elems(cout, cin, ksize);
}

int yolo_pack_weights(const float* w_oihw, void* w_packed, int cout, int cin, int ksize, int dtype, void* stream) {
    if (!w_oihw || !w_packed) return fail(YOLO_ERR_ARG, "pack_weights: null pointer");
    if (dtype == YOLO_F16 || dtype == YOLO_BF16) {
        if (!yolo_packed_weight_bytes(cout, cin, ksize, dtype)) return fail(YOLO_ERR_UNSUPPORTED, "pack_weights: 16-bit needs cin %% 32 == 0");
        return h16_pack(w_oihw, w_packed, cout, cin, ksize, dtype, (hipStream_t)stream);
    }
    if (dtype != YOLO_F32) return fail(YOLO_ERR_UNSUPPORTED, "pack_weights: dtype %d", dtype);
    if (!yolo_packed_weight_elems(cout, cin, ksize)) return fail(YOLO_ERR_ARG, "pack_weights: bad shape");
    const long long total = (long long)v0_packed_elems(cout, cin, ksize);
    const int grid = (int)((total + 255) / 256 < 4096 ? (total + 255) / 256 : 4096);
    hipLaunchKernelGGL(pack_weights_f32, dim3(grid), dim3(256), 0, (hipStream_t)stream, w_oihw, (float*)w_packed, cout, cin,
                       ksize, cin_pad_of(cin), kpad_of(cin, ksize), total);
    int rc = check_launch("pack_weights");
    if (rc) return rc;
    const size_t frag = v2_frag_elems(cout, cin, ksize);
    if (frag && (rc = v2_pack(w_oihw, (float*)w_packed + total, cout, cin, ksize, (hipStream_t)stream))) return rc;
    if (wino_weight_elems(cout, cin, ksize)) return wino_pack(w_oihw, (float*)w_packed + total + frag, cout, cin, (hipStream_t)stream);
    return YOLO_OK;
}

int yolo_pack_weights_batch(const yolo_pack_item* items, int n, int dgrad, int dtype, void* stream) {
    if (n < 0 || (n > 0 && !items)) return fail(YOLO_ERR_ARG, "pack_weights_batch: bad arguments");
    if (dtype != YOLO_F16 && dtype != YOLO_BF16) {            // fp32 layouts: one launch (pair) per item as before
        for (int i = 0; i < n; ++i) {
            const yolo_pack_item& q = items[i];
            const int rc = dgrad ? yolo_pack_weights_dgrad(q.w_oihw, q.w_packed, q.cout, q.cin, q.ksize, 1, dtype, stream)
                                 : yolo_pack_weights(q.w_oihw, q.w_packed, q.cout, q.cin, q.ksize, dtype, stream);
            if (rc) return rc;
        }
        return YOLO_OK;
    }
    if (n > 4096) return fail(YOLO_ERR_ARG, "pack_weights_batch: too many items");
    const float* w[4096]; void* wf[4096]; int cout[4096], cin[4096], ks[4096];
    for (int i = 0; i < n; ++i) {
        const yolo_pack_item& q = items[i];
        if (!q.w_oihw || !q.w_packed) return fail(YOLO_ERR_ARG, "pack_weights_batch: null pointer in item %d", i);
        const size_t bytes = dgrad ? yolo_packed_dgrad_bytes(q.cout, q.cin, q.ksize, 1, dtype) : yolo_packed_weight_bytes(q.cout, q.cin, q.ksize, dtype);
        if (!bytes) return fail(YOLO_ERR_UNSUPPORTED, "pack_weights_batch: item %d (%d->%d k%d) has no 16-bit layout", i, q.cin, q.cout, q.ksize);
        w[i] = q.w_oihw; wf[i] = q.w_packed; cout[i] = q.cout; cin[i] = q.cin; ks[i] = q.ksize;
    }
    return h16_pack_batch(w, wf, cout, cin, ks, n, dgrad, dtype, (hipStream_t)stream);
}

int yolo_unpack_weights(const void* w_packed, float* w_oihw, int cout, int cin, int ksize, int dtype, void* stream) {
    if (!w_oihw || !w_packed) return fail(YOLO_ERR_ARG, "unpack_weights: null pointer");
    if (dtype != YOLO_F32) return fail(YOLO_ERR_UNSUPPORTED, "unpack_weights: dtype %d", dtype);
    if (!yolo_packed_weight_elems(cout, cin, ksize)) return fail(YOLO_ERR_ARG, "unpack_weights: bad shape");
    const long long total = (long long)cout * cin * ksize * ksize;
    const int grid = (int)((total + 255) / 256 < 4096 ? (total + 255) / 256 : 4096);
    hipLaunchKernelGGL(unpack_weights_f32, dim3(grid), dim3(256), 0, (hipStream_t)stream, (const float*)w_packed, w_oihw,
                       cout, cin, ksize, cin_pad_of(cin), kpad_of(cin, ksize), total);
    return check_launch("unpack_weights");
}

int yolo_bn_fold(const float* gamma, const float* beta, const float* mean, const float* var, float eps, float* scale,
                 float* shift, int c, void* stream) {
    if (!scale || !shift || c <= 0) return fail(YOLO_ERR_ARG, "bn_fold: bad arguments");
    if (gamma && (!beta || !mean || !var)) return fail(YOLO_ERR_ARG, "bn_fold: gamma without beta/mean/var");
    hipLaunchKernelGGL(bn_fold_kernel, dim3(ceil_div(c, 256)), dim3(256), 0, (hipStream_t)stream, gamma, beta, mean, var, eps,
                       scale, shift, c);
    return check_launch("bn_fold");
}

int yolo_nchw_to_nhwc(const float* x, void* y, int n, int c, int h, int w, int c_pad, int dtype, int32_t* nan_flag,
                      void* stream) {
    if (!x || !y || n <= 0 || c <= 0 || h <= 0 || w <= 0 || c_pad < c) return fail(YOLO_ERR_ARG, "nchw_to_nhwc: bad arguments");
    if (dtype < 0 || dtype > 2) return fail(YOLO_ERR_UNSUPPORTED, "nchw_to_nhwc: dtype %d", dtype);
    const int hw = h * w;
    if (c_pad <= 8 && dtype == YOLO_F32) {
        const long long total = (long long)n * hw;
        const int grid = (int)((total + 255) / 256 < 8192 ? (total + 255) / 256 : 8192);
        hipLaunchKernelGGL(nchw_to_nhwc_small, dim3(grid), dim3(256), 0, (hipStream_t)stream, x, (float*)y, n, c, hw, c_pad,
                           nan_flag);
    } else if (c_pad == 8 && c <= 8 && dtype != YOLO_F32) {
        const long long total = (long long)n * hw;
        const int grid = (int)((total + 255) / 256 < 16384 ? (total + 255) / 256 : 16384);
        hipLaunchKernelGGL(nchw_to_nhwc_small_h16, dim3(grid), dim3(256), 0, (hipStream_t)stream, x, (unsigned short*)y, n, c, hw, dtype,
                           nan_flag);
    } else {
        dim3 grid(ceil_div(hw, 32), ceil_div(c_pad, 32), n), block(32, 8);
        hipLaunchKernelGGL(nchw_to_nhwc_tiled, grid, block, 0, (hipStream_t)stream, x, y, c, hw, c_pad, dtype, nan_flag);
    }
    return check_launch("nchw_to_nhwc");
}

int yolo_nhwc_to_nchw(const void* x, float* y, int n, int c, int h, int w, int x_ld, int x_off, int dtype, void* stream) {
    if (!x || !y || n <= 0 || c <= 0 || h <= 0 || w <= 0 || x_ld < c) return fail(YOLO_ERR_ARG, "nhwc_to_nchw: bad arguments");
    if (dtype < 0 || dtype > 2) return fail(YOLO_ERR_UNSUPPORTED, "nhwc_to_nchw: dtype %d", dtype);
    const int hw = h * w;
    dim3 grid(ceil_div(hw, 32), ceil_div(c, 32), n), block(32, 8);
    hipLaunchKernelGGL(nhwc_to_nchw_tiled, grid, block, 0, (hipStream_t)stream, x, y, c, hw, x_ld, x_off, dtype);
    return check_launch("nhwc_to_nchw");
}

}  // extern "C"
