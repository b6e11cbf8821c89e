// bn_train.hip — training-mode BatchNorm + activation, forward and backward (HBM-bound passes).
//
// Replaces (reference file:line): nn.BatchNorm2d in train mode + LeakyReLU(0.1)/Mish inside
// CNNBlock.forward code/model.py:61-66,84 and their autograd backward under train.py:67
// (`grad_scaler.scale(loss).backward()`); the `x + layer(x)` skip add of model.py:118.
//
// Batch statistics break the single-pass epilogue fusion of the inference kernels: the conv writes
// the raw pre-BN tensor z once, then
//   bn_stats_partial  + bn_stats_finalize : per-channel mean / biased variance, running-stat update
//                                           (momentum 0.1, unbiased variance), folded scale/shift
//   bn_act_fwd                            : y = act(z*scale + shift) [+ residual]   (1 read + 1 write)
// and in backward, with u = z*scale + shift recomputed from z (nothing but z is saved per block):
//   bn_bwd_partial + bn_bwd_finalize      : dbeta = sum(du), dgamma = sum(du * zhat), du = dy*act'(u)
//   bn_bwd_apply                          : dz = gamma*invstd * (du - dbeta/n - zhat*dgamma/n)
// Reductions are deterministic (the reference seeds everything and sets
// cudnn.deterministic): fixed pixel ranges per block, fp32 per-thread partials over short runs,
// fp64 across threads and blocks in a fixed order. No float atomics.
// Algorithmic bytes per element (fp32): stats 4 R; fwd apply 4 R + 4 W (+4 R residual);
// bwd partial 8 R; bwd apply 8 R + 4 W.
#include "common.h"

namespace yolo {

__device__ __forceinline__ float act_fwd(float u, int act) {
    if (act == YOLO_ACT_LEAKY) return u > 0.f ? u : u * 0.1f;
    if (act == YOLO_ACT_MISH) {
        const float sp = u > 20.f ? u : log1pf(expf(u));
        return u * tanhf(sp);
    }
    return u;
}

__device__ __forceinline__ float act_grad(float u, int act) {
    if (act == YOLO_ACT_LEAKY) return u > 0.f ? 1.f : 0.1f;
    if (act == YOLO_ACT_MISH) {                      // d/du [u * tanh(softplus(u))]
        const float sp = u > 20.f ? u : log1pf(expf(u));
        const float t = tanhf(sp);
        const float sg = 1.f / (1.f + expf(-u));
        return t + u * (1.f - t * t) * sg;
    }
    return 1.f;
}

// Thread layout shared by the reductions: C4 = C/4 vector channels; thread t owns vector channel
// t % VC (VC = min(C4, 256)) and pixel lane t / VC; a block covers every channel of its pixel range.
struct RedGeom { int vc, lanes, passes; };
__device__ __forceinline__ RedGeom red_geom(int c4) {
    RedGeom g;
    g.vc = c4 < 256 ? c4 : 256;
    g.lanes = 256 / g.vc;
    g.passes = (c4 + g.vc - 1) / g.vc;
    return g;
}

// partial[blk][c][2] (double): sum z, sum z^2 over the block's pixel range
template <typename T>
__global__ __launch_bounds__(256) void bn_stats_partial(const typename Elt<T>::S* __restrict__ z, int m, int c, int ld, int off,
                                                        int pix_per_block, double* __restrict__ partial) {
    __shared__ double red[256][2];
    const int c4 = c >> 2;
    const RedGeom g = red_geom(c4);
    const int t = threadIdx.x;
    const int v = t % g.vc, lane = t / g.vc;
    const int p0 = blockIdx.x * pix_per_block;
    const int p1 = p0 + pix_per_block < m ? p0 + pix_per_block : m;
    for (int pass = 0; pass < g.passes; ++pass) {
        const int vch = pass * g.vc + v;
        double s[4] = {0, 0, 0, 0}, q[4] = {0, 0, 0, 0};
        if (vch < c4 && lane < g.lanes) {
            float fs[4] = {0, 0, 0, 0}, fq[4] = {0, 0, 0, 0};
            int run = 0;
            const typename Elt<T>::S* zp = z + off + vch * 4;
            int p = p0 + lane;
            // 4 independent 16-byte loads in flight per thread (one-at-a-time left the kernel at ~45 % of
            // the achievable HBM rate: latency-bound, not bandwidth-bound)
            for (; p + 3 * g.lanes < p1; p += 4 * g.lanes) {
                f32x4 x[4];
#pragma unroll
                for (int u = 0; u < 4; ++u) x[u] = Elt<T>::ld4(zp + (size_t)(p + u * g.lanes) * ld);
#pragma unroll
                for (int u = 0; u < 4; ++u)
#pragma unroll
                    for (int e = 0; e < 4; ++e) { fs[e] += x[u][e]; fq[e] += x[u][e] * x[u][e]; }
                run += 4;
                if (run >= 64) {                         // flush short fp32 runs into fp64
#pragma unroll
                    for (int e = 0; e < 4; ++e) { s[e] += fs[e]; q[e] += fq[e]; fs[e] = 0.f; fq[e] = 0.f; }
                    run = 0;
                }
            }
            for (; p < p1; p += g.lanes) {
                const f32x4 x = Elt<T>::ld4(zp + (size_t)p * ld);
#pragma unroll
                for (int e = 0; e < 4; ++e) { fs[e] += x[e]; fq[e] += x[e] * x[e]; }
            }
#pragma unroll
            for (int e = 0; e < 4; ++e) { s[e] += fs[e]; q[e] += fq[e]; }
        }
        // reduce over pixel lanes (fixed order) through LDS, one channel component at a time
#pragma unroll
        for (int e = 0; e < 4; ++e) {
            __syncthreads();
            red[t][0] = s[e];
            red[t][1] = q[e];
            __syncthreads();
            if (lane == 0 && vch < c4) {
                double a = 0, b = 0;
                for (int l = 0; l < g.lanes; ++l) { a += red[l * g.vc + v][0]; b += red[l * g.vc + v][1]; }
                double* dst = partial + ((size_t)blockIdx.x * c + vch * 4 + e) * 2;
                dst[0] = a;
                dst[1] = b;
            }
        }
    }
}

// Fixed-order reduction of partial[nblk][c][2] for the finalize kernels: a block handles 16 channels
// with 16 lanes each; lane l sums blocks l, l+16, ... (independent loads, pipelined), then lanes are
// added in order 0..15 through LDS. (One thread per channel walking 1024 partials took ~65 us.)
__device__ __forceinline__ bool reduce_partials(const double* __restrict__ partial, int nblk, int c, int* ch_out,
                                                double* s_out, double* q_out) {
    __shared__ double red[16][16][2];
    const int lc = threadIdx.x & 15, lane = threadIdx.x >> 4;
    const int ch = blockIdx.x * 16 + lc;
    double s = 0, q = 0;
    if (ch < c)
        for (int b = lane; b < nblk; b += 16) {
            s += partial[((size_t)b * c + ch) * 2];
            q += partial[((size_t)b * c + ch) * 2 + 1];
        }
    red[lane][lc][0] = s;
    red[lane][lc][1] = q;
    __syncthreads();
    if (lane != 0 || ch >= c) return false;
    s = 0; q = 0;
    for (int l = 0; l < 16; ++l) { s += red[l][lc][0]; q += red[l][lc][1]; }
    *ch_out = ch; *s_out = s; *q_out = q;
    return true;
}

__global__ void bn_stats_finalize(const double* __restrict__ partial, int nblk, int m, int c, float momentum, float eps,
                                  const float* __restrict__ gamma, const float* __restrict__ beta,
                                  float* __restrict__ running_mean, float* __restrict__ running_var,
                                  float* __restrict__ mean, float* __restrict__ invstd, float* __restrict__ scale,
                                  float* __restrict__ shift) {
    int ch;
    double s, q;
    if (!reduce_partials(partial, nblk, c, &ch, &s, &q)) return;
    const double mu = s / m;
    double var = q / m - mu * mu;                       // biased (what normalisation uses)
    if (var < 0) var = 0;
    const float is = 1.0f / sqrtf((float)var + eps);
    mean[ch] = (float)mu;
    invstd[ch] = is;
    // u = (z - mean) * scale + beta is evaluated in that order by the train kernels (like PyTorch): the
    // folded form z*scale + (beta - mean*scale) cancels badly when |mean| >> std, and the backward chain
    // through 70 BatchNorm layers amplifies that noise (measured 8x the reference's own fp32 noise).
    scale[ch] = gamma[ch] * is;
    shift[ch] = beta[ch];
    if (running_mean) {
        const double unbiased = m > 1 ? var * m / (m - 1) : var;
        running_mean[ch] = (1.f - momentum) * running_mean[ch] + momentum * (float)mu;
        running_var[ch] = (1.f - momentum) * running_var[ch] + momentum * (float)unbiased;
    }
}

// y = act(z*scale + shift) [+ residual], float4 per thread, same output modes as the conv epilogue
template <typename T>
__global__ __launch_bounds__(256) void bn_act_fwd_kernel(const typename Elt<T>::S* __restrict__ z, int z_ld, int z_off,
                                                         const float* __restrict__ mean,
                                                         const float* __restrict__ scale, const float* __restrict__ shift,
                                                         const typename Elt<T>::S* __restrict__ res, int r_ld, int r_off,
                                                         typename Elt<T>::S* __restrict__ y, int y_ld, int y_off, long long m, int c,
                                                         int Ho, int Wo, int act, int out_mode, int* nan_flag) {
    const int c4 = c >> 2;
    const long long total = m * c4;
    bool bad = false;
    for (long long i = blockIdx.x * 256LL + threadIdx.x; i < total; i += (long long)gridDim.x * 256) {
        const long long p = i / c4;
        const int ch = (int)(i - p * c4) * 4;
        const f32x4 x = Elt<T>::ld4(z + (size_t)p * z_ld + z_off + ch);
        const f32x4 sc = *reinterpret_cast<const f32x4*>(scale + ch);
        const f32x4 sh = *reinterpret_cast<const f32x4*>(shift + ch);
        f32x4 mu = {0.f, 0.f, 0.f, 0.f};
        if (mean) mu = *reinterpret_cast<const f32x4*>(mean + ch);
        f32x4 v;
#pragma unroll
        for (int e = 0; e < 4; ++e) v[e] = act_fwd((x[e] - mu[e]) * sc[e] + sh[e], act);
        if (res) v += Elt<T>::ld4(res + (size_t)p * r_ld + r_off + ch);
        bad |= (v[0] != v[0]) | (v[1] != v[1]) | (v[2] != v[2]) | (v[3] != v[3]);
        if (out_mode == YOLO_OUT_UPSAMPLE2X) {
            const long long hw = (long long)Ho * Wo;
            const long long img = p / hw;
            const int rem = (int)(p - img * hw);
            const int ho = rem / Wo, wo = rem - ho * Wo;
            const int W2 = 2 * Wo;
            typename Elt<T>::S* d = y + ((size_t)(img * 2 * Ho + 2 * ho) * W2 + 2 * wo) * y_ld + y_off + ch;
            Elt<T>::st4(d, v);
            Elt<T>::st4(d + y_ld, v);
            Elt<T>::st4(d + (size_t)W2 * y_ld, v);
            Elt<T>::st4(d + (size_t)(W2 + 1) * y_ld, v);
        } else {
            Elt<T>::st4(y + (size_t)p * y_ld + y_off + ch, v);
        }
    }
    if (bad && nan_flag) atomicOr(nan_flag, 2);
}

// partial[blk][c][2]: sum du, sum du*zhat.  gamma == nullptr: bare conv (du = dy, only sum du is used)
template <typename T>
__global__ __launch_bounds__(256) void bn_bwd_partial(const typename Elt<T>::S* __restrict__ dy, int dy_ld, int dy_off,
                                                      const typename Elt<T>::S* __restrict__ z, int z_ld, int z_off,
                                                      const float* __restrict__ mean, const float* __restrict__ invstd,
                                                      const float* __restrict__ scale, const float* __restrict__ shift,
                                                      int m, int c, int act, int pix_per_block, double* __restrict__ partial) {
    __shared__ double red[256][2];
    const int c4 = c >> 2;
    const RedGeom g = red_geom(c4);
    const int t = threadIdx.x;
    const int v = t % g.vc, lane = t / g.vc;
    const int p0 = blockIdx.x * pix_per_block;
    const int p1 = p0 + pix_per_block < m ? p0 + pix_per_block : m;
    for (int pass = 0; pass < g.passes; ++pass) {
        const int vch = pass * g.vc + v;
        double s[4] = {0, 0, 0, 0}, q[4] = {0, 0, 0, 0};
        if (vch < c4 && lane < g.lanes) {
            f32x4 mu = {0, 0, 0, 0}, is = {1, 1, 1, 1}, sc = {1, 1, 1, 1}, sh = {0, 0, 0, 0};
            if (mean) {
                mu = *reinterpret_cast<const f32x4*>(mean + vch * 4);
                is = *reinterpret_cast<const f32x4*>(invstd + vch * 4);
                sc = *reinterpret_cast<const f32x4*>(scale + vch * 4);
                sh = *reinterpret_cast<const f32x4*>(shift + vch * 4);
            }
            float fs[4] = {0, 0, 0, 0}, fq[4] = {0, 0, 0, 0};
            int run = 0;
            auto accum = [&](const f32x4& d, const f32x4& x) {
#pragma unroll
                for (int e = 0; e < 4; ++e) {
                    const float du = mean ? d[e] * act_grad((x[e] - mu[e]) * sc[e] + sh[e], act) : d[e];
                    fs[e] += du;
                    fq[e] += du * ((x[e] - mu[e]) * is[e]);
                }
            };
            int p = p0 + lane;
            for (; p + g.lanes < p1; p += 2 * g.lanes) {          // 4 loads in flight (2 pixels x {dy, z})
                const f32x4 d0 = Elt<T>::ld4(dy + (size_t)p * dy_ld + dy_off + vch * 4);
                const f32x4 d1 = Elt<T>::ld4(dy + (size_t)(p + g.lanes) * dy_ld + dy_off + vch * 4);
                f32x4 x0 = {0, 0, 0, 0}, x1 = {0, 0, 0, 0};
                if (mean) {
                    x0 = Elt<T>::ld4(z + (size_t)p * z_ld + z_off + vch * 4);
                    x1 = Elt<T>::ld4(z + (size_t)(p + g.lanes) * z_ld + z_off + vch * 4);
                }
                accum(d0, x0);
                accum(d1, x1);
                run += 2;
                if (run >= 64) {
#pragma unroll
                    for (int e = 0; e < 4; ++e) { s[e] += fs[e]; q[e] += fq[e]; fs[e] = 0.f; fq[e] = 0.f; }
                    run = 0;
                }
            }
            for (; p < p1; p += g.lanes) {
                const f32x4 d = Elt<T>::ld4(dy + (size_t)p * dy_ld + dy_off + vch * 4);
                f32x4 x = {0, 0, 0, 0};
                if (mean) x = Elt<T>::ld4(z + (size_t)p * z_ld + z_off + vch * 4);
                accum(d, x);
            }
#pragma unroll
            for (int e = 0; e < 4; ++e) { s[e] += fs[e]; q[e] += fq[e]; }
        }
#pragma unroll
        for (int e = 0; e < 4; ++e) {
            __syncthreads();
            red[t][0] = s[e];
            red[t][1] = q[e];
            __syncthreads();
            if (lane == 0 && vch < c4) {
                double a = 0, b = 0;
                for (int l = 0; l < g.lanes; ++l) { a += red[l * g.vc + v][0]; b += red[l * g.vc + v][1]; }
                double* dst = partial + ((size_t)blockIdx.x * c + vch * 4 + e) * 2;
                dst[0] = a;
                dst[1] = b;
            }
        }
    }
}

// dbeta/dgamma (or dbias) + per-channel coefficients for the apply pass:
//   dz = k0 * (du - mean(du) - zhat * mean(du*zhat)),  k0 = gamma*invstd,  zhat = (z - mean)*invstd
__global__ void bn_bwd_finalize(const double* __restrict__ partial, int nblk, int m, int c, const float* __restrict__ gamma,
                                const float* __restrict__ mean, const float* __restrict__ invstd, float* __restrict__ dgamma,
                                float* __restrict__ dbeta, float* __restrict__ coef) {
    int ch;
    double s, q;
    if (!reduce_partials(partial, nblk, c, &ch, &s, &q)) return;
    dbeta[ch] = (float)s;
    if (!gamma) return;
    dgamma[ch] = (float)q;
    coef[ch * 3 + 0] = gamma[ch] * invstd[ch];          // k0
    coef[ch * 3 + 1] = (float)(s / m);                  // mean(du)
    coef[ch * 3 + 2] = (float)(q / m);                  // mean(du * zhat)
}

template <typename T>
__global__ __launch_bounds__(256) void bn_bwd_apply(const typename Elt<T>::S* __restrict__ dy, int dy_ld, int dy_off,
                                                    const typename Elt<T>::S* __restrict__ z, int z_ld, int z_off,
                                                    const float* __restrict__ mean, const float* __restrict__ invstd,
                                                    const float* __restrict__ scale, const float* __restrict__ shift,
                                                    const float* __restrict__ coef, typename Elt<T>::S* __restrict__ dz, int dz_ld,
                                                    int dz_off, long long m, int c, int act) {
    const int c4 = c >> 2;
    const long long total = m * c4;
    for (long long i = blockIdx.x * 256LL + threadIdx.x; i < total; i += (long long)gridDim.x * 256) {
        const long long p = i / c4;
        const int ch = (int)(i - p * c4) * 4;
        const f32x4 d = Elt<T>::ld4(dy + (size_t)p * dy_ld + dy_off + ch);
        const f32x4 x = Elt<T>::ld4(z + (size_t)p * z_ld + z_off + ch);
        const f32x4 sc = *reinterpret_cast<const f32x4*>(scale + ch);
        const f32x4 sh = *reinterpret_cast<const f32x4*>(shift + ch);
        const f32x4 mu = *reinterpret_cast<const f32x4*>(mean + ch);
        const f32x4 is = *reinterpret_cast<const f32x4*>(invstd + ch);
        f32x4 o;
#pragma unroll
        for (int e = 0; e < 4; ++e) {
            const float xc = x[e] - mu[e];
            const float du = d[e] * act_grad(xc * sc[e] + sh[e], act);
            o[e] = coef[(ch + e) * 3] * (du - coef[(ch + e) * 3 + 1] - xc * is[e] * coef[(ch + e) * 3 + 2]);
        }
        Elt<T>::st4(dz + (size_t)p * dz_ld + dz_off + ch, o);
    }
}

// gradient of nn.Upsample(scale_factor=2, nearest): each source pixel sums its 2x2 destinations
template <typename T>
__global__ __launch_bounds__(256) void upsample2x_bwd_kernel(const typename Elt<T>::S* __restrict__ dup, int d_ld, int d_off,
                                                             typename Elt<T>::S* __restrict__ dx, int x_ld, int x_off, long long m,
                                                             int c, int Ho, int Wo) {
    const int c4 = c >> 2;
    const long long total = m * c4;
    for (long long i = blockIdx.x * 256LL + threadIdx.x; i < total; i += (long long)gridDim.x * 256) {
        const long long p = i / c4;
        const int ch = (int)(i - p * c4) * 4;
        const long long hw = (long long)Ho * Wo;
        const long long img = p / hw;
        const int rem = (int)(p - img * hw);
        const int ho = rem / Wo, wo = rem - ho * Wo;
        const int W2 = 2 * Wo;
        const typename Elt<T>::S* s = dup + ((size_t)(img * 2 * Ho + 2 * ho) * W2 + 2 * wo) * d_ld + d_off + ch;
        f32x4 v = Elt<T>::ld4(s);
        v += Elt<T>::ld4(s + d_ld);
        v += Elt<T>::ld4(s + (size_t)W2 * d_ld);
        v += Elt<T>::ld4(s + (size_t)(W2 + 1) * d_ld);
        Elt<T>::st4(dx + (size_t)p * x_ld + x_off + ch, v);
    }
}

static int red_blocks(int m, int* pix_per_block) {
    int nblk = (m + 255) / 256;
    if (nblk > 2048) nblk = 2048;
    if (nblk < 1) nblk = 1;
    *pix_per_block = (m + nblk - 1) / nblk;
    return (m + *pix_per_block - 1) / *pix_per_block;
}

static int ew_grid(long long total) {
    long long b = (total + 255) / 256;
    return (int)(b < 1 ? 1 : (b > 8192 ? 8192 : b));
}

}  // namespace yolo

using namespace yolo;

extern "C" {

size_t yolo_bn_workspace_bytes(int m, int c) {
    if (m <= 0 || c <= 0) return 0;
    int ppb;
    const int nblk = red_blocks(m, &ppb);
    return (size_t)nblk * c * 2 * sizeof(double) + (size_t)c * 3 * sizeof(float);
}

int yolo_bn_stats(const void* z, int m, int c, int ld, int off, const float* gamma, const float* beta, float momentum,
                  float eps, float* running_mean, float* running_var, float* mean, float* invstd, float* scale,
                  float* shift, int dtype, void* workspace, size_t workspace_bytes, void* stream) {
    if (!z || !gamma || !beta || !mean || !invstd || !scale || !shift || !workspace) return fail(YOLO_ERR_ARG, "bn_stats: null pointer");
    if (m <= 0 || c <= 0 || (c & 3) || (ld & 3) || (off & 3) || ld < c) return fail(YOLO_ERR_ARG, "bn_stats: c/ld/off must be multiples of 4");
    if (workspace_bytes < yolo_bn_workspace_bytes(m, c)) return fail(YOLO_ERR_WORKSPACE, "bn_stats: workspace too small");
    int ppb;
    const int nblk = red_blocks(m, &ppb);
    hipStream_t s = (hipStream_t)stream;
    YOLO_DISPATCH_DTYPE(dtype, "bn_stats",
        hipLaunchKernelGGL(bn_stats_partial<T>, dim3(nblk), dim3(256), 0, s, (const Elt<T>::S*)z, m, c, ld, off, ppb, (double*)workspace));
    int rc = check_launch("bn_stats_partial");
    if (rc) return rc;
    hipLaunchKernelGGL(bn_stats_finalize, dim3(ceil_div(c, 16)), dim3(256), 0, s, (const double*)workspace, nblk, m, c, momentum,
                       eps, gamma, beta, running_mean, running_var, mean, invstd, scale, shift);
    return check_launch("bn_stats_finalize");
}

int yolo_bn_act_fwd(const void* z, int z_ld, int z_off, const float* mean, const float* scale, const float* shift, const void* residual,
                    int r_ld, int r_off, void* y, int y_ld, int y_off, int n, int h, int w, int c, int act, int out_mode,
                    int dtype, int32_t* nan_flag, void* stream) {
    if (!z || !scale || !shift || !y) return fail(YOLO_ERR_ARG, "bn_act_fwd: null pointer");
    if ((c & 3) || (z_ld & 3) || (z_off & 3) || (y_ld & 3) || (y_off & 3) || (residual && ((r_ld & 3) || (r_off & 3))))
        return fail(YOLO_ERR_ARG, "bn_act_fwd: channel counts / strides must be multiples of 4");
    if (out_mode != YOLO_OUT_NHWC && out_mode != YOLO_OUT_UPSAMPLE2X) return fail(YOLO_ERR_ARG, "bn_act_fwd: out_mode");
    const long long m = (long long)n * h * w;
    YOLO_DISPATCH_DTYPE(dtype, "bn_act_fwd",
        hipLaunchKernelGGL(bn_act_fwd_kernel<T>, dim3(ew_grid(m * (c / 4))), dim3(256), 0, (hipStream_t)stream, (const Elt<T>::S*)z, z_ld,
                           z_off, mean, scale, shift, (const Elt<T>::S*)residual, r_ld, r_off, (Elt<T>::S*)y, y_ld, y_off, m, c, h, w, act,
                           out_mode, nan_flag));
    return check_launch("bn_act_fwd");
}

int yolo_bn_act_bwd(const void* dy, int dy_ld, int dy_off, const void* z, int z_ld, int z_off, const float* gamma,
                    const float* mean, const float* invstd, const float* scale, const float* shift, int m, int c, int act,
                    float* dgamma, float* dbeta, void* dz, int dz_ld, int dz_off, int dtype, void* workspace, size_t workspace_bytes,
                    void* stream) {
    if (!dy || !dbeta || !workspace) return fail(YOLO_ERR_ARG, "bn_act_bwd: null pointer");
    if (gamma && (!z || !mean || !invstd || !scale || !shift || !dgamma || !dz)) return fail(YOLO_ERR_ARG, "bn_act_bwd: null pointer");
    if (m <= 0 || c <= 0 || (c & 3) || (dy_ld & 3) || (dy_off & 3)) return fail(YOLO_ERR_ARG, "bn_act_bwd: c/ld/off must be multiples of 4");
    if (workspace_bytes < yolo_bn_workspace_bytes(m, c)) return fail(YOLO_ERR_WORKSPACE, "bn_act_bwd: workspace too small");
    int ppb;
    const int nblk = red_blocks(m, &ppb);
    hipStream_t s = (hipStream_t)stream;
    double* part = (double*)workspace;
    float* coef = (float*)((char*)workspace + (size_t)nblk * c * 2 * sizeof(double));
    YOLO_DISPATCH_DTYPE(dtype, "bn_act_bwd",
        hipLaunchKernelGGL(bn_bwd_partial<T>, dim3(nblk), dim3(256), 0, s, (const Elt<T>::S*)dy, dy_ld, dy_off, (const Elt<T>::S*)z, z_ld, z_off,
                           gamma ? mean : nullptr, invstd, scale, shift, m, c, act, ppb, part));
    int rc = check_launch("bn_bwd_partial");
    if (rc) return rc;
    hipLaunchKernelGGL(bn_bwd_finalize, dim3(ceil_div(c, 16)), dim3(256), 0, s, part, nblk, m, c, gamma, mean, invstd, dgamma, dbeta, coef);
    rc = check_launch("bn_bwd_finalize");
    if (rc || !gamma) return rc;
    YOLO_DISPATCH_DTYPE(dtype, "bn_act_bwd",
        hipLaunchKernelGGL(bn_bwd_apply<T>, dim3(ew_grid((long long)m * (c / 4))), dim3(256), 0, s, (const Elt<T>::S*)dy, dy_ld, dy_off,
                           (const Elt<T>::S*)z, z_ld, z_off, mean, invstd, scale, shift, coef, (Elt<T>::S*)dz, dz_ld, dz_off, (long long)m, c, act));
    return check_launch("bn_bwd_apply");
}

int yolo_upsample2x_bwd(const void* dup, int d_ld, int d_off, void* dx, int x_ld, int x_off, int n, int h, int w, int c,
                        int dtype, void* stream) {
    if (!dup || !dx || (c & 3) || (d_ld & 3) || (d_off & 3) || (x_ld & 3) || (x_off & 3)) return fail(YOLO_ERR_ARG, "upsample2x_bwd: bad arguments");
    const long long m = (long long)n * h * w;
    YOLO_DISPATCH_DTYPE(dtype, "upsample2x_bwd",
        hipLaunchKernelGGL(upsample2x_bwd_kernel<T>, dim3(ew_grid(m * (c / 4))), dim3(256), 0, (hipStream_t)stream, (const Elt<T>::S*)dup, d_ld,
                           d_off, (Elt<T>::S*)dx, x_ld, x_off, m, c, h, w));
    return check_launch("upsample2x_bwd");
}

}  // extern "C"
