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


// VN consecutive per-channel parameters (fp32) as 16-byte loads
template <int VN>
__device__ __forceinline__ void ldp(const float* __restrict__ p, float (&v)[VN]) {
#pragma unroll
    for (int k = 0; k < VN / 4; ++k) {
        const f32x4 r = *reinterpret_cast<const f32x4*>(p + 4 * k);
        v[4 * k] = r[0]; v[4 * k + 1] = r[1]; v[4 * k + 2] = r[2]; v[4 * k + 3] = r[3];
    }
}

// Thread layout shared by the reductions: every thread moves 16-byte vectors (VN = 4 fp32 or 8 halfs);
// CV = C/VN vector channels; thread t owns vector channel t % VC (VC = min(CV, 256)) and pixel lane
// t / VC; a block covers every channel of its pixel range.
struct RedGeom { int vc, lanes, passes; };
constexpr int RED_MLP = 8;          // loads in flight per thread in the reductions
constexpr int RED_RUN = 16;         // pixels per thread and block (the fp32 run in front of the fp64 tree)
__device__ __forceinline__ RedGeom red_geom(int cv) {
    RedGeom g;
    g.vc = cv < 256 ? cv : 256;
    g.lanes = 256 / g.vc;
    g.passes = (cv + g.vc - 1) / g.vc;
    return g;
}

// fixed-order reduction over the pixel lanes of a block through LDS: every thread parks its 2 VN fp64 sums (planar, so the
// stores and the reads below are conflict-free), ONE barrier, then the vc * 2 VN outputs are spread over all 256 threads,
// each adding its `lanes` terms in lane order. (One component at a time with only the lane-0 threads adding cost 2 VN
// barrier pairs per block: a third of the kernel on the 20-40 MB layers.)
template <int VN>
__device__ __forceinline__ void block_reduce_store(double (&s)[VN], double (&q)[VN], const RedGeom& g, int t,
                                                   int vch0, int cv, double* __restrict__ dst /* partial + blk*c*2 */) {
    __shared__ double red[2 * VN][256];
    __syncthreads();                                    // the previous pass has finished reading
#pragma unroll
    for (int e = 0; e < VN; ++e) {
        red[2 * e][t] = s[e];
        red[2 * e + 1][t] = q[e];
    }
    __syncthreads();
    const int outs = g.vc * 2 * VN;
    for (int o = t; o < outs; o += 256) {
        const int v = o % g.vc, k = o / g.vc;           // k = 2 e + {0: first sum, 1: second sum}
        if (vch0 + v >= cv) continue;
        double a = 0;
        for (int l = 0; l < g.lanes; ++l) a += red[k][l * g.vc + v];
        dst[((size_t)(vch0 + v) * VN + (k >> 1)) * 2 + (k & 1)] = a;
    }
}

// partial[blk][c][2] (double): sum z, sum z^2 over the block's pixel range
template <typename T, bool LONG>
__global__ __launch_bounds__(256) void bn_stats_partial(const typename Elt<T>::S* __restrict__ z, int m, int c, int ld, int off,
                                                        int pix_per_block, double* __restrict__ partial) {
    constexpr int VN = Vec16<T>::VN;
    const int cv = c / VN;
    const RedGeom g = red_geom(cv);
    const int t = threadIdx.x;
    const int v = t % g.vc, lane = t / g.vc;
    const int p0 = blockIdx.x * pix_per_block;
    const int p1 = p0 + pix_per_block < m ? p0 + pix_per_block : m;
    for (int pass = 0; pass < g.passes; ++pass) {
        const int vch = pass * g.vc + v;
        double s[VN], q[VN];
#pragma unroll
        for (int e = 0; e < VN; ++e) { s[e] = 0; q[e] = 0; }
        const bool own = vch < cv && lane < g.lanes;
        if (own) {
            float fs[VN], fq[VN];
#pragma unroll
            for (int e = 0; e < VN; ++e) { fs[e] = 0.f; fq[e] = 0.f; }
            int run = 0;
            const typename Elt<T>::S* zp = z + off + vch * VN;
            int p = p0 + lane;
            // RED_MLP independent 16-byte loads in flight per thread (one-at-a-time left the kernel at ~45 % of the achievable
            // HBM rate, four at ~2.7 TB/s with the 1-2 blocks per CU of these grids: latency-bound, not bandwidth-bound)
            for (; p + (RED_MLP - 1) * g.lanes < p1; p += RED_MLP * g.lanes) {
                float x[RED_MLP][VN];
#pragma unroll
                for (int u = 0; u < RED_MLP; ++u) Vec16<T>::ld(zp + (size_t)(p + u * g.lanes) * ld, x[u]);
#pragma unroll
                for (int u = 0; u < RED_MLP; ++u)
#pragma unroll
                    for (int e = 0; e < VN; ++e) { fs[e] += x[u][e]; fq[e] += x[u][e] * x[u][e]; }
                if constexpr (LONG) {                    // only huge inputs: flush fp32 runs into fp64 (costs 4*VN VGPRs)
                    run += RED_MLP;
                    if (run >= 64) {
#pragma unroll
                        for (int e = 0; e < VN; ++e) { s[e] += fs[e]; q[e] += fq[e]; fs[e] = 0.f; fq[e] = 0.f; }
                        run = 0;
                    }
                }
            }
            for (; p < p1; p += g.lanes) {
                float x[VN];
                Vec16<T>::ld(zp + (size_t)p * ld, x);
#pragma unroll
                for (int e = 0; e < VN; ++e) { fs[e] += x[e]; fq[e] += x[e] * x[e]; }
            }
#pragma unroll
            for (int e = 0; e < VN; ++e) { s[e] += fs[e]; q[e] += fq[e]; }
        }
        block_reduce_store<VN>(s, q, g, t, pass * g.vc, cv, partial + (size_t)blockIdx.x * c * 2);
    }
}

// Fixed-order reduction of partial[nblk][c][2] for the finalize kernels: a block handles FIN_CH channels with
// 256 / FIN_CH lanes each; lane l sums blocks l, l + lanes, ... (independent 16-byte loads, pipelined), then the lanes are
// added in order through LDS. (One thread per channel walking 1024 partials took ~65 us; 16 channels x 16 lanes per block
// left a c = 256 layer on 16 CUs with ~40 dependent additions per thread: 12 us for 2.8 MB.)
constexpr int FIN_CH = 4;
constexpr int FIN_LANES = 256 / FIN_CH;
typedef double f64x2 __attribute__((ext_vector_type(2)));
__device__ __forceinline__ bool reduce_partials(const double* __restrict__ partial, int nblk, int c, int* ch_out,
                                                double* s_out, double* q_out) {
    __shared__ double red[FIN_LANES][FIN_CH][2];
    const int lc = threadIdx.x % FIN_CH, lane = threadIdx.x / FIN_CH;
    const int ch = blockIdx.x * FIN_CH + lc;
    double s = 0, q = 0;
    if (ch < c) {
        const double* src = partial + (size_t)ch * 2;
        const size_t step = (size_t)c * 2;
        int b = lane;
        for (; b + 3 * FIN_LANES < nblk; b += 4 * FIN_LANES) {     // four loads in flight (hipcc leaves the plain loop at one
            f64x2 v[4];                                            // load + s_waitcnt vmcnt(0) per iteration); same adding order
#pragma unroll
            for (int u = 0; u < 4; ++u) v[u] = *reinterpret_cast<const f64x2*>(src + (size_t)(b + u * FIN_LANES) * step);
#pragma unroll
            for (int u = 0; u < 4; ++u) { s += v[u][0]; q += v[u][1]; }
        }
        for (; b < nblk; b += FIN_LANES) {
            const f64x2 v = *reinterpret_cast<const f64x2*>(src + (size_t)b * step);
            s += v[0];
            q += v[1];
        }
    }
    red[lane][lc][0] = s;
    red[lane][lc][1] = q;
    __syncthreads();
    if (lane != 0 || ch >= c) return false;
    s = 0; q = 0;
#pragma unroll 4
    for (int l = 0; l < FIN_LANES; ++l) { s += red[l][lc][0]; q += red[l][lc][1]; }   // (fully unrolled, hipcc spilled 64 LDS reads to scratch)
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

// The same finalize from the per-wave partial sums a convolution's epilogue wrote (conv_h16.hip: d_epilogue_stats):
// partial[row][2][ld] fp32, row = a wave's 64 pixels; fp64 across rows in a fixed order (lane l adds rows l, l + 64, ...,
// then the lanes in order), i.e. deterministic like the path above.
__global__ void bn_stats_finalize_rows(const float* __restrict__ partial, int nrows, int ld, int m, int c, float momentum, float eps,
                                       const float* __restrict__ gamma, const float* __restrict__ beta,
                                       float* __restrict__ running_mean, float* __restrict__ running_var,
                                       float* __restrict__ mean, float* __restrict__ invstd, float* __restrict__ scale,
                                       float* __restrict__ shift) {
    __shared__ double red[FIN_LANES][FIN_CH][2];
    const int lc = threadIdx.x % FIN_CH, lane = threadIdx.x / FIN_CH;
    const int ch = blockIdx.x * FIN_CH + lc;
    double s = 0, q = 0;
    if (ch < c) {
        const float* src = partial + ch;
        const size_t step = (size_t)2 * ld;
        int r = lane;
        for (; r + 3 * FIN_LANES < nrows; r += 4 * FIN_LANES) {
            float a[4], b[4];
#pragma unroll
            for (int u = 0; u < 4; ++u) { a[u] = src[(size_t)(r + u * FIN_LANES) * step]; b[u] = src[(size_t)(r + u * FIN_LANES) * step + ld]; }
#pragma unroll
            for (int u = 0; u < 4; ++u) { s += a[u]; q += b[u]; }
        }
        for (; r < nrows; r += FIN_LANES) { s += src[(size_t)r * step]; q += src[(size_t)r * step + ld]; }
    }
    red[lane][lc][0] = s;
    red[lane][lc][1] = q;
    __syncthreads();
    if (lane != 0 || ch >= c) return;
    s = 0; q = 0;
#pragma unroll 4
    for (int l = 0; l < FIN_LANES; ++l) { s += red[l][lc][0]; q += red[l][lc][1]; }
    const double mu = s / m;
    double var = q / m - mu * mu;
    if (var < 0) var = 0;
    const float is = 1.0f / sqrtf((float)var + eps);
    mean[ch] = (float)mu;
    invstd[ch] = is;
    scale[ch] = gamma[ch] * is;
    shift[ch] = beta[ch];
    if (running_mean) {
        const double unbiased = m > 1 ? var * m / (m - 1) : var;
        running_mean[ch] = (1.f - momentum) * running_mean[ch] + momentum * (float)mu;
        running_var[ch] = (1.f - momentum) * running_var[ch] + momentum * (float)unbiased;
    }
}

// y = act((z - mean)*scale + shift) [+ residual], same output modes as the conv epilogue. Thread layout of the reductions
// (a thread owns one 16-byte channel vector and walks pixels): the per-channel parameters are loaded ONCE per thread, no
// index division per element, AP_MLP pixels in flight. (The first version was a flat grid-stride loop over (pixel, vector)
// pairs: 3 parameter vectors + a 64-bit division per 16-byte load, 4.3 TB/s.)
constexpr int AP_MLP = 4;
template <typename T, int ACT>
__global__ __launch_bounds__(256) void bn_act_fwd_kernel(const typename Elt<T>::S* __restrict__ z, int z_ld, int z_off,
                                                         const float* __restrict__ mean,
                                                         const float* __restrict__ scale, const float* __restrict__ shift,
                                                         const typename Elt<T>::S* __restrict__ res, int r_ld, int r_off,
                                                         typename Elt<T>::S* __restrict__ y, int y_ld, int y_off, int m, int c,
                                                         int Ho, int Wo, int out_mode, int pix_per_block, int* nan_flag) {
    constexpr int VN = Vec16<T>::VN;
    const int cv = c / VN;
    const RedGeom g = red_geom(cv);
    const int t = threadIdx.x;
    const int v = t % g.vc, lane = t / g.vc;
    const int p0 = blockIdx.x * pix_per_block;
    const int p1 = p0 + pix_per_block < m ? p0 + pix_per_block : m;
    bool bad = false;
    for (int pass = 0; pass < g.passes; ++pass) {
        const int vch = pass * g.vc + v;
        if (vch >= cv || lane >= g.lanes) continue;
        const int ch = vch * VN;
        float mu[VN], sc[VN], sh[VN];
        ldp<VN>(scale + ch, sc);
        ldp<VN>(shift + ch, sh);
        if (mean) ldp<VN>(mean + ch, mu);
        else {
#pragma unroll
            for (int e = 0; e < VN; ++e) mu[e] = 0.f;
        }
        const typename Elt<T>::S* zp = z + z_off + ch;
        const typename Elt<T>::S* rp = res ? res + r_off + ch : nullptr;
        auto emit = [&](int p, const float (&x)[VN], const float (&r)[VN]) {
            float o[VN];
#pragma unroll
            for (int e = 0; e < VN; ++e) {
                o[e] = act_c<ACT>((x[e] - mu[e]) * sc[e] + sh[e]) + r[e];
                bad |= (o[e] != o[e]);
            }
            if (out_mode == YOLO_OUT_UPSAMPLE2X) {
                const int hw = Ho * Wo;
                const int img = p / hw;
                const int rem = p - img * hw;
                const int ho = rem / Wo, wo = rem - ho * Wo;
                const int W2 = 2 * Wo;
                typename Elt<T>::S* d = y + ((size_t)(img * 2 * Ho + 2 * ho) * W2 + 2 * wo) * y_ld + y_off + ch;
                Vec16<T>::st(d, o);
                Vec16<T>::st(d + y_ld, o);
                Vec16<T>::st(d + (size_t)W2 * y_ld, o);
                Vec16<T>::st(d + (size_t)(W2 + 1) * y_ld, o);
            } else {
                Vec16<T>::st(y + (size_t)p * y_ld + y_off + ch, o);
            }
        };
        int p = p0 + lane;
        for (; p + (AP_MLP - 1) * g.lanes < p1; p += AP_MLP * g.lanes) {
            float x[AP_MLP][VN], r[AP_MLP][VN];
#pragma unroll
            for (int u = 0; u < AP_MLP; ++u) Vec16<T>::ld(zp + (size_t)(p + u * g.lanes) * z_ld, x[u]);
            if (rp) {
#pragma unroll
                for (int u = 0; u < AP_MLP; ++u) Vec16<T>::ld(rp + (size_t)(p + u * g.lanes) * r_ld, r[u]);
            } else {
#pragma unroll
                for (int u = 0; u < AP_MLP; ++u)
#pragma unroll
                    for (int e = 0; e < VN; ++e) r[u][e] = 0.f;
            }
#pragma unroll
            for (int u = 0; u < AP_MLP; ++u) emit(p + u * g.lanes, x[u], r[u]);
        }
        for (; p < p1; p += g.lanes) {
            float x[VN], r[VN];
            Vec16<T>::ld(zp + (size_t)p * z_ld, x);
            if (rp) Vec16<T>::ld(rp + (size_t)p * r_ld, r);
            else {
#pragma unroll
                for (int e = 0; e < VN; ++e) r[e] = 0.f;
            }
            emit(p, x, r);
        }
    }
    if (bad && nan_flag) atomicOr(nan_flag, 2);
}

// partial[blk][c][2]: sum du, sum du*zhat.  mean == nullptr: bare conv (du = dy, only sum du is used)
template <typename T, bool LONG, int ACT>
__global__ __launch_bounds__(256) void bn_bwd_partial(const typename Elt<T>::S* __restrict__ dy, int dy_ld, int dy_off,
                                                      const typename Elt<T>::S* __restrict__ z, int z_ld, int z_off,
                                                      const float* __restrict__ mean, const float* __restrict__ invstd,
                                                      const float* __restrict__ scale, const float* __restrict__ shift,
                                                      int m, int c, int pix_per_block, double* __restrict__ partial) {
    constexpr int VN = Vec16<T>::VN;
    const int cv = c / VN;
    const RedGeom g = red_geom(cv);
    const int t = threadIdx.x;
    const int v = t % g.vc, lane = t / g.vc;
    const int p0 = blockIdx.x * pix_per_block;
    const int p1 = p0 + pix_per_block < m ? p0 + pix_per_block : m;
    for (int pass = 0; pass < g.passes; ++pass) {
        const int vch = pass * g.vc + v;
        double s[VN], q[VN];
#pragma unroll
        for (int e = 0; e < VN; ++e) { s[e] = 0; q[e] = 0; }
        if (vch < cv && lane < g.lanes) {
            float mu[VN], is[VN], sc[VN], sh[VN], fs[VN], fq[VN];
#pragma unroll
            for (int e = 0; e < VN; ++e) { mu[e] = 0.f; is[e] = 1.f; sc[e] = 1.f; sh[e] = 0.f; fs[e] = 0.f; fq[e] = 0.f; }
            if (mean) {
                ldp<VN>(mean + vch * VN, mu);
                ldp<VN>(invstd + vch * VN, is);
                ldp<VN>(scale + vch * VN, sc);
                ldp<VN>(shift + vch * VN, sh);
            }
            int run = 0;
            auto accum = [&](const float (&d)[VN], const float (&x)[VN]) {
#pragma unroll
                for (int e = 0; e < VN; ++e) {
                    const float du = mean ? d[e] * act_grad_c<ACT>((x[e] - mu[e]) * sc[e] + sh[e]) : d[e];
                    fs[e] += du;
                    fq[e] += du * ((x[e] - mu[e]) * is[e]);
                }
            };
            const typename Elt<T>::S* dp = dy + dy_off + vch * VN;
            const typename Elt<T>::S* zp = z + z_off + vch * VN;
            int p = p0 + lane;
            constexpr int NP = RED_MLP / 2;                       // RED_MLP loads in flight (NP pixels x {dy, z})
            for (; p + (NP - 1) * g.lanes < p1; p += NP * g.lanes) {
                float d[NP][VN], x[NP][VN];
#pragma unroll
                for (int u = 0; u < NP; ++u) Vec16<T>::ld(dp + (size_t)(p + u * g.lanes) * dy_ld, d[u]);
                if (mean) {
#pragma unroll
                    for (int u = 0; u < NP; ++u) Vec16<T>::ld(zp + (size_t)(p + u * g.lanes) * z_ld, x[u]);
                } else {
#pragma unroll
                    for (int u = 0; u < NP; ++u)
#pragma unroll
                        for (int e = 0; e < VN; ++e) x[u][e] = 0.f;
                }
#pragma unroll
                for (int u = 0; u < NP; ++u) accum(d[u], x[u]);
                if constexpr (LONG) {
                    run += NP;
                    if (run >= 64) {
#pragma unroll
                        for (int e = 0; e < VN; ++e) { s[e] += fs[e]; q[e] += fq[e]; fs[e] = 0.f; fq[e] = 0.f; }
                        run = 0;
                    }
                }
            }
            for (; p < p1; p += g.lanes) {
                float d[VN], x[VN];
                Vec16<T>::ld(dp + (size_t)p * dy_ld, d);
                if (mean) Vec16<T>::ld(zp + (size_t)p * z_ld, x);
                else {
#pragma unroll
                    for (int e = 0; e < VN; ++e) x[e] = 0.f;
                }
                accum(d, x);
            }
#pragma unroll
            for (int e = 0; e < VN; ++e) { s[e] += fs[e]; q[e] += fq[e]; }
        }
        block_reduce_store<VN>(s, q, g, t, pass * g.vc, cv, partial + (size_t)blockIdx.x * c * 2);
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
    coef[ch] = gamma[ch] * invstd[ch];                  // k0          (planar [3][c]: vector loads in the apply pass)
    coef[c + ch] = (float)(s / m);                      // mean(du)
    coef[2 * c + ch] = (float)(q / m);                  // mean(du * zhat)
}

// The same from the per-wave sums an input-gradient convolution's epilogue wrote (conv_h16.hip: d_epilogue_bstats):
// partial[row][2][ld] fp32 = sum(du), sum(du * (z - mean)); fp64 across rows in the fixed order of bn_stats_finalize_rows.
__global__ void bn_bwd_finalize_rows(const float* __restrict__ partial, int nrows, int ld, int m, int c, const float* __restrict__ gamma,
                                     const float* __restrict__ invstd, float* __restrict__ dgamma, float* __restrict__ dbeta,
                                     float* __restrict__ coef) {
    __shared__ double red[FIN_LANES][FIN_CH][2];
    const int lc = threadIdx.x % FIN_CH, lane = threadIdx.x / FIN_CH;
    const int ch = blockIdx.x * FIN_CH + lc;
    double s = 0, q = 0;
    if (ch < c) {
        const float* src = partial + ch;
        const size_t step = (size_t)2 * ld;
        int r = lane;
        for (; r + 3 * FIN_LANES < nrows; r += 4 * FIN_LANES) {
            float a[4], b[4];
#pragma unroll
            for (int u = 0; u < 4; ++u) { a[u] = src[(size_t)(r + u * FIN_LANES) * step]; b[u] = src[(size_t)(r + u * FIN_LANES) * step + ld]; }
#pragma unroll
            for (int u = 0; u < 4; ++u) { s += a[u]; q += b[u]; }
        }
        for (; r < nrows; r += FIN_LANES) { s += src[(size_t)r * step]; q += src[(size_t)r * step + ld]; }
    }
    red[lane][lc][0] = s;
    red[lane][lc][1] = q;
    __syncthreads();
    if (lane != 0 || ch >= c) return;
    s = 0; q = 0;
#pragma unroll 4
    for (int l = 0; l < FIN_LANES; ++l) { s += red[l][lc][0]; q += red[l][lc][1]; }
    q *= (double)invstd[ch];                            // sum(du * zhat)
    dbeta[ch] = (float)s;
    dgamma[ch] = (float)q;
    coef[ch] = gamma[ch] * invstd[ch];
    coef[c + ch] = (float)(s / m);
    coef[2 * c + ch] = (float)(q / m);
}

template <typename T, int ACT>
__global__ __launch_bounds__(256) void bn_bwd_apply(const typename Elt<T>::S* __restrict__ dy, int dy_ld, int dy_off,
                                                    const typename Elt<T>::S* __restrict__ z, int z_ld, int z_off,
                                                    const float* __restrict__ mean, const float* __restrict__ invstd,
                                                    const float* __restrict__ scale, const float* __restrict__ shift,
                                                    const float* __restrict__ coef, typename Elt<T>::S* __restrict__ dz, int dz_ld,
                                                    int dz_off, int m, int c, int pix_per_block) {
    constexpr int VN = Vec16<T>::VN;
    const int cv = c / VN;
    const RedGeom g = red_geom(cv);
    const int t = threadIdx.x;
    const int v = t % g.vc, lane = t / g.vc;
    const int p0 = blockIdx.x * pix_per_block;
    const int p1 = p0 + pix_per_block < m ? p0 + pix_per_block : m;
    for (int pass = 0; pass < g.passes; ++pass) {
        const int vch = pass * g.vc + v;
        if (vch >= cv || lane >= g.lanes) continue;
        const int ch = vch * VN;
        float mu[VN], is[VN], sc[VN], sh[VN], k0[VN], k1[VN], k2[VN];      // once per thread (see bn_act_fwd_kernel)
        ldp<VN>(mean + ch, mu);
        ldp<VN>(invstd + ch, is);
        ldp<VN>(scale + ch, sc);
        ldp<VN>(shift + ch, sh);
        ldp<VN>(coef + ch, k0);
        ldp<VN>(coef + c + ch, k1);
        ldp<VN>(coef + 2 * c + ch, k2);
        const typename Elt<T>::S* dp = dy + dy_off + ch;
        const typename Elt<T>::S* zp = z + z_off + ch;
        typename Elt<T>::S* op = dz + dz_off + ch;
        auto emit = [&](int p, const float (&d)[VN], const float (&x)[VN]) {
            float o[VN];
#pragma unroll
            for (int e = 0; e < VN; ++e) {
                const float xc = x[e] - mu[e];
                const float du = d[e] * act_grad_c<ACT>(xc * sc[e] + sh[e]);
                o[e] = k0[e] * (du - k1[e] - xc * is[e] * k2[e]);
            }
            Vec16<T>::st(op + (size_t)p * dz_ld, o);
        };
        constexpr int NP = AP_MLP / 2;
        int p = p0 + lane;
        for (; p + (NP - 1) * g.lanes < p1; p += NP * g.lanes) {
            float d[NP][VN], x[NP][VN];
#pragma unroll
            for (int u = 0; u < NP; ++u) {
                Vec16<T>::ld(dp + (size_t)(p + u * g.lanes) * dy_ld, d[u]);
                Vec16<T>::ld(zp + (size_t)(p + u * g.lanes) * z_ld, x[u]);
            }
#pragma unroll
            for (int u = 0; u < NP; ++u) emit(p + u * g.lanes, d[u], x[u]);
        }
        for (; p < p1; p += g.lanes) {
            float d[VN], x[VN];
            Vec16<T>::ld(dp + (size_t)p * dy_ld, d);
            Vec16<T>::ld(zp + (size_t)p * z_ld, x);
            emit(p, d, x);
        }
    }
}

// gradient of nn.Upsample(scale_factor=2, nearest): each source pixel sums its 2x2 destinations
template <typename T>
__global__ __launch_bounds__(256) void upsample2x_bwd_kernel(const typename Elt<T>::S* __restrict__ dup, int d_ld, int d_off,
                                                             typename Elt<T>::S* __restrict__ dx, int x_ld, int x_off, long long m,
                                                             int c, int Ho, int Wo) {
    constexpr int VN = Vec16<T>::VN;
    const int cv = c / VN;
    const long long total = m * cv;
    for (long long i = blockIdx.x * 256LL + threadIdx.x; i < total; i += (long long)gridDim.x * 256) {
        const long long p = i / cv;
        const int ch = (int)(i - p * cv) * VN;
        const long long hw = (long long)Ho * Wo;
        const long long img = p / hw;
        const int rem = (int)(p - img * hw);
        const int ho = rem / Wo, wo = rem - ho * Wo;
        const int W2 = 2 * Wo;
        const typename Elt<T>::S* s = dup + ((size_t)(img * 2 * Ho + 2 * ho) * W2 + 2 * wo) * d_ld + d_off + ch;
        float a[VN], b[VN], cc[VN], d[VN];
        Vec16<T>::ld(s, a);
        Vec16<T>::ld(s + d_ld, b);
        Vec16<T>::ld(s + (size_t)W2 * d_ld, cc);
        Vec16<T>::ld(s + (size_t)(W2 + 1) * d_ld, d);
#pragma unroll
        for (int e = 0; e < VN; ++e) a[e] = (a[e] + b[e]) + (cc[e] + d[e]);
        Vec16<T>::st(dx + (size_t)p * x_ld + x_off + ch, a);
    }
}

// Blocks of the reductions: each thread accumulates a short fp32 run (RED_RUN pixels) before the fp64 tree, so the
// kernels need no per-thread fp64 state (that cost 3 of 8 waves/SIMD); `*long_run` selects the variant with
// in-loop fp64 flushes for inputs so large that the block cap makes the runs long.
static int red_blocks(int m, int c, int vn, int* pix_per_block, bool* long_run) {
    const int cv = c / vn;
    const int lanes = 256 / (cv < 256 ? cv : 256);
    int ppb = RED_RUN * lanes;
    int nblk = (m + ppb - 1) / ppb;
    if (nblk > 4096) nblk = 4096;
    if (nblk < 1) nblk = 1;
    ppb = (m + nblk - 1) / nblk;
    *pix_per_block = ppb;
    if (long_run) *long_run = ppb / lanes > 256;
    return (m + ppb - 1) / ppb;
}

// Blocks of the apply passes: AP_RUN pixels per thread, at most 8192 blocks
constexpr int AP_RUN = 8;
static int ap_blocks(int m, int c, int vn, int* pix_per_block) {
    const int cv = c / vn;
    const int lanes = 256 / (cv < 256 ? cv : 256);
    int ppb = AP_RUN * lanes;
    int nblk = (m + ppb - 1) / ppb;
    if (nblk > 8192) nblk = 8192;
    if (nblk < 1) nblk = 1;
    ppb = (m + nblk - 1) / nblk;
    *pix_per_block = ppb;
    return (m + ppb - 1) / ppb;
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
    const int c4 = (c + 3) & ~3;
    const int nblk = red_blocks(m, c4, 4, &ppb, nullptr);      // fp32 vectors give the most blocks: upper bound for every dtype
    return (size_t)nblk * c * 2 * sizeof(double) + (size_t)c * 3 * sizeof(float);
}

int yolo_bn_stats(const void* z, int m, int c, int ld, int off, const float* gamma, const float* beta, float momentum,
                  float eps, float* running_mean, float* running_var, float* mean, float* invstd, float* scale,
                  float* shift, int dtype, void* workspace, size_t workspace_bytes, void* stream) {
    if (!z || !gamma || !beta || !mean || !invstd || !scale || !shift || !workspace) return fail(YOLO_ERR_ARG, "bn_stats: null pointer");
    const int vn = dtype == YOLO_F32 ? 4 : 8;
    if (m <= 0 || c <= 0 || (c % vn) || (ld % vn) || (off % vn) || ld < c) return fail(YOLO_ERR_ARG, "bn_stats: c/ld/off must be multiples of %d", vn);
    if (workspace_bytes < yolo_bn_workspace_bytes(m, c)) return fail(YOLO_ERR_WORKSPACE, "bn_stats: workspace too small");
    int ppb;
    bool lr;
    const int nblk = red_blocks(m, c, vn, &ppb, &lr);
    hipStream_t s = (hipStream_t)stream;
    YOLO_DISPATCH_DTYPE(dtype, "bn_stats",
        if (lr) hipLaunchKernelGGL((bn_stats_partial<T, true>), dim3(nblk), dim3(256), 0, s, (const Elt<T>::S*)z, m, c, ld, off, ppb, (double*)workspace);
        else hipLaunchKernelGGL((bn_stats_partial<T, false>), dim3(nblk), dim3(256), 0, s, (const Elt<T>::S*)z, m, c, ld, off, ppb, (double*)workspace));
    int rc = check_launch("bn_stats_partial");
    if (rc) return rc;
    hipLaunchKernelGGL(bn_stats_finalize, dim3(ceil_div(c, FIN_CH)), dim3(256), 0, s, (const double*)workspace, nblk, m, c, momentum,
                       eps, gamma, beta, running_mean, running_var, mean, invstd, scale, shift);
    return check_launch("bn_stats_finalize");
}

/* rows / channel stride of the BatchNorm partial sums yolo_conv_fwd_stats would write for this convolution (0 rows: it has no
 * fused-statistics kernel and the caller runs yolo_bn_stats on z instead) */
int yolo_conv_stats_rows(const yolo_conv_desc* d, int* ld) {
    if (ld) *ld = 0;
    if (!d || d->dtype == YOLO_F32) return 0;
    int rl[2] = {0, 0};
    if (conv_h16_launch_stats(d, nullptr, nullptr, nullptr, nullptr, nullptr, nullptr, nullptr, nullptr, rl, nullptr, nullptr, nullptr) != YOLO_OK) {
        return 0;
    }
    if (ld) *ld = rl[1];
    return rl[0];
}

/* the same question for an input-gradient launch (a stride-1 convolution with the flipped weights, identity epilogue, optional
 * accumulate): rows of BatchNorm-BACKWARD partial sums yolo_conv_dgrad_bstats would write (0: not available) */
int yolo_conv_bstats_rows(const yolo_conv_desc* d, int* ld) {
    if (ld) *ld = 0;
    if (!d || d->dtype == YOLO_F32) return 0;
    int rl[2] = {0, 0};
    const ConvBStats dry = {nullptr, 0, 0, nullptr, nullptr, nullptr, 0};
    if (conv_h16_launch_stats(d, nullptr, nullptr, nullptr, nullptr, nullptr, nullptr, nullptr, nullptr, rl, nullptr, nullptr, &dry) != YOLO_OK) {
        return 0;
    }
    if (ld) *ld = rl[1];
    return rl[0];
}

int yolo_conv_dgrad_bstats(const yolo_conv_desc* d, const void* dz, const void* w_packed, const void* residual, void* dx, const void* z,
                           int z_ld, int z_off, const float* mean, const float* scale, const float* shift, int act, float* stats,
                           size_t stats_bytes, void* stream) {
    if (!d || !dz || !w_packed || !dx || !z || !mean || !scale || !shift || !stats) return fail(YOLO_ERR_ARG, "conv_dgrad_bstats: null pointer");
    if (d->dtype == YOLO_F32) return fail(YOLO_ERR_UNSUPPORTED, "conv_dgrad_bstats: 16-bit convolutions only");
    if (((d->flags & YOLO_FLAG_RESIDUAL) != 0) != (residual != nullptr)) return fail(YOLO_ERR_ARG, "conv_dgrad_bstats: residual pointer and flag disagree");
    const ConvBStats bs = {z, z_ld, z_off, mean, scale, shift, act};
    int rl[2];
    const int rc = conv_h16_launch_stats(d, dz, w_packed, nullptr, nullptr, residual, dx, nullptr, stats, rl, &stats_bytes, (hipStream_t)stream, &bs);
    if (rc == YOLO_OK && rl[0] == 0) return fail(YOLO_ERR_UNSUPPORTED, "conv_dgrad_bstats: no fused-statistics kernel for this convolution");
    return rc;
}

int yolo_conv_fwd_stats(const yolo_conv_desc* d, const void* x, const void* w_packed, void* z, float* stats, size_t stats_bytes,
                        void* stream) {
    if (!d || !x || !w_packed || !z || !stats) return fail(YOLO_ERR_ARG, "conv_fwd_stats: null pointer");
    if (d->dtype == YOLO_F32) return fail(YOLO_ERR_UNSUPPORTED, "conv_fwd_stats: 16-bit convolutions only");
    int rl[2];
    return conv_h16_launch_stats(d, x, w_packed, nullptr, nullptr, nullptr, z, nullptr, stats, rl, &stats_bytes, (hipStream_t)stream, nullptr);
}

int yolo_bn_stats_from_partials(const float* partial, int rows, int ld, int m, int c, const float* gamma, const float* beta,
                                float momentum, float eps, float* running_mean, float* running_var, float* mean, float* invstd,
                                float* scale, float* shift, void* stream) {
    if (!partial || !gamma || !beta || !mean || !invstd || !scale || !shift) return fail(YOLO_ERR_ARG, "bn_stats_from_partials: null pointer");
    if (rows <= 0 || m <= 0 || c <= 0 || ld < c) return fail(YOLO_ERR_ARG, "bn_stats_from_partials: bad shape");
    hipLaunchKernelGGL(bn_stats_finalize_rows, dim3(ceil_div(c, FIN_CH)), dim3(256), 0, (hipStream_t)stream, partial, rows, ld, m, c,
                       momentum, eps, gamma, beta, running_mean, running_var, mean, invstd, scale, shift);
    return check_launch("bn_stats_finalize_rows");
}

int yolo_bn_act_fwd(const void* z, int z_ld, int z_off, const float* mean, const float* scale, const float* shift, const void* residual,
                    int r_ld, int r_off, void* y, int y_ld, int y_off, int n, int h, int w, int c, int act, int out_mode,
                    int dtype, int32_t* nan_flag, void* stream) {
    if (!z || !scale || !shift || !y) return fail(YOLO_ERR_ARG, "bn_act_fwd: null pointer");
    const int vn = dtype == YOLO_F32 ? 4 : 8;
    if ((c % vn) || (z_ld % vn) || (z_off % vn) || (y_ld % vn) || (y_off % vn) || (residual && ((r_ld % vn) || (r_off % vn))))
        return fail(YOLO_ERR_ARG, "bn_act_fwd: channel counts / strides must be multiples of %d", vn);
    if (out_mode != YOLO_OUT_NHWC && out_mode != YOLO_OUT_UPSAMPLE2X) return fail(YOLO_ERR_ARG, "bn_act_fwd: out_mode");
    const long long m = (long long)n * h * w;
    if (m > 0x7fffffffLL) return fail(YOLO_ERR_UNSUPPORTED, "bn_act_fwd: too many pixels");
    int ppb;
    const int nblk = ap_blocks((int)m, c, vn, &ppb);
    YOLO_DISPATCH_DTYPE(dtype, "bn_act_fwd",
        YOLO_SWITCH_ACT(act,
            hipLaunchKernelGGL((bn_act_fwd_kernel<T, ACT>), dim3(nblk), dim3(256), 0, (hipStream_t)stream, (const Elt<T>::S*)z,
                               z_ld, z_off, mean, scale, shift, (const Elt<T>::S*)residual, r_ld, r_off, (Elt<T>::S*)y, y_ld, y_off, (int)m, c, h, w,
                               out_mode, ppb, nan_flag)));
    return check_launch("bn_act_fwd");
}

int yolo_bn_act_bwd(const void* dy, int dy_ld, int dy_off, const void* z, int z_ld, int z_off, const float* gamma,
                    const float* mean, const float* invstd, const float* scale, const float* shift, int m, int c, int act,
                    float* dgamma, float* dbeta, void* dz, int dz_ld, int dz_off, int dtype, void* workspace, size_t workspace_bytes,
                    void* stream) {
    if (!dy || !dbeta || !workspace) return fail(YOLO_ERR_ARG, "bn_act_bwd: null pointer");
    if (gamma && (!z || !mean || !invstd || !scale || !shift || !dgamma || !dz)) return fail(YOLO_ERR_ARG, "bn_act_bwd: null pointer");
    const int vn = dtype == YOLO_F32 ? 4 : 8;
    if (m <= 0 || c <= 0 || (c % vn) || (dy_ld % vn) || (dy_off % vn) || (gamma && ((z_ld % vn) || (z_off % vn) || (dz_ld % vn) || (dz_off % vn))))
        return fail(YOLO_ERR_ARG, "bn_act_bwd: c/ld/off must be multiples of %d", vn);
    if (workspace_bytes < yolo_bn_workspace_bytes(m, c)) return fail(YOLO_ERR_WORKSPACE, "bn_act_bwd: workspace too small");
    int ppb;
    bool lr;
    const int nblk = red_blocks(m, c, vn, &ppb, &lr);
    hipStream_t s = (hipStream_t)stream;
    double* part = (double*)workspace;
    float* coef = (float*)((char*)workspace + (size_t)nblk * c * 2 * sizeof(double));
    YOLO_DISPATCH_DTYPE(dtype, "bn_act_bwd",
        YOLO_SWITCH_ACT(act,
            if (lr) hipLaunchKernelGGL((bn_bwd_partial<T, true, ACT>), dim3(nblk), dim3(256), 0, s, (const Elt<T>::S*)dy, dy_ld, dy_off,
                                       (const Elt<T>::S*)z, z_ld, z_off, gamma ? mean : nullptr, invstd, scale, shift, m, c, ppb, part);
            else hipLaunchKernelGGL((bn_bwd_partial<T, false, ACT>), dim3(nblk), dim3(256), 0, s, (const Elt<T>::S*)dy, dy_ld, dy_off,
                                    (const Elt<T>::S*)z, z_ld, z_off, gamma ? mean : nullptr, invstd, scale, shift, m, c, ppb, part)));
    int rc = check_launch("bn_bwd_partial");
    if (rc) return rc;
    hipLaunchKernelGGL(bn_bwd_finalize, dim3(ceil_div(c, FIN_CH)), dim3(256), 0, s, part, nblk, m, c, gamma, mean, invstd, dgamma, dbeta, coef);
    rc = check_launch("bn_bwd_finalize");
    if (rc || !gamma) return rc;
    int appb;
    const int anblk = ap_blocks(m, c, vn, &appb);
    YOLO_DISPATCH_DTYPE(dtype, "bn_act_bwd",
        YOLO_SWITCH_ACT(act,
            hipLaunchKernelGGL((bn_bwd_apply<T, ACT>), dim3(anblk), dim3(256), 0, s, (const Elt<T>::S*)dy, dy_ld, dy_off,
                               (const Elt<T>::S*)z, z_ld, z_off, mean, invstd, scale, shift, coef, (Elt<T>::S*)dz, dz_ld, dz_off, m, c, appb)));
    return check_launch("bn_bwd_apply");
}

/* yolo_bn_act_bwd with the reduction pass already done by the convolution that wrote dy (yolo_conv_dgrad_bstats): `rows` are
 * its partial sums [nrows][2][rows_ld], followed in the same buffer by room for the 3 c per-channel coefficients */
int yolo_bn_act_bwd_rows(const void* dy, int dy_ld, int dy_off, const void* z, int z_ld, int z_off, const float* gamma,
                         const float* mean, const float* invstd, const float* scale, const float* shift, int m, int c, int act,
                         float* dgamma, float* dbeta, void* dz, int dz_ld, int dz_off, int dtype, float* rows, int nrows, int rows_ld,
                         void* stream) {
    if (!dy || !z || !gamma || !mean || !invstd || !scale || !shift || !dgamma || !dbeta || !dz || !rows) return fail(YOLO_ERR_ARG, "bn_act_bwd_rows: null pointer");
    if (dtype == YOLO_F32) return fail(YOLO_ERR_UNSUPPORTED, "bn_act_bwd_rows: 16-bit only");
    if (m <= 0 || c <= 0 || (c % 8) || (dy_ld % 8) || (dy_off % 8) || (z_ld % 8) || (z_off % 8) || (dz_ld % 8) || (dz_off % 8) || nrows <= 0 || rows_ld < c)
        return fail(YOLO_ERR_ARG, "bn_act_bwd_rows: bad shape");
    hipStream_t s = (hipStream_t)stream;
    float* coef = rows + (size_t)nrows * 2 * rows_ld;
    hipLaunchKernelGGL(bn_bwd_finalize_rows, dim3(ceil_div(c, FIN_CH)), dim3(256), 0, s, rows, nrows, rows_ld, m, c, gamma, invstd, dgamma, dbeta, coef);
    if (int rc = check_launch("bn_bwd_finalize_rows")) return rc;
    int appb;
    const int anblk = ap_blocks(m, c, 8, &appb);
    YOLO_DISPATCH_DTYPE(dtype, "bn_act_bwd_rows",
        YOLO_SWITCH_ACT(act,
            hipLaunchKernelGGL((bn_bwd_apply<T, ACT>), dim3(anblk), dim3(256), 0, s, (const Elt<T>::S*)dy, dy_ld, dy_off,
                               (const Elt<T>::S*)z, z_ld, z_off, mean, invstd, scale, shift, coef, (Elt<T>::S*)dz, dz_ld, dz_off, m, c, appb)));
    return check_launch("bn_bwd_apply");
}

int yolo_upsample2x_bwd(const void* dup, int d_ld, int d_off, void* dx, int x_ld, int x_off, int n, int h, int w, int c,
                        int dtype, void* stream) {
    const int vn = dtype == YOLO_F32 ? 4 : 8;
    if (!dup || !dx || (c % vn) || (d_ld % vn) || (d_off % vn) || (x_ld % vn) || (x_off % vn)) return fail(YOLO_ERR_ARG, "upsample2x_bwd: bad arguments");
    const long long m = (long long)n * h * w;
    YOLO_DISPATCH_DTYPE(dtype, "upsample2x_bwd",
        hipLaunchKernelGGL(upsample2x_bwd_kernel<T>, dim3(ew_grid(m * (c / vn))), dim3(256), 0, (hipStream_t)stream, (const Elt<T>::S*)dup, d_ld,
                           d_off, (Elt<T>::S*)dx, x_ld, x_off, m, c, h, w));
    return check_launch("upsample2x_bwd");
}

}  // extern "C"
