// dgrad_f32.hip — input gradient of the convolution blocks (fp32).
//
// Replaces the autograd backward of nn.Conv2d w.r.t. its input inside `loss.backward()`
// (reference: code/train.py:67; conv definitions code/model.py:60,105-106,136-137,203-204).
//
// Stride 1 (70 of the 75 convs): dx = conv(dz, W') with W'[ci][co][kh][kw] = W[co][ci][2-kh][2-kw] is
// itself a stride-1 "same" convolution, so it runs on the FORWARD kernels (conv_f32_v2.hip /
// conv_f32.hip) over weights packed by yolo_pack_weights_dgrad (transpose + tap flip); the kernels'
// residual-add epilogue accumulates the other gradient contributions of the same tensor (skip
// connection, route/concat slice, second consumer).
//
// Stride 2 (the 5 down-sampling convs) is a transposed convolution:
//     dx[n,hi,wi,ci] = sum_{kh,kw,co : (hi+1-kh), (wi+1-kw) even} dz[n,(hi+1-kh)/2,(wi+1-kw)/2,co] W[co,ci,kh,kw]
// Gathering it naively wastes 3/4 of the MFMAs on taps whose parity does not match. This kernel
// enumerates the dx pixels PARITY-CLASS-MAJOR (class = (hi&1, wi&1)): within a class the valid taps
// are the same for every pixel — 1, 2, 2 and 4 taps — so each block loops over exactly its class's
// taps and no matrix work is spent on zeros. Same register-staged 64x64 implicit-GEMM structure as
// conv_igemm_f32<64,64> (M = dx pixels of one class, N = Cin, K = (valid taps, Cout)).
#include "common.h"

namespace yolo {

typedef float f32x16 __attribute__((ext_vector_type(16)));

struct DgradS2Args {
    const void* dz;       // (N, Ho, Wo, Cout) gradient of the raw conv output          (element type T)
    const float* w;       // [Cin_pad128][9*Cout] : W'[ci][tap*Cout + co] = W[co][ci][tap]   (fp32)
    const void* res;      // optional other contribution to dx (same geometry as dx)      (T)
    void* dx;             // (N, 2Ho, 2Wo, Cin)                                            (T)
    int N, Ho, Wo, Cin, Cout;
    int dz_ld, dz_off, dx_ld, dx_off, r_ld, r_off;
    int Mc;               // pixels per class = N*Ho*Wo
    int tiles_per_class, tiles_n, Kpad;
};

constexpr int DLD = 36;

template <typename T>
__global__ __launch_bounds__(256) void dgrad_s2_f32_kernel(const DgradS2Args p) {
    typedef typename Elt<T>::S S;
    const S* gz = reinterpret_cast<const S*>(p.dz);
    const S* gres = reinterpret_cast<const S*>(p.res);
    S* gdx = reinterpret_cast<S*>(p.dx);
    constexpr int BM = 64, BN = 64;
    __shared__ __attribute__((aligned(16))) float As[2][BM][DLD];
    __shared__ __attribute__((aligned(16))) float Bs[2][BN][DLD];
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int wm = wave >> 1, wn = wave & 1;
    const int tile_n = blockIdx.x % p.tiles_n;
    const int tile_mc = blockIdx.x / p.tiles_n;
    const int cls = tile_mc / p.tiles_per_class;
    const int tile_m = tile_mc - cls * p.tiles_per_class;
    const int ph = cls >> 1, pw = cls & 1;
    const int m0 = tile_m * BM, n0 = tile_n * BN;

    // valid taps of this class: kh has the parity of (ph+1); source offset dh = (ph+1-kh)/2 in {0,1}.
    // ph = 0: kh = 1 (dh 0).  ph = 1: kh = 0 (dh 1), kh = 2 (dh 0).  Same along w.
    const int nth = ph ? 2 : 1, ntw = pw ? 2 : 1;
    const int ntaps = nth * ntw;
    const int chunks = p.Cout / 32;
    const int KT = ntaps * chunks;

    const int chunk = tid & 7, lrow = tid >> 3;
    const int HoWo = p.Ho * p.Wo;
    long long a_base[2];
    int a_r[2], a_c[2];
    bool a_ok[2];
#pragma unroll
    for (int i = 0; i < 2; ++i) {
        const int m = m0 + lrow + 32 * i;
        a_ok[i] = m < p.Mc;
        const int mm = a_ok[i] ? m : 0;
        const int n = mm / HoWo;
        const int rem = mm - n * HoWo;
        a_r[i] = rem / p.Wo;
        a_c[i] = rem - a_r[i] * p.Wo;
        a_base[i] = ((long long)(n * p.Ho + a_r[i]) * p.Wo + a_c[i]) * p.dz_ld + p.dz_off + chunk * 4;
    }
    const float* wrow = p.w + (size_t)(n0 + lrow) * p.Kpad + chunk * 4;

    f32x4 ra[2], rb[2];
    auto load_global = [&](int kt) {
        const int t = kt / chunks, c0 = (kt - t * chunks) * 32;
        const int qi = t / ntw, qj = t - qi * ntw;
        const int kh = ph ? (qi ? 2 : 0) : 1, kw = pw ? (qj ? 2 : 0) : 1;
        const int dh = ph ? (qi ? 0 : 1) : 0, dw = pw ? (qj ? 0 : 1) : 0;
        const int id = kh * 3 + kw;
        const long long toff = (long long)(dh * p.Wo + dw) * p.dz_ld + c0;
#pragma unroll
        for (int i = 0; i < 2; ++i) {
            const bool v = a_ok[i] && a_r[i] + dh < p.Ho && a_c[i] + dw < p.Wo;
            f32x4 z = {0.f, 0.f, 0.f, 0.f};
            ra[i] = v ? Elt<T>::ld4(gz + a_base[i] + toff) : z;
        }
#pragma unroll
        for (int i = 0; i < 2; ++i)
            rb[i] = *reinterpret_cast<const f32x4*>(wrow + (size_t)(32 * i) * p.Kpad + id * p.Cout + c0);
    };
    auto store_lds = [&](int buf) {
#pragma unroll
        for (int i = 0; i < 2; ++i) {
            *reinterpret_cast<f32x4*>(&As[buf][lrow + 32 * i][chunk * 4]) = ra[i];
            *reinterpret_cast<f32x4*>(&Bs[buf][lrow + 32 * i][chunk * 4]) = rb[i];
        }
    };

    f32x16 acc;
#pragma unroll
    for (int r = 0; r < 16; ++r) acc[r] = 0.f;
    const int frow = lane & 31, fh = lane >> 5;

    load_global(0);
    store_lds(0);
    __syncthreads();
    for (int kt = 0; kt < KT; ++kt) {
        const int cur = kt & 1;
        if (kt + 1 < KT) load_global(kt + 1);
#pragma unroll
        for (int s = 0; s < 4; ++s) {
            const f32x4 af = *reinterpret_cast<const f32x4*>(&As[cur][wm * 32 + frow][s * 8 + 4 * fh]);
            const f32x4 bf = *reinterpret_cast<const f32x4*>(&Bs[cur][wn * 32 + frow][s * 8 + 4 * fh]);
#pragma unroll
            for (int e = 0; e < 4; ++e) acc = __builtin_amdgcn_mfma_f32_32x32x2f32(af[e], bf[e], acc, 0, 0, 0);
        }
        if (kt + 1 < KT) store_lds(cur ^ 1);
        __syncthreads();
    }

    const int n = n0 + wn * 32 + frow;
    if (n >= p.Cin) return;
    const int H2 = 2 * p.Ho, W2 = 2 * p.Wo;
#pragma unroll
    for (int r = 0; r < 16; ++r) {
        const int m = m0 + wm * 32 + (r & 3) + 8 * (r >> 2) + 4 * fh;
        if (m >= p.Mc) continue;
        const int img = m / HoWo;
        const int rem = m - img * HoWo;
        const int rr = rem / p.Wo, cc = rem - rr * p.Wo;
        const size_t pix = (size_t)(img * H2 + 2 * rr + ph) * W2 + 2 * cc + pw;
        float v = acc[r];
        if (gres) v += Elt<T>::ld(gres + pix * p.r_ld + p.r_off + n);
        Elt<T>::st(gdx + pix * p.dx_ld + p.dx_off + n, v);
    }
}

// weights for the gradient convolutions, from OIHW W[cout][cin][k][k]:
//   flip = 1 (stride-1 dgrad on the forward kernels): W'[ci][tap][co] = W[co][ci][k*k-1-tap]
//   flip = 0 (stride-2 transposed-conv kernel above):  W'[ci][tap][co] = W[co][ci][tap]
// written in the row-major layout [cin_pad128][round_up(k*k*coutp, 32)] (coutp = cout rounded up to 32,
// pad channels zero), K index = tap*coutp + co.
__global__ void pack_dgrad_rowmajor(const float* __restrict__ w, float* __restrict__ wp, int cout, int cin, int ks, int coutp,
                                    int kpad, int flip, long long total) {
    const int taps = ks * ks;
    for (long long i = blockIdx.x * (long long)blockDim.x + threadIdx.x; i < total; i += (long long)gridDim.x * blockDim.x) {
        const int ci = (int)(i / kpad);
        const int k = (int)(i - (long long)ci * kpad);
        const int tap = k / coutp, co = k - tap * coutp;
        float v = 0.f;
        if (ci < cin && tap < taps && co < cout) v = w[((size_t)co * cin + ci) * taps + (flip ? taps - 1 - tap : tap)];
        wp[i] = v;
    }
}

// fragment-order copy for the patch kernel: [n_tile32 over ci][kt][s][lane][e], k channel = co
__global__ void pack_dgrad_frag(const float* __restrict__ w, float* __restrict__ wf, int cout, int cin, int ks, int KT,
                                long long total) {
    const int taps = ks * ks;
    for (long long i = blockIdx.x * (long long)blockDim.x + threadIdx.x; i < total; i += (long long)gridDim.x * blockDim.x) {
        const int e = (int)(i & 3);
        const int lane = (int)((i >> 2) & 63);
        const int s = (int)((i >> 8) & 3);
        const long long rest = i >> 10;
        const int kt = (int)(rest % KT);
        const int nt = (int)(rest / KT);
        const int ci = nt * 32 + (lane & 31);
        const int chunk = kt / taps, tap = kt - chunk * taps;
        const int co = chunk * 32 + s * 8 + 4 * (lane >> 5) + e;
        wf[i] = (ci < cin && co < cout) ? w[((size_t)co * cin + ci) * taps + (taps - 1 - tap)] : 0.f;
    }
}

// (B,3,g,g,D) head-layout gradient -> NHWC (B,g,g,ld) with channel a*D+k, pad channels zeroed
template <typename T>
__global__ void head_grad_to_nhwc_kernel(const float* __restrict__ dp, long long sb, long long sa, long long sy, long long sx,
                                         long long sk, typename Elt<T>::S* __restrict__ out, int B, int g, int D, int ld) {
    const long long total = (long long)B * g * g * ld;
    for (long long i = blockIdx.x * (long long)blockDim.x + threadIdx.x; i < total; i += (long long)gridDim.x * blockDim.x) {
        const int ch = (int)(i % ld);
        const long long pix = i / ld;
        const int x = (int)(pix % g);
        const int y = (int)((pix / g) % g);
        const int b = (int)(pix / ((long long)g * g));
        float v = 0.f;
        if (ch < 3 * D) {
            const int a = ch / D, k = ch - a * D;
            v = dp[b * sb + a * sa + y * sy + x * sx + k * sk];
        }
        Elt<T>::st(out + i, v);
    }
}

}  // namespace yolo

using namespace yolo;

extern "C" {

/* packed size of the gradient-convolution weights of a conv (cout, cin, k): row-major part + (k-flipped)
 * fragment-order part; the latter only exists for flip = 1 layouts and cout rounded up to 32. */
size_t yolo_packed_dgrad_bytes(int cout, int cin, int ksize, int flip, int dtype) {
    if (cout <= 0 || cin <= 0 || (ksize != 1 && ksize != 3)) return 0;
    const int coutp = round_up(cout, 32);
    if (flip && dtype != YOLO_F32) return h16_frag_elems(cin, coutp, ksize) * 2;
    if (!flip && dtype != YOLO_F32) return (ksize == 3 && cout % 32 == 0) ? h16_dgrad_s2_elems(cout, cin) * 2 : 0;
    return (v0_packed_elems(cin, coutp, ksize) + v2_frag_elems(cin, coutp, ksize) + (flip ? wino_weight_elems(cin, coutp, ksize) : 0)) * sizeof(float);
}

int yolo_pack_weights_dgrad(const float* w_oihw, void* w_packed, int cout, int cin, int ksize, int flip, int dtype, void* stream) {
    if (!w_oihw || !w_packed || !yolo_packed_dgrad_bytes(cout, cin, ksize, flip, dtype)) return fail(YOLO_ERR_ARG, "pack_weights_dgrad: bad arguments");
    if (flip && dtype != YOLO_F32) return h16_pack_dgrad(w_oihw, w_packed, cout, cin, ksize, dtype, (hipStream_t)stream);
    if (!flip && dtype != YOLO_F32) return h16_pack_dgrad_s2(w_oihw, w_packed, cout, cin, dtype, (hipStream_t)stream);
    const int coutp = round_up(cout, 32);
    const int kpad = kpad_of(coutp, ksize);
    const long long total = (long long)v0_packed_elems(cin, coutp, ksize);
    hipStream_t s = (hipStream_t)stream;
    int grid = (int)((total + 255) / 256 < 8192 ? (total + 255) / 256 : 8192);
    hipLaunchKernelGGL(pack_dgrad_rowmajor, dim3(grid), dim3(256), 0, s, w_oihw, (float*)w_packed, cout, cin, ksize, coutp, kpad, flip, total);
    int rc = check_launch("pack_dgrad_rowmajor");
    if (rc || !flip) return rc;
    const long long ftotal = (long long)v2_frag_elems(cin, coutp, ksize);
    const int KT = (coutp / 32) * ksize * ksize;
    grid = (int)((ftotal + 255) / 256 < 8192 ? (ftotal + 255) / 256 : 8192);
    hipLaunchKernelGGL(pack_dgrad_frag, dim3(grid), dim3(256), 0, s, w_oihw, (float*)w_packed + total, cout, cin, ksize, KT, ftotal);
    rc = check_launch("pack_dgrad_frag");
    // third section, as in yolo_pack_weights: the Winograd-domain filters of the gradient convolution (yolo_conv_fwd_ws)
    if (rc || !wino_weight_elems(cin, coutp, ksize)) return rc;
    return wino_pack_dgrad(w_oihw, (float*)w_packed + total + ftotal, cout, cin, s);
}

/* dx (N,2Ho,2Wo,cin) = transposed 3x3 stride-2 conv of dz (N,Ho,Wo,cout) [+ residual]; w_packed from
 * yolo_pack_weights_dgrad(flip = 0). cout % 32 == 0. */
int yolo_conv_dgrad_s2(const void* dz, int dz_ld, int dz_off, const void* w_packed, const void* residual, int r_ld, int r_off,
                       void* dx, int dx_ld, int dx_off, int n, int ho, int wo, int cin, int cout, int dtype, void* stream) {
    if (!dz || !w_packed || !dx) return fail(YOLO_ERR_ARG, "dgrad_s2: null pointer");
    if (dtype == YOLO_BF16 || dtype == YOLO_F16)
        return dgrad_s2_h16_launch(dz, dz_ld, dz_off, w_packed, residual, r_ld, r_off, dx, dx_ld, dx_off, n, ho, wo, cin, cout, dtype,
                                   (hipStream_t)stream);
    if (n <= 0 || ho <= 0 || wo <= 0 || cin <= 0 || cout <= 0 || cout % 32) return fail(YOLO_ERR_UNSUPPORTED, "dgrad_s2: cout %% 32 != 0");
    if ((dz_ld & 3) || (dz_off & 3)) return fail(YOLO_ERR_ARG, "dgrad_s2: dz_ld/dz_off must be multiples of 4");
    DgradS2Args a;
    a.dz = dz; a.w = (const float*)w_packed; a.res = residual; a.dx = dx;
    a.N = n; a.Ho = ho; a.Wo = wo; a.Cin = cin; a.Cout = cout;
    a.dz_ld = dz_ld; a.dz_off = dz_off; a.dx_ld = dx_ld; a.dx_off = dx_off; a.r_ld = r_ld; a.r_off = r_off;
    const long long mc = (long long)n * ho * wo;
    if (mc * 4 > 0x7fffffffLL) return fail(YOLO_ERR_UNSUPPORTED, "dgrad_s2: too many pixels");
    a.Mc = (int)mc;
    a.tiles_per_class = ceil_div(a.Mc, 64);
    a.tiles_n = ceil_div(cin, 64);
    a.Kpad = kpad_of(cout, 3);
    if (dtype != YOLO_F32) return fail(YOLO_ERR_ARG, "dgrad_s2: unknown dtype %d", dtype);
    hipLaunchKernelGGL(dgrad_s2_f32_kernel<float>, dim3(4 * a.tiles_per_class * a.tiles_n), dim3(256), 0, (hipStream_t)stream, a);
    return check_launch("dgrad_s2_f32");
}

int yolo_head_grad_to_nhwc(const float* dp, const int64_t* strides5, void* out, int b, int g, int d, int ld, int dtype, void* stream) {
    if (!dp || !strides5 || !out || b <= 0 || g <= 0 || d <= 0 || ld < 3 * d) return fail(YOLO_ERR_ARG, "head_grad_to_nhwc: bad arguments");
    const long long total = (long long)b * g * g * ld;
    const int grid = (int)((total + 255) / 256 < 8192 ? (total + 255) / 256 : 8192);
    YOLO_DISPATCH_DTYPE(dtype, "head_grad_to_nhwc",
        hipLaunchKernelGGL(head_grad_to_nhwc_kernel<T>, dim3(grid), dim3(256), 0, (hipStream_t)stream, dp, (long long)strides5[0],
                           (long long)strides5[1], (long long)strides5[2], (long long)strides5[3], (long long)strides5[4], (Elt<T>::S*)out,
                           b, g, d, ld));
    return check_launch("head_grad_to_nhwc");
}

}  // extern "C"
