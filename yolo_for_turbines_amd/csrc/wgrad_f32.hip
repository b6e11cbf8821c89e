// wgrad_f32.hip — weight gradient of the convolution blocks (fp32, f32 matrix cores).
//
// Replaces the autograd backward of nn.Conv2d w.r.t. its weight inside `loss.backward()`
// (reference: code/train.py:67; conv definition code/model.py:60):
//     dW[co][ci][kh][kw] = sum_{n,ho,wo} dz[n,ho,wo,co] * x[n, ho*s+kh-p, wo*s+kw-p, ci]
// GEMM view: M = Cout, N = (tap, ci), K = all output pixels (86k .. 5.5M at batch 32): the output is
// tiny, the reduction is huge, so K is split over blocks (grid.y) into fixed pixel ranges; every
// slice writes its own fp32 partial and `wgrad_reduce` adds the slices in a fixed order and emits
// OIHW — deterministic, no float atomics (SURVEY §5 determinism).
// Both operands are NHWC rows (channels contiguous, pixel = K index), which is exactly the
// [k][m] / [k][n] LDS image the 32x32x2 f32 MFMA wants: lane l reads A[i = l&31][k = l>>5] as one
// ds_read_b32 at row k, column i — consecutive lanes, consecutive banks, no transpose anywhere.
#include "common.h"

namespace yolo {

typedef float f32x16 __attribute__((ext_vector_type(16)));

struct WgradArgs {
    const void* dz;         // element type T of the kernel template
    const void* x;
    float* partial;
    int N, H, W, Ho, Wo, M;
    int Cin, Cout;          // Cin = padded to 4
    int ks, stride, pad;
    int dz_ld, dz_off, x_ld, x_off;
    int Kp;                 // ks*ks*Cin
    int total_steps, steps_per_slice;
    int tiles_n, tiles_per_tap;
    int cout_pad;           // rows of one partial slice
};

constexpr int WBK = 32;     // pixels per K step

template <typename T, int BM, int BN, bool SMALLC>
__global__ __launch_bounds__(256) void wgrad_f32_kernel(const WgradArgs p) {
    typedef typename Elt<T>::S S;
    const S* gz = reinterpret_cast<const S*>(p.dz);
    const S* gx = reinterpret_cast<const S*>(p.x);
    constexpr int WM = BM / 2, WN = BN / 2, TM = WM / 32, TN = WN / 32;
    constexpr int A4 = BM / 4, B4 = BN / 4;                   // float4 columns
    constexpr int AR = 256 / A4, BR = 256 / B4;               // rows per pass
    constexpr int AP = WBK / AR, BP = WBK / BR;               // passes
    __shared__ __attribute__((aligned(16))) float As[2][WBK][BM];
    __shared__ __attribute__((aligned(16))) float Bs[2][WBK][BN];

    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int wm = wave >> 1, wn = wave & 1;
    const int tile_n = blockIdx.x % p.tiles_n, tile_m = blockIdx.x / p.tiles_n;
    const int co0 = tile_m * BM;
    int tap = 0, ci0 = 0;
    if (!SMALLC) { tap = tile_n / p.tiles_per_tap; ci0 = (tile_n - tap * p.tiles_per_tap) * BN; }
    const int kh = tap / p.ks, kw = tap - kh * p.ks;
    const int step0 = blockIdx.y * p.steps_per_slice;
    const int step1 = step0 + p.steps_per_slice < p.total_steps ? step0 + p.steps_per_slice : p.total_steps;
    const int HoWo = p.Ho * p.Wo;
    const int co_lim = (p.Cout + 3) & ~3;

    const int a_c4 = tid % A4, a_r = tid / A4;
    const int b_c4 = tid % B4, b_r = tid / B4;
    f32x4 ra[AP], rb[BP];

    auto load = [&](int step) {
        const int p0 = step * WBK;
#pragma unroll
        for (int i = 0; i < AP; ++i) {
            const int pix = p0 + a_r + i * AR;
            const int co = co0 + a_c4 * 4;
            f32x4 z = {0.f, 0.f, 0.f, 0.f};
            ra[i] = (pix < p.M && co < co_lim) ? Elt<T>::ld4(gz + (size_t)pix * p.dz_ld + p.dz_off + co) : z;
        }
#pragma unroll
        for (int i = 0; i < BP; ++i) {
            const int pix = p0 + b_r + i * BR;
            f32x4 z = {0.f, 0.f, 0.f, 0.f};
            rb[i] = z;
            if (pix < p.M) {
                const int n = pix / HoWo;
                const int rem = pix - n * HoWo;
                const int ho = rem / p.Wo, wo = rem - ho * p.Wo;
                int t = tap, ci = ci0 + b_c4 * 4, th = kh, tw = kw;
                if (SMALLC) { t = b_c4; ci = 0; th = t / p.ks; tw = t - th * p.ks; }     // one 4-channel tap per float4 column
                const int hi = ho * p.stride + th - p.pad, wi = wo * p.stride + tw - p.pad;
                if (t < p.ks * p.ks && ci < p.Cin && (unsigned)hi < (unsigned)p.H && (unsigned)wi < (unsigned)p.W)
                    rb[i] = Elt<T>::ld4(gx + ((size_t)(n * p.H + hi) * p.W + wi) * p.x_ld + p.x_off + ci);
            }
        }
    };
    auto store = [&](int buf) {
#pragma unroll
        for (int i = 0; i < AP; ++i) *reinterpret_cast<f32x4*>(&As[buf][a_r + i * AR][a_c4 * 4]) = ra[i];
#pragma unroll
        for (int i = 0; i < BP; ++i) *reinterpret_cast<f32x4*>(&Bs[buf][b_r + i * BR][b_c4 * 4]) = rb[i];
    };

    f32x16 acc[TM][TN];
#pragma unroll
    for (int i = 0; i < TM; ++i)
#pragma unroll
        for (int j = 0; j < TN; ++j)
#pragma unroll
            for (int r = 0; r < 16; ++r) acc[i][j][r] = 0.f;

    const int frow = lane & 31, fh = lane >> 5;
    if (step0 < step1) {
        load(step0);
        store(0);
    }
    __syncthreads();
    for (int step = step0; step < step1; ++step) {
        const int cur = (step - step0) & 1;
        if (step + 1 < step1) load(step + 1);
#pragma unroll
        for (int kk = 0; kk < WBK / 2; ++kk) {
            float a[TM], b[TN];
#pragma unroll
            for (int i = 0; i < TM; ++i) a[i] = As[cur][2 * kk + fh][wm * WM + i * 32 + frow];
#pragma unroll
            for (int j = 0; j < TN; ++j) b[j] = Bs[cur][2 * kk + fh][wn * WN + j * 32 + frow];
#pragma unroll
            for (int i = 0; i < TM; ++i)
#pragma unroll
                for (int j = 0; j < TN; ++j) acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x2f32(a[i], b[j], acc[i][j], 0, 0, 0);
        }
        if (step + 1 < step1) store(cur ^ 1);
        __syncthreads();
    }

    // partial[slice][co][k], k = tap*Cin + ci (row-major, same K order as the packed forward weights)
    float* out = p.partial + (size_t)blockIdx.y * p.cout_pad * p.Kp;
#pragma unroll
    for (int j = 0; j < TN; ++j) {
        const int col = wn * WN + j * 32 + frow;
        int k;
        bool kv;
        if (SMALLC) { k = col; kv = col < p.Kp; }
        else { const int ci = ci0 + col; k = tap * p.Cin + ci; kv = ci < p.Cin; }
#pragma unroll
        for (int i = 0; i < TM; ++i)
#pragma unroll
            for (int r = 0; r < 16; ++r) {
                const int co = co0 + wm * WM + i * 32 + (r & 3) + 8 * (r >> 2) + 4 * fh;
                if (kv && co < p.cout_pad) out[(size_t)co * p.Kp + k] = acc[i][j][r];
            }
    }
}

// dW (OIHW) = sum over slices, fixed order. Threads walk the PARTIAL layout ([co][tap][ci], ci fastest) four ci at a
// time: every slice is read with coalesced 16-byte loads (the first version walked the OIHW output order and read the
// slices with a stride of cin_pad floats — 2.1 TB/s on 75 MB per layer); the 4-byte OIHW writes are 1/nslices of the bytes.
// G lanes share one output vector: lane g adds slices g, g + G, ... (two chains), the G lane sums are added in lane order
// through LDS. G = 1 for big weight tensors (enough vectors to fill the chip); G = 16 for the small ones, where one thread
// walking 256-512 slices serially took 35-41 us for a few KB of output. The order depends only on (nslices, G): deterministic.
template <int G>
__global__ __launch_bounds__(256) void wgrad_reduce(const float* __restrict__ partial, float* __restrict__ dw, int nslices, int cout, int cin,
                                                    int cin_pad, int taps, int cout_pad, long long total4) {
    const size_t slice = (size_t)cout_pad * taps * cin_pad;
    const int c4n = cin_pad >> 2;
    __shared__ f32x4 red[256];
    const long long nthreads = total4 * G;
    for (long long base = blockIdx.x * 256LL; base < nthreads; base += (long long)gridDim.x * 256) {
        const long long id = base + threadIdx.x;
        const long long i = id / G;
        const int g = (int)(id - i * G);
        const bool live = id < nthreads;
        f32x4 s0 = {0.f, 0.f, 0.f, 0.f}, s1 = s0;
        int c4 = 0, t = 0, co = 0;
        if (live) {
            c4 = (int)(i % c4n);
            const long long r = i / c4n;
            t = (int)(r % taps);
            co = (int)(r / taps);
            const float* src = partial + ((size_t)co * taps + t) * cin_pad + c4 * 4;
            int k = g;
            for (; k + G < nslices; k += 2 * G) {             // two independent chains, order fixed
                s0 += *reinterpret_cast<const f32x4*>(src + (size_t)k * slice);
                s1 += *reinterpret_cast<const f32x4*>(src + (size_t)(k + G) * slice);
            }
            if (k < nslices) s0 += *reinterpret_cast<const f32x4*>(src + (size_t)k * slice);
            s0 += s1;
        }
        if (G > 1) {
            __syncthreads();
            red[threadIdx.x] = s0;
            __syncthreads();
            if (g == 0) {
#pragma unroll
                for (int l = 1; l < G; ++l) s0 += red[threadIdx.x + l];
            }
        }
        if (live && g == 0) {
#pragma unroll
            for (int e = 0; e < 4; ++e) {
                const int ci = c4 * 4 + e;
                if (ci < cin) dw[((size_t)co * cin + ci) * taps + t] = s0[e];
            }
        }
    }
}

static int launch_wgrad_reduce(const float* partial, float* dw, int nslices, int cout, int cin, int cin_pad, int taps, int cout_pad,
                               long long total4, hipStream_t s) {
    // vectors x lanes >= ~64k threads: the small tensors of the high-resolution layers have 128-512 slices and < 10k vectors
    if (total4 < 16384 && nslices >= 32) {
        const long long nt = total4 * 16;
        const int grid = (int)((nt + 255) / 256 < 8192 ? (nt + 255) / 256 : 8192);
        hipLaunchKernelGGL(wgrad_reduce<16>, dim3(grid), dim3(256), 0, s, partial, dw, nslices, cout, cin, cin_pad, taps, cout_pad, total4);
    } else if (total4 < 65536 && nslices >= 8) {
        const long long nt = total4 * 4;
        const int grid = (int)((nt + 255) / 256 < 8192 ? (nt + 255) / 256 : 8192);
        hipLaunchKernelGGL(wgrad_reduce<4>, dim3(grid), dim3(256), 0, s, partial, dw, nslices, cout, cin, cin_pad, taps, cout_pad, total4);
    } else {
        const int grid = (int)((total4 + 255) / 256 < 8192 ? (total4 + 255) / 256 : 8192);
        hipLaunchKernelGGL(wgrad_reduce<1>, dim3(grid), dim3(256), 0, s, partial, dw, nslices, cout, cin, cin_pad, taps, cout_pad, total4);
    }
    return check_launch("wgrad_reduce");
}

struct WgradPlan { int bm, bn, smallc, tiles_m, tiles_n, tiles_per_tap, total_steps, nslices, steps_per_slice, cout_pad, kp; };

static WgradPlan plan_wgrad(int n, int h, int w, int cin, int cout, int ks, int stride) {
    WgradPlan q;
    const int pad = ks / 2;
    const int ho = (h + 2 * pad - ks) / stride + 1, wo = (w + 2 * pad - ks) / stride + 1;
    const long long M = (long long)n * ho * wo;
    const int cp = cin_pad_of(cin);
    q.smallc = cp == 4;
    q.bm = cout > 64 ? 128 : 64;
    q.bn = q.smallc ? 64 : (cp > 64 ? 128 : 64);
    q.tiles_m = ceil_div(cout, q.bm);
    q.tiles_per_tap = q.smallc ? 1 : ceil_div(cp, q.bn);
    q.tiles_n = q.smallc ? 1 : ks * ks * q.tiles_per_tap;
    q.total_steps = (int)((M + WBK - 1) / WBK);
    int want = ceil_div(1536, q.tiles_m * q.tiles_n);
    int maxs = ceil_div(q.total_steps, 8);                 // at least 8 K steps (256 pixels) per slice
    if (maxs < 1) maxs = 1;
    q.nslices = want < 1 ? 1 : (want > maxs ? maxs : want);
    if (q.nslices > 512) q.nslices = 512;
    q.steps_per_slice = ceil_div(q.total_steps, q.nslices);
    q.nslices = ceil_div(q.total_steps, q.steps_per_slice);
    q.cout_pad = q.tiles_m * q.bm;
    q.kp = ks * ks * cp;
    return q;
}

}  // namespace yolo

using namespace yolo;

extern "C" {

size_t yolo_wgrad_workspace_bytes(int n, int h, int w, int cin, int cout, int ksize, int stride, int dtype) {
    if (dtype != YOLO_F32 && dtype != YOLO_BF16 && dtype != YOLO_F16) return 0;
    if (n <= 0 || h <= 0 || w <= 0 || cin <= 0 || cout <= 0 || (ksize != 1 && ksize != 3) || (stride != 1 && stride != 2)) return 0;
    const WgradPlan q = plan_wgrad(n, h, w, cin, cout, ksize, stride);
    size_t need = (size_t)q.nslices * q.cout_pad * q.kp * sizeof(float);
    if (dtype != YOLO_F32) {
        const size_t nh = wgrad_h16_workspace(n, h, w, cin, cout, ksize, stride);
        if (nh > need) need = nh;
        if (ksize == 3 && stride == 1) {
            const size_t nd = wgrad_dma_workspace(n, h, w, cin, cout);
            if (nd > need) need = nd;
            if (cin <= 3 && cout <= 32 && (w & 15) == 0) {
                const size_t nst = wgrad_stem_workspace(n, h, w);
                if (nst > need) need = nst;
            }
        }
    }
    return need;
}

/* probe of the gfx950 transposing LDS read used by the 16-bit wgrad kernel (tests only):
 * in: [64][ld] 16-bit image, out: [64 lanes][8] what each lane receives for the 32x32x16 operand */
int yolo_debug_tr_probe(const void* in, void* out, int ld, void* stream) {
    if (!in || !out || ld < 32 || ld > 128 || (ld & 3)) return fail(YOLO_ERR_ARG, "tr_probe: bad arguments");
    return tr_probe_launch(in, out, ld, (hipStream_t)stream);
}

/* dz: NHWC gradient of the raw conv output (n, ho, wo, cout) with channel stride dz_ld (>= cout rounded
 * up to 4; any padding channels must be zero); x: NHWC conv input; dw: OIHW fp32 out. */
int yolo_conv_wgrad(const void* dz, int dz_ld, int dz_off, const void* x, int x_ld, int x_off, float* dw_oihw, int n, int h,
                    int w, int cin, int cout, int ksize, int stride, int dtype, void* workspace, size_t workspace_bytes, void* stream) {
    if (!dz || !x || !dw_oihw || !workspace) return fail(YOLO_ERR_ARG, "wgrad: null pointer");
    const size_t need = yolo_wgrad_workspace_bytes(n, h, w, cin, cout, ksize, stride, dtype);
    if (!need) return fail(YOLO_ERR_ARG, "wgrad: bad shape");
    if (workspace_bytes < need) return fail(YOLO_ERR_WORKSPACE, "wgrad: workspace %zu < %zu", workspace_bytes, need);
    const int cp = cin_pad_of(cin);
    if ((dz_ld & 3) || (dz_off & 3) || (x_ld & 3) || (x_off & 3) || x_ld < cp || dz_ld < ((cout + 3) & ~3))
        return fail(YOLO_ERR_ARG, "wgrad: ld/off must be multiples of 4 and cover the padded channels");
    hipStream_t s = (hipStream_t)stream;
    const long long total = (long long)cout * ksize * ksize * (cp / 4);            // float4 groups of the partial layout
    if (dtype != YOLO_F32 && wgrad_stem_eligible(n, h, w, cin, cout, ksize, stride, dz_ld, dz_off, x_ld, x_off))
        return wgrad_stem_launch(dz, dz_ld, dz_off, x, x_off, (float*)workspace, dw_oihw, n, h, w, cin, cout, dtype, s);
    if (dtype != YOLO_F32 && wgrad_dma_eligible(cin, cout, ksize, stride, dz_ld, dz_off, x_ld, x_off))
        return wgrad_dma_launch(dz, dz_ld, dz_off, x, x_ld, x_off, (float*)workspace, dw_oihw, n, h, w, cin, cout, dtype, s);
    if (dtype != YOLO_F32 && wgrad_h16_eligible(cin, cout, ksize, stride, dz_ld, dz_off, x_ld, x_off)) {
        int cout_pad = 0;
        const int ns = wgrad_h16_launch(dz, dz_ld, dz_off, x, x_ld, x_off, (float*)workspace, n, h, w, cin, cout, ksize, stride, dtype,
                                        &cout_pad, s);
        if (ns < 0) return ns;
        return launch_wgrad_reduce((const float*)workspace, dw_oihw, ns, cout, cin, cp, ksize * ksize, cout_pad, total, s);
    }
    const WgradPlan q = plan_wgrad(n, h, w, cin, cout, ksize, stride);
    WgradArgs a;
    a.dz = dz; a.x = x; a.partial = (float*)workspace;
    a.N = n; a.H = h; a.W = w; a.ks = ksize; a.stride = stride; a.pad = ksize / 2;
    a.Ho = (h + 2 * a.pad - ksize) / stride + 1; a.Wo = (w + 2 * a.pad - ksize) / stride + 1;
    const long long M = (long long)n * a.Ho * a.Wo;
    if (M > 0x7fffffffLL) return fail(YOLO_ERR_UNSUPPORTED, "wgrad: too many pixels");
    a.M = (int)M; a.Cin = cp; a.Cout = cout;
    a.dz_ld = dz_ld; a.dz_off = dz_off; a.x_ld = x_ld; a.x_off = x_off;
    a.Kp = q.kp; a.total_steps = q.total_steps; a.steps_per_slice = q.steps_per_slice;
    a.tiles_n = q.tiles_n; a.tiles_per_tap = q.tiles_per_tap; a.cout_pad = q.cout_pad;
    dim3 grid(q.tiles_m * q.tiles_n, q.nslices), block(256);
    YOLO_DISPATCH_DTYPE(dtype, "wgrad",
        if (q.smallc) {
            if (q.bm == 128) hipLaunchKernelGGL((wgrad_f32_kernel<T, 128, 64, true>), grid, block, 0, s, a);
            else hipLaunchKernelGGL((wgrad_f32_kernel<T, 64, 64, true>), grid, block, 0, s, a);
        } else if (q.bm == 128 && q.bn == 128) hipLaunchKernelGGL((wgrad_f32_kernel<T, 128, 128, false>), grid, block, 0, s, a);
        else if (q.bm == 128) hipLaunchKernelGGL((wgrad_f32_kernel<T, 128, 64, false>), grid, block, 0, s, a);
        else if (q.bn == 128) hipLaunchKernelGGL((wgrad_f32_kernel<T, 64, 128, false>), grid, block, 0, s, a);
        else hipLaunchKernelGGL((wgrad_f32_kernel<T, 64, 64, false>), grid, block, 0, s, a));
    int rc = check_launch("wgrad_f32");
    if (rc) return rc;
    return launch_wgrad_reduce((const float*)workspace, dw_oihw, q.nslices, cout, cin, cp, ksize * ksize, q.cout_pad, total, s);
}

}  // extern "C"
