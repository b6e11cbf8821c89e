// wgrad_h16.hip — weight gradient of the convolution blocks on the bf16 / f16 matrix cores.
//
// Replaces the autograd backward of nn.Conv2d w.r.t. its weight (reference: code/train.py:67 under the
// autocast context of train.py:53; conv definition code/model.py:60):
//     dW[co][ci][kh][kw] = sum_{n,ho,wo} dz[n,ho,wo,co] * x[n, ho*s+kh-p, wo*s+kw-p, ci]
// GEMM view: M = Cout, N = Cin (per tap), K = output pixels. Both operands are NHWC, i.e. the contraction
// index (pixel) is the STRIDED one, while v_mfma_f32_32x32x16_{bf16,f16} wants 8 consecutive k per lane.
// gfx950's transposing LDS read does that for free: tiles are staged [pixel][channel] exactly as they lie
// in HBM (16-byte coalesced loads, ds_write_b128) and both operands are fetched with ds_read_b64_tr_b16
// (4 pixels x 16 channels per 16-lane group, delivered channel-major).
//
// Data movement ("patch" structure, like the forward kernels): a block owns a 64(co) x 64(ci) tile of dW
// for ALL 9 taps (9 x 64 x 64 fp32 accumulators = 144 VGPRs/lane) and walks K in tiles of TH x 16 output
// pixels. Per K tile it stages the dz tile and the x patch with halo ONCE; the 9 taps are 9 constant row
// offsets into the same patch, so x is read from HBM/L2 ~1.7x instead of 9x and dz once instead of 9x.
// Tile width 16 makes one tile row exactly one k16 MFMA step and every LDS offset a compile-time immediate.
// 1x1 convs use the same code with a 128 x 128 tile, one tap, linear pixel tiles.
// K is split over blockIdx.y slices; each slice writes an fp32 partial in the layout of wgrad_f32.hip and
// the same fixed-order `wgrad_reduce` emits OIHW — deterministic, no float atomics.
// LDS rows are padded to (channels*2 + 64) bytes: the 4 pixel rows of one transposing read then fall in
// 4 disjoint 16-bank windows (conflict-free for stride-1 patches).
#include "common.h"

namespace yolo {

typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef unsigned int u32x4 __attribute__((ext_vector_type(4)));
typedef short s16x4 __attribute__((ext_vector_type(4)));
typedef short s16x8 __attribute__((ext_vector_type(8)));
typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));
typedef _Float16 f16x8 __attribute__((ext_vector_type(8)));

struct WgradHArgs {
    const unsigned short* dz;
    const unsigned short* x;
    float* partial;
    int N, H, W, Ho, Wo, M;
    int Cp, Cout, cin8, cout8;       // Cp = cin rounded up to 4 (partial layout); *8 = rounded up to 8 (valid 16-B pieces)
    int dz_ld, dz_off, x_ld, x_off;
    int Kp, cout_pad;
    int tiles_n, ntile;              // ci tiles; co tiles * ci tiles
    int th_tiles, tw_tiles, total_tiles, tiles_per_slice, nslices;
};

template <typename T> struct WTraits;
template <> struct WTraits<__bf16> {
    static __device__ __forceinline__ f32x16 mfma(s16x8 a, s16x8 b, f32x16 c) {
        return __builtin_amdgcn_mfma_f32_32x32x16_bf16(__builtin_bit_cast(bf16x8, a), __builtin_bit_cast(bf16x8, b), c, 0, 0, 0);
    }
};
template <> struct WTraits<_Float16> {
    static __device__ __forceinline__ f32x16 mfma(s16x8 a, s16x8 b, f32x16 c) {
        return __builtin_amdgcn_mfma_f32_32x32x16_f16(__builtin_bit_cast(f16x8, a), __builtin_bit_cast(f16x8, b), c, 0, 0, 0);
    }
};

// 4 pixel rows x 16 channel columns of a [pixel][channel] LDS image, channel-major into the lane:
// element e of the result is (row e of the block, column lane%16). Row addresses come from lanes 4q+p.
__device__ __forceinline__ s16x4 tr_read(const char* lds_base, int byte_off) {
    return __builtin_amdgcn_ds_read_tr16_b64_v4i16((__attribute__((address_space(3))) s16x4*)(lds_base + byte_off));
}

template <typename T, int KS, int STRIDE>
__global__ __launch_bounds__(256) void wgrad_patch_h16(const WgradHArgs p) {
    constexpr int TAPS = KS * KS;
    constexpr int TT = KS == 3 ? 1 : 2;                     // 32x32 tiles per wave along co and along ci
    constexpr int BM = 64 * TT, BN = 64 * TT;
    constexpr int TH = KS == 3 ? (STRIDE == 1 ? 4 : 2) : 2;  // tile rows = k16 steps per K tile
    constexpr int KPX = TH * 16;
    constexpr int PAD = KS / 2;
    constexpr int PC = STRIDE * 15 + KS, PR = STRIDE * (TH - 1) + KS;
    constexpr int PATCH = PR * PC;
    constexpr int GROW = BM * 2 + 64, XROW = BN * 2 + 64;   // LDS bytes per pixel row
    constexpr int GP = BM / 8, XP = BN / 8;                 // 16-byte pieces per pixel
    constexpr int NG = (KPX * GP + 255) / 256, NX = (PATCH * XP + 255) / 256;
    constexpr int BUF = KPX * GROW + PATCH * XROW;
    extern __shared__ __attribute__((aligned(16))) char smem[];

    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int wm = wave >> 1, wn = wave & 1;

    // ---- block -> (dW tile, K slice); slices of one pixel range share an XCD (and its L2) when possible
    int slice, tile;
    if ((p.nslices & 7) == 0) {
        const int xcd = blockIdx.x & 7, j = blockIdx.x >> 3;
        slice = (j / p.ntile) * 8 + xcd;
        tile = j % p.ntile;
    } else {
        slice = blockIdx.x / p.ntile;
        tile = blockIdx.x % p.ntile;
    }
    const int co0 = (tile / p.tiles_n) * BM, ci0 = (tile % p.tiles_n) * BN;
    const int t0 = slice * p.tiles_per_slice;
    const int t1 = t0 + p.tiles_per_slice < p.total_tiles ? t0 + p.tiles_per_slice : p.total_tiles;

    // ---- staging roles (fixed per thread; only the tile origin changes per K tile)
    int g_px[NG], g_c8[NG], x_pr[NX], x_pc[NX], x_c8[NX];
#pragma unroll
    for (int i = 0; i < NG; ++i) { const int id = tid + 256 * i; g_px[i] = id / GP; g_c8[i] = id % GP; }
#pragma unroll
    for (int i = 0; i < NX; ++i) {
        const int id = tid + 256 * i;
        const int ppx = id / XP;
        x_c8[i] = id % XP;
        x_pr[i] = ppx / PC;
        x_pc[i] = ppx % PC;
    }
    u32x4 rg[NG], rx[NX];
    bool vg[NG], vx[NX];

    auto stage_load = [&](int t) {
        int n, ho0, wo0;
        if (KS == 3) {
            const int tw = t % p.tw_tiles, r = t / p.tw_tiles;
            const int th = r % p.th_tiles;
            n = r / p.th_tiles; ho0 = th * TH; wo0 = tw * 16;
        } else { n = 0; ho0 = 0; wo0 = t * KPX; }          // 1x1: linear pixels, "row" = 16 consecutive pixels
#pragma unroll
        for (int i = 0; i < NG; ++i) {
            const int r = g_px[i] >> 4, c = g_px[i] & 15;
            const int co = co0 + g_c8[i] * 8;
            long long pix;
            bool v;
            if (KS == 3) { v = ho0 + r < p.Ho && wo0 + c < p.Wo; pix = (long long)(n * p.Ho + ho0 + r) * p.Wo + wo0 + c; }
            else { pix = (long long)wo0 + g_px[i]; v = pix < p.M; }
            v = v && g_px[i] < KPX && co < p.cout8;
            vg[i] = v;
            if (!v) { pix = 0; }
            rg[i] = *reinterpret_cast<const u32x4*>(p.dz + (size_t)pix * p.dz_ld + p.dz_off + (v ? co : 0));
        }
#pragma unroll
        for (int i = 0; i < NX; ++i) {
            const int ci = ci0 + x_c8[i] * 8;
            long long pix;
            bool v;
            if (KS == 3) {
                const int hi = STRIDE * ho0 - PAD + x_pr[i], wi = STRIDE * wo0 - PAD + x_pc[i];
                v = (unsigned)hi < (unsigned)p.H && (unsigned)wi < (unsigned)p.W && x_pr[i] < PR;
                pix = (long long)(n * p.H + hi) * p.W + wi;
            } else {
                pix = (long long)wo0 + x_pr[i] * PC + x_pc[i];
                v = pix < p.M && x_pr[i] < PR;
            }
            v = v && ci < p.cin8;
            vx[i] = v;
            if (!v) { pix = 0; }
            rx[i] = *reinterpret_cast<const u32x4*>(p.x + (size_t)pix * p.x_ld + p.x_off + (v ? ci : 0));
        }
    };
    auto stage_store = [&](int buf) {
        char* G = smem + buf * BUF;
        char* X = G + KPX * GROW;
        const u32x4 z = {0u, 0u, 0u, 0u};
#pragma unroll
        for (int i = 0; i < NG; ++i)
            if (g_px[i] < KPX) *reinterpret_cast<u32x4*>(G + g_px[i] * GROW + g_c8[i] * 16) = vg[i] ? rg[i] : z;
#pragma unroll
        for (int i = 0; i < NX; ++i)
            if (x_pr[i] < PR) *reinterpret_cast<u32x4*>(X + (x_pr[i] * PC + x_pc[i]) * XROW + x_c8[i] * 16) = vx[i] ? rx[i] : z;
    };

    // ---- operand addresses: lane (q, pp) of its 16-lane group supplies row q, columns 4pp..4pp+3
    const int lg = lane & 15, q = lg >> 2, pp = lg & 3;
    const int kl = 8 * (lane >> 5) + q;                       // this lane's pixel column inside a tile row (+4 for the 2nd read)
    const int cb = 16 * ((lane >> 4) & 1) + 4 * pp;           // channel column inside a 32-wide MFMA tile
    const int a_off = kl * GROW + (wm * 32 * TT + cb) * 2;
    const int b_off = KPX * GROW + STRIDE * kl * XROW + (wn * 32 * TT + cb) * 2;

    f32x16 acc[TAPS][TT][TT];
#pragma unroll
    for (int t = 0; t < TAPS; ++t)
#pragma unroll
        for (int i = 0; i < TT; ++i)
#pragma unroll
            for (int j = 0; j < TT; ++j)
#pragma unroll
                for (int r = 0; r < 16; ++r) acc[t][i][j][r] = 0.f;

    if (t0 < t1) {
        stage_load(t0);
        stage_store(0);
    }
    __syncthreads();
    for (int t = t0; t < t1; ++t) {
        const int cur = (t - t0) & 1;
        stage_load(t + 1 < t1 ? t + 1 : t);                 // unconditional (clamped) prefetch of the next K tile
        const char* base = smem + cur * BUF;
#pragma unroll
        for (int k16 = 0; k16 < TH; ++k16) {
            s16x8 a[TT];
#pragma unroll
            for (int i = 0; i < TT; ++i) {
                const s16x4 lo = tr_read(base, a_off + (16 * k16) * GROW + i * 64);
                const s16x4 hi = tr_read(base, a_off + (16 * k16 + 4) * GROW + i * 64);
                a[i] = __builtin_shufflevector(lo, hi, 0, 1, 2, 3, 4, 5, 6, 7);
            }
#pragma unroll
            for (int tap = 0; tap < TAPS; ++tap) {
                const int dh = tap / KS, dw = tap % KS;
#pragma unroll
                for (int j = 0; j < TT; ++j) {
                    const int o = ((STRIDE * k16 + dh) * PC + dw) * XROW + j * 64;
                    const s16x4 lo = tr_read(base, b_off + o);
                    const s16x4 hi = tr_read(base, b_off + o + STRIDE * 4 * XROW);
                    const s16x8 b = __builtin_shufflevector(lo, hi, 0, 1, 2, 3, 4, 5, 6, 7);
#pragma unroll
                    for (int i = 0; i < TT; ++i) acc[tap][i][j] = WTraits<T>::mfma(a[i], b, acc[tap][i][j]);
                }
            }
        }
        stage_store(cur ^ 1);
        __syncthreads();
    }

    // ---- partial[slice][co][tap*Cp + ci] (same layout as wgrad_f32.hip; reduced by wgrad_reduce)
    float* out = p.partial + (size_t)slice * p.cout_pad * p.Kp;
    const int fh = lane >> 5, frow = lane & 31;
#pragma unroll
    for (int tap = 0; tap < TAPS; ++tap)
#pragma unroll
        for (int j = 0; j < TT; ++j) {
            const int ci = ci0 + wn * 32 * TT + j * 32 + frow;
            if (ci >= p.Cp) continue;
#pragma unroll
            for (int i = 0; i < TT; ++i)
#pragma unroll
                for (int r = 0; r < 16; ++r) {
                    const int co = co0 + wm * 32 * TT + i * 32 + (r & 3) + 8 * (r >> 2) + 4 * fh;
                    out[(size_t)co * p.Kp + tap * p.Cp + ci] = acc[tap][i][j][r];
                }
        }
}

// probe of the transposing read (tests): out[lane][e] for a [64 rows][32 cols] image with row stride `ld` elements
__global__ void tr_probe_kernel(const unsigned short* in, unsigned short* out, int ld) {
    __shared__ __attribute__((aligned(16))) unsigned short img[64 * 128];
    for (int i = threadIdx.x; i < 64 * ld; i += 64) img[i] = in[i];
    __syncthreads();
    const int l = threadIdx.x, lg = l & 15, q = lg >> 2, pp = lg & 3;
    const int row = 8 * (l >> 5) + q, col = 16 * ((l >> 4) & 1) + 4 * pp;
    const s16x4 lo = tr_read(reinterpret_cast<const char*>(img), (row * ld + col) * 2);
    const s16x4 hi = tr_read(reinterpret_cast<const char*>(img), ((row + 4) * ld + col) * 2);
#pragma unroll
    for (int e = 0; e < 4; ++e) { out[l * 8 + e] = lo[e]; out[l * 8 + 4 + e] = hi[e]; }
}

struct WgradHPlan { int bm, tiles_m, tiles_n, th, th_tiles, tw_tiles, total_tiles, nslices, tiles_per_slice, cout_pad, kp; };

static WgradHPlan plan_wgrad_h(int n, int h, int w, int cin, int cout, int ks, int stride) {
    WgradHPlan q;
    const int pad = ks / 2;
    const int ho = (h + 2 * pad - ks) / stride + 1, wo = (w + 2 * pad - ks) / stride + 1;
    q.bm = ks == 3 ? 64 : 128;
    q.tiles_m = ceil_div(cout, q.bm);
    q.tiles_n = ceil_div(cin, q.bm);
    q.th = ks == 3 ? (stride == 1 ? 4 : 2) : 2;
    if (ks == 3) {
        q.th_tiles = ceil_div(ho, q.th);
        q.tw_tiles = ceil_div(wo, 16);
        q.total_tiles = n * q.th_tiles * q.tw_tiles;
    } else {
        q.th_tiles = q.tw_tiles = 1;
        q.total_tiles = (int)(((long long)n * ho * wo + q.th * 16 - 1) / (q.th * 16));
    }
    const int ntile = q.tiles_m * q.tiles_n;
    int want = ceil_div(512, ntile);                       // one resident wave of blocks: partial traffic = blocks x tile
    int maxs = ceil_div(q.total_tiles, 8);                  // at least 8 K tiles per slice
    if (maxs < 1) maxs = 1;
    int ns = want < 1 ? 1 : (want > maxs ? maxs : want);
    if (ns > 512) ns = 512;                                 // (was 256: the one-tile layers - stem, 32->64 - then ran ONE block per CU and
                                                            //  streamed their 350 MB of dz at 0.66 TB/s: 537 us each)
    if (ns >= 8) ns = ns / 8 * 8;                           // multiples of 8: XCD-aware block mapping
    q.tiles_per_slice = ceil_div(q.total_tiles, ns);
    q.nslices = ns;                                         // trailing slices may be empty (they write zeros)
    q.cout_pad = q.tiles_m * q.bm;
    q.kp = ks * ks * cin_pad_of(cin);
    return q;
}

// cin need not be a multiple of 8 (the 3-channel stem): x is read in 8-channel pieces, so x_ld must cover cin rounded
// up to 8 and the pad channels of x must be zero (the NHWC boundary copy zero-fills them). With few channels most of the
// 64-wide ci tile is zeros, which only wastes matrix issue slots: that layer is bound by reading dz from HBM.
bool wgrad_h16_eligible(int cin, int cout, int ks, int stride, int dz_ld, int dz_off, int x_ld, int x_off) {
    if (cin < 1 || x_ld < round_up(cin, 8)) return false;
    if (ks == 1 && stride != 1) return false;
    if ((dz_ld & 7) || (dz_off & 7) || (x_ld & 7) || (x_off & 7)) return false;
    if (dz_ld < round_up(cout, 8)) return false;
    return true;
}

size_t wgrad_h16_workspace(int n, int h, int w, int cin, int cout, int ks, int stride) {
    const WgradHPlan q = plan_wgrad_h(n, h, w, cin, cout, ks, stride);
    return (size_t)q.nslices * q.cout_pad * q.kp * sizeof(float);
}

template <typename T, int KS, int STRIDE>
static int launch_wh(const WgradHArgs& a, hipStream_t s) {
    constexpr int TT = KS == 3 ? 1 : 2;
    constexpr int TH = KS == 3 ? (STRIDE == 1 ? 4 : 2) : 2;
    constexpr int PC = STRIDE * 15 + KS, PR = STRIDE * (TH - 1) + KS;
    constexpr int ROW = 64 * TT * 2 + 64;
    const size_t lds = 2 * (size_t)(TH * 16 * ROW + PR * PC * ROW);
    static LdsOnce once;
    if (lds > 65536)
        if (int rc = reserve_lds(once, reinterpret_cast<const void*>(&wgrad_patch_h16<T, KS, STRIDE>), lds, "wgrad_patch_h16")) return rc;
    hipLaunchKernelGGL((wgrad_patch_h16<T, KS, STRIDE>), dim3(a.ntile * a.nslices), dim3(256), lds, s, a);
    return check_launch("wgrad_patch_h16");
}

// returns the number of partial slices written (> 0) or a negative error code
int wgrad_h16_launch(const void* dz, int dz_ld, int dz_off, const void* x, int x_ld, int x_off, float* partial, int n, int h, int w,
                     int cin, int cout, int ks, int stride, int dtype, int* cout_pad, hipStream_t s) {
    const WgradHPlan q = plan_wgrad_h(n, h, w, cin, cout, ks, stride);
    WgradHArgs a;
    a.dz = (const unsigned short*)dz; a.x = (const unsigned short*)x; a.partial = partial;
    const int pad = ks / 2;
    a.N = n; a.H = h; a.W = w;
    a.Ho = (h + 2 * pad - ks) / stride + 1; a.Wo = (w + 2 * pad - ks) / stride + 1;
    const long long M = (long long)n * a.Ho * a.Wo;
    if (M > 0x7fffffffLL) return fail(YOLO_ERR_UNSUPPORTED, "wgrad: too many pixels");
    a.M = (int)M;
    a.Cp = cin_pad_of(cin); a.Cout = cout; a.cin8 = round_up(cin, 8); a.cout8 = round_up(cout, 8);
    a.dz_ld = dz_ld; a.dz_off = dz_off; a.x_ld = x_ld; a.x_off = x_off;
    a.Kp = q.kp; a.cout_pad = q.cout_pad;
    a.tiles_n = q.tiles_n; a.ntile = q.tiles_m * q.tiles_n;
    a.th_tiles = q.th_tiles; a.tw_tiles = q.tw_tiles; a.total_tiles = q.total_tiles;
    a.tiles_per_slice = q.tiles_per_slice; a.nslices = q.nslices;
    *cout_pad = q.cout_pad;
    int rc;
    if (dtype == YOLO_BF16) {
        if (ks == 1) rc = launch_wh<__bf16, 1, 1>(a, s);
        else if (stride == 1) rc = launch_wh<__bf16, 3, 1>(a, s);
        else rc = launch_wh<__bf16, 3, 2>(a, s);
    } else {
        if (ks == 1) rc = launch_wh<_Float16, 1, 1>(a, s);
        else if (stride == 1) rc = launch_wh<_Float16, 3, 1>(a, s);
        else rc = launch_wh<_Float16, 3, 2>(a, s);
    }
    return rc ? rc : q.nslices;
}

int tr_probe_launch(const void* in, void* out, int ld, hipStream_t s) {
    hipLaunchKernelGGL(tr_probe_kernel, dim3(1), dim3(64), 0, s, (const unsigned short*)in, (unsigned short*)out, ld);
    return check_launch("tr_probe");
}

}  // namespace yolo
