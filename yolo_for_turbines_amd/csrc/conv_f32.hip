// conv_f32.hip — fp32 implicit-GEMM convolution on the gfx950 f32 matrix cores.
//
// Replaces (reference file:line): CNNBlock.forward  code/model.py:80-86  (Conv2d -> BN(eval) ->
// LeakyReLU/Mish, or bare Conv2d+bias), the residual add of ResidualBlock.forward :115-121, the
// nn.Upsample + torch.cat writer of YOLOv3.forward :189-191 and the head reshape/permute :145-148.
//
// GEMM view:  M = N*Ho*Wo output pixels, N = Cout, K = k*k*Cin with K index (kh, kw, ci).
// A (activations, NHWC) is gathered straight from HBM/L2 into LDS — no im2col buffer: one
// 32-wide K step lies inside a single filter tap because Cin % 32 == 0 for every layer but the
// first (Cin = 3 padded to 4, where each 16-byte chunk is one tap).  B is the packed weight
// matrix [Cout_pad][K_pad] (yolo_pack_weights).  Arithmetic: v_mfma_f32_32x32x2_f32, exact
// fp32 products and fp32 accumulation (bit-for-bit an fmaf chain), which is what lets the fp32
// path meet the 1e-3 parity bar with margin.
//
// Block = 256 threads = 4 waves (2 x 2), wave tile = (BM/2) x (BN/2) built from 32x32 MFMA
// tiles.  LDS rows are 32 floats padded to 36 so the ds_read_b128 fragment reads of 16 lanes
// (16 different rows, same column chunk) hit 16 different 16-byte slots.  Register-staged
// double buffering: global loads of step t+1 are issued before the MFMAs of step t and written
// to the other LDS buffer after them; one barrier per K step.
#include "common.h"

namespace yolo {

typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef float f32x4 __attribute__((ext_vector_type(4)));

struct ConvArgs {
    const float* x;
    const float* w;
    const float* scale;
    const float* shift;
    const float* res;
    float* y;
    int* nan_flag;
    int N, H, W, Cin, Cout, Ho, Wo, M;
    int ks, stride, pad;
    int x_ld, x_off, y_ld, y_off, r_ld, r_off;
    int Kpad, KT;
    int act, out_mode, flags;
    int nc5;
    int tiles_n;
};

constexpr int BK = 32;
constexpr int LDS_LD = 36;   // padded row length (floats)


template <int BM, int BN, bool SMALLC>
__global__ __launch_bounds__(256) void conv_igemm_f32(const ConvArgs p) {
    constexpr int WM = BM / 2, WN = BN / 2;      // wave tile
    constexpr int TM = WM / 32, TN = WN / 32;    // 32x32 MFMA tiles per wave
    constexpr int RA = BM / 32, RB = BN / 32;    // rows staged per thread
    extern __shared__ __attribute__((aligned(16))) char smem_raw[];
    float* As = reinterpret_cast<float*>(smem_raw);              // [2][BM][LDS_LD]
    float* Bs = As + 2 * BM * LDS_LD;                            // [2][BN][LDS_LD]

    const int tid = threadIdx.x;
    const int lane = tid & 63;
    const int wave = tid >> 6;
    const int wm = wave >> 1, wn = wave & 1;
    const int tile_n = blockIdx.x % p.tiles_n;
    const int tile_m = blockIdx.x / p.tiles_n;
    const int m0 = tile_m * BM, n0 = tile_n * BN;

    // ---------------------------------------------------------------- staging geometry
    const int chunk = tid & 7;       // 16-byte chunk inside the 32-float K step
    const int lrow = tid >> 3;       // 0..31
    long long a_base[RA];
    unsigned a_mask[RA];
    const int HoWo = p.Ho * p.Wo;
#pragma unroll
    for (int i = 0; i < RA; ++i) {
        const int m = m0 + lrow + 32 * i;
        const bool mv = m < p.M;
        const int mm = mv ? m : 0;
        const int n = mm / HoWo;
        const int rem = mm - n * HoWo;
        const int ho = rem / p.Wo;
        const int wo = rem - ho * p.Wo;
        const int hi0 = ho * p.stride - p.pad, wi0 = wo * p.stride - p.pad;
        a_base[i] = ((long long)(n * p.H + hi0) * p.W + wi0) * p.x_ld + p.x_off;
        unsigned mk = 0;
        for (int kh = 0; kh < p.ks; ++kh)
            for (int kw = 0; kw < p.ks; ++kw)
                if (mv && (unsigned)(hi0 + kh) < (unsigned)p.H && (unsigned)(wi0 + kw) < (unsigned)p.W)
                    mk |= 1u << (kh * p.ks + kw);
        a_mask[i] = mk;
    }
    const float* wrow = p.w + (size_t)(n0 + lrow) * p.Kpad + chunk * 4;

    f32x4 ra[RA], rb[RB];
    auto load_global = [&](int kt) {
        int tap, coff;
        if (SMALLC) {
            tap = kt * 8 + chunk;
            coff = 0;
        } else {
            const int kg = kt * BK;
            tap = kg / p.Cin;
            coff = kg - tap * p.Cin + chunk * 4;
        }
        const int kh = tap / p.ks, kw = tap - kh * p.ks;
        const long long toff = (long long)(kh * p.W + kw) * p.x_ld + coff;
#pragma unroll
        for (int i = 0; i < RA; ++i) {
            const bool v = (tap < 9) && ((a_mask[i] >> tap) & 1u);
            f32x4 z = {0.f, 0.f, 0.f, 0.f};
            ra[i] = v ? *reinterpret_cast<const f32x4*>(p.x + a_base[i] + toff) : z;
        }
#pragma unroll
        for (int i = 0; i < RB; ++i)
            rb[i] = *reinterpret_cast<const f32x4*>(wrow + (size_t)(32 * i) * p.Kpad + kt * BK);
    };
    auto store_lds = [&](int buf) {
        float* a = As + buf * BM * LDS_LD + lrow * LDS_LD + chunk * 4;
        float* b = Bs + buf * BN * LDS_LD + lrow * LDS_LD + chunk * 4;
#pragma unroll
        for (int i = 0; i < RA; ++i) *reinterpret_cast<f32x4*>(a + 32 * i * LDS_LD) = ra[i];
#pragma unroll
        for (int i = 0; i < RB; ++i) *reinterpret_cast<f32x4*>(b + 32 * i * LDS_LD) = rb[i];
    };

    f32x16 acc[TM][TN];
#pragma unroll
    for (int i = 0; i < TM; ++i)
#pragma unroll
        for (int j = 0; j < TN; ++j)
#pragma unroll
            for (int r = 0; r < 16; ++r) acc[i][j][r] = 0.f;

    // fragment read offsets: lane l -> row (l & 31), K chunk 4*(l >> 5) inside each 8-wide sub-step
    const int frow = lane & 31, fh = lane >> 5;
    const int a_frag = (wm * WM + frow) * LDS_LD + 4 * fh;
    const int b_frag = (wn * WN + frow) * LDS_LD + 4 * fh;

    load_global(0);
    store_lds(0);
    __syncthreads();

    for (int kt = 0; kt < p.KT; ++kt) {
        const int cur = kt & 1;
        if (kt + 1 < p.KT) load_global(kt + 1);
        const float* Ab = As + cur * BM * LDS_LD + a_frag;
        const float* Bb = Bs + cur * BN * LDS_LD + b_frag;
#pragma unroll
        for (int s = 0; s < 4; ++s) {
            f32x4 af[TM], bf[TN];
#pragma unroll
            for (int i = 0; i < TM; ++i) af[i] = *reinterpret_cast<const f32x4*>(Ab + i * 32 * LDS_LD + s * 8);
#pragma unroll
            for (int j = 0; j < TN; ++j) bf[j] = *reinterpret_cast<const f32x4*>(Bb + j * 32 * LDS_LD + s * 8);
#pragma unroll
            for (int e = 0; e < 4; ++e)
#pragma unroll
                for (int i = 0; i < TM; ++i)
#pragma unroll
                    for (int j = 0; j < TN; ++j)
                        acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x2f32(af[i][e], bf[j][e], acc[i][j], 0, 0, 0);
        }
        if (kt + 1 < p.KT) store_lds(cur ^ 1);
        __syncthreads();
    }

    // ---------------------------------------------------------------------- epilogue
    // C/D map of the 32x32 tile: column (N = cout) = lane & 31, row (M = pixel) =
    // (r & 3) + 8 * (r >> 2) + 4 * (lane >> 5).
    const bool has_res = p.flags & YOLO_FLAG_RESIDUAL;
    const bool nan_chk = p.flags & YOLO_FLAG_NANCHECK;
    bool saw_nan = false;
    // NHWC outputs with cout % 4 == 0 (every BN block): scale/shift/activation in registers, transpose the BM x BN tile
    // through the (now idle) operand LDS, and let every lane move 16 contiguous bytes of one pixel row. The direct stores
    // below are 4-byte pieces of different rows per lane; on the short 1x1 blocks they were most of the block's life.
    const bool aligned4 = ((p.y_ld | p.y_off) & 3) == 0 && (!has_res || ((p.r_ld | p.r_off) & 3) == 0);
    if (p.out_mode != YOLO_OUT_HEAD && (p.Cout & 3) == 0 && aligned4) {
        constexpr int OLD = BN + 4;
        float* ost = As;                                   // [BM][OLD] floats <= 2 * (BM + BN) * LDS_LD
#pragma unroll
        for (int j = 0; j < TN; ++j) {
            const int n = n0 + wn * WN + j * 32 + frow;
            const bool nv = n < p.Cout;
            const float sc = nv ? p.scale[n] : 0.f;
            const float sh = nv ? p.shift[n] : 0.f;
            float* dst = ost + wn * WN + j * 32 + frow;
            YOLO_SWITCH_ACT(p.act,
                _Pragma("unroll") for (int i = 0; i < TM; ++i)
                    _Pragma("unroll") for (int r = 0; r < 16; ++r) {
                        const int row = wm * WM + i * 32 + (r & 3) + 8 * (r >> 2) + 4 * fh;
                        dst[row * OLD] = act_c<ACT>(acc[i][j][r] * sc + sh);
                    })
        }
        __syncthreads();
        constexpr int C4 = BN / 4;
#pragma unroll 4
        for (int idx = tid; idx < BM * C4; idx += 256) {
            const int row = idx / C4, c4 = idx - row * C4;
            const int m = m0 + row, n = n0 + c4 * 4;
            if (m >= p.M || n >= p.Cout) continue;
            f32x4 v = *reinterpret_cast<const f32x4*>(ost + row * OLD + c4 * 4);
            if (has_res) v += *reinterpret_cast<const f32x4*>(p.res + (size_t)m * p.r_ld + p.r_off + n);
            if (nan_chk && (v[0] != v[0] || v[1] != v[1] || v[2] != v[2] || v[3] != v[3])) saw_nan = true;
            if (p.out_mode == YOLO_OUT_NHWC) {
                *reinterpret_cast<f32x4*>(p.y + (size_t)m * p.y_ld + p.y_off + n) = v;
            } else {                                        // YOLO_OUT_UPSAMPLE2X
                const int img = m / HoWo;
                const int rem = m - img * HoWo;
                const int ho = rem / p.Wo;
                const int wo = rem - ho * p.Wo;
                const int W2 = 2 * p.Wo;
                float* d = p.y + ((size_t)(img * 2 * p.Ho + 2 * ho) * W2 + 2 * wo) * p.y_ld + p.y_off + n;
                *reinterpret_cast<f32x4*>(d) = v;
                *reinterpret_cast<f32x4*>(d + p.y_ld) = v;
                *reinterpret_cast<f32x4*>(d + (size_t)W2 * p.y_ld) = v;
                *reinterpret_cast<f32x4*>(d + (size_t)(W2 + 1) * p.y_ld) = v;
            }
        }
        if (nan_chk && saw_nan) atomicOr(p.nan_flag, 2);
        return;
    }
#pragma unroll
    for (int j = 0; j < TN; ++j) {
        const int n = n0 + wn * WN + j * 32 + frow;
        const bool nv = n < p.Cout;
        const float sc = nv ? p.scale[n] : 0.f;
        const float sh = nv ? p.shift[n] : 0.f;
        int head_a = 0, head_k = 0;
        if (p.out_mode == YOLO_OUT_HEAD) {
            head_a = n / p.nc5;
            head_k = n - head_a * p.nc5;
        }
        YOLO_SWITCH_ACT(p.act,                          // activation chosen once, outside the element loops
            _Pragma("unroll") for (int i = 0; i < TM; ++i)
                _Pragma("unroll") for (int r = 0; r < 16; ++r) acc[i][j][r] = act_c<ACT>(acc[i][j][r] * sc + sh);)
#pragma unroll
        for (int i = 0; i < TM; ++i) {
#pragma unroll
            for (int r = 0; r < 16; ++r) {
                const int m = m0 + wm * WM + i * 32 + (r & 3) + 8 * (r >> 2) + 4 * fh;
                if (m >= p.M || !nv) continue;
                float v = acc[i][j][r];
                if (has_res) v += p.res[(size_t)m * p.r_ld + p.r_off + n];
                if (nan_chk && v != v) saw_nan = true;
                if (p.out_mode == YOLO_OUT_NHWC) {
                    p.y[(size_t)m * p.y_ld + p.y_off + n] = v;
                } else {
                    const int img = m / HoWo;
                    const int rem = m - img * HoWo;
                    const int ho = rem / p.Wo;
                    const int wo = rem - ho * p.Wo;
                    if (p.out_mode == YOLO_OUT_UPSAMPLE2X) {
                        const int W2 = 2 * p.Wo;
                        float* d = p.y + ((size_t)(img * 2 * p.Ho + 2 * ho) * W2 + 2 * wo) * p.y_ld + p.y_off + n;
                        d[0] = v;
                        d[p.y_ld] = v;
                        d[(size_t)W2 * p.y_ld] = v;
                        d[(size_t)(W2 + 1) * p.y_ld] = v;
                    } else {  // YOLO_OUT_HEAD: (B,3,g,g,5+nc)
                        p.y[((size_t)((img * 3 + head_a) * p.Ho + ho) * p.Wo + wo) * p.nc5 + head_k] = v;
                    }
                }
            }
        }
    }
    if (nan_chk && saw_nan) atomicOr(p.nan_flag, 2);
}

// ------------------------------------------------------------------------------ host side


constexpr int kNumTiles = 10;     // 1-4: register-staged tiles above; 5/6: conv_f32_v2.hip with BN = 64/128; 7: BN = 64, one patch buffer (3 blocks/CU); 8-10: 16-bit only (conv3_dma_h16)
constexpr int kTileRs = 12;       // conv1_rs_f32.hip (fp32 1x1, weights in registers)
constexpr int kTileWino = 13;     // conv_wino_f32.hip (fp32 3x3 stride 1, Winograd F(2x2, 3x3); needs the caller's workspace)
constexpr int kMaxTileId = 31;    // ids above kNumTiles select timing probes of the diagnostic library (make probes); the product library runs tile 8 for them

static size_t lds_bytes(int bm, int bn) { return (size_t)2 * (bm + bn) * LDS_LD * sizeof(float); }

template <int BM, int BN>
static int launch_tile(const ConvArgs& a, bool smallc, hipStream_t s) {
    const int tiles_m = ceil_div(a.M, BM);
    ConvArgs p = a;
    p.tiles_n = ceil_div(a.Cout, BN);
    dim3 grid(tiles_m * p.tiles_n), block(256);
    const size_t lds = lds_bytes(BM, BN);
    if (smallc)
        hipLaunchKernelGGL((conv_igemm_f32<BM, BN, true>), grid, block, lds, s, p);
    else
        hipLaunchKernelGGL((conv_igemm_f32<BM, BN, false>), grid, block, lds, s, p);
    return check_launch("conv_igemm_f32");
}

// Measured on MI355X (tools/conv_bench.py, batch 32). The register-staged kernel of this file is
// latency-bound, so among its tiles 64x64 (4+ resident blocks per CU) wins on every YOLOv3 shape;
// stride-1 layers with cin % 32 == 0 go to the patch/fragment-stream kernel (ids 5, 6).
static int pick_direct_tile(const yolo_conv_desc* d);
static int pick_tile(const yolo_conv_desc* d) {
    if (wino_eligible(d)) return kTileWino;         // with a workspace (yolo_conv_fwd_ws / the launch table); without one: the ids below
    return pick_direct_tile(d);
}

static int pick_direct_tile(const yolo_conv_desc* d) {
    if (v2_eligible(d) && d->ksize == 3) {
        // Measured (tools/conv_bench.py --tile 5,6,7, batch 32): BN = 64 with ONE patch buffer (tile 7: 37 KB of LDS,
        // 148 registers -> 3 blocks per CU instead of 2, one extra barrier per 32-channel chunk) wins by 4-16 % at 208x208,
        // 104x104, 52x52 and 13x13 — three co-resident blocks cover each other's prologue/epilogue and 752 blocks fill
        // 768 slots in one round at 13x13. At 26x26 (K = 2304, 1440 such blocks) BN = 128 with two buffers stays 9 % ahead.
        const int ho = d->h;
        if (d->cout > 64 && ho <= 26 && ho > 13 && v2_blocks(d, 128) >= 640) return 6;
        return 7;
    }
    return 4;       // 1x1 (few K steps, prologue-dominated) and stride-2 layers: 64x64 register-staged tile
}

static int validate(const yolo_conv_desc* d) {
    if (!d) return fail(YOLO_ERR_ARG, "conv: null descriptor");
    if (d->dtype < 0 || d->dtype > 2) return fail(YOLO_ERR_ARG, "conv: dtype %d", d->dtype);
    if (d->ksize != 1 && d->ksize != 3) return fail(YOLO_ERR_UNSUPPORTED, "conv: ksize %d", d->ksize);
    if (d->stride != 1 && d->stride != 2) return fail(YOLO_ERR_UNSUPPORTED, "conv: stride %d", d->stride);
    if (d->n <= 0 || d->h <= 0 || d->w <= 0 || d->cin <= 0 || d->cout <= 0) return fail(YOLO_ERR_ARG, "conv: bad shape");
    const int cp = cin_pad_of(d->cin);
    if (cp != 4 && cp % 32 != 0) return fail(YOLO_ERR_UNSUPPORTED, "conv: cin %d (need <= 4 or a multiple of 32)", d->cin);
    if (d->x_ld < cp + 0 || (d->x_ld & 3) || (d->x_off & 3)) return fail(YOLO_ERR_ARG, "conv: x_ld/x_off must be multiples of 4 and x_ld >= cin_pad");
    if (d->out_mode == YOLO_OUT_HEAD && d->cout % 3 != 0) return fail(YOLO_ERR_ARG, "conv: head cout %% 3 != 0");
    if (d->out_mode < 0 || d->out_mode > 2) return fail(YOLO_ERR_ARG, "conv: out_mode");
    if (d->tile < 0 || d->tile > kMaxTileId) return fail(YOLO_ERR_ARG, "conv: tile id");
    return YOLO_OK;
}

static int conv_fwd_impl(const yolo_conv_desc* d, const void* x, const void* w, const float* scale, const float* shift,
                         const void* residual, void* y, void* ws, size_t ws_bytes, int32_t* nan_flag, hipStream_t s) {
    int rc = validate(d);
    if (rc) return rc;
    if (!x || !w || !scale || !shift || !y) return fail(YOLO_ERR_ARG, "conv: null pointer");
    if ((d->flags & YOLO_FLAG_RESIDUAL) && !residual) return fail(YOLO_ERR_ARG, "conv: residual flag without pointer");
    if ((d->flags & YOLO_FLAG_NANCHECK) && !nan_flag) return fail(YOLO_ERR_ARG, "conv: nancheck flag without pointer");
    if (d->dtype != YOLO_F32) return conv_h16_launch(d, x, w, scale, shift, residual, y, nan_flag, s);
    ConvArgs a;
    a.x = (const float*)x; a.w = (const float*)w; a.scale = scale; a.shift = shift;
    a.res = (const float*)residual; a.y = (float*)y; a.nan_flag = nan_flag;
    a.N = d->n; a.H = d->h; a.W = d->w; a.Cin = cin_pad_of(d->cin); a.Cout = d->cout;
    a.ks = d->ksize; a.stride = d->stride; a.pad = d->ksize / 2;
    a.Ho = (d->h + 2 * a.pad - d->ksize) / d->stride + 1;
    a.Wo = (d->w + 2 * a.pad - d->ksize) / d->stride + 1;
    const long long M = (long long)d->n * a.Ho * a.Wo;
    if (M > 0x7fffffffLL) return fail(YOLO_ERR_UNSUPPORTED, "conv: N*Ho*Wo exceeds int32");
    a.M = (int)M;
    a.x_ld = d->x_ld; a.x_off = d->x_off; a.y_ld = d->y_ld; a.y_off = d->y_off; a.r_ld = d->r_ld; a.r_off = d->r_off;
    a.Kpad = kpad_of(d->cin, d->ksize);
    a.KT = a.Kpad / BK;
    a.act = d->act; a.out_mode = d->out_mode; a.flags = d->flags;
    a.nc5 = d->out_mode == YOLO_OUT_HEAD ? d->cout / 3 : 1;
    a.tiles_n = 0;
    const bool smallc = a.Cin == 4;
    // 1x1 with 256 / 384 / 512 input channels and a multiple of 128 output channels: weights stationary in registers (tile 0 =
    // heuristic, or tile 12 explicitly; tiles 1-4 keep the register-staged kernel for A/B)
    if ((d->tile == 0 || d->tile == kTileRs) && conv1_rs_eligible(d, residual)) return conv1_rs_launch(d, x, w, scale, shift, residual, y, nan_flag, s);
    if (d->tile == kTileRs) return fail(YOLO_ERR_UNSUPPORTED, "conv: tile 12 needs a 1x1 with 256 / 384 / 512 input channels and cout %% 128 == 0");
    // 3x3 stride 1 by Winograd F(2x2, 3x3): tile 13 explicitly, or the heuristic when the caller brought a large enough workspace
    // (yolo_conv_fwd itself has none: the library allocates nothing)
    if (d->tile == kTileWino || d->tile == kTileWino + 1 || (d->tile >= 16 && d->ksize == 3 && wino_supported(d)) ||      // 16+: timing probes of the diagnostic library (make wstamps)
        (d->tile == 0 && wino_eligible(d) && ws && ws_bytes >= wino_workspace_bytes(d))) {
        if (!wino_supported(d)) return fail(YOLO_ERR_UNSUPPORTED, "conv: tile 13 needs fp32 3x3 stride 1 with NHWC output and channel counts %% 4 == 0");
        const float* U = (const float*)w + v0_packed_elems(d->cout, d->cin, d->ksize) + v2_frag_elems(d->cout, d->cin, d->ksize);
        return conv_wino_launch(d, x, U, scale, shift, residual, y, ws, ws_bytes, nan_flag, s);
    }
    const int t = d->tile ? d->tile : pick_direct_tile(d);
    if (t >= 8) return fail(YOLO_ERR_UNSUPPORTED, "conv: tile ids from 8 up are 16-bit kernels (conv3_dma_h16)");
    if (t >= 5) {
        if (!v2_eligible(d)) return fail(YOLO_ERR_UNSUPPORTED, "conv: tile %d needs stride 1 and cin %% 32 == 0", t);
        const float* wf = (const float*)w + v0_packed_elems(d->cout, d->cin, d->ksize);
        return conv_v2_launch(d, x, wf, scale, shift, residual, y, nan_flag, t == 6 ? 128 : 64, t == 7, s);
    }
    switch (t) {
        case 1: return launch_tile<128, 128>(a, smallc, s);
        case 2: return launch_tile<128, 64>(a, smallc, s);
        case 3: return launch_tile<64, 128>(a, smallc, s);
        default: return launch_tile<64, 64>(a, smallc, s);
    }
}

}  // namespace yolo

extern "C" {

int yolo_conv_num_tiles(void) { return yolo::kNumTiles; }

int yolo_conv_pick_tile(const yolo_conv_desc* d) {
    int rc = yolo::validate(d);
    if (rc) return rc;
    return yolo::pick_tile(d);
}

int yolo_conv_fwd(const yolo_conv_desc* d, const void* x, const void* w_packed, const float* scale, const float* shift,
                  const void* residual, void* y, int32_t* nan_flag, void* stream) {
    return yolo::conv_fwd_impl(d, x, w_packed, scale, shift, residual, y, nullptr, 0, nan_flag, (hipStream_t)stream);
}

size_t yolo_conv_workspace_bytes(const yolo_conv_desc* d) {
    if (yolo::validate(d)) return 0;
    if (d->tile == yolo::kTileWino || d->tile == yolo::kTileWino + 1 || d->tile >= 16 || (d->tile == 0 && yolo::wino_eligible(d))) return yolo::wino_workspace_bytes(d);
    return 0;
}

int yolo_conv_fwd_ws(const yolo_conv_desc* d, const void* x, const void* w_packed, const float* scale, const float* shift,
                     const void* residual, void* y, void* workspace, size_t workspace_bytes, int32_t* nan_flag, void* stream) {
    return yolo::conv_fwd_impl(d, x, w_packed, scale, shift, residual, y, workspace, workspace_bytes, nan_flag, (hipStream_t)stream);
}

int yolo_conv_fwd_batch(const yolo_conv_op* ops, int n_ops, int32_t* nan_flag, void* stream) {
    if (!ops && n_ops > 0) return yolo::fail(YOLO_ERR_ARG, "conv batch: null ops");
    for (int i = 0; i < n_ops; ++i) {
        const yolo_conv_op& o = ops[i];
        int rc = yolo::conv_fwd_impl(&o.d, (const void*)o.x, (const void*)o.w_packed, (const float*)o.scale,
                                     (const float*)o.shift, (const void*)o.residual, (void*)o.y, (void*)o.workspace,
                                     (size_t)o.workspace_bytes, nan_flag, (hipStream_t)stream);
        if (rc) return rc;
    }
    return YOLO_OK;
}

}  // extern "C"
