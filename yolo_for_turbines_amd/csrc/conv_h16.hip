// conv_h16.hip — bf16 / fp16 convolution blocks (fp32 accumulation) for the reduced-precision configs
// (BASELINE configs 4-5: bf16 fine-tune forward, fp16 inference; the reference reaches them through
// torch.autocast, code/train.py:53).
//
// Same fused block as the fp32 kernels (reference: CNNBlock.forward code/model.py:80-86, residual add
// :115-121, upsample+concat :189-191, head permute :145-148) and the same "patch + fragment stream"
// data movement as conv_f32_v2.hip, on v_mfma_f32_32x32x16_{bf16,f16}:
//  * activations NHWC 16-bit; a block owns TH x TW <= 128 output pixels (global rows) and stages, per
//    32-channel chunk, the input patch with halo in LDS once for all taps — stride 1 AND stride 2
//    (patch (S*(TH-1)+3 [+2 per image crossed]) x (S*(TW-1)+3)), 1x1 as the degenerate linear case;
//  * weights in MFMA-fragment order [n_tile32][kstep][2][64 lanes][8 halfs]: one contiguous 1 KiB load
//    per wave per 16 k-values, in a 3-deep register ring (a K step is only 8 MFMAs = 256 cycles, so the
//    loads are issued two K steps ahead); every in-loop load unconditional, taps compile-time,
//    sched_barrier after the prefetch group (see conv_f32_v2.hip for why);
//  * accumulators and the whole epilogue (scale/shift = folded BatchNorm, LeakyReLU/Mish, residual)
//    in fp32; one rounding to 16-bit at the store; detection heads are written in fp32.
// The matrix rate is 16x the fp32 path, so this kernel is bound by operand delivery (weight fragments
// through L1/L2) and, for 1x1 layers, by HBM; see DESIGN.md for the measured numbers.
#include "common.h"
#include <cstdlib>

namespace yolo {

typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef unsigned int u32x4 __attribute__((ext_vector_type(4)));
typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));
typedef _Float16 f16x8 __attribute__((ext_vector_type(8)));

constexpr int H_PIX_BYTES = 80;          // 32 channels x 2 B + 16 B pad per patch pixel in LDS
constexpr int H_NI = 8;                  // staged pixels per 4-lane group -> patch <= 512 pixels
constexpr int H_PATCH_CAP = 64 * H_NI;

struct ConvHArgs {
    const unsigned short* x;
    const unsigned short* wf;
    const float* scale;
    const float* shift;
    const unsigned short* res;
    void* y;
    int* nan_flag;
    int H, W, Hin, Win, rows_total;      // output tiling view (1x1: H = 1, W = M); input dims
    int Cin, Cout;
    int x_ld, x_off, y_ld, y_off, r_ld, r_off;
    int TH, TW, PC, patch_cap;
    int bufmask, mtab_off;   // bufmask 1: two patch buffers; 0: one (stride-2 3x3, see launch_h). mtab_off: byte offset of mtab in LDS
    int tiles_w, tiles_n, nblocks;
    int KT, nchunks;
    int act, out_mode, flags, nc5;
    int Ho, Wo;
    int first_wave, stagger;
    int cls_ph, cls_pw;                  // MASK kernels (stride-2 input gradient): output pixel (2r+ph, 2c+pw)
    // magic multipliers of the prologue's index divisions (a wave64 integer division is ~40 VALU instructions;
    // ~20 of them per thread were most of a 10k-cycle prologue in front of 9k cycles of matrix work)
    unsigned mg_H, mg_TW, mg_PC, mg_tn, mg_tw, mg_Hp;
};

// x / d for 0 <= x < 2^31 with mg = ceil(2^32 / d) (d >= 2) or 0 (d == 1): the estimate is q or q + 1, one fix-up
__device__ __forceinline__ int fdiv(int x, unsigned mg, int d) {
    if (!mg) return x;
    const int q = (int)__umulhi((unsigned)x, mg);
    return (long long)q * d > x ? q - 1 : q;
}
static unsigned magic_of(int d) { return d <= 1 ? 0u : (unsigned)((0x100000000ULL + (unsigned)d - 1) / (unsigned)d); }

template <typename T> struct HTraits;
template <> struct HTraits<__bf16> {
    typedef bf16x8 vec;
    static __device__ __forceinline__ f32x16 mfma(vec a, vec b, f32x16 c) { return __builtin_amdgcn_mfma_f32_32x32x16_bf16(a, b, c, 0, 0, 0); }
    static __device__ __forceinline__ float to_f32(unsigned short v) { return __uint_as_float((unsigned)v << 16); }
    static __device__ __forceinline__ unsigned short from_f32(float f) { __bf16 h = (__bf16)f; return *reinterpret_cast<unsigned short*>(&h); }
};
template <> struct HTraits<_Float16> {
    typedef f16x8 vec;
    static __device__ __forceinline__ f32x16 mfma(vec a, vec b, f32x16 c) { return __builtin_amdgcn_mfma_f32_32x32x16_f16(a, b, c, 0, 0, 0); }
    static __device__ __forceinline__ float to_f32(unsigned short v) { _Float16 h = *reinterpret_cast<_Float16*>(&v); return (float)h; }
    static __device__ __forceinline__ unsigned short from_f32(float f) { _Float16 h = (_Float16)f; return *reinterpret_cast<unsigned short*>(&h); }
};

template <typename T, int TN>
struct HCtx {
    const unsigned short* wfrag[TN];
    int a_off[2];               // LDS byte offset of this lane's pixel for m-tile 0/1 (+16*h)
    int pix[H_NI];
    int KT;
};

// one K step = 32 channels of one tap = 2 MFMA k16-steps per 32x32 tile
template <typename T, int KS, int TN, int TAP>
__device__ __forceinline__ void h_kstep(const ConvHArgs& p, const HCtx<T, TN>& c, int chunk, char* patch,
                                        u32x4 (&ring)[3][2][TN], u32x4 (&stage)[H_NI], u32x4 (&af)[2][2],
                                        f32x16 (&acc)[2][TN], int tid) {
    typedef typename HTraits<T>::vec vec;
    constexpr int TAPS = KS * KS;
    constexpr int PF_TAP = TAPS > 2 ? TAPS - 2 : 0;
    constexpr int CUR = TAP % 3, NXT2 = (TAP + 2) % 3;
    const int kt = chunk * TAPS + TAP;
    const int kta = kt + 2 < c.KT ? kt + 2 : c.KT - 1;      // clamped: unconditional loads
#pragma unroll
    for (int s = 0; s < 2; ++s)
#pragma unroll
        for (int j = 0; j < TN; ++j)
            ring[NXT2][s][j] = *reinterpret_cast<const u32x4*>(c.wfrag[j] + ((size_t)kta * 2 + s) * 512);
    if (TAP == PF_TAP) {
        const int cn = chunk + 1 < p.nchunks ? chunk + 1 : chunk;
        const int coff = p.x_off + cn * 32 + (tid & 3) * 8;
#pragma unroll
        for (int i = 0; i < H_NI; ++i) {
            const int px = c.pix[i] < 0 ? 0 : c.pix[i];
            stage[i] = *reinterpret_cast<const u32x4*>(p.x + (size_t)px * p.x_ld + coff);
        }
    }
    __builtin_amdgcn_sched_barrier(0);
    constexpr int kh = TAP / KS, kw = TAP % KS;
    constexpr int nkh = (TAP + 1) / KS, nkw = (TAP + 1) % KS;
    const char* Ab_next = patch + (chunk & p.bufmask) * (p.patch_cap * H_PIX_BYTES) + (nkh * p.PC + nkw) * H_PIX_BYTES;
    (void)kh; (void)kw;
    // A fragments of this step were read during the previous one (af); read the next step's now
    u32x4 an[2][2];
#pragma unroll
    for (int i = 0; i < 2; ++i)
#pragma unroll
        for (int s = 0; s < 2; ++s) an[i][s] = af[i][s];
    if (TAP + 1 < TAPS) {
#pragma unroll
        for (int i = 0; i < 2; ++i)
#pragma unroll
            for (int s = 0; s < 2; ++s) an[i][s] = *reinterpret_cast<const u32x4*>(Ab_next + c.a_off[i] + s * 32);
    }
    // keep the next step's A reads HERE, ahead of this step's 8 MFMAs: left free, the scheduler sinks them to just
    // before their first use and every K step starts with an exposed LDS round trip (seen in the ISA)
    __builtin_amdgcn_sched_barrier(0);
#pragma unroll
    for (int s = 0; s < 2; ++s)
#pragma unroll
        for (int j = 0; j < TN; ++j) {
            const vec b = __builtin_bit_cast(vec, ring[CUR][s][j]);
            acc[0][j] = HTraits<T>::mfma(__builtin_bit_cast(vec, af[0][s]), b, acc[0][j]);
            acc[1][j] = HTraits<T>::mfma(__builtin_bit_cast(vec, af[1][s]), b, acc[1][j]);
        }
#pragma unroll
    for (int i = 0; i < 2; ++i)
#pragma unroll
        for (int s = 0; s < 2; ++s) af[i][s] = an[i][s];
    if (TAP == TAPS - 1) {
        if (!p.bufmask) __syncthreads();             // one buffer: every wave has finished reading this chunk
        char* dst = patch + ((chunk + 1) & p.bufmask) * (p.patch_cap * H_PIX_BYTES) + (tid >> 2) * H_PIX_BYTES + (tid & 3) * 16;
#pragma unroll
        for (int i = 0; i < H_NI; ++i) {
            u32x4 z = {0u, 0u, 0u, 0u};
            if ((tid >> 2) + 64 * i < p.patch_cap) *reinterpret_cast<u32x4*>(dst + 64 * i * H_PIX_BYTES) = c.pix[i] < 0 ? z : stage[i];
        }
        __syncthreads();
        const char* An = patch + ((chunk + 1) & p.bufmask) * (p.patch_cap * H_PIX_BYTES);
#pragma unroll
        for (int i = 0; i < 2; ++i)
#pragma unroll
            for (int s = 0; s < 2; ++s) af[i][s] = *reinterpret_cast<const u32x4*>(An + c.a_off[i] + s * 32);
    }
}

template <typename T, int KS, int TN, int TAP>
__device__ __forceinline__ void h_chunk(const ConvHArgs& p, const HCtx<T, TN>& c, int chunk, char* patch,
                                        u32x4 (&ring)[3][2][TN], u32x4 (&stage)[H_NI], u32x4 (&af)[2][2],
                                        f32x16 (&acc)[2][TN], int tid) {
    if constexpr (TAP < KS * KS) {
        h_kstep<T, KS, TN, TAP>(p, c, chunk, patch, ring, stage, af, acc, tid);
        h_chunk<T, KS, TN, TAP + 1>(p, c, chunk, patch, ring, stage, af, acc, tid);
    }
}

// ---- tap subsets (stride-2 input gradient, see dgrad_s2_h16 below) ------------------------------------
// MASK selects taps of the 3x3 window (bit kh*3+kw); the K loop runs over the set bits only. The ring slot
// must be compile-time, so three chunks are unrolled (R = running K-step index mod 3).
constexpr int mask_count(int m) { int n = 0; for (int b = 0; b < 9; ++b) n += (m >> b) & 1; return n; }
constexpr int mask_nth(int m, int n) { for (int b = 0; b < 9; ++b) if ((m >> b) & 1) { if (n == 0) return b; --n; } return 0; }

template <typename T, int TN, int MASK, int TI, int R>
__device__ __forceinline__ void h_kstep_m(const ConvHArgs& p, const HCtx<T, TN>& c, int chunk, char* patch,
                                          u32x4 (&ring)[3][2][TN], u32x4 (&stage)[H_NI], u32x4 (&af)[2][2],
                                          f32x16 (&acc)[2][TN], int tid) {
    typedef typename HTraits<T>::vec vec;
    constexpr int NT = mask_count(MASK);
    constexpr int PF_T = NT > 2 ? NT - 2 : 0;
    constexpr int CUR = R % 3, NXT2 = (R + 2) % 3;
    constexpr int TAP0 = mask_nth(MASK, 0);
    const int kt = chunk * NT + TI;
    const int kta = kt + 2 < c.KT ? kt + 2 : c.KT - 1;
#pragma unroll
    for (int s = 0; s < 2; ++s)
#pragma unroll
        for (int j = 0; j < TN; ++j)
            ring[NXT2][s][j] = *reinterpret_cast<const u32x4*>(c.wfrag[j] + ((size_t)kta * 2 + s) * 512);
    if (TI == PF_T) {
        const int cn = chunk + 1 < p.nchunks ? chunk + 1 : chunk;
        const int coff = p.x_off + cn * 32 + (tid & 3) * 8;
#pragma unroll
        for (int i = 0; i < H_NI; ++i) {
            const int px = c.pix[i] < 0 ? 0 : c.pix[i];
            stage[i] = *reinterpret_cast<const u32x4*>(p.x + (size_t)px * p.x_ld + coff);
        }
    }
    __builtin_amdgcn_sched_barrier(0);
    u32x4 an[2][2];
#pragma unroll
    for (int i = 0; i < 2; ++i)
#pragma unroll
        for (int s = 0; s < 2; ++s) an[i][s] = af[i][s];
    if (TI + 1 < NT) {                                   // a_off already points at the first tap of the set
        constexpr int NTAP = mask_nth(MASK, TI + 1 < NT ? TI + 1 : 0);
        constexpr int dkh = NTAP / 3 - TAP0 / 3, dkw = NTAP % 3 - TAP0 % 3;
        const char* Ab_next = patch + (chunk & 1) * (p.patch_cap * H_PIX_BYTES) + (dkh * p.PC + dkw) * H_PIX_BYTES;
#pragma unroll
        for (int i = 0; i < 2; ++i)
#pragma unroll
            for (int s = 0; s < 2; ++s) an[i][s] = *reinterpret_cast<const u32x4*>(Ab_next + c.a_off[i] + s * 32);
    }
    // keep the next step's A reads HERE, ahead of this step's 8 MFMAs: left free, the scheduler sinks them to just
    // before their first use and every K step starts with an exposed LDS round trip (seen in the ISA)
    __builtin_amdgcn_sched_barrier(0);
#pragma unroll
    for (int s = 0; s < 2; ++s)
#pragma unroll
        for (int j = 0; j < TN; ++j) {
            const vec b = __builtin_bit_cast(vec, ring[CUR][s][j]);
            acc[0][j] = HTraits<T>::mfma(__builtin_bit_cast(vec, af[0][s]), b, acc[0][j]);
            acc[1][j] = HTraits<T>::mfma(__builtin_bit_cast(vec, af[1][s]), b, acc[1][j]);
        }
#pragma unroll
    for (int i = 0; i < 2; ++i)
#pragma unroll
        for (int s = 0; s < 2; ++s) af[i][s] = an[i][s];
    if (TI == NT - 1) {
        char* dst = patch + ((chunk + 1) & 1) * (p.patch_cap * H_PIX_BYTES) + (tid >> 2) * H_PIX_BYTES + (tid & 3) * 16;
#pragma unroll
        for (int i = 0; i < H_NI; ++i) {
            u32x4 z = {0u, 0u, 0u, 0u};
            if ((tid >> 2) + 64 * i < p.patch_cap) *reinterpret_cast<u32x4*>(dst + 64 * i * H_PIX_BYTES) = c.pix[i] < 0 ? z : stage[i];
        }
        __syncthreads();
        const char* An = patch + ((chunk + 1) & 1) * (p.patch_cap * H_PIX_BYTES);
#pragma unroll
        for (int i = 0; i < 2; ++i)
#pragma unroll
            for (int s = 0; s < 2; ++s) af[i][s] = *reinterpret_cast<const u32x4*>(An + c.a_off[i] + s * 32);
    }
}

template <typename T, int TN, int MASK, int CC, int TI>
__device__ __forceinline__ void h_chunk_m(const ConvHArgs& p, const HCtx<T, TN>& c, int chunk, char* patch,
                                          u32x4 (&ring)[3][2][TN], u32x4 (&stage)[H_NI], u32x4 (&af)[2][2],
                                          f32x16 (&acc)[2][TN], int tid) {
    constexpr int NT = mask_count(MASK);
    if constexpr (TI < NT) {
        h_kstep_m<T, TN, MASK, TI, (CC * NT + TI) % 3>(p, c, chunk, patch, ring, stage, af, acc, tid);
        h_chunk_m<T, TN, MASK, CC, TI + 1>(p, c, chunk, patch, ring, stage, af, acc, tid);
    }
}

// 1x1: one tap per chunk -> unroll three chunks so the ring index stays compile-time. Activations are fetched TWO chunks
// ahead into a 3-slot register rotation (slots = pairs of stage[]): a chunk is only 8-16 MFMAs (~300 cycles), so with the
// usual one-chunk distance every chunk waited out a full L2 round trip (stamps: 700-1300 cycles per chunk).
template <typename T, int TN, int R>
__device__ __forceinline__ void h_kstep_1x1(const ConvHArgs& p, const HCtx<T, TN>& c, int chunk, char* patch,
                                            u32x4 (&ring)[3][2][TN], u32x4 (&stage)[H_NI], u32x4 (&af)[2][2],
                                            f32x16 (&acc)[2][TN], int tid) {
    typedef typename HTraits<T>::vec vec;
    constexpr int CUR = R % 3, NXT2 = (R + 2) % 3;
    constexpr int S_LOAD = ((R + 2) % 3) * 2, S_WRITE = ((R + 1) % 3) * 2;      // chunk + 2 arrives, chunk + 1 goes to LDS
    static_assert(H_NI >= 6, "three 2-entry slots");
    const int kta = chunk + 2 < c.KT ? chunk + 2 : c.KT - 1;
#pragma unroll
    for (int s = 0; s < 2; ++s)
#pragma unroll
        for (int j = 0; j < TN; ++j)
            ring[NXT2][s][j] = *reinterpret_cast<const u32x4*>(c.wfrag[j] + ((size_t)kta * 2 + s) * 512);
    {
        const int cn = chunk + 2 < p.nchunks ? chunk + 2 : p.nchunks - 1;
        const int coff = p.x_off + cn * 32 + (tid & 3) * 8;
#pragma unroll
        for (int i = 0; i < 2; ++i) {                   // 1x1 patch = 128 pixels = 2 passes of 64
            const int px = c.pix[i] < 0 ? 0 : c.pix[i];
            stage[S_LOAD + i] = *reinterpret_cast<const u32x4*>(p.x + (size_t)px * p.x_ld + coff);
        }
    }
    __builtin_amdgcn_sched_barrier(0);
#pragma unroll
    for (int s = 0; s < 2; ++s)
#pragma unroll
        for (int j = 0; j < TN; ++j) {
            const vec b = __builtin_bit_cast(vec, ring[CUR][s][j]);
            acc[0][j] = HTraits<T>::mfma(__builtin_bit_cast(vec, af[0][s]), b, acc[0][j]);
            acc[1][j] = HTraits<T>::mfma(__builtin_bit_cast(vec, af[1][s]), b, acc[1][j]);
        }
    char* dst = patch + ((chunk + 1) & 1) * (p.patch_cap * H_PIX_BYTES) + (tid >> 2) * H_PIX_BYTES + (tid & 3) * 16;
#pragma unroll
    for (int i = 0; i < 2; ++i) {
        u32x4 z = {0u, 0u, 0u, 0u};
        *reinterpret_cast<u32x4*>(dst + 64 * i * H_PIX_BYTES) = c.pix[i] < 0 ? z : stage[S_WRITE + i];
    }
    __syncthreads();
    const char* An = patch + ((chunk + 1) & 1) * (p.patch_cap * H_PIX_BYTES);
#pragma unroll
    for (int i = 0; i < 2; ++i)
#pragma unroll
        for (int s = 0; s < 2; ++s) af[i][s] = *reinterpret_cast<const u32x4*>(An + c.a_off[i] + s * 32);
}

template <typename T, int KS, int STRIDE, int BN, int MASK>
__device__ __forceinline__ void conv_patch_h16_body(const ConvHArgs& p);

template <typename T, int KS, int STRIDE, int BN, int MASK = 0>
__global__ __launch_bounds__(256) void conv_patch_h16(const ConvHArgs p) { conv_patch_h16_body<T, KS, STRIDE, BN, MASK>(p); }

// Register cap for the 64-wide variants. Measured with per-block stamps: a CU held THREE blocks of the 1x1 variant
// (128 VGPRs + 32 AGPRs = 160) but never more than TWO of the 3x3 variant at 132 + 32 = 164, although the compiler's
// occupancy estimate says 3 for both (and LDS allows 4: tools/lds_occ_probe.hip) — the hardware allocates registers in
// coarser granules than the estimate assumes. With this attribute the compiler keeps the accumulators in VGPRs and lands at
// 154 (3x3) / 108 (1x1) registers in total; worth 1-2 % on the 64-wide layers.
template <typename T, int KS, int STRIDE, int MASK = 0>
__global__ __launch_bounds__(256) __attribute__((amdgpu_num_vgpr(128))) void conv_patch_h16_n64(const ConvHArgs p) {
    conv_patch_h16_body<T, KS, STRIDE, 64, MASK>(p);
}

template <typename T, int KS, int STRIDE, int BN, int MASK>
__device__ __forceinline__ void conv_patch_h16_body(const ConvHArgs& p) {
    constexpr int TN = BN / 64;
    static_assert(MASK == 0 || (KS == 3 && STRIDE == 1), "tap subsets are defined on the 3x3 stride-1 window");
    extern __shared__ __attribute__((aligned(16))) char smem_raw[];
    char* patch = smem_raw;                                             // [2 or 1][patch_cap][80 B]
    int* mtab = reinterpret_cast<int*>(patch + p.mtab_off);

    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int wm = wave >> 1, wn = wave & 1;
    const int fh = lane >> 5, frow = lane & 31;
#ifdef H16_STAMPS   // diagnostic build (make stamps): per-block phase stamps into the buffer passed as nan_flag
    const unsigned long long st0 = __builtin_amdgcn_s_memtime();
#endif

    if (p.stagger > 0 && (int)blockIdx.x < p.first_wave) {             // see conv_f32_v2.hip
        unsigned hw;
        asm volatile("s_getreg_b32 %0, hwreg(HW_REG_HW_ID)" : "=s"(hw));
        const int slot = (hw >> 16) & 15;
        for (int i = 0; i < slot * p.stagger; ++i) __builtin_amdgcn_s_sleep(32);
    }
    int bid = blockIdx.x;
    {
        const int nb = p.nblocks, q = nb / 8, r = nb % 8, xcd = bid % 8;
        bid = (xcd < r ? xcd * (q + 1) : r * (q + 1) + (xcd - r) * q) + bid / 8;
    }
    const int sp = fdiv(bid, p.mg_tn, p.tiles_n);
    const int n_tile = bid - sp * p.tiles_n;
    const int r_tile = fdiv(sp, p.mg_tw, p.tiles_w);
    const int w_tile = sp - r_tile * p.tiles_w;
    const int g0 = r_tile * p.TH, c0 = w_tile * p.TW;
    const int g_last = (g0 + p.TH < p.rows_total ? g0 + p.TH : p.rows_total) - 1;
    const int Hp = p.Hin + 2;
    auto vrow = [&](int g) {
        if (KS != 3) return g;
        const int n = fdiv(g, p.mg_H, p.H);
        return n * Hp + STRIDE * (g - n * p.H);
    };
    const int v0 = vrow(g0);
    const int PR = vrow(g_last) + (KS == 3 ? 3 : 1) - v0;

    HCtx<T, TN> c;
    c.KT = p.KT;
    {   // staged patch pixels of this 4-lane group: idx = (tid >> 2) + 64 i. Closed form per entry (two magic divisions):
        // the incremental version with carry loops was ~800 VALU instructions, 4-6k cycles of a 10k-cycle prologue
#pragma unroll
        for (int i = 0; i < H_NI; ++i) {
            const int idx = (tid >> 2) + 64 * i;
            const int pr = fdiv(idx, p.mg_PC, p.PC), pc = idx - pr * p.PC;
            int pix = -1;
            if (pr < PR) {
                if (KS == 3) {
                    const int vv = v0 + pr;
                    const int n = fdiv(vv, p.mg_Hp, Hp), yy = vv - n * Hp;
                    const int hi = yy - 1, wi = STRIDE * c0 + pc - 1;
                    if ((unsigned)hi < (unsigned)p.Hin && (unsigned)wi < (unsigned)p.Win) pix = (n * p.Hin + hi) * p.Win + wi;
                } else {
                    const int wi = c0 + pc;
                    if (wi < p.W) pix = wi;
                }
            }
            c.pix[i] = pix;
        }
    }
#pragma unroll
    for (int i = 0; i < 2; ++i) {
        const int pp = wm * 64 + i * 32 + frow;
        const int r = fdiv(pp, p.mg_TW, p.TW), cc = pp - r * p.TW;
        const int g = g0 + r;
        const bool ok = pp < p.TH * p.TW && g <= g_last && c0 + cc < p.W;
        c.a_off[i] = (ok ? ((vrow(g) - v0) * p.PC + STRIDE * cc) * H_PIX_BYTES : 0) + 16 * fh;
        if (MASK) c.a_off[i] += ((mask_nth(MASK, 0) / 3) * p.PC + mask_nth(MASK, 0) % 3) * H_PIX_BYTES;
    }
    if (tid < 128) {
        const int r = fdiv(tid, p.mg_TW, p.TW), cc = tid - r * p.TW;
        const int g = g0 + r;
        int m = -1;
        if (tid < p.TH * p.TW && g <= g_last && c0 + cc < p.W) {
            if (MASK) {                               // parity class: dx pixel (2r + ph, 2c + pw) of image n
                const int n = fdiv(g, p.mg_H, p.H), rr = g - n * p.H;
                m = (n * 2 * p.H + 2 * rr + p.cls_ph) * (2 * p.W) + 2 * (c0 + cc) + p.cls_pw;
            } else {
                m = g * p.W + c0 + cc;
            }
        }
        mtab[tid] = m;
    }
#pragma unroll
    for (int j = 0; j < TN; ++j) {
        const int nt = n_tile * (BN / 32) + j * 2 + wn;          // pass j of the epilogue = 64 CONTIGUOUS channels (full 128-B lines)
        c.wfrag[j] = p.wf + (size_t)nt * p.KT * 1024 + lane * 8;
    }

    f32x16 acc[2][TN];
#pragma unroll
    for (int i = 0; i < 2; ++i)
#pragma unroll
        for (int j = 0; j < TN; ++j)
#pragma unroll
            for (int r = 0; r < 16; ++r) acc[i][j][r] = 0.f;

    u32x4 ring[3][2][TN], stage[H_NI], af[2][2];
    // prologue: weight fragments of steps 0 and 1, patch of chunk 0
#pragma unroll
    for (int q = 0; q < 2; ++q) {
        const int kq = q < p.KT ? q : p.KT - 1;
#pragma unroll
        for (int s = 0; s < 2; ++s)
#pragma unroll
            for (int j = 0; j < TN; ++j)
                ring[q][s][j] = *reinterpret_cast<const u32x4*>(c.wfrag[j] + ((size_t)kq * 2 + s) * 512);
    }
    {
        const int coff = p.x_off + (tid & 3) * 8;
        char* dst = patch + (tid >> 2) * H_PIX_BYTES + (tid & 3) * 16;
#pragma unroll
        for (int i = 0; i < H_NI; ++i) {
            const int px = c.pix[i] < 0 ? 0 : c.pix[i];
            stage[i] = *reinterpret_cast<const u32x4*>(p.x + (size_t)px * p.x_ld + coff);
        }
#pragma unroll
        for (int i = 0; i < H_NI; ++i) {
            u32x4 z = {0u, 0u, 0u, 0u};
            if ((tid >> 2) + 64 * i < p.patch_cap) *reinterpret_cast<u32x4*>(dst + 64 * i * H_PIX_BYTES) = c.pix[i] < 0 ? z : stage[i];
        }
    }
    if constexpr (KS == 1 && MASK == 0) {             // 1x1: chunk 1 into register slot 1 (h_kstep_1x1 runs two chunks ahead)
        const int cn = 1 < p.nchunks ? 1 : 0;
        const int coff = p.x_off + cn * 32 + (tid & 3) * 8;
#pragma unroll
        for (int i = 0; i < 2; ++i) {
            const int px = c.pix[i] < 0 ? 0 : c.pix[i];
            stage[2 + i] = *reinterpret_cast<const u32x4*>(p.x + (size_t)px * p.x_ld + coff);
        }
    }
    __syncthreads();
#pragma unroll
    for (int i = 0; i < 2; ++i)
#pragma unroll
        for (int s = 0; s < 2; ++s) af[i][s] = *reinterpret_cast<const u32x4*>(patch + c.a_off[i] + s * 32);

#ifdef H16_STAMPS
    asm volatile("s_nop 0" ::: "memory");
    const unsigned long long st1 = __builtin_amdgcn_s_memtime();
#endif
    if constexpr (MASK != 0) {
        int chunk = 0;
        for (; chunk + 3 <= p.nchunks; chunk += 3) {
            h_chunk_m<T, TN, MASK, 0, 0>(p, c, chunk, patch, ring, stage, af, acc, tid);
            h_chunk_m<T, TN, MASK, 1, 0>(p, c, chunk + 1, patch, ring, stage, af, acc, tid);
            h_chunk_m<T, TN, MASK, 2, 0>(p, c, chunk + 2, patch, ring, stage, af, acc, tid);
        }
        if (chunk < p.nchunks) h_chunk_m<T, TN, MASK, 0, 0>(p, c, chunk, patch, ring, stage, af, acc, tid);
        if (chunk + 1 < p.nchunks) h_chunk_m<T, TN, MASK, 1, 0>(p, c, chunk + 1, patch, ring, stage, af, acc, tid);
    } else if constexpr (KS == 3) {
        for (int chunk = 0; chunk < p.nchunks; ++chunk) h_chunk<T, 3, TN, 0>(p, c, chunk, patch, ring, stage, af, acc, tid);
    } else {
        int chunk = 0;
        for (; chunk + 3 <= p.nchunks; chunk += 3) {
            h_kstep_1x1<T, TN, 0>(p, c, chunk, patch, ring, stage, af, acc, tid);
            h_kstep_1x1<T, TN, 1>(p, c, chunk + 1, patch, ring, stage, af, acc, tid);
            h_kstep_1x1<T, TN, 2>(p, c, chunk + 2, patch, ring, stage, af, acc, tid);
        }
        if (chunk < p.nchunks) h_kstep_1x1<T, TN, 0>(p, c, chunk, patch, ring, stage, af, acc, tid);
        if (chunk + 1 < p.nchunks) h_kstep_1x1<T, TN, 1>(p, c, chunk + 1, patch, ring, stage, af, acc, tid);
    }

#ifdef H16_STAMPS
    asm volatile("s_nop 0" ::: "memory");
    const unsigned long long st2 = __builtin_amdgcn_s_memtime();
#endif
    // ---------------------------------------------------------------------- epilogue (fp32 math)
    const bool has_res = p.flags & YOLO_FLAG_RESIDUAL;
    const bool nan_chk = p.flags & YOLO_FLAG_NANCHECK;
    const int HoWo = p.Ho * p.Wo;
    constexpr int OLD = 68;
    float* ost = reinterpret_cast<float*>(patch);                     // [128][68] fp32 = 34,816 B
    const bool vec_ok = (p.out_mode != YOLO_OUT_HEAD) && (p.Cout % 8 == 0);
    bool saw_nan = false;
    __syncthreads();
#pragma unroll
    for (int j = 0; j < TN; ++j) {
        {
            const int n = n_tile * BN + j * 64 + wn * 32 + frow;
            const bool nv = n < p.Cout;
            const float sc = nv ? (MASK ? 1.f : p.scale[n]) : 0.f;       // gradient kernels: plain accumulation
            const float sh = nv ? (MASK ? 0.f : p.shift[n]) : 0.f;
            float* dst = ost + wn * 32 + frow;
            YOLO_SWITCH_ACT(p.act,
                _Pragma("unroll") for (int i = 0; i < 2; ++i)
                    _Pragma("unroll") for (int r = 0; r < 16; ++r) {
                        const int row = wm * 64 + i * 32 + (r & 3) + 8 * (r >> 2) + 4 * fh;
                        dst[row * OLD] = act_c<ACT>(acc[i][j][r] * sc + sh);
                    })
        }
        __syncthreads();
        if (vec_ok) {
            unsigned short* yo = reinterpret_cast<unsigned short*>(p.y);
#pragma unroll
            for (int it = 0; it < 4; ++it) {
                const int idx = tid + 256 * it;
                const int row = idx >> 3, c8 = idx & 7;
                const int m = mtab[row];
                const int n = n_tile * BN + j * 64 + c8 * 8;
                if (m < 0 || n >= p.Cout) continue;
                const f32x4 v0 = *reinterpret_cast<const f32x4*>(ost + row * OLD + c8 * 8);
                const f32x4 v1 = *reinterpret_cast<const f32x4*>(ost + row * OLD + c8 * 8 + 4);
                float v[8] = {v0[0], v0[1], v0[2], v0[3], v1[0], v1[1], v1[2], v1[3]};
                if (has_res) {
                    const u32x4 rr = *reinterpret_cast<const u32x4*>(p.res + (size_t)m * p.r_ld + p.r_off + n);
#pragma unroll
                    for (int e = 0; e < 4; ++e) {
                        v[2 * e] += HTraits<T>::to_f32((unsigned short)(rr[e] & 0xffffu));
                        v[2 * e + 1] += HTraits<T>::to_f32((unsigned short)(rr[e] >> 16));
                    }
                }
                u32x4 o;
#pragma unroll
                for (int e = 0; e < 4; ++e) {
                    if (nan_chk && (v[2 * e] != v[2 * e] || v[2 * e + 1] != v[2 * e + 1])) saw_nan = true;
                    o[e] = (unsigned)HTraits<T>::from_f32(v[2 * e]) | ((unsigned)HTraits<T>::from_f32(v[2 * e + 1]) << 16);
                }
                if (p.out_mode == YOLO_OUT_NHWC) {
                    *reinterpret_cast<u32x4*>(yo + (size_t)m * p.y_ld + p.y_off + n) = o;
                } else {
                    const int img = m / HoWo;
                    const int rem = m - img * HoWo;
                    const int ho = rem / p.Wo;
                    const int wo2 = rem - ho * p.Wo;
                    const int W2 = 2 * p.Wo;
                    unsigned short* d = yo + ((size_t)(img * 2 * p.Ho + 2 * ho) * W2 + 2 * wo2) * p.y_ld + p.y_off + n;
                    *reinterpret_cast<u32x4*>(d) = o;
                    *reinterpret_cast<u32x4*>(d + p.y_ld) = o;
                    *reinterpret_cast<u32x4*>(d + (size_t)W2 * p.y_ld) = o;
                    *reinterpret_cast<u32x4*>(d + (size_t)(W2 + 1) * p.y_ld) = o;
                }
            }
        } else {                                    // detection heads: fp32 (B,3,g,g,5+nc); odd channel counts
            for (int it = 0; it < 32; ++it) {
                const int idx = tid + 256 * it;
                const int row = idx >> 6, col = idx & 63;
                const int m = mtab[row];
                const int n = n_tile * BN + j * 64 + col;
                if (m < 0 || n >= p.Cout) continue;
                float v = ost[row * OLD + col];
                if (has_res) v += HTraits<T>::to_f32(p.res[(size_t)m * p.r_ld + p.r_off + n]);
                if (nan_chk && v != v) saw_nan = true;
                const int img = m / HoWo;
                const int rem = m - img * HoWo;
                const int ho = rem / p.Wo;
                const int wo2 = rem - ho * p.Wo;
                if (p.out_mode == YOLO_OUT_HEAD) {
                    const int head_a = n / p.nc5, head_k = n - head_a * p.nc5;
                    reinterpret_cast<float*>(p.y)[((size_t)((img * 3 + head_a) * p.Ho + ho) * p.Wo + wo2) * p.nc5 + head_k] = v;
                } else if (p.out_mode == YOLO_OUT_NHWC) {
                    reinterpret_cast<unsigned short*>(p.y)[(size_t)m * p.y_ld + p.y_off + n] = HTraits<T>::from_f32(v);
                } else {
                    const int W2 = 2 * p.Wo;
                    unsigned short* d = reinterpret_cast<unsigned short*>(p.y) + ((size_t)(img * 2 * p.Ho + 2 * ho) * W2 + 2 * wo2) * p.y_ld + p.y_off + n;
                    const unsigned short hv = HTraits<T>::from_f32(v);
                    d[0] = hv; d[p.y_ld] = hv; d[(size_t)W2 * p.y_ld] = hv; d[(size_t)(W2 + 1) * p.y_ld] = hv;
                }
            }
        }
        if (j + 1 < TN) __syncthreads();
    }
    if (nan_chk && saw_nan) atomicOr(p.nan_flag, 2);
#ifdef H16_STAMPS
    {
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        const unsigned long long st3 = __builtin_amdgcn_s_memtime();
        if (tid == 0) {
            unsigned hw, xcc;
            asm volatile("s_getreg_b32 %0, hwreg(HW_REG_HW_ID)" : "=s"(hw));
            asm volatile("s_getreg_b32 %0, hwreg(HW_REG_XCC_ID)" : "=s"(xcc));
            unsigned long long* o = reinterpret_cast<unsigned long long*>(p.nan_flag) + (size_t)blockIdx.x * 6;
            o[0] = st0; o[1] = st1; o[2] = st2; o[3] = st3; o[4] = hw; o[5] = xcc;
        }
    }
#endif
}

// fragment-order 16-bit weights: [n_tile32][kt][s(2)][lane(64)][e(8)], n = nt*32 + (lane&31),
// ci = chunk*32 + s*16 + 8*(lane>>5) + e, (chunk, tap) = divmod(kt, ks*ks)
template <typename T>
__global__ void pack_weights_frag_h16(const float* __restrict__ w, unsigned short* __restrict__ wf, int cout, int cin, int ks,
                                      int KT, long long total) {
    const int taps = ks * ks;
    for (long long i = blockIdx.x * (long long)blockDim.x + threadIdx.x; i < total; i += (long long)gridDim.x * blockDim.x) {
        const int e = (int)(i & 7);
        const int lane = (int)((i >> 3) & 63);
        const int s = (int)((i >> 9) & 1);
        const long long rest = i >> 10;
        const int kt = (int)(rest % KT);
        const int nt = (int)(rest / KT);
        const int n = nt * 32 + (lane & 31);
        const int chunk = kt / taps, tap = kt - chunk * taps;
        const int ci = chunk * 32 + s * 16 + 8 * (lane >> 5) + e;
        const float v = (n < cout && ci < cin) ? w[((size_t)n * cin + ci) * taps + tap] : 0.f;
        wf[i] = HTraits<T>::from_f32(v);
    }
}

// same fragment order for the stride-1 input-gradient convolution dx = conv(dz, W'):
// n = ci, k channel = co, W'[ci][co][tap] = W[co][ci][taps-1-tap]  (see dgrad_f32.hip)
template <typename T>
__global__ void pack_dgrad_frag_h16(const float* __restrict__ w, unsigned short* __restrict__ wf, int cout, int cin, int ks,
                                    int KT, long long total) {
    const int taps = ks * ks;
    for (long long i = blockIdx.x * (long long)blockDim.x + threadIdx.x; i < total; i += (long long)gridDim.x * blockDim.x) {
        const int e = (int)(i & 7);
        const int lane = (int)((i >> 3) & 63);
        const int s = (int)((i >> 9) & 1);
        const long long rest = i >> 10;
        const int kt = (int)(rest % KT);
        const int nt = (int)(rest / KT);
        const int ci = nt * 32 + (lane & 31);
        const int chunk = kt / taps, tap = kt - chunk * taps;
        const int co = chunk * 32 + s * 16 + 8 * (lane >> 5) + e;
        const float v = (ci < cin && co < cout) ? w[((size_t)co * cin + ci) * taps + (taps - 1 - tap)] : 0.f;
        wf[i] = HTraits<T>::from_f32(v);
    }
}

// ---- many layers in ONE launch: an optimizer step changes every weight tensor, and 75 + 70 separate ~6 us pack launches
// per fine-tune step were 3 % of the bf16 step (the conversion itself is 0.1 ms of HBM time). Items ride in the kernel
// argument; a block finds its item by a scan of the (<= 48) first-block numbers.
constexpr int H_PACK_BATCH = 48;
struct PackItemH { const float* w; unsigned short* wf; int cout, cin, ks, KT; long long total; int first_block, nblocks; };
struct PackBatchH { PackItemH it[H_PACK_BATCH]; int n; };

template <typename T, bool DGRAD>
__global__ void pack_batch_h16(const PackBatchH b) {
    int k = 0;
    while (k + 1 < b.n && (int)blockIdx.x >= b.it[k + 1].first_block) ++k;
    const PackItemH& q = b.it[k];
    const int taps = q.ks * q.ks;
    const long long start = ((long long)blockIdx.x - q.first_block) * blockDim.x + threadIdx.x;
    for (long long i = start; i < q.total; i += (long long)q.nblocks * blockDim.x) {
        const int e = (int)(i & 7);
        const int lane = (int)((i >> 3) & 63);
        const int s = (int)((i >> 9) & 1);
        const long long rest = i >> 10;
        const int kt = (int)(rest % q.KT);
        const int nt = (int)(rest / q.KT);
        const int chunk = kt / taps, tap = kt - chunk * taps;
        const int a = nt * 32 + (lane & 31);                      // GEMM n: output channel (forward) / input channel (dgrad)
        const int c = chunk * 32 + s * 16 + 8 * (lane >> 5) + e;  // GEMM k channel
        float v = 0.f;
        if (DGRAD) { if (a < q.cin && c < q.cout) v = q.w[((size_t)c * q.cin + a) * taps + (taps - 1 - tap)]; }
        else       { if (a < q.cout && c < q.cin) v = q.w[((size_t)a * q.cin + c) * taps + tap]; }
        q.wf[i] = HTraits<T>::from_f32(v);
    }
}

// ------------------------------------------------------------------------------ host side
static const bool g_h_stagger = !(getenv("YOLO_NO_STAGGER"));

size_t h16_frag_elems(int cout, int cin, int ks) {
    const int cinp = round_up(cin, 32);
    return (size_t)(round_up(cout, 128) / 32) * (cinp / 32) * ks * ks * 1024;
}

int h16_pack(const float* w_oihw, void* wf, int cout, int cin, int ks, int dtype, hipStream_t s) {
    const long long total = (long long)h16_frag_elems(cout, cin, ks);
    const int KT = (round_up(cin, 32) / 32) * ks * ks;
    const int grid = (int)((total + 255) / 256 < 8192 ? (total + 255) / 256 : 8192);
    if (dtype == YOLO_BF16)
        hipLaunchKernelGGL(pack_weights_frag_h16<__bf16>, dim3(grid), dim3(256), 0, s, w_oihw, (unsigned short*)wf, cout, cin, ks, KT, total);
    else
        hipLaunchKernelGGL(pack_weights_frag_h16<_Float16>, dim3(grid), dim3(256), 0, s, w_oihw, (unsigned short*)wf, cout, cin, ks, KT, total);
    return check_launch("pack_weights_frag_h16");
}

int h16_pack_dgrad(const float* w_oihw, void* wf, int cout, int cin, int ks, int dtype, hipStream_t s) {
    const int coutp = round_up(cout, 32);
    const long long total = (long long)h16_frag_elems(cin, coutp, ks);
    const int KT = (coutp / 32) * ks * ks;
    const int grid = (int)((total + 255) / 256 < 8192 ? (total + 255) / 256 : 8192);
    if (dtype == YOLO_BF16)
        hipLaunchKernelGGL(pack_dgrad_frag_h16<__bf16>, dim3(grid), dim3(256), 0, s, w_oihw, (unsigned short*)wf, cout, cin, ks, KT, total);
    else
        hipLaunchKernelGGL(pack_dgrad_frag_h16<_Float16>, dim3(grid), dim3(256), 0, s, w_oihw, (unsigned short*)wf, cout, cin, ks, KT, total);
    return check_launch("pack_dgrad_frag_h16");
}

// items: host array. dgrad = 1: the flipped / transposed stride-1 input-gradient weights (h16_pack_dgrad layout)
int h16_pack_batch(const float* const* w, void* const* wf, const int* cout, const int* cin, const int* ks, int n, int dgrad, int dtype,
                   hipStream_t s) {
    for (int base = 0; base < n; base += H_PACK_BATCH) {
        PackBatchH b;
        b.n = n - base < H_PACK_BATCH ? n - base : H_PACK_BATCH;
        int blocks = 0;
        for (int j = 0; j < b.n; ++j) {
            const int i = base + j;
            PackItemH& q = b.it[j];
            q.w = w[i]; q.wf = (unsigned short*)wf[i]; q.cout = cout[i]; q.cin = cin[i]; q.ks = ks[i];
            if (dgrad) {
                const int coutp = round_up(cout[i], 32);
                q.total = (long long)h16_frag_elems(cin[i], coutp, ks[i]);
                q.KT = (coutp / 32) * ks[i] * ks[i];
            } else {
                q.total = (long long)h16_frag_elems(cout[i], cin[i], ks[i]);
                q.KT = (round_up(cin[i], 32) / 32) * ks[i] * ks[i];
            }
            const long long nb = (q.total + 255) / 256;
            q.nblocks = (int)(nb < 1024 ? nb : 1024);
            q.first_block = blocks;
            blocks += q.nblocks;
        }
        if (dtype == YOLO_BF16) {
            if (dgrad) hipLaunchKernelGGL((pack_batch_h16<__bf16, true>), dim3(blocks), dim3(256), 0, s, b);
            else hipLaunchKernelGGL((pack_batch_h16<__bf16, false>), dim3(blocks), dim3(256), 0, s, b);
        } else {
            if (dgrad) hipLaunchKernelGGL((pack_batch_h16<_Float16, true>), dim3(blocks), dim3(256), 0, s, b);
            else hipLaunchKernelGGL((pack_batch_h16<_Float16, false>), dim3(blocks), dim3(256), 0, s, b);
        }
        const int rc = check_launch("pack_batch_h16");
        if (rc) return rc;
    }
    return YOLO_OK;
}

// ---- stride-2 input gradient (transposed conv) as four stride-1 tap-subset convolutions over dz -------------
//   dx[n, 2r+ph, 2c+pw, ci] = sum_{dh <= ph, dw <= pw, co} dz[n, r+dh, c+dw, co] * W[co, ci, ph+1-2dh, pw+1-2dw]
// In the 3x3 window of the patch kernel (pad 1) the offset (dh, dw) is tap (1+dh, 1+dw): class (ph, pw) uses the
// taps {1, 1+ph} x {1, 1+pw} — 1, 2, 2 and 4 of them, 9 in total, so no matrix work is spent on structural zeros.
constexpr int cls_mask(int ph, int pw) {
    int m = 0;
    for (int dh = 0; dh <= ph; ++dh)
        for (int dw = 0; dw <= pw; ++dw) m |= 1 << ((1 + dh) * 3 + 1 + dw);
    return m;
}
static size_t cls_frag_elems(int cin, int cout, int cls) {       // N = cin (dx channels), K = cout
    const int nt = mask_count(cls_mask(cls >> 1, cls & 1));
    return (size_t)(round_up(cin, 128) / 32) * (cout / 32) * nt * 1024;
}

template <typename T>
__global__ void pack_dgrad_s2_cls_h16(const float* __restrict__ w, unsigned short* __restrict__ wf, int cout, int cin, int ph, int pw,
                                      int NT, int KT, long long total) {
    for (long long i = blockIdx.x * (long long)blockDim.x + threadIdx.x; i < total; i += (long long)gridDim.x * blockDim.x) {
        const int e = (int)(i & 7);
        const int lane = (int)((i >> 3) & 63);
        const int s = (int)((i >> 9) & 1);
        const long long rest = i >> 10;
        const int kt = (int)(rest % KT);
        const int nt = (int)(rest / KT);
        const int ci = nt * 32 + (lane & 31);
        const int chunk = kt / NT, t = kt - chunk * NT;
        const int dh = pw ? t / 2 : t, dw = pw ? t % 2 : 0;       // taps in window order: dh-major, dw-minor
        const int kh = ph + 1 - 2 * dh, kw = pw + 1 - 2 * dw;
        const int co = chunk * 32 + s * 16 + 8 * (lane >> 5) + e;
        const float v = (ci < cin && co < cout) ? w[((size_t)co * cin + ci) * 9 + kh * 3 + kw] : 0.f;
        wf[i] = HTraits<T>::from_f32(v);
    }
}

size_t h16_dgrad_s2_elems(int cout, int cin) {
    size_t n = 0;
    for (int cls = 0; cls < 4; ++cls) n += cls_frag_elems(cin, cout, cls);
    return n;
}

int h16_pack_dgrad_s2(const float* w_oihw, void* wf, int cout, int cin, int dtype, hipStream_t s) {
    unsigned short* dst = (unsigned short*)wf;
    for (int cls = 0; cls < 4; ++cls) {
        const int ph = cls >> 1, pw = cls & 1;
        const int NT = mask_count(cls_mask(ph, pw));
        const long long total = (long long)cls_frag_elems(cin, cout, cls);
        const int KT = (cout / 32) * NT;
        const int grid = (int)((total + 255) / 256 < 8192 ? (total + 255) / 256 : 8192);
        if (dtype == YOLO_BF16)
            hipLaunchKernelGGL(pack_dgrad_s2_cls_h16<__bf16>, dim3(grid), dim3(256), 0, s, w_oihw, dst, cout, cin, ph, pw, NT, KT, total);
        else
            hipLaunchKernelGGL(pack_dgrad_s2_cls_h16<_Float16>, dim3(grid), dim3(256), 0, s, w_oihw, dst, cout, cin, ph, pw, NT, KT, total);
        const int rc = check_launch("pack_dgrad_s2_cls_h16");
        if (rc) return rc;
        dst += total;
    }
    return YOLO_OK;
}

static void fill_magics(ConvHArgs& a) {
    a.mg_H = magic_of(a.H); a.mg_TW = magic_of(a.TW); a.mg_PC = magic_of(a.PC);
    a.mg_tn = magic_of(a.tiles_n); a.mg_tw = magic_of(a.tiles_w); a.mg_Hp = magic_of(a.Hin + 2);
}

static void pick_tile_h(int Hin, int Hout, int Wout, int ks, int stride, int* th, int* tw, int* prmax) {
    if (ks == 1) { *th = 1; *tw = 128; *prmax = 1; return; }
    double best = -1;
    *th = 1; *tw = 1; *prmax = 3 + 2;
    for (int TW = 1; TW <= (Wout < 126 ? Wout : 126); ++TW) {
        int TH = 128 / TW;
        int pr = 0;
        while (TH >= 1) {
            const int cross = (TH + Hout - 1) / Hout;
            pr = stride * (TH - 1) + 3 + 2 * cross;
            if (pr * (stride * (TW - 1) + 3) <= H_PATCH_CAP) break;
            --TH;
        }
        if (TH < 1) continue;
        const double eff = ((double)Wout / (ceil_div(Wout, TW) * TW)) * (TH * TW / 128.0);
        if (eff > best + 1e-9) { best = eff; *th = TH; *tw = TW; *prmax = pr; }
    }
    (void)Hin;
}

template <typename T, int KS, int STRIDE, int BN>
static int launch_h(ConvHArgs& a, hipStream_t s) {
    a.tiles_n = ceil_div(a.Cout, BN);
    const int tiles_r = ceil_div(a.rows_total, a.TH);
    a.nblocks = a.tiles_n * a.tiles_w * tiles_r;
    fill_magics(a);
    a.first_wave = 2 * 256;
    const long mfma_cycles = (long)a.KT * 8 * (BN / 64) / 2 * 32;      // one block's matrix cycles per wave
    a.stagger = g_h_stagger ? (int)((mfma_cycles + 1024) / 2048) : 0;   // s_sleep 32 = 2048 cycles
    // Stride-2 3x3: the patch of 128 output pixels is ~500 input pixels, and two buffers of it (82 KB) leave ONE block per
    // CU (measured: >= 82 KB -> 1, 42-52 KB -> 3), i.e. nothing to overlap a block's staging and epilogue with. One buffer
    // + one more barrier per 32-channel chunk instead; the region also holds the epilogue's 128 x 68 fp32 staging tile.
    a.bufmask = (KS == 3 && STRIDE == 2) ? 0 : 1;
    size_t patch_bytes = (size_t)(a.bufmask + 1) * a.patch_cap * H_PIX_BYTES;
    if (patch_bytes < 128 * 68 * sizeof(float)) patch_bytes = 128 * 68 * sizeof(float);
    a.mtab_off = (int)patch_bytes;
    const size_t lds = patch_bytes + 128 * sizeof(int);
    if constexpr (BN == 64) hipLaunchKernelGGL((conv_patch_h16_n64<T, KS, STRIDE>), dim3(a.nblocks), dim3(256), lds, s, a);
    else hipLaunchKernelGGL((conv_patch_h16<T, KS, STRIDE, BN>), dim3(a.nblocks), dim3(256), lds, s, a);
    return check_launch("conv_patch_h16");
}

template <typename T, int BN, int MASK>
static int launch_cls(ConvHArgs& a, hipStream_t s) {
    a.tiles_n = ceil_div(a.Cout, BN);
    const int tiles_r = ceil_div(a.rows_total, a.TH);
    a.nblocks = a.tiles_n * a.tiles_w * tiles_r;
    fill_magics(a);
    a.first_wave = 2 * 256;
    const long mfma_cycles = (long)a.KT * 8 * (BN / 64) / 2 * 32;
    a.stagger = g_h_stagger ? (int)((mfma_cycles + 1024) / 2048) : 0;
    a.bufmask = 1;
    a.mtab_off = 2 * a.patch_cap * H_PIX_BYTES;
    const size_t lds = (size_t)a.mtab_off + 128 * sizeof(int);
    if constexpr (BN == 64) hipLaunchKernelGGL((conv_patch_h16_n64<T, 3, 1, MASK>), dim3(a.nblocks), dim3(256), lds, s, a);
    else hipLaunchKernelGGL((conv_patch_h16<T, 3, 1, BN, MASK>), dim3(a.nblocks), dim3(256), lds, s, a);
    return check_launch("conv_patch_h16 (dgrad s2 class)");
}

template <typename T>
static int dgrad_s2_classes(ConvHArgs& a, const unsigned short* wf, int cin, int cout, int bn, hipStream_t s) {
    for (int cls = 0; cls < 4; ++cls) {
        a.cls_ph = cls >> 1; a.cls_pw = cls & 1;
        a.wf = wf;
        a.KT = a.nchunks * mask_count(cls_mask(a.cls_ph, a.cls_pw));
        int rc;
        if (bn == 128) {
            rc = cls == 0 ? launch_cls<T, 128, cls_mask(0, 0)>(a, s) : cls == 1 ? launch_cls<T, 128, cls_mask(0, 1)>(a, s)
               : cls == 2 ? launch_cls<T, 128, cls_mask(1, 0)>(a, s) : launch_cls<T, 128, cls_mask(1, 1)>(a, s);
        } else {
            rc = cls == 0 ? launch_cls<T, 64, cls_mask(0, 0)>(a, s) : cls == 1 ? launch_cls<T, 64, cls_mask(0, 1)>(a, s)
               : cls == 2 ? launch_cls<T, 64, cls_mask(1, 0)>(a, s) : launch_cls<T, 64, cls_mask(1, 1)>(a, s);
        }
        if (rc) return rc;
        wf += cls_frag_elems(cin, cout, cls);
    }
    return YOLO_OK;
}

// dx (n, 2ho, 2wo, cin) [+ residual] from dz (n, ho, wo, cout), weights from h16_pack_dgrad_s2
int dgrad_s2_h16_launch(const void* dz, int dz_ld, int dz_off, const void* wf, const void* residual, int r_ld, int r_off, void* dx,
                        int dx_ld, int dx_off, int n, int ho, int wo, int cin, int cout, int dtype, hipStream_t s) {
    if (cout % 32) return fail(YOLO_ERR_UNSUPPORTED, "dgrad_s2 (16-bit): cout %d must be a multiple of 32", cout);
    if ((dz_ld & 7) || (dz_off & 7)) return fail(YOLO_ERR_ARG, "dgrad_s2 (16-bit): dz_ld/dz_off must be multiples of 8");
    ConvHArgs a;
    a.x = (const unsigned short*)dz; a.scale = nullptr; a.shift = nullptr;
    a.res = (const unsigned short*)residual; a.y = dx; a.nan_flag = nullptr;
    a.Cin = cout; a.Cout = cin;
    a.x_ld = dz_ld; a.x_off = dz_off; a.y_ld = dx_ld; a.y_off = dx_off; a.r_ld = r_ld; a.r_off = r_off;
    a.Hin = ho; a.Win = wo; a.Ho = ho; a.Wo = wo;
    const long long M = (long long)n * ho * wo;
    if (M * 4 > 0x7fffffffLL) return fail(YOLO_ERR_UNSUPPORTED, "dgrad_s2: too many pixels");
    int prmax = 1;
    a.H = ho; a.W = wo; a.rows_total = n * ho;
    pick_tile_h(ho, ho, wo, 3, 1, &a.TH, &a.TW, &prmax);
    a.PC = a.TW + 2;
    a.patch_cap = round_up(prmax * a.PC, 8);               // see conv_h16_launch: keep two patch buffers under 40 KB when possible
    if (a.patch_cap < 224) a.patch_cap = 224;
    if (a.patch_cap > H_PATCH_CAP) return fail(YOLO_ERR_UNSUPPORTED, "dgrad_s2 (16-bit): patch too large");
    a.tiles_w = ceil_div(a.W, a.TW);
    a.nchunks = cout / 32;
    a.act = YOLO_ACT_NONE; a.out_mode = YOLO_OUT_NHWC; a.flags = residual ? YOLO_FLAG_RESIDUAL : 0;
    a.nc5 = 1;
    const int bn = (cin > 64 && ho <= 52) ? 128 : 64;
    if (dtype == YOLO_BF16) return dgrad_s2_classes<__bf16>(a, (const unsigned short*)wf, cin, cout, bn, s);
    return dgrad_s2_classes<_Float16>(a, (const unsigned short*)wf, cin, cout, bn, s);
}

template <typename T>
static int dispatch_h(ConvHArgs& a, int ks, int stride, int bn, hipStream_t s) {
    if (ks == 1) return bn == 128 ? launch_h<T, 1, 1, 128>(a, s) : launch_h<T, 1, 1, 64>(a, s);
    if (stride == 1) return bn == 128 ? launch_h<T, 3, 1, 128>(a, s) : launch_h<T, 3, 1, 64>(a, s);
    return bn == 128 ? launch_h<T, 3, 2, 128>(a, s) : launch_h<T, 3, 2, 64>(a, s);
}

int conv_h16_launch(const yolo_conv_desc* d, const void* x, const void* wf, const float* scale, const float* shift,
                    const void* residual, void* y, int32_t* nan_flag, hipStream_t s) {
    if (d->cin % 32) return fail(YOLO_ERR_UNSUPPORTED, "conv (16-bit): cin %d must be a multiple of 32", d->cin);
    if (d->ksize == 1 && d->stride != 1) return fail(YOLO_ERR_UNSUPPORTED, "conv (16-bit): strided 1x1");
    if ((d->x_ld & 7) || (d->x_off & 7)) return fail(YOLO_ERR_ARG, "conv (16-bit): x_ld/x_off must be multiples of 8");
    ConvHArgs a;
    a.x = (const unsigned short*)x; a.wf = (const unsigned short*)wf; a.scale = scale; a.shift = shift;
    a.res = (const unsigned short*)residual; a.y = y; a.nan_flag = nan_flag;
    a.Cin = d->cin; a.Cout = d->cout;
    a.x_ld = d->x_ld; a.x_off = d->x_off; a.y_ld = d->y_ld; a.y_off = d->y_off; a.r_ld = d->r_ld; a.r_off = d->r_off;
    const int pad = d->ksize / 2;
    a.Hin = d->h; a.Win = d->w;
    a.Ho = (d->h + 2 * pad - d->ksize) / d->stride + 1;
    a.Wo = (d->w + 2 * pad - d->ksize) / d->stride + 1;
    if (d->stride == 2 && ((d->h & 1) || (d->w & 1))) return fail(YOLO_ERR_UNSUPPORTED, "conv (16-bit): stride 2 needs even H, W");
    const long long M = (long long)d->n * a.Ho * a.Wo;
    if (M > 0x7fffffffLL) return fail(YOLO_ERR_UNSUPPORTED, "conv: N*H*W exceeds int32");
    int prmax = 1;
    if (d->ksize == 1) {
        a.H = 1; a.W = (int)M; a.rows_total = 1; a.TH = 1; a.TW = 128; a.PC = 128;
    } else {
        a.H = a.Ho; a.W = a.Wo; a.rows_total = d->n * a.Ho;
        pick_tile_h(d->h, a.Ho, a.Wo, 3, d->stride, &a.TH, &a.TW, &prmax);
        a.PC = d->stride * (a.TW - 1) + 3;
    }
    // Not rounded up to the staging granularity of 64 pixels (the stores are guarded): measured with per-block stamps, a CU
    // never held more than TWO of the 64-wide 3x3 blocks although registers and the occupancy API allow three — their
    // 2 x 256 x 80 B = 41.5 KB of LDS did not pack three to a CU (consistent with allocations not straddling the two 80 KB
    // halves of the 160 KB LDS: 2 x 41.5 KB > 80 KB), while the 36.4 KB 1x1 blocks did run three deep.
    a.patch_cap = round_up(prmax * a.PC, 8);
    if (a.patch_cap < 224) a.patch_cap = 224;              // epilogue stages 128 x 68 fp32 in the patch region
    if (a.patch_cap > H_PATCH_CAP) return fail(YOLO_ERR_UNSUPPORTED, "conv (16-bit): patch too large");
    a.tiles_w = ceil_div(a.W, a.TW);
    a.nchunks = d->cin / 32;
    a.KT = a.nchunks * d->ksize * d->ksize;
    a.act = d->act; a.out_mode = d->out_mode; a.flags = d->flags;
    a.nc5 = d->out_mode == YOLO_OUT_HEAD ? d->cout / 3 : 1;
    // measured (tools/conv_bench.py --dtype bf16 --tile 5,6, batch 32): 1x1 layers are latency/HBM-bound and want more,
    // smaller blocks (BN = 64); every 3x3 with more than 64 output channels gains from BN = 128 (64->128 @104: 81 vs 98 us,
    // 64->128 s2 @208: 119 vs 131 us, 13x13 .. 52x52: 15-20 %) - the earlier "only up to 52x52" rule predated the epilogue fixes
    const int auto_bn = (d->ksize == 3 && d->cout > 64) ? 128 : 64;   // same-box A/B of the whole forward: +0.7 % at 416x416, +3.8 % at 608x608
    const int bn = d->tile == 5 ? 64 : (d->tile == 6 ? 128 : auto_bn);
    if (d->dtype == YOLO_BF16) return dispatch_h<__bf16>(a, d->ksize, d->stride, bn, s);
    return dispatch_h<_Float16>(a, d->ksize, d->stride, bn, s);
}

}  // namespace yolo
