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
    int prio;                            // conv3_dma_h16: prologue / epilogue at s_setprio 2 (A/B switch YOLO_DMA_PRIO=0)
    unsigned qperm;                      // conv3_dma_h16: nibble q = pixel quad of lane quad q within a 32-pixel m-tile
    int cls_ph, cls_pw;                  // MASK kernels (stride-2 input gradient): output pixel (2r+ph, 2c+pw)
    float* stats = nullptr;              // DMA kernels, training: per-wave BatchNorm partial sums [row][2][stats_ld] (null: ordinary epilogue)
    int stats_ld = 0;
    // backward statistics (input-gradient launches): the block that PRODUCED this convolution's input - its conv output z and
    // BatchNorm tables. Non-null: the epilogue (identity [+ residual]) also sums du = dx * act'(bn(z)) and du * (z - mean) per channel
    const unsigned short* bz = nullptr;
    const float* bmean = nullptr;
    const float* bscale = nullptr;
    const float* bshift = nullptr;
    int bz_ld = 0, bz_off = 0, bact = 0;
    // magic multipliers of the prologue's index divisions (a wave64 integer division is ~40 VALU instructions;
    // ~20 of them per thread were most of a 10k-cycle prologue in front of 9k cycles of matrix work)
    unsigned mg_H, mg_TW, mg_PC, mg_tn, mg_tw, mg_Hp;
};

// x / d for 0 <= x < 2^31 with mg = ceil(2^32 / d) (d >= 2) or 0 (d == 1): the estimate is q or q + 1, one fix-up
__device__ __forceinline__ int fdiv(int x, unsigned mg, int d) {
    if (!mg) return x;
    const unsigned q = __umulhi((unsigned)x, mg);
    return (int)(q * (unsigned)d > (unsigned)x ? q - 1 : q);       // q*d <= x + d < 2^32: no 64-bit multiply needed
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

// two fp32 -> one dword of two 16-bit values (low half = a): ONE v_cvt_pk_{bf16,f16}_f32 instead of two conversions + shift + or
template <typename T> __device__ __forceinline__ unsigned pack2(float a, float b);
template <> __device__ __forceinline__ unsigned pack2<__bf16>(float a, float b) {
    typedef float f2 __attribute__((ext_vector_type(2)));
    typedef __bf16 b2 __attribute__((ext_vector_type(2)));
    const f2 v = {a, b};
    return __builtin_bit_cast(unsigned, __builtin_convertvector(v, b2));
}
template <> __device__ __forceinline__ unsigned pack2<_Float16>(float a, float b) {
    typedef float f2 __attribute__((ext_vector_type(2)));
    typedef _Float16 h2 __attribute__((ext_vector_type(2)));
    const f2 v = {a, b};
    return __builtin_bit_cast(unsigned, __builtin_convertvector(v, h2));
}

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
    constexpr int NI = KS == 1 ? 2 : H_NI;                             // staged pixels per 4-lane group that can be live (1x1: 128-pixel patch)
    static_assert(MASK == 0 || (KS == 3 && STRIDE == 1), "tap subsets are defined on the 3x3 stride-1 window");
    extern __shared__ __attribute__((aligned(16))) char smem_raw[];
    char* patch = smem_raw;                                             // [2 or 1][patch_cap][80 B]
    int* mtab = reinterpret_cast<int*>(patch + p.mtab_off);             // [128] output pixel of tile row, [128] head-layout base

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

    // ---- prologue. Order matters: a 16-bit block's matrix work is ~9k cycles, so every exposed memory round trip counts.
    // (1) weight fragments of K steps 0 and 1 and the folded BatchNorm scale / shift need nothing but n_tile: request them
    //     FIRST, so they travel while the patch indices are computed (the index math used to run in front of every load)
    HCtx<T, TN> c;
    c.KT = p.KT;
#pragma unroll
    for (int j = 0; j < TN; ++j) {
        const int nt = n_tile * (BN / 32) + j * 2 + wn;          // pass j of the epilogue = 64 CONTIGUOUS channels (full 128-B lines)
        c.wfrag[j] = p.wf + (size_t)nt * p.KT * 1024 + lane * 8;
    }
    u32x4 ring[3][2][TN], stage[H_NI], af[2][2];
#pragma unroll
    for (int q = 0; q < 2; ++q) {
        const int kq = q < p.KT ? q : p.KT - 1;
#pragma unroll
        for (int s = 0; s < 2; ++s)
#pragma unroll
            for (int j = 0; j < TN; ++j)
                ring[q][s][j] = *reinterpret_cast<const u32x4*>(c.wfrag[j] + ((size_t)kq * 2 + s) * 512);
    }
    float sc[TN], sh[TN];                                       // this lane's output channel of pass j: n_tile*BN + j*64 + wn*32 + frow
#pragma unroll
    for (int j = 0; j < TN; ++j) {
        sc[j] = 1.f; sh[j] = 0.f;                              // gradient kernels (MASK): plain accumulation
        if (!MASK) {
            const int n = n_tile * BN + j * 64 + wn * 32 + frow;
            const int ncl = n < p.Cout ? n : p.Cout - 1;       // clamped: unconditional loads
            sc[j] = p.scale[ncl];
            sh[j] = p.shift[ncl];
            if (n >= p.Cout) { sc[j] = 0.f; sh[j] = 0.f; }
        }
    }
    __builtin_amdgcn_sched_barrier(0);

    // (2) patch geometry. Branch-free: every entry is computed for every lane and invalidated by a select (the nested
    //     ifs compiled to ~20 exec-mask branches per lane)
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
#pragma unroll
    for (int i = 0; i < H_NI; ++i) c.pix[i] = -1;
#pragma unroll
    for (int i = 0; i < NI; ++i) {   // staged patch pixels of this 4-lane group: idx = (tid >> 2) + 64 i
        const int idx = (tid >> 2) + 64 * i;
        const int pr = fdiv(idx, p.mg_PC, p.PC), pc = idx - pr * p.PC;
        int pix;
        bool ok;
        if (KS == 3) {
            const int vv = v0 + pr;
            const int n = fdiv(vv, p.mg_Hp, Hp), yy = vv - n * Hp;
            const int hi = yy - 1, wi = STRIDE * c0 + pc - 1;
            ok = (pr < PR) & ((unsigned)hi < (unsigned)p.Hin) & ((unsigned)wi < (unsigned)p.Win);
            pix = (n * p.Hin + hi) * p.Win + wi;
        } else {
            pix = c0 + pc;
            ok = (pr < PR) & (pix < p.W);
        }
        c.pix[i] = ok ? pix : -1;
    }
    // (3) patch of chunk 0 (and, 1x1, of chunk 1 into register slot 1: h_kstep_1x1 runs two chunks ahead)
    {
        const int coff = p.x_off + (tid & 3) * 8;
#pragma unroll
        for (int i = 0; i < NI; ++i) {
            const int px = c.pix[i] < 0 ? 0 : c.pix[i];
            stage[i] = *reinterpret_cast<const u32x4*>(p.x + (size_t)px * p.x_ld + coff);
        }
    }
    if constexpr (KS == 1 && MASK == 0) {
        const int cn = 1 < p.nchunks ? 1 : 0;
        const int coff = p.x_off + cn * 32 + (tid & 3) * 8;
#pragma unroll
        for (int i = 0; i < 2; ++i) {
            const int px = c.pix[i] < 0 ? 0 : c.pix[i];
            stage[2 + i] = *reinterpret_cast<const u32x4*>(p.x + (size_t)px * p.x_ld + coff);
        }
    }
    __builtin_amdgcn_sched_barrier(0);
    // (4) while those loads are in flight: A-fragment offsets and the tile-row -> output-pixel table of the epilogue
#pragma unroll
    for (int i = 0; i < 2; ++i) {
        const int pp = wm * 64 + i * 32 + frow;
        const int r = fdiv(pp, p.mg_TW, p.TW), cc = pp - r * p.TW;
        const int g = g0 + r;
        const bool ok = (pp < p.TH * p.TW) & (g <= g_last) & (c0 + cc < p.W);
        c.a_off[i] = (ok ? ((vrow(g) - v0) * p.PC + STRIDE * cc) * H_PIX_BYTES : 0) + 16 * fh;
        if (MASK) c.a_off[i] += ((mask_nth(MASK, 0) / 3) * p.PC + mask_nth(MASK, 0) % 3) * H_PIX_BYTES;
    }
    if (tid < 128) {
        const int r = fdiv(tid, p.mg_TW, p.TW), cc = tid - r * p.TW;
        const int g = g0 + r;
        int m = -1, mh = 0;
        if (tid < p.TH * p.TW && g <= g_last && c0 + cc < p.W) {
            if (MASK) {                               // parity class: dx pixel (2r + ph, 2c + pw) of image n
                const int n = fdiv(g, p.mg_H, p.H), rr = g - n * p.H;
                m = (n * 2 * p.H + 2 * rr + p.cls_ph) * (2 * p.W) + 2 * (c0 + cc) + p.cls_pw;
            } else {
                m = g * p.W + c0 + cc;
                if (p.out_mode == YOLO_OUT_HEAD) mh = m + 2 * (m / (p.Ho * p.Wo)) * (p.Ho * p.Wo);   // (img*3)*HoWo + pixel
            }
        }
        mtab[tid] = m;
        mtab[128 + tid] = mh;
    }
    {
        char* dst = patch + (tid >> 2) * H_PIX_BYTES + (tid & 3) * 16;
#pragma unroll
        for (int i = 0; i < NI; ++i) {
            u32x4 z = {0u, 0u, 0u, 0u};
            if (i < 3 || (tid >> 2) + 64 * i < p.patch_cap)              // patch_cap >= 224: the first three always fit
                *reinterpret_cast<u32x4*>(dst + 64 * i * H_PIX_BYTES) = c.pix[i] < 0 ? z : stage[i];
        }
    }
    __syncthreads();
#pragma unroll
    for (int i = 0; i < 2; ++i)
#pragma unroll
        for (int s = 0; s < 2; ++s) af[i][s] = *reinterpret_cast<const u32x4*>(patch + c.a_off[i] + s * 32);
#pragma unroll
    for (int j = 0; j < TN; ++j) asm volatile("" : "+v"(sc[j]), "+v"(sh[j]));   // pinned here: not re-loaded in the epilogue

    f32x16 acc[2][TN];
#pragma unroll
    for (int i = 0; i < 2; ++i)
#pragma unroll
        for (int j = 0; j < TN; ++j)
#pragma unroll
            for (int r = 0; r < 16; ++r) acc[i][j][r] = 0.f;

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
    // No memory round trip may sit on the critical path here: scale / shift came with the prologue, the residual rows of
    // BOTH 64-channel passes are requested before the accumulators go through LDS, and nothing ever waits for a store
    // (an s_waitcnt vmcnt(0) in front of a late load also waits for every store issued before it).
    const bool has_res = p.flags & YOLO_FLAG_RESIDUAL;
    const bool nan_chk = p.flags & YOLO_FLAG_NANCHECK;
    constexpr int OLD = 68;
    float* ost = reinterpret_cast<float*>(patch);                     // [128][68] fp32 = 34,816 B
    const bool vec_ok = (p.out_mode != YOLO_OUT_HEAD) && (p.Cout % 8 == 0);
    bool saw_nan = false;
    __syncthreads();                                                  // every wave is done reading the patch
    const int c8 = tid & 7;
    int mrow[4];
    u32x4 rr[TN][4];
    if (vec_ok) {
#pragma unroll
        for (int it = 0; it < 4; ++it) mrow[it] = mtab[(tid >> 3) + 32 * it];
#pragma unroll
        for (int j = 0; j < TN; ++j)
#pragma unroll
            for (int it = 0; it < 4; ++it) {
                const u32x4 z = {0u, 0u, 0u, 0u};
                rr[j][it] = z;
            }
        if (has_res) {
#pragma unroll
            for (int j = 0; j < TN; ++j)
#pragma unroll
                for (int it = 0; it < 4; ++it) {
                    const int n = n_tile * BN + j * 64 + c8 * 8;
                    const int mc = mrow[it] < 0 ? 0 : mrow[it];
                    const int ncl = n < p.Cout ? n : 0;             // clamped: unconditional loads, discarded below
                    rr[j][it] = *reinterpret_cast<const u32x4*>(p.res + (size_t)mc * p.r_ld + p.r_off + ncl);
                }
        }
    }
#pragma unroll
    for (int j = 0; j < TN; ++j) {
        {
            float* dst = ost + wn * 32 + frow;
            YOLO_SWITCH_ACT(p.act,
                _Pragma("unroll") for (int i = 0; i < 2; ++i)
                    _Pragma("unroll") for (int r = 0; r < 16; ++r) {
                        const int row = wm * 64 + i * 32 + (r & 3) + 8 * (r >> 2) + 4 * fh;
                        dst[row * OLD] = act_c<ACT>(acc[i][j][r] * sc[j] + sh[j]);
                    })
        }
        __syncthreads();
        if (vec_ok) {
            unsigned short* yo = reinterpret_cast<unsigned short*>(p.y);
            const int n = n_tile * BN + j * 64 + c8 * 8;
            f32x4 va[4], vb[4];
            if (j == 0 && has_res) {
                // all residual rows (both passes) are awaited HERE, before the first store is issued: a later wait for a
                // pass-2 row would be counted against the stores issued in between (one in-order counter for loads and stores)
#pragma unroll
                for (int jj = 0; jj < TN; ++jj)
#pragma unroll
                    for (int it = 0; it < 4; ++it) asm volatile("" : "+v"(rr[jj][it]));
            }
#pragma unroll
            for (int it = 0; it < 4; ++it) {                           // all LDS reads first: one latency, not four
                const int row = (tid >> 3) + 32 * it;
                va[it] = *reinterpret_cast<const f32x4*>(ost + row * OLD + c8 * 8);
                vb[it] = *reinterpret_cast<const f32x4*>(ost + row * OLD + c8 * 8 + 4);
            }
#pragma unroll
            for (int it = 0; it < 4; ++it) {
                const int m = mrow[it];
                if (m < 0 || n >= p.Cout) continue;
                float v[8] = {va[it][0], va[it][1], va[it][2], va[it][3], vb[it][0], vb[it][1], vb[it][2], vb[it][3]};
                if (has_res) {
#pragma unroll
                    for (int e = 0; e < 4; ++e) {
                        v[2 * e] += HTraits<T>::to_f32((unsigned short)(rr[j][it][e] & 0xffffu));
                        v[2 * e + 1] += HTraits<T>::to_f32((unsigned short)(rr[j][it][e] >> 16));
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
                    const int HoWo = p.Ho * p.Wo;
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
        } else if (p.out_mode == YOLO_OUT_HEAD && !has_res) {   // detection heads: fp32 (B,3,g,g,5+nc), channel = a*(5+nc) + k
            const int col = tid & 63;
            const int n = n_tile * BN + j * 64 + col;
            const int head_a = n / p.nc5, head_k = n - head_a * p.nc5;
            const int HoWo = p.Ho * p.Wo;
            float* yo = reinterpret_cast<float*>(p.y);
            if (n < p.Cout) {
#pragma unroll 8
                for (int it = 0; it < 32; ++it) {
                    const int row = (tid >> 6) + 4 * it;
                    if (mtab[row] < 0) continue;
                    const float v = ost[row * OLD + col];
                    if (nan_chk && v != v) saw_nan = true;
                    yo[(size_t)(mtab[128 + row] + head_a * HoWo) * p.nc5 + head_k] = v;
                }
            }
        } else {                                    // odd channel counts outside the heads (block-level tests)
            const int HoWo = p.Ho * p.Wo;
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

// =====================================================================================================
// conv3_dma_h16 — 3x3 stride-1 blocks with EVERY operand delivered by LDS-DMA (global_load_lds_dwordx4).
//
// Why (per-block stamps of conv_patch_h16, 128->256 @52x52, batch 32): a block's main loop takes ~19.5k cycles whether
// or not the second resident block is computing — 2 x the 9.2k cycles of its matrix work. The weight fragments travel
// L2 -> VGPR with a look-ahead of ~2 K steps (~580 matrix cycles), less than the L2 round trip under load, and a deeper
// REGISTER ring does not fit. Alone on its SIMDs a wave therefore runs at half rate, so the prologue / epilogue of one
// block is never covered by the other. Here the weights stream through a D_SLOTS-deep ring in LDS instead (shared by
// the four waves: half the L2 traffic, D_P K steps = ~1,000 matrix cycles of look-ahead, no staging registers), and the
// activation patch comes the same way, so the loop contains no register-destination load at all: every wait is a
// counted s_waitcnt vmcnt(N) in front of ONE raw s_barrier per K step (cdna_hip_programming.md "Pipelining across
// barriers"; a __syncthreads() would drain the DMA queue).
//   LDS: [2][256 px][64 B] patch (chunk double buffer) | [D_SLOTS][BN/32][2 KiB] weight ring | tile-row tables.
//   * patch rows are 64 B (32 channels) with the 16-byte granules XOR-swizzled by (pixel >> 2) & 3: an LDS-DMA image is
//     lane-linear, so the swizzle is applied to the per-lane SOURCE address and again in the fragment read; 16
//     consecutive pixels then cover all 16 granule slots of the 256-B bank row (conflict-free ds_read_b128);
//   * halo pixels outside the image read a zero page (g_zero_page) instead of being masked;
//   * the weight ring holds the fragment-order stream as it lies in HBM: wave w copies n-tile w of the block, every
//     wave reads its B fragments back lane-linearly.
//   K step t:  [DMA weights t+D_P] [tap 4: DMA patch of the next chunk] [ds_read A/B of step t+1] [8 MFMAs of step t]
//              [s_waitcnt vmcnt: own DMAs of step t+2 landed] [s_barrier].
// =====================================================================================================
constexpr int D_NI = 4;                          // patch DMA rounds of 64 pixels
constexpr int D_PATCH_PIX = 64 * D_NI;           // 256 pixels
constexpr int D_PATCH_BYTES = D_PATCH_PIX * 64;  // 16 KiB per buffer
constexpr int D_P = 4;                           // weight K steps in flight
constexpr int D_SLOTS = D_P + 1;
constexpr int D_PF_TAP = 4;                      // tap at which the next chunk's patch is requested

__device__ __attribute__((aligned(256))) unsigned int g_zero_page[1024 + 16];   // 4 KiB + 64 B of zeros: Cin <= 2048

typedef const __attribute__((address_space(1))) void* gptr_t;
typedef __attribute__((address_space(3))) void* lptr_t;
__device__ __forceinline__ void glds16(const void* g, void* l) {
    __builtin_amdgcn_global_load_lds((gptr_t)g, (lptr_t)l, 16, 0, 0);
}
template <int N> __device__ __forceinline__ void wait_vmcnt() { asm volatile("s_waitcnt vmcnt(%0)" ::"n"(N) : "memory"); }

// SPEED PROBE ONLY (tile 9, results are garbage): the same FLOPs per K step issued as v_mfma_f32_16x16x32 instead of
// 32x32x16 — MI355X_MICROARCH.md "DVFS give-back" (7): where the chip holds its clock down under load, the 16x16x32 shape
// sustained ~1.15x the FLOP/s at equal cycles. Two 16x16x32 (16 cycles each) per 32x32x16 (32 cycles), on 4-register slices.
typedef float f32x4v __attribute__((ext_vector_type(4)));
template <typename T, int S>
__device__ __forceinline__ void probe16(const u32x4& a, const u32x4& b, f32x16& acc) {
    typedef typename HTraits<T>::vec vec;
    f32x4v c0 = {acc[8 * S + 0], acc[8 * S + 1], acc[8 * S + 2], acc[8 * S + 3]};
    f32x4v c1 = {acc[8 * S + 4], acc[8 * S + 5], acc[8 * S + 6], acc[8 * S + 7]};
    if constexpr (sizeof(T) == 2 && __is_same(T, __bf16)) {
        c0 = __builtin_amdgcn_mfma_f32_16x16x32_bf16(__builtin_bit_cast(vec, a), __builtin_bit_cast(vec, b), c0, 0, 0, 0);
        c1 = __builtin_amdgcn_mfma_f32_16x16x32_bf16(__builtin_bit_cast(vec, a), __builtin_bit_cast(vec, b), c1, 0, 0, 0);
    } else {
        c0 = __builtin_amdgcn_mfma_f32_16x16x32_f16(__builtin_bit_cast(vec, a), __builtin_bit_cast(vec, b), c0, 0, 0, 0);
        c1 = __builtin_amdgcn_mfma_f32_16x16x32_f16(__builtin_bit_cast(vec, a), __builtin_bit_cast(vec, b), c1, 0, 0, 0);
    }
#pragma unroll
    for (int e = 0; e < 4; ++e) { acc[8 * S + e] = c0[e]; acc[8 * S + 4 + e] = c1[e]; }
}

template <typename T, int TN>
struct DCtx {
    const unsigned short* wsrc;      // this wave's n-tile of the fragment stream (+ lane * 8)
    const unsigned short* psrc[D_NI];// this lane's source granule of patch round i, chunk 0
    int p0[2];                       // patch pixel (tap 0,0) of this lane's row in m-tile 0 / 1
    int KT, PC;
};

struct DRes {                       // residual rows of the epilogue, requested from inside the last chunk (see d_kstep)
    const unsigned short* rptr[2];  // row of this lane's pixel in m-tile 0 / 1 (+ r_off), null-safe (pixel clamped)
    int ch0;                        // first of this lane's 8 channels (+ j * 64 + kp * 16)
    bool has_res;
};

template <typename T, int BN, int TAP, int PROBE, bool LAST>
__device__ __forceinline__ void d_kstep(const ConvHArgs& p, const DCtx<T, BN / 64>& c, int chunk, char* patch, char* wring,
                                        const unsigned short*& wp, int& slot_w, int& slot_r, u32x4 (&af)[2][2], u32x4 (&bf)[2][BN / 64],
                                        f32x16 (&acc)[2][BN / 64], int wave, int lane, int wn, int fh, const DRes& rs,
                                        u32x4 (&rr)[2][BN / 64][2]) {
    typedef typename HTraits<T>::vec vec;
    constexpr int TN = BN / 64;
    constexpr bool P16 = PROBE & 16;                        // PROBE bits (diagnostic builds only, results are garbage): 1 no s_barrier per
    constexpr bool NOBAR = PROBE & 1, NOWDMA = PROBE & 2;   // step, 2 no weight DMA, 4 no patch fragment reads, 8 no weight fragment reads,
    constexpr bool NOAREAD = PROBE & 4, NOBREAD = PROBE & 8;// 16 the 16x16x32 MFMA shape
    static_assert(TN == 2, "the interleave below is written for 2 x 2 tiles per wave");
    constexpr int SLOT_BYTES = (BN / 32) * 2048;
    // The WEIGHT fragment is the MFMA's A operand and the activation fragment its B operand (the two operand layouts are
    // mirror images, so the same packed streams serve either way): D = [channel][pixel], i.e. a lane owns ONE pixel and 16
    // channels of it in runs of 4 — the layout the register epilogue below stores from without an LDS round trip.
    // A lone wave must keep its matrix pipe fed by itself (the other resident block is in its prologue / epilogue half of the
    // time), so nothing is issued in a burst: the DMA requests and the 8 fragment reads of step t + 1 sit one per MFMA gap
    // (an MFMA occupies the pipe for 32 cycles and the issue port for 8 of them).
#define D_MFMA(i, j, s) \
    if constexpr (P16) probe16<T, s>(bf[s][j], af[i][s], acc[i][j]); \
    else acc[i][j] = HTraits<T>::mfma(__builtin_bit_cast(vec, bf[s][j]), __builtin_bit_cast(vec, af[i][s]), acc[i][j])
    u32x4 an[2][2], bn[2][TN];
    constexpr int NTAP = (TAP + 1) % 9;
    constexpr int nkh = NTAP / 3, nkw = NTAP % 3;
    const int nchunk = TAP == 8 ? chunk + 1 : chunk;
    const char* pb = patch + (nchunk & 1) * D_PATCH_BYTES;
    const char* wb = wring + slot_r * SLOT_BYTES + wn * 2048 + lane * 16;
    __builtin_amdgcn_sched_barrier(0);
    D_MFMA(0, 0, 0);
    constexpr bool last = LAST;                             // the last chunk of a tile is its own instantiation
    constexpr bool fetch = (TAP < 9 - D_P || !last) && !NOWDMA;   // nothing to fetch in the last D_P steps
    if (LAST && TAP == 7 && rs.has_res) {                   // residual rows, first half (see the note at the wait below)
#pragma unroll
        for (int j = 0; j < TN; ++j)
#pragma unroll
            for (int kp = 0; kp < 2; ++kp) {
                const int ch = rs.ch0 + j * 64 + kp * 16;
                rr[0][j][kp] = *reinterpret_cast<const u32x4*>(rs.rptr[0] + (ch < p.Cout ? ch : 0));   // clamped, discarded in the epilogue
            }
    }
    char* wdst = wring + slot_w * SLOT_BYTES + wave * 2048;
    if (fetch) glds16(wp, wdst);                            // (1) weights of step t + D_P -> ring slot slot_w: first KiB here ...
    __builtin_amdgcn_sched_barrier(0);
    D_MFMA(1, 0, 0);
    {   // (2) activation fragments of step t + 1 (landed and made visible by the wait + barrier that closed step t - 1)
        const int px = c.p0[0] + nkh * c.PC + nkw;
        const int a0 = (px << 6) | ((((px >> 2) ^ fh) & 3) << 4);              // granule (s = 0) = fh, swizzled
        if (NOAREAD) { an[0][0] = af[0][0]; an[0][1] = af[0][1]; } else {
        an[0][0] = *reinterpret_cast<const u32x4*>(pb + a0);
        an[0][1] = *reinterpret_cast<const u32x4*>(pb + (a0 ^ 32));            // granule 2 + fh
        }
    }
    __builtin_amdgcn_sched_barrier(0);
    D_MFMA(0, 1, 0);
    {
        const int px = c.p0[1] + nkh * c.PC + nkw;
        const int a0 = (px << 6) | ((((px >> 2) ^ fh) & 3) << 4);
        if (NOAREAD) { an[1][0] = af[1][0]; an[1][1] = af[1][1]; } else {
        an[1][0] = *reinterpret_cast<const u32x4*>(pb + a0);
        an[1][1] = *reinterpret_cast<const u32x4*>(pb + (a0 ^ 32));
        }
    }
    __builtin_amdgcn_sched_barrier(0);
    D_MFMA(1, 1, 0);
    if (NOBREAD) { bn[0][0] = bf[0][0]; bn[1][0] = bf[1][0]; bn[0][1] = bf[0][1]; bn[1][1] = bf[1][1]; }
    if (!NOBREAD) {
    bn[0][0] = *reinterpret_cast<const u32x4*>(wb);                            // (3) weight fragments of step t + 1
    bn[1][0] = *reinterpret_cast<const u32x4*>(wb + 1024);
    }
    __builtin_amdgcn_sched_barrier(0);
    D_MFMA(0, 0, 1);
    if (fetch) {                                            // ... second KiB four MFMAs later (a request costs ~60 cycles of issue, an
        glds16(wp + 512, wdst + 1024);                      //     MFMA covers 32: two in one gap leave the matrix pipe idle)
        wp += 1024;
        slot_w = slot_w + 1 == D_SLOTS ? 0 : slot_w + 1;
    }
    if (!NOBREAD) {
    bn[0][1] = *reinterpret_cast<const u32x4*>(wb + 4096);
    bn[1][1] = *reinterpret_cast<const u32x4*>(wb + 4096 + 1024);
    }
    slot_r = slot_r + 1 == D_SLOTS ? 0 : slot_r + 1;
    __builtin_amdgcn_sched_barrier(0);
    D_MFMA(1, 0, 1);
    if (LAST && TAP == 7 && rs.has_res) {                   // residual rows, second half
#pragma unroll
        for (int j = 0; j < TN; ++j)
#pragma unroll
            for (int kp = 0; kp < 2; ++kp) {
                const int ch = rs.ch0 + j * 64 + kp * 16;
                rr[1][j][kp] = *reinterpret_cast<const u32x4*>(rs.rptr[1] + (ch < p.Cout ? ch : 0));
            }
    }
    if (TAP == D_PF_TAP && !last) {   // (4) patch of the next chunk
        char* dst = patch + ((chunk + 1) & 1) * D_PATCH_BYTES + wave * 1024;
#pragma unroll
        for (int i = 0; i < D_NI; ++i) glds16(c.psrc[i] + (chunk + 1) * 32, dst + i * 4096);
    }
    __builtin_amdgcn_sched_barrier(0);
    D_MFMA(0, 1, 1);
    D_MFMA(1, 1, 1);
#undef D_MFMA
#pragma unroll
    for (int i = 0; i < 2; ++i)
#pragma unroll
        for (int s = 0; s < 2; ++s) af[i][s] = an[i][s];
#pragma unroll
    for (int s = 0; s < 2; ++s)
#pragma unroll
        for (int j = 0; j < TN; ++j) bf[s][j] = bn[s][j];
    __builtin_amdgcn_sched_barrier(0);
    // (5) own DMAs of step t + 2 have landed (2 weight ops per step are younger: steps t - 1 and t; + the patch ops if they
    //     were issued in one of those two steps), then the block-wide rendezvous that makes every wave's pieces visible.
    //     In the last chunk nothing is issued from tap 9 - D_P on (and no patch): the counts shrink with the queue, and
    //     the epilogue finds it empty
    //     From tap 6 of the last chunk the queue is empty: the residual rows of the epilogue are requested in tap 7 (ordinary
    //     loads: with no DMA pending hipcc counts them normally) and have the last two K steps + the epilogue's arithmetic
    //     to arrive; taps 7 and 8 wait for nothing.
    static_assert(D_P == 4 && D_PF_TAP == 4, "wait counts below");
    if (TAP == 4) { if (last) wait_vmcnt<4>(); else wait_vmcnt<4 + D_NI>(); }
    else if (TAP == 5) { if (last) wait_vmcnt<2>(); else wait_vmcnt<4 + D_NI>(); }
    else if (TAP == 6) { if (last) wait_vmcnt<0>(); else wait_vmcnt<4>(); }
    else if (TAP >= 7) { if (!last) wait_vmcnt<4>(); }
    else wait_vmcnt<4>();
    if (!NOBAR) __builtin_amdgcn_s_barrier();
    __builtin_amdgcn_sched_barrier(0);
}

template <typename T, int BN, int TAP, int PROBE, bool LAST>
__device__ __forceinline__ void d_chunk(const ConvHArgs& p, const DCtx<T, BN / 64>& c, int chunk, char* patch, char* wring,
                                        const unsigned short*& wp, int& slot_w, int& slot_r, u32x4 (&af)[2][2], u32x4 (&bf)[2][BN / 64],
                                        f32x16 (&acc)[2][BN / 64], int wave, int lane, int wn, int fh, const DRes& rs,
                                        u32x4 (&rr)[2][BN / 64][2]) {
    if constexpr (TAP < 9) {
        d_kstep<T, BN, TAP, PROBE, LAST>(p, c, chunk, patch, wring, wp, slot_w, slot_r, af, bf, acc, wave, lane, wn, fh, rs, rr);
        d_chunk<T, BN, TAP + 1, PROBE, LAST>(p, c, chunk, patch, wring, wp, slot_w, slot_r, af, bf, acc, wave, lane, wn, fh, rs, rr);
    }
}

// ---- epilogue of conv3_dma_h16 (fp32 math, from registers)
// acc[i][j]: rows = the 32 channels of this wave's n-tile j, columns = the 32 pixels of m-tile i. A lane owns pixel
// (lane & 31) and channels 8g + 4h + {0..3} (g = 0..3, h = lane >> 5). Scale / shift / activation in that layout; then
// one v_permlane32_swap per register pair exchanges halves so that lanes 0-31 hold channels 8k .. 8k+7 and lanes 32-63
// channels 8k+8 .. 8k+15 of their pixel (k = 0, 2): 16 contiguous bytes of output per lane -> ONE 16-byte store (and one
// 16-byte residual row, requested inside the last chunk) per lane, pixel and 16 channels. No LDS round trip, no barrier
// (cdna_hip_programming.md T21). Phases: (A) arithmetic of all four tiles, (B) ALL residual adds, (C) per 16-byte group: NaN guard,
// one rounding, store - nothing that could wait on memory sits between two stores.
template <typename T, int BN, int ACT, bool RES>
__device__ __forceinline__ bool d_epilogue(const ConvHArgs& p, const f32x16 (&acc)[2][BN / 64], const u32x4 (&rr)[2][BN / 64][2],
                                           const float* sstab, const int (&mpix)[2], const size_t (&ooff)[2], int ch0, int wn, int fh) {
    constexpr int TN = BN / 64;
    float w[2][TN][2][8];           // w[i][j][kp][0..7] = this lane's 8 consecutive output channels (ch0 + j*64 + kp*16 ...) of pixel mpix[i]
#pragma unroll
    for (int j = 0; j < TN; ++j) {
        // folded BatchNorm scale / shift of channels 8g + 4h + {0..3}: broadcast reads of the table the prologue staged
        f32x4 sc4[4], sh4[4];
#pragma unroll
        for (int g = 0; g < 4; ++g) {
            sc4[g] = *reinterpret_cast<const f32x4*>(sstab + j * 64 + wn * 32 + 8 * g + 4 * fh);
            sh4[g] = *reinterpret_cast<const f32x4*>(sstab + BN + j * 64 + wn * 32 + 8 * g + 4 * fh);
        }
#pragma unroll
        for (int i = 0; i < 2; ++i) {
            float v[16];
#pragma unroll
            for (int r = 0; r < 16; ++r) v[r] = act_c<ACT>(acc[i][j][r] * sc4[r >> 2][r & 3] + sh4[r >> 2][r & 3]);
            // half exchange on the fp32 values (one rounding, after the residual add): group pairs (0,1) and (2,3)
#pragma unroll
            for (int kp = 0; kp < 2; ++kp)
#pragma unroll
                for (int e = 0; e < 4; ++e) {
                    const auto sw = __builtin_amdgcn_permlane32_swap(__float_as_uint(v[8 * kp + e]), __float_as_uint(v[8 * kp + 4 + e]), false, false);
                    w[i][j][kp][e] = __uint_as_float(sw[0]);          // lanes 0-31: own group 2kp | lanes 32-63: lower half's group 2kp+1
                    w[i][j][kp][4 + e] = __uint_as_float(sw[1]);      // lanes 0-31: upper half's group 2kp | lanes 32-63: own group 2kp+1
                }
            __builtin_amdgcn_sched_barrier(0);
        }
    }
    __builtin_amdgcn_sched_barrier(0);      // phases stay phases: overlapped by the scheduler they were all live at once (250 VGPRs)
    if (RES) {
#pragma unroll
        for (int i = 0; i < 2; ++i)
#pragma unroll
            for (int j = 0; j < TN; ++j)
#pragma unroll
                for (int kp = 0; kp < 2; ++kp)
#pragma unroll
                    for (int e = 0; e < 4; ++e) {
                        w[i][j][kp][2 * e] += HTraits<T>::to_f32((unsigned short)(rr[i][j][kp][e] & 0xffffu));
                        w[i][j][kp][2 * e + 1] += HTraits<T>::to_f32((unsigned short)(rr[i][j][kp][e] >> 16));
                    }
        __builtin_amdgcn_sched_barrier(0);
    }
    bool saw_nan = false;
    unsigned short* yo = reinterpret_cast<unsigned short*>(p.y);
#pragma unroll
    for (int i = 0; i < 2; ++i)
#pragma unroll
        for (int j = 0; j < TN; ++j)
#pragma unroll
            for (int kp = 0; kp < 2; ++kp) {
                u32x4 o;
#pragma unroll
                for (int e = 0; e < 4; ++e) {
                    const float x0 = w[i][j][kp][2 * e], x1 = w[i][j][kp][2 * e + 1];
                    saw_nan |= __builtin_isunordered(x0, x1);          // one v_cmp_u_f32 per pair
                    o[e] = pack2<T>(x0, x1);
                }
                if (mpix[i] < 0 || ch0 + j * 64 + kp * 16 >= p.Cout) continue;
                unsigned short* d = yo + ooff[i] + j * 64 + kp * 16;
                *reinterpret_cast<u32x4*>(d) = o;
                if (p.out_mode != YOLO_OUT_NHWC) {
                    const size_t W2 = 2 * (size_t)p.Wo;
                    *reinterpret_cast<u32x4*>(d + p.y_ld) = o;
                    *reinterpret_cast<u32x4*>(d + W2 * p.y_ld) = o;
                    *reinterpret_cast<u32x4*>(d + (W2 + 1) * p.y_ld) = o;
                }
            }
    return saw_nan;
}


// ---- epilogue of the train-mode forward: raw convolution output z (no scale / shift / activation / residual) AND the
// BatchNorm partial sums of this wave's 64 pixels x 64 channels, so that the statistics pass over z (one full read of every
// conv output: 0.7 ms of the 17 ms bf16 step) disappears. Sums are taken of the ROUNDED values, i.e. of exactly what is
// stored and normalised later (the reference's batch_norm sees the 16-bit conv output too).
// After the half exchange of d_epilogue a lane holds 8 consecutive channels of ONE pixel per (n-tile j, channel pair kp) and
// m-tile i: 2 (sum, sum of squares) x 2 x 2 x 8 = 64 per-lane values, each to be added over the 32 pixels (lanes) of its half.
// A reduce-scatter butterfly does that in 31 + 31 adds instead of 64 x 5: every level pairs two registers and two lane groups,
// each group keeps one register of the pair and receives the partner group's copy of it:
//   level 16: v_permlane16_swap (odd rows of X <-> even rows of Y), pair = (sum, sum of squares)  -> bit 4 of the lane = quantity
//   level  8: DPP row_mirror (l <-> 15 - l),       pair = n-tile 0 / 1                           -> bit 3 = j
//   level  4: DPP row_half_mirror (l <-> 7 - l),   pair = channel pair kp 0 / 1                  -> bit 2 = kp
//   level  2: DPP quad_perm [2,3,0,1],             pair = channels e / e + 4                      -> bit 1
//   level  1: DPP quad_perm [1,0,3,2],             pair = channels e / e + 2                      -> bit 0
// leaving two values (channels c, c + 1) per lane: one 8-byte store into stats[row][quantity][channel].
template <int CTRL> __device__ __forceinline__ float dpp_f(float v) {
    return __int_as_float(__builtin_amdgcn_update_dpp(0, __float_as_int(v), CTRL, 0xf, 0xf, true));
}
template <int BN>
__device__ __forceinline__ void stats_reduce_store(const ConvHArgs& p, const float (&sq)[2][BN / 64][2][8], int ch0, int lane, int row) {
    constexpr int TN = BN / 64;
    // level 16
    float l8[TN][2][8];
#pragma unroll
    for (int j = 0; j < TN; ++j)
#pragma unroll
        for (int kp = 0; kp < 2; ++kp)
#pragma unroll
            for (int e = 0; e < 8; ++e) {
                const auto sw = __builtin_amdgcn_permlane16_swap(__float_as_uint(sq[0][j][kp][e]), __float_as_uint(sq[1][j][kp][e]), false, false);
                l8[j][kp][e] = __uint_as_float(sw[0]) + __uint_as_float(sw[1]);
            }
    const bool b3 = lane & 8, b2 = lane & 4, b1 = lane & 2, b0 = lane & 1;
    float l4[2][8];
#pragma unroll
    for (int kp = 0; kp < 2; ++kp)
#pragma unroll
        for (int e = 0; e < 8; ++e) {
            const float t0 = l8[0][kp][e] + dpp_f<0x140>(l8[0][kp][e]);       // row_mirror
            const float t1 = l8[1][kp][e] + dpp_f<0x140>(l8[1][kp][e]);
            l4[kp][e] = b3 ? t1 : t0;
        }
    float l2[8];
#pragma unroll
    for (int e = 0; e < 8; ++e) {
        const float t0 = l4[0][e] + dpp_f<0x141>(l4[0][e]);                   // row_half_mirror
        const float t1 = l4[1][e] + dpp_f<0x141>(l4[1][e]);
        l2[e] = b2 ? t1 : t0;
    }
    float l1[4];
#pragma unroll
    for (int e = 0; e < 4; ++e) {
        const float t0 = l2[e] + dpp_f<0x4E>(l2[e]);                          // quad_perm [2,3,0,1]
        const float t1 = l2[e + 4] + dpp_f<0x4E>(l2[e + 4]);
        l1[e] = b1 ? t1 : t0;
    }
    float l0[2];
#pragma unroll
    for (int e = 0; e < 2; ++e) {
        const float t0 = l1[e] + dpp_f<0xB1>(l1[e]);                          // quad_perm [1,0,3,2]
        const float t1 = l1[e + 2] + dpp_f<0xB1>(l1[e + 2]);
        l0[e] = b0 ? t1 : t0;
    }
    const int qty = (lane >> 4) & 1;
    const int ch = ch0 + (b3 ? 64 : 0) + (b2 ? 16 : 0) + (b1 ? 4 : 0) + (b0 ? 2 : 0);
    typedef float f32x2 __attribute__((ext_vector_type(2)));
    const f32x2 out = {l0[0], l0[1]};
    *reinterpret_cast<f32x2*>(p.stats + ((size_t)row * 2 + qty) * p.stats_ld + ch) = out;      // stats_ld covers the padded channel tiles
}

template <typename T, int BN>
__device__ __forceinline__ void d_epilogue_stats(const ConvHArgs& p, const f32x16 (&acc)[2][BN / 64], const int (&mpix)[2],
                                                 const size_t (&ooff)[2], int ch0, int lane, int row) {
    constexpr int TN = BN / 64;
    static_assert(TN == 2, "two n-tiles per wave");
    float sq[2][TN][2][8];                                  // [quantity][j][kp][e]
#pragma unroll
    for (int a = 0; a < 2; ++a)
#pragma unroll
        for (int j = 0; j < TN; ++j)
#pragma unroll
            for (int kp = 0; kp < 2; ++kp)
#pragma unroll
                for (int e = 0; e < 8; ++e) sq[a][j][kp][e] = 0.f;
    unsigned short* yo = reinterpret_cast<unsigned short*>(p.y);
#pragma unroll
    for (int i = 0; i < 2; ++i) {
        const float live = mpix[i] < 0 ? 0.f : 1.f;        // tile padding: the lane computed a duplicate of pixel 0, counts for nothing
#pragma unroll
        for (int j = 0; j < TN; ++j) {
#pragma unroll
            for (int kp = 0; kp < 2; ++kp) {
                float w[8];
#pragma unroll
                for (int e = 0; e < 4; ++e) {
                    const auto sw = __builtin_amdgcn_permlane32_swap(__float_as_uint(acc[i][j][8 * kp + e]), __float_as_uint(acc[i][j][8 * kp + 4 + e]), false, false);
                    w[e] = __uint_as_float(sw[0]);
                    w[4 + e] = __uint_as_float(sw[1]);
                }
                u32x4 o;
#pragma unroll
                for (int e = 0; e < 4; ++e) {
                    o[e] = pack2<T>(w[2 * e], w[2 * e + 1]);
                    const float r0 = HTraits<T>::to_f32((unsigned short)(o[e] & 0xffffu)) * live;
                    const float r1 = HTraits<T>::to_f32((unsigned short)(o[e] >> 16)) * live;
                    sq[0][j][kp][2 * e] += r0;
                    sq[0][j][kp][2 * e + 1] += r1;
                    sq[1][j][kp][2 * e] = __builtin_fmaf(r0, r0, sq[1][j][kp][2 * e]);
                    sq[1][j][kp][2 * e + 1] = __builtin_fmaf(r1, r1, sq[1][j][kp][2 * e + 1]);
                }
                if (mpix[i] >= 0 && ch0 + j * 64 + kp * 16 < p.Cout) *reinterpret_cast<u32x4*>(yo + ooff[i] + j * 64 + kp * 16) = o;
            }
        }
        __builtin_amdgcn_sched_barrier(0);
    }
    stats_reduce_store<BN>(p, sq, ch0, lane, row);
}

// ---- epilogue of an input-gradient launch that ALSO takes the BatchNorm-backward sums of the block that produced this
// convolution's input (the block whose output gradient dx is): identity epilogue [+ the running gradient], rounded once, and
// of exactly those rounded values  sum(du)  and  sum(du * (z - mean))  with  du = dx * act'((z - mean) * scale + shift)  -
// the formula (and the fp32 operation order) of bn_bwd_partial, whose pass over dx and z this replaces. z is read here
// once (a 16-byte row per lane, pixel and 8 channels, like the residual). Same per-wave rows as d_epilogue_stats.
template <typename T, int BN, int ACT>
__device__ __forceinline__ void d_epilogue_bstats(const ConvHArgs& p, const f32x16 (&acc)[2][BN / 64], const u32x4 (&rr)[2][BN / 64][2],
                                                  bool has_res, const int (&mpix)[2], const size_t (&ooff)[2], int ch0, int lane, int row) {
    constexpr int TN = BN / 64;
    unsigned short* yo = reinterpret_cast<unsigned short*>(p.y);
    size_t zoff[2];
#pragma unroll
    for (int i = 0; i < 2; ++i) zoff[i] = (size_t)(mpix[i] < 0 ? 0 : mpix[i]) * p.bz_ld + p.bz_off;
    const bool b3 = lane & 8, b2 = lane & 4, b1 = lane & 2;
    float* srow = p.stats + ((size_t)row * 2 + ((lane >> 4) & 1)) * p.stats_ld + (b3 ? 4 : 0) + (b2 ? 2 : 0) + (b1 ? 1 : 0);
#pragma unroll
    for (int j = 0; j < TN; ++j) {
#pragma unroll
        for (int kp = 0; kp < 2; ++kp) {
            const int cb = ch0 + j * 64 + kp * 16;
            const bool chan_ok = cb < p.Cout;
            const int cbs = chan_ok ? cb : 0;
            f32x4 mu[2], sc[2], sh[2];
#pragma unroll
            for (int h = 0; h < 2; ++h) {
                mu[h] = *reinterpret_cast<const f32x4*>(p.bmean + cbs + 4 * h);
                sc[h] = *reinterpret_cast<const f32x4*>(p.bscale + cbs + 4 * h);
                sh[h] = *reinterpret_cast<const f32x4*>(p.bshift + cbs + 4 * h);
            }
            float sq[2][8];                                 // this group's 8 channels: sum(du), sum(du * (z - mean)) over the lane's two pixels
#pragma unroll
            for (int i = 0; i < 2; ++i) {
                const u32x4 zv = *reinterpret_cast<const u32x4*>(p.bz + zoff[i] + cbs);
                float w[8];
#pragma unroll
                for (int e = 0; e < 4; ++e) {
                    const auto sw = __builtin_amdgcn_permlane32_swap(__float_as_uint(acc[i][j][8 * kp + e]), __float_as_uint(acc[i][j][8 * kp + 4 + e]), false, false);
                    w[e] = __uint_as_float(sw[0]);
                    w[4 + e] = __uint_as_float(sw[1]);
                }
                if (has_res) {
#pragma unroll
                    for (int e = 0; e < 4; ++e) {
                        w[2 * e] += HTraits<T>::to_f32((unsigned short)(rr[i][j][kp][e] & 0xffffu));
                        w[2 * e + 1] += HTraits<T>::to_f32((unsigned short)(rr[i][j][kp][e] >> 16));
                    }
                }
                const float live = (mpix[i] >= 0 && chan_ok) ? 1.f : 0.f;
                u32x4 o;
#pragma unroll
                for (int e = 0; e < 4; ++e) {
                    o[e] = pack2<T>(w[2 * e], w[2 * e + 1]);
#pragma unroll
                    for (int h = 0; h < 2; ++h) {
                        const int c = 2 * e + h;
                        const float r = HTraits<T>::to_f32((unsigned short)(h ? o[e] >> 16 : o[e] & 0xffffu));
                        const float zc = HTraits<T>::to_f32((unsigned short)(h ? zv[e] >> 16 : zv[e] & 0xffffu)) - mu[c >> 2][c & 3];
                        const float du = r * act_grad_c<ACT>(zc * sc[c >> 2][c & 3] + sh[c >> 2][c & 3]) * live;
                        sq[0][c] = i == 0 ? du : sq[0][c] + du;
                        sq[1][c] = i == 0 ? du * zc : __builtin_fmaf(du, zc, sq[1][c]);
                    }
                }
                if (mpix[i] >= 0 && chan_ok) *reinterpret_cast<u32x4*>(yo + ooff[i] + j * 64 + kp * 16) = o;
            }
            // the 32 pixels of this half, per group (16 live values instead of 64 for all four groups at once - the kernel must
            // stay under 256 VGPRs): reduce-scatter as in stats_reduce_store, pairing (quantity), (c, c+4), (c, c+2), (c, c+1)
            float l8[8];
#pragma unroll
            for (int e = 0; e < 8; ++e) {
                const auto sw = __builtin_amdgcn_permlane16_swap(__float_as_uint(sq[0][e]), __float_as_uint(sq[1][e]), false, false);
                l8[e] = __uint_as_float(sw[0]) + __uint_as_float(sw[1]);
            }
            float l4[4];
#pragma unroll
            for (int e = 0; e < 4; ++e) {
                const float t0 = l8[e] + dpp_f<0x140>(l8[e]);
                const float t1 = l8[e + 4] + dpp_f<0x140>(l8[e + 4]);
                l4[e] = b3 ? t1 : t0;
            }
            float l2[2];
#pragma unroll
            for (int e = 0; e < 2; ++e) {
                const float t0 = l4[e] + dpp_f<0x141>(l4[e]);
                const float t1 = l4[e + 2] + dpp_f<0x141>(l4[e + 2]);
                l2[e] = b2 ? t1 : t0;
            }
            const float t0 = l2[0] + dpp_f<0x4E>(l2[0]);
            const float t1 = l2[1] + dpp_f<0x4E>(l2[1]);
            float l1 = b1 ? t1 : t0;
            l1 += dpp_f<0xB1>(l1);
            if (!(lane & 1)) srow[cb] = l1;                // stats_ld covers the padded channel tiles
            __builtin_amdgcn_sched_barrier(0);
        }
    }
}

// ---- epilogue of the FUSED stride-2 input gradient (conv1_dma_h16<GATH = 2>): the GEMM's output channel n = class * C + c
// (class = (ph, pw) parity of the dx pixel inside the 2 x 2 block of dz pixel m, C = p.H channels of dx), identity epilogue,
// optional accumulate into what is already there (p.res / r_ld / r_off address the same pixels of the running gradient).
// Each lane holds 8 consecutive n per (j, kp): one class, 8 consecutive channels -> a 16-byte store at pixel (2 row + ph, 2 col + pw).
template <typename T, int BN>
__device__ __forceinline__ void d_epilogue_s2g(const ConvHArgs& p, const f32x16 (&acc)[2][BN / 64], const int (&mpix)[2], int ch0) {
    constexpr int TN = BN / 64;
    const bool has_res = p.flags & YOLO_FLAG_RESIDUAL;
    unsigned short* yo = reinterpret_cast<unsigned short*>(p.y);
    const int W2 = 2 * p.Win;
#pragma unroll
    for (int i = 0; i < 2; ++i) {
        const int m = mpix[i] < 0 ? 0 : mpix[i];
        const int img = fdiv(m, p.mg_PC, p.PC), rem = m - img * p.PC;
        const int row = fdiv(rem, p.mg_TW, p.TW), col = rem - row * p.TW;
        const size_t blk = (size_t)(img * 2 * p.Hin + 2 * row) * W2 + 2 * col;          // dx pixel (2 row, 2 col)
#pragma unroll
        for (int j = 0; j < TN; ++j)
#pragma unroll
            for (int kp = 0; kp < 2; ++kp) {
                float w[8];
#pragma unroll
                for (int e = 0; e < 4; ++e) {
                    const auto sw = __builtin_amdgcn_permlane32_swap(__float_as_uint(acc[i][j][8 * kp + e]), __float_as_uint(acc[i][j][8 * kp + 4 + e]), false, false);
                    w[e] = __uint_as_float(sw[0]);
                    w[4 + e] = __uint_as_float(sw[1]);
                }
                const int nb = ch0 + j * 64 + kp * 16;
                const int cls = fdiv(nb, p.mg_H, p.H), c = nb - cls * p.H;
                const size_t pix = blk + (size_t)(cls >> 1) * W2 + (cls & 1);
                if (mpix[i] < 0 || nb >= p.Cout) continue;
                if (has_res) {
                    const u32x4 r4 = *reinterpret_cast<const u32x4*>(p.res + pix * p.r_ld + p.r_off + c);
#pragma unroll
                    for (int e = 0; e < 4; ++e) {
                        w[2 * e] += HTraits<T>::to_f32((unsigned short)(r4[e] & 0xffffu));
                        w[2 * e + 1] += HTraits<T>::to_f32((unsigned short)(r4[e] >> 16));
                    }
                }
                u32x4 o;
#pragma unroll
                for (int e = 0; e < 4; ++e) o[e] = pack2<T>(w[2 * e], w[2 * e + 1]);
                *reinterpret_cast<u32x4*>(yo + pix * p.y_ld + p.y_off + c) = o;
            }
    }
}

template <typename T, int BN, int PROBE = 0>
__global__ __launch_bounds__(256) void conv3_dma_h16(const ConvHArgs p) {
    constexpr int TN = BN / 64;
    constexpr int SLOT_BYTES = (BN / 32) * 2048;
    static_assert(BN / 32 == 4, "one weight n-tile per wave");
    extern __shared__ __attribute__((aligned(16))) char smem_raw[];
    char* patch = smem_raw;                                             // [2][256 px][64 B]
    char* wring = smem_raw + 2 * D_PATCH_BYTES;                         // [D_SLOTS][BN/32][2 KiB]
    int* mtab = reinterpret_cast<int*>(wring + D_SLOTS * SLOT_BYTES);   // [128] output pixel of tile row, [128] head-layout base
    float* sstab = reinterpret_cast<float*>(mtab + 256);                // [BN] scale, [BN] shift

    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);         // wave id on the scalar unit: the DMA destinations need no VALU
    const int wm = wave >> 1, wn = wave & 1;
    const int fh = lane >> 5, frow = lane & 31;
#ifdef H16_STAMPS
    const unsigned long long st0 = __builtin_amdgcn_s_memtime();
#endif

    if (p.stagger > 0 && (int)blockIdx.x < p.first_wave) {             // see conv_f32_v2.hip
        unsigned hw;
        asm volatile("s_getreg_b32 %0, hwreg(HW_REG_HW_ID)" : "=s"(hw));
        const int slot = (hw >> 16) & 15;
        for (int i = 0; i < slot * p.stagger; ++i) __builtin_amdgcn_s_sleep(32);
    }
#ifdef H16_STAMPS
    const unsigned long long st0b = __builtin_amdgcn_s_memtime();
#endif
    int bid = blockIdx.x;
    {
        const int nb = p.nblocks, q = nb / 8, r = nb % 8, xcd = bid % 8;
        bid = (xcd < r ? xcd * (q + 1) : r * (q + 1) + (xcd - r) * q) + bid / 8;
    }
    const int sp = fdiv(bid, p.mg_tn, p.tiles_n);
    const int n_tile = bid - sp * p.tiles_n;

    // The other resident block is usually in its main loop: its waves need the issue port for 8 of every 32 cycles (one
    // MFMA), this wave's prologue / epilogue needs it all the time. Priority, then age, arbitrates the port between the two
    // waves of a SIMD (MI355X_MICROARCH.md): take it while there is no matrix work here, give it back for the loop.
    if (p.prio) __builtin_amdgcn_s_setprio(2);
    DCtx<T, TN> c;
    c.KT = p.KT;
    c.PC = p.PC;
    c.wsrc = p.wf + (size_t)(n_tile * (BN / 32) + wave) * p.KT * 1024 + lane * 8;
    // ---- prologue: weight steps 0 and 1 leave at once (they need nothing but n_tile); steps 2 .. D_P-1 follow the patch,
    //      so that the first wait can leave them in flight (one in-order counter)
    auto issue_w = [&](int q) {
        const int kq = q < p.KT ? q : p.KT - 1;
        const unsigned short* src = c.wsrc + (size_t)kq * 1024;
        char* dst = wring + q * SLOT_BYTES + wave * 2048;
        glds16(src, dst);
        glds16(src + 512, dst + 1024);
    };
    issue_w(0);
    issue_w(1);
    // folded BatchNorm scale / shift of the block's BN channels: by LDS-DMA too (4 bytes per lane; waves 0-1 scale, 2-3 shift).
    // An ordinary load here would be awaited with vmcnt(0) — hipcc does not count a register load apart from pending DMAs
    {
        const int n = n_tile * BN + (wave & 1) * 64 + lane;
        const int ncl = n < p.Cout ? n : p.Cout - 1;
        __builtin_amdgcn_global_load_lds((gptr_t)((wave < 2 ? p.scale : p.shift) + ncl), (lptr_t)(sstab + wave * 64), 4, 0, 0);
    }
    // patch geometry (as conv_patch_h16, KS = 3, stride 1)
    const int r_tile = fdiv(sp, p.mg_tw, p.tiles_w);
    const int w_tile = sp - r_tile * p.tiles_w;
    const int g0 = r_tile * p.TH, c0 = w_tile * p.TW;
    const int g_last = (g0 + p.TH < p.rows_total ? g0 + p.TH : p.rows_total) - 1;
    const int Hp = p.Hin + 2;
    auto vrow = [&](int g) {
        const int n = fdiv(g, p.mg_H, p.H);
        return n * Hp + (g - n * p.H);
    };
    const int v0 = vrow(g0);
    const int PR = vrow(g_last) + 3 - v0;
    {
        const int gs = (tid & 3) ^ ((tid >> 4) & 3);           // source granule of LDS granule (pixel (tid>>2) + 64 i, slot tid & 3)
        const unsigned short* zp = reinterpret_cast<const unsigned short*>(g_zero_page) + gs * 8;
#pragma unroll
        for (int i = 0; i < D_NI; ++i) {
            const int idx = (tid >> 2) + 64 * i;
            const int pr = fdiv(idx, p.mg_PC, p.PC), pc = idx - pr * p.PC;
            const int vv = v0 + pr;
            const int n = fdiv(vv, p.mg_Hp, Hp), yy = vv - n * Hp;
            const int hi = yy - 1, wi = c0 + pc - 1;
            const bool ok = (pr < PR) & ((unsigned)hi < (unsigned)p.Hin) & ((unsigned)wi < (unsigned)p.Win);
            const int pix = (n * p.Hin + hi) * p.Win + wi;
            c.psrc[i] = ok ? p.x + (size_t)pix * p.x_ld + p.x_off + gs * 8 : zp;
        }
        char* dst = patch + wave * 1024;
#pragma unroll
        for (int i = 0; i < D_NI; ++i) glds16(c.psrc[i], dst + i * 4096);
    }
#pragma unroll
    for (int q = 2; q < D_P; ++q) issue_w(q);
    if constexpr ((PROBE & 96) != 0) {   // NEGATIVE ablation: 1k (bit 32) / 2k (bit 64) cycles of extra quarter-rate VALU work per wave
        int xx = tid | 1;                // and tile, while the first operands travel: is a block's VALU time hidden or additive?
#pragma unroll
        for (int k = 0; k < ((PROBE & 64) ? 128 : 64); ++k) asm volatile("v_mul_lo_u32 %0, %0, %1" : "+v"(xx) : "v"(tid | 3));
    }
    // while those travel: fragment rows and the tile-row -> output-pixel tables of the epilogue
#pragma unroll
    for (int i = 0; i < 2; ++i) {
        const int pp = wm * 64 + i * 32 + ((((p.qperm >> ((frow >> 2) * 4)) & 7) << 2) | (frow & 3));
        const int r = fdiv(pp, p.mg_TW, p.TW), cc = pp - r * p.TW;
        const int g = g0 + r;
        const bool ok = (pp < p.TH * p.TW) & (g <= g_last) & (c0 + cc < p.W);
        c.p0[i] = ok ? (vrow(g) - v0) * p.PC + cc : 0;
    }
    if (tid < 128) {
        const int pp = (tid & ~31) | (((p.qperm >> (((tid & 31) >> 2) * 4)) & 7) << 2) | (tid & 3);
        const int r = fdiv(pp, p.mg_TW, p.TW), cc = pp - r * p.TW;
        const int g = g0 + r;
        int m = -1, mh = 0;
        if (pp < p.TH * p.TW && g <= g_last && c0 + cc < p.W) {
            m = g * p.W + c0 + cc;
            if (p.out_mode == YOLO_OUT_HEAD) mh = m + 2 * (m / (p.Ho * p.Wo)) * (p.Ho * p.Wo);
        }
        mtab[tid] = m;
        mtab[128 + tid] = mh;
    }
    // weights of steps 0 and 1 and the patch of chunk 0 must have landed; steps 2 .. D_P-1 (the 2 (D_P - 2) youngest ops)
    // stay in flight — the same count the loop keeps
    wait_vmcnt<2 * (D_P - 2)>();
    __builtin_amdgcn_s_barrier();
    __builtin_amdgcn_sched_barrier(0);
    u32x4 af[2][2], bf[2][TN];
#pragma unroll
    for (int i = 0; i < 2; ++i) {
        const int px = c.p0[i];
        const int a0 = (px << 6) | ((((px >> 2) ^ fh) & 3) << 4);
        af[i][0] = *reinterpret_cast<const u32x4*>(patch + a0);
        af[i][1] = *reinterpret_cast<const u32x4*>(patch + (a0 ^ 32));
    }
#pragma unroll
    for (int j = 0; j < TN; ++j) {
        bf[0][j] = *reinterpret_cast<const u32x4*>(wring + wn * 2048 + lane * 16 + j * 4096);
        bf[1][j] = *reinterpret_cast<const u32x4*>(wring + wn * 2048 + lane * 16 + j * 4096 + 1024);
    }
    f32x16 acc[2][TN];
#pragma unroll
    for (int i = 0; i < 2; ++i)
#pragma unroll
        for (int j = 0; j < TN; ++j)
#pragma unroll
            for (int r = 0; r < 16; ++r) acc[i][j][r] = 0.f;

#ifdef H16_STAMPS
    asm volatile("s_nop 0" ::: "memory");
    const unsigned long long st1 = __builtin_amdgcn_s_memtime();
#endif
    int slot_w = D_P % D_SLOTS, slot_r = 1;
    const unsigned short* wp = c.wsrc + (size_t)D_P * 1024;         // weights of step D_P: advanced by one step per request
    __builtin_amdgcn_s_setprio(0);
    // what the residual requests inside the last chunk need: this lane's output pixels and first channel
    const bool has_res = p.flags & YOLO_FLAG_RESIDUAL;
    int mpix[2];
#pragma unroll
    for (int i = 0; i < 2; ++i) mpix[i] = mtab[wm * 64 + i * 32 + frow];
    DRes rs;
    rs.ch0 = n_tile * BN + wn * 32 + 8 * fh;
    rs.has_res = has_res;
    u32x4 rr[2][TN][2];
#pragma unroll
    for (int i = 0; i < 2; ++i) {
        rs.rptr[i] = p.res + (size_t)(mpix[i] < 0 ? 0 : mpix[i]) * p.r_ld + p.r_off;
#pragma unroll
        for (int j = 0; j < TN; ++j)
#pragma unroll
            for (int kp = 0; kp < 2; ++kp) {
                const u32x4 z = {0u, 0u, 0u, 0u};
                rr[i][j][kp] = z;
            }
    }
    for (int chunk = 0; chunk + 1 < p.nchunks; ++chunk)
        d_chunk<T, BN, 0, PROBE, false>(p, c, chunk, patch, wring, wp, slot_w, slot_r, af, bf, acc, wave, lane, wn, fh, rs, rr);
    d_chunk<T, BN, 0, PROBE, true>(p, c, p.nchunks - 1, patch, wring, wp, slot_w, slot_r, af, bf, acc, wave, lane, wn, fh, rs, rr);
    if (p.prio) __builtin_amdgcn_s_setprio(2);
#ifdef H16_STAMPS
    asm volatile("s_nop 0" ::: "memory");
    const unsigned long long st2 = __builtin_amdgcn_s_memtime();
#endif

    // ---------------------------------------------------------------------- epilogue (fp32 math, from registers)
    // acc[i][j]: rows = the 32 channels of this wave's n-tile j, columns = the 32 pixels of m-tile i. A lane owns pixel
    // (lane & 31) and channels 8g + 4h + {0..3} (g = 0..3, h = lane >> 5). Scale / shift / activation in that layout; then
    // one v_permlane32_swap per register pair exchanges halves so that lanes 0-31 hold channels 8k .. 8k+7 and lanes 32-63
    // channels 8k+8 .. 8k+15 of their pixel (k = 0, 2): 16 contiguous bytes of output per lane -> ONE 16-byte store (and one
    // 16-byte residual load) per lane, pixel and 16 channels. No LDS round trip, no barrier (cdna_hip_programming.md T21).
    const bool nan_chk = p.flags & YOLO_FLAG_NANCHECK;
    const int ch0 = rs.ch0;                                           // + j * 64 + 16 * kp: first of this lane's 8 output channels
    size_t ooff[2];
#pragma unroll
    for (int i = 0; i < 2; ++i) {
        const int m = mpix[i] < 0 ? 0 : mpix[i];
        if (p.out_mode == YOLO_OUT_NHWC) {
            ooff[i] = (size_t)m * p.y_ld + p.y_off + ch0;
        } else {                                                      // 2x nearest upsample into the concat buffer
            const int HoWo = p.Ho * p.Wo;
            const int img = m / HoWo;
            const int rem = m - img * HoWo;
            const int ho = rem / p.Wo;
            const int wo2 = rem - ho * p.Wo;
            ooff[i] = ((size_t)(img * 2 * p.Ho + 2 * ho) * (2 * p.Wo) + 2 * wo2) * p.y_ld + p.y_off + ch0;
        }
    }
    // One straight-line instance per (activation, residual): chosen by ONE wave-uniform switch here. With the switch inside the
    // tile loops hipcc merged the variants through ~190 v_mov and a branch per tile, and with the residual add between the
    // stores every add waited for the stores before it (s_waitcnt vmcnt(0): one in-order counter, and the rows were requested
    // in another basic block) — ~500 cycles per store group on the 23 residual layers.
    bool saw_nan = false;
    if (p.stats != nullptr) {                                         // train-mode forward: raw z + BatchNorm partial sums
        if (p.bz == nullptr) d_epilogue_stats<T, BN>(p, acc, mpix, ooff, ch0, lane, sp * 2 + wm);
        else if (p.bact == YOLO_ACT_LEAKY) d_epilogue_bstats<T, BN, YOLO_ACT_LEAKY>(p, acc, rr, has_res, mpix, ooff, ch0, lane, sp * 2 + wm);
        else d_epilogue_bstats<T, BN, YOLO_ACT_MISH>(p, acc, rr, has_res, mpix, ooff, ch0, lane, sp * 2 + wm);
    } else {
    YOLO_SWITCH_ACT(p.act, saw_nan = has_res ? (d_epilogue<T, BN, ACT, true>(p, acc, rr, sstab, mpix, ooff, ch0, wn, fh))
                                             : (d_epilogue<T, BN, ACT, false>(p, acc, rr, sstab, mpix, ooff, ch0, wn, fh)));
    }
    if (nan_chk && saw_nan) atomicOr(p.nan_flag, 2);
#ifdef H16_STAMPS
    {
        const unsigned long long st3 = __builtin_amdgcn_s_memtime();      // stores issued, not awaited
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        const unsigned long long st4 = __builtin_amdgcn_s_memtime();
        if (tid == 0) {
            unsigned hw, xcc;
            asm volatile("s_getreg_b32 %0, hwreg(HW_REG_HW_ID)" : "=s"(hw));
            asm volatile("s_getreg_b32 %0, hwreg(HW_REG_XCC_ID)" : "=s"(xcc));
            unsigned long long* o = reinterpret_cast<unsigned long long*>(p.nan_flag) + (size_t)blockIdx.x * 6;
            o[0] = st0b; o[1] = st1; o[2] = st2; o[3] = st3; o[4] = hw; o[5] = xcc | ((st4 - st3) << 8) | ((st0b - st0) << 36);
        }
    }
#endif
}

// =====================================================================================================
// conv1_dma_h16 — 1x1 layers with >= 128 output channels on the machinery of conv3_dma_h16.
//
// Per-layer times of the 16-bit forward: every 1x1 launch of conv_patch_h16 took 19-21 us whatever its size (52x52 256->128:
// 66 MB of traffic and 5.7 GFLOP; 13x13 1024->512: 17 MB) - 8-32 K steps per block, each staged through registers, behind a
// prologue and an LDS epilogue longer than the matrix work. Here a block owns 128 consecutive output pixels x 128 output
// channels; both operands stream through 5-slot LDS rings by LDS-DMA (activations: [128 px][64 B] per 32-channel step, the
// 16-byte granules XOR-swizzled by (pixel >> 2) & 3 on the source side; weights: the packed fragment stream), 4 K steps
// ahead, one counted s_waitcnt vmcnt + one s_barrier per step. 5 x 16 KiB = 80 KiB of LDS exactly (two blocks per CU), so the
// folded scale / shift table is parked in the ring slot that step KT would have used, requested when the last group begins.
// Epilogue: d_epilogue (register layout, permlane32 swap, 16-byte stores). Needs Cin >= 128 (KT >= 4), Cout % 8 == 0,
// no head layout.
// =====================================================================================================
constexpr int E_SLOTS = 5;
constexpr int E_P = 4;                           // K steps in flight
constexpr int E_A_BYTES = 128 * 64;              // activation slab of one step
constexpr int E_W_BYTES = 4 * 2048;              // weight slab of one step (BN = 128)
constexpr int E_SLOT_BYTES = E_A_BYTES + E_W_BYTES;

// GATH = 1 (3x3 stride 2 as a GEMM with gathered rows, see conv1_dma_h16): source of this lane's activation granule for the
// K step (chunk, tap): the pixel's base + the tap's offset, or the zero page where the tap falls outside the image (only the
// top row / left column can: H and W are even)
struct EGather {
    const unsigned short* zp;
    int vmask[2];                    // per pixel row of this lane: bit 0 = output row > 0, bit 1 = output column > 0
    int tap, chunk;                  // of the NEXT step to request
};
template <int GATH>
__device__ __forceinline__ const unsigned short* e_gsrc(const ConvHArgs& p, const unsigned short* base, int vmask, const EGather& g) {
    if constexpr (GATH == 1) {                               // 3x3 stride 2 forward: tap (kh, kw) of the input window
        const int kh = (g.tap * 11) >> 5, kw = g.tap - 3 * kh;
        const int off = ((kh - 1) * p.Win + (kw - 1)) * p.x_ld + g.chunk * 32;
        const bool ok = (kh > 0 || (vmask & 1)) && (kw > 0 || (vmask & 2));
        return ok ? base + off : g.zp;
    } else {                                                 // stride-2 input gradient: neighbour (dr, dc) of the dz pixel
        const int dr = g.tap >> 1, dc = g.tap & 1;
        const int off = (dr * p.Win + dc) * p.x_ld + g.chunk * 32;
        const bool ok = (!dr || (vmask & 1)) && (!dc || (vmask & 2));
        return ok ? base + off : g.zp;
    }
}
template <int GATH> __device__ __forceinline__ void e_gnext(EGather& g) {
    constexpr int last = GATH == 1 ? 8 : 3;
    g.tap = g.tap == last ? 0 : g.tap + 1;
    g.chunk += g.tap == 0;
}

// LU = -1: a step of the steady loop (requests step t + E_P); LU = 0..3: the last four steps (nothing left to request)
template <typename T, int LU, int GATH = 0>
__device__ __forceinline__ void e_step(const ConvHArgs& p, char* ring, const unsigned short* const (&asrc)[2], const unsigned short* wsrc,
                                       int t, int& slot_w, int& slot_r, u32x4 (&af)[2][2], u32x4 (&bf)[2][2], f32x16 (&acc)[2][2],
                                       int wave, int lane, int wn, const int (&aoff)[2][2], const DRes& rs, u32x4 (&rr)[2][2][2],
                                       const float* ss_src, int ss_slot, EGather& eg) {
    typedef typename HTraits<T>::vec vec;
#define E_MFMA(i, j, s) acc[i][j] = HTraits<T>::mfma(__builtin_bit_cast(vec, bf[s][j]), __builtin_bit_cast(vec, af[i][s]), acc[i][j])
    constexpr bool fetch = LU < 0;
    u32x4 an[2][2], bn[2][2];
    const char* ab = ring + slot_r * E_SLOT_BYTES;
    const char* wb = ab + E_A_BYTES + wn * 2048 + lane * 16;
    char* dst = ring + slot_w * E_SLOT_BYTES + wave * 2048;
    __builtin_amdgcn_sched_barrier(0);
    E_MFMA(0, 0, 0);
    if (LU == 0)                                              // scale / shift -> the slot step KT would have used (4 bytes per lane)
        __builtin_amdgcn_global_load_lds((gptr_t)ss_src, (lptr_t)(ring + ss_slot * E_SLOT_BYTES + wave * 256), 4, 0, 0);
    if (fetch) glds16(wsrc + (size_t)(t + E_P) * 1024, dst + E_A_BYTES);
    if (LU == 1 && rs.has_res) {                              // residual rows: nothing but scale / shift is requested after them
#pragma unroll
        for (int j = 0; j < 2; ++j)
#pragma unroll
            for (int kp = 0; kp < 2; ++kp) {
                const int ch = rs.ch0 + j * 64 + kp * 16;
                rr[0][j][kp] = *reinterpret_cast<const u32x4*>(rs.rptr[0] + (ch < p.Cout ? ch : 0));
            }
    }
    __builtin_amdgcn_sched_barrier(0);
    E_MFMA(1, 0, 0);
    an[0][0] = *reinterpret_cast<const u32x4*>(ab + aoff[0][0]);
    an[0][1] = *reinterpret_cast<const u32x4*>(ab + aoff[0][1]);
    __builtin_amdgcn_sched_barrier(0);
    E_MFMA(0, 1, 0);
    if (fetch) glds16(wsrc + (size_t)(t + E_P) * 1024 + 512, dst + E_A_BYTES + 1024);
    an[1][0] = *reinterpret_cast<const u32x4*>(ab + aoff[1][0]);
    an[1][1] = *reinterpret_cast<const u32x4*>(ab + aoff[1][1]);
    __builtin_amdgcn_sched_barrier(0);
    E_MFMA(1, 1, 0);
    bn[0][0] = *reinterpret_cast<const u32x4*>(wb);
    bn[1][0] = *reinterpret_cast<const u32x4*>(wb + 1024);
    __builtin_amdgcn_sched_barrier(0);
    E_MFMA(0, 0, 1);
    if (fetch) {
        if constexpr (GATH != 0) glds16(e_gsrc<GATH>(p, asrc[0], eg.vmask[0], eg), dst);
        else glds16(asrc[0] + (size_t)(t + E_P) * 32, dst);
    }
    bn[0][1] = *reinterpret_cast<const u32x4*>(wb + 4096);
    bn[1][1] = *reinterpret_cast<const u32x4*>(wb + 4096 + 1024);
    __builtin_amdgcn_sched_barrier(0);
    E_MFMA(1, 0, 1);
    if (LU == 1 && rs.has_res) {
#pragma unroll
        for (int j = 0; j < 2; ++j)
#pragma unroll
            for (int kp = 0; kp < 2; ++kp) {
                const int ch = rs.ch0 + j * 64 + kp * 16;
                rr[1][j][kp] = *reinterpret_cast<const u32x4*>(rs.rptr[1] + (ch < p.Cout ? ch : 0));
            }
    }
    __builtin_amdgcn_sched_barrier(0);
    E_MFMA(0, 1, 1);
    if (fetch) {
        if constexpr (GATH != 0) {
            glds16(e_gsrc<GATH>(p, asrc[1], eg.vmask[1], eg), dst + 1024);
            e_gnext<GATH>(eg);                                // K order of the fragment stream: chunk-major, the taps inside
        } else {
            glds16(asrc[1] + (size_t)(t + E_P) * 32, dst + 1024);
        }
    }
    __builtin_amdgcn_sched_barrier(0);
    E_MFMA(1, 1, 1);
#undef E_MFMA
#pragma unroll
    for (int i = 0; i < 2; ++i)
#pragma unroll
        for (int q = 0; q < 2; ++q) { af[i][q] = an[i][q]; bf[q][i] = bn[q][i]; }
    slot_r = slot_r + 1 == E_SLOTS ? 0 : slot_r + 1;
    slot_w = slot_w + 1 == E_SLOTS ? 0 : slot_w + 1;
    __builtin_amdgcn_sched_barrier(0);
    // own requests of step t + 2 have landed. Steady loop: steps t + 3 and t + 4 (8 requests) are younger. Last group: step
    // KT - 4 leaves step KT - 1 and the scale / shift request (5), step KT - 3 everything but the residual rows; the last two
    // steps read what is already there and need neither a wait nor a rendezvous
    if (LU < 0) wait_vmcnt<8>();
    else if (LU == 0) wait_vmcnt<5>();
    else if (LU == 1) { if (rs.has_res) wait_vmcnt<8>(); else wait_vmcnt<0>(); }
    if (LU < 2) __builtin_amdgcn_s_barrier();
    __builtin_amdgcn_sched_barrier(0);
}

// GATH = 1: the same kernel as a GEMM with GATHERED activation rows = the 3x3 STRIDE-2 blocks (model.py:20-45: the five
// downsampling layers). z[r, c] = sum over (tap, ci) of x[2r + kh - 1, 2c + kw - 1, ci] W[co, ci, kh, kw] is a product with
// K = 9 Cin whose A row for output pixel m and K step (chunk, tap) is 32 consecutive channels of ONE input pixel: per lane a
// base pointer (pixel (2r, 2c)) plus a wave-uniform offset per step, and the zero page for the taps that leave the image at the
// top row / left column. The weights are the ordinary 3x3 fragment stream (chunk-major, taps inside). No patch, no halo
// re-reads beyond L2: round 2's register-staged stride-2 kernel ran these layers at 81-113 us (450-630 TF).
template <typename T, int BN, int GATH = 0>
__global__ __launch_bounds__(256) void conv1_dma_h16(const ConvHArgs p) {
    static_assert(BN == 128, "4 waves x (2 x 2) tiles of 32 x 32");
    extern __shared__ __attribute__((aligned(16))) char smem_raw[];
    char* ring = smem_raw;                                              // [E_SLOTS][ 128 px x 64 B | 4 x 2 KiB ]

    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int wm = wave >> 1, wn = wave & 1;
    const int fh = lane >> 5, frow = lane & 31;
    int bid = blockIdx.x;
    {
        const int nb = p.nblocks, q = nb / 8, r = nb % 8, xcd = bid % 8;
        bid = (xcd < r ? xcd * (q + 1) : r * (q + 1) + (xcd - r) * q) + bid / 8;
    }
    const int sp = fdiv(bid, p.mg_tn, p.tiles_n);                      // pixel tile; the n tiles of one pixel tile are neighbours
    const int n_tile = bid - sp * p.tiles_n;
    if (p.prio) __builtin_amdgcn_s_setprio(2);
    const int M = p.W;                                                  // 1x1: the tiling view is one row of M pixels
    const unsigned short* wsrc = p.wf + (size_t)(n_tile * (BN / 32) + wave) * p.KT * 1024 + lane * 8;
    // this lane's two activation rows (DMA rounds 2 wave, 2 wave + 1 of 16 pixels x 4 granules), clamped to the last pixel
    const unsigned short* asrc[2];
    EGather eg;
    eg.tap = 0; eg.chunk = 0; eg.vmask[0] = eg.vmask[1] = 3;
#pragma unroll
    for (int r = 0; r < 2; ++r) {
        const int px = 16 * (2 * wave + r) + (lane >> 2);
        int m = sp * 128 + px;
        m = m < M ? m : M - 1;
        const int gs = (lane & 3) ^ ((px >> 2) & 3);
        if constexpr (GATH == 1) {                                     // output pixel m = (img, orow, ocol) -> input pixel (2 orow, 2 ocol)
            const int img = fdiv(m, p.mg_PC, p.PC), rem = m - img * p.PC;
            const int orow = fdiv(rem, p.mg_TW, p.TW), ocol = rem - orow * p.TW;
            eg.vmask[r] = (orow > 0 ? 1 : 0) | (ocol > 0 ? 2 : 0);
            asrc[r] = p.x + ((size_t)(img * p.Hin + 2 * orow) * p.Win + 2 * ocol) * p.x_ld + p.x_off + gs * 8;
        } else if constexpr (GATH == 2) {                              // dz pixel m = (img, row, col): neighbours below / right exist?
            const int img = fdiv(m, p.mg_PC, p.PC), rem = m - img * p.PC;
            const int row = fdiv(rem, p.mg_TW, p.TW), col = rem - row * p.TW;
            eg.vmask[r] = (row < p.Hin - 1 ? 1 : 0) | (col < p.Win - 1 ? 2 : 0);
            asrc[r] = p.x + (size_t)m * p.x_ld + p.x_off + gs * 8;
        } else {
            asrc[r] = p.x + (size_t)m * p.x_ld + p.x_off + gs * 8;
        }
    }
    eg.zp = reinterpret_cast<const unsigned short*>(g_zero_page);   // (any granule of the zero page is zeros)
#pragma unroll
    for (int q = 0; q < E_P; ++q) {                                    // steps 0 .. 3, four requests each
        char* dst = ring + q * E_SLOT_BYTES + wave * 2048;
        glds16(wsrc + (size_t)q * 1024, dst + E_A_BYTES);
        glds16(wsrc + (size_t)q * 1024 + 512, dst + E_A_BYTES + 1024);
        if constexpr (GATH != 0) {
            glds16(e_gsrc<GATH>(p, asrc[0], eg.vmask[0], eg), dst);
            glds16(e_gsrc<GATH>(p, asrc[1], eg.vmask[1], eg), dst + 1024);
            e_gnext<GATH>(eg);
        } else {
            glds16(asrc[0] + (size_t)q * 32, dst);
            glds16(asrc[1] + (size_t)q * 32, dst + 1024);
        }
    }
    // fragment rows: byte offsets of this lane's two pixels x two k16 halves inside an activation slab, output pixels
    int aoff[2][2], mpix[2];
#pragma unroll
    for (int i = 0; i < 2; ++i) {
        const int px = wm * 64 + i * 32 + frow;
#pragma unroll
        for (int q = 0; q < 2; ++q) aoff[i][q] = (px << 6) | ((((2 * q + fh) ^ (px >> 2)) & 3) << 4);
        const int m = sp * 128 + px;
        mpix[i] = m < M ? m : -1;
    }
    const bool has_res = p.flags & YOLO_FLAG_RESIDUAL;
    DRes rs;
    rs.ch0 = n_tile * BN + wn * 32 + 8 * fh;
    rs.has_res = GATH == 2 ? false : has_res;                           // (GATH 2: the epilogue reads the running gradient itself)
    u32x4 rr[2][2][2];
#pragma unroll
    for (int i = 0; i < 2; ++i) {
        rs.rptr[i] = p.res + (size_t)(mpix[i] < 0 ? 0 : mpix[i]) * p.r_ld + p.r_off;
#pragma unroll
        for (int j = 0; j < 2; ++j)
#pragma unroll
            for (int kp = 0; kp < 2; ++kp) {
                const u32x4 z = {0u, 0u, 0u, 0u};
                rr[i][j][kp] = z;
            }
    }
    const float* ss_src;                                               // waves 0-1: scale, 2-3: shift of channel (wave & 1) * 64 + lane
    {
        const int n = n_tile * BN + (wave & 1) * 64 + lane;
        ss_src = (wave < 2 ? p.scale : p.shift) + (n < p.Cout ? n : p.Cout - 1);
    }
    const int ss_slot = p.KT % E_SLOTS;
    // step 0 landed; steps 1 .. 3 (12 requests) stay in flight
    wait_vmcnt<12>();
    __builtin_amdgcn_s_barrier();
    __builtin_amdgcn_sched_barrier(0);
    u32x4 af[2][2], bf[2][2];
#pragma unroll
    for (int i = 0; i < 2; ++i) {
        af[i][0] = *reinterpret_cast<const u32x4*>(ring + aoff[i][0]);
        af[i][1] = *reinterpret_cast<const u32x4*>(ring + aoff[i][1]);
    }
#pragma unroll
    for (int j = 0; j < 2; ++j) {
        bf[0][j] = *reinterpret_cast<const u32x4*>(ring + E_A_BYTES + wn * 2048 + lane * 16 + j * 4096);
        bf[1][j] = *reinterpret_cast<const u32x4*>(ring + E_A_BYTES + wn * 2048 + lane * 16 + j * 4096 + 1024);
    }
    f32x16 acc[2][2];
#pragma unroll
    for (int i = 0; i < 2; ++i)
#pragma unroll
        for (int j = 0; j < 2; ++j)
#pragma unroll
            for (int r = 0; r < 16; ++r) acc[i][j][r] = 0.f;
    // the loop's invariant at the top of step t: step t + 1 has landed and is visible (step t reads its fragments)
    wait_vmcnt<8>();
    __builtin_amdgcn_s_barrier();
    __builtin_amdgcn_sched_barrier(0);
    __builtin_amdgcn_s_setprio(0);
    int slot_w = E_P % E_SLOTS, slot_r = 1;
    int t = 0;
    for (; t + 4 < p.KT; ++t)
        e_step<T, -1, GATH>(p, ring, asrc, wsrc, t, slot_w, slot_r, af, bf, acc, wave, lane, wn, aoff, rs, rr, ss_src, ss_slot, eg);
    e_step<T, 0, GATH>(p, ring, asrc, wsrc, t, slot_w, slot_r, af, bf, acc, wave, lane, wn, aoff, rs, rr, ss_src, ss_slot, eg);
    e_step<T, 1, GATH>(p, ring, asrc, wsrc, t + 1, slot_w, slot_r, af, bf, acc, wave, lane, wn, aoff, rs, rr, ss_src, ss_slot, eg);
    e_step<T, 2, GATH>(p, ring, asrc, wsrc, t + 2, slot_w, slot_r, af, bf, acc, wave, lane, wn, aoff, rs, rr, ss_src, ss_slot, eg);
    e_step<T, 3, GATH>(p, ring, asrc, wsrc, t + 3, slot_w, slot_r, af, bf, acc, wave, lane, wn, aoff, rs, rr, ss_src, ss_slot, eg);
    if (p.prio) __builtin_amdgcn_s_setprio(2);

    const float* sstab = reinterpret_cast<const float*>(ring + ss_slot * E_SLOT_BYTES);   // [BN] scale, [BN] shift
    const bool nan_chk = p.flags & YOLO_FLAG_NANCHECK;
    const int ch0 = rs.ch0;
    size_t ooff[2];
#pragma unroll
    for (int i = 0; i < 2; ++i) {
        const int m = mpix[i] < 0 ? 0 : mpix[i];
        if (p.out_mode == YOLO_OUT_NHWC) {
            ooff[i] = (size_t)m * p.y_ld + p.y_off + ch0;
        } else {                                                      // 2x nearest upsample into the concat buffer
            const int HoWo = p.Ho * p.Wo;
            const int img = m / HoWo;
            const int rem = m - img * HoWo;
            const int ho = rem / p.Wo;
            const int wo2 = rem - ho * p.Wo;
            ooff[i] = ((size_t)(img * 2 * p.Ho + 2 * ho) * (2 * p.Wo) + 2 * wo2) * p.y_ld + p.y_off + ch0;
        }
    }
    bool saw_nan = false;
    if constexpr (GATH == 2) {                                        // stride-2 input gradient: the four parity classes of a 2 x 2 block
        d_epilogue_s2g<T, BN>(p, acc, mpix, ch0);
    } else {
    if (p.stats != nullptr) {                                         // train-mode forward: raw z + BatchNorm partial sums
        if (p.bz == nullptr) d_epilogue_stats<T, BN>(p, acc, mpix, ooff, ch0, lane, sp * 2 + wm);
        else if (p.bact == YOLO_ACT_LEAKY) d_epilogue_bstats<T, BN, YOLO_ACT_LEAKY>(p, acc, rr, has_res, mpix, ooff, ch0, lane, sp * 2 + wm);
        else d_epilogue_bstats<T, BN, YOLO_ACT_MISH>(p, acc, rr, has_res, mpix, ooff, ch0, lane, sp * 2 + wm);
    } else {
    YOLO_SWITCH_ACT(p.act, saw_nan = has_res ? (d_epilogue<T, BN, ACT, true>(p, acc, rr, sstab, mpix, ooff, ch0, wn, fh))
                                             : (d_epilogue<T, BN, ACT, false>(p, acc, rr, sstab, mpix, ooff, ch0, wn, fh)));
    }
    }
    if (nan_chk && saw_nan) atomicOr(p.nan_flag, 2);
}

#ifdef H16_PROBES
// ---- conv3_dmap_h16: the same K step and epilogue in a PERSISTENT block. DIAGNOSTIC LIBRARY ONLY (make probes, tile id 11):
// correct (stress-tested with a grid of 7 blocks) and measured at 0 to -7 % against conv3_dma_h16 - see DESIGN 4.5. --------------
// Per-block stamps of conv3_dma_h16 (128->256 @52x52): 2.1k cycles between a block's end and its successor's start, 3.8k of
// prologue (mostly the latency of the first patch + two weight steps), 19.5k of main loop, 4.4k of epilogue. Here a block walks
// tiles t = blockIdx.x, + gridDim.x, ... and requests the NEXT tile's first operands (weight steps 0-1, scale / shift, patch of
// chunk 0) right after its main loop, when only the accumulators are live, so that they travel during the epilogue:
//   * every wave has passed the last K step's barrier, so nobody reads the weight ring or the patch any more; scale / shift
//     are double-buffered by tile parity (the epilogue still reads this tile's);
//   * weight steps 2-3 follow the epilogue's stores, so that the loop-top wait stays the counted vmcnt(4): everything older
//     than those four requests - the prefetches AND the stores - has completed (the store drain overlaps the lane-row math);
//   * the lane-row geometry (p0, output pixels) and the patch source pointers of the next tile are recomputed after the
//     epilogue instead of being carried through it (the epilogue peaks at ~250 VGPRs).
template <typename T, int BN, int ACT, bool RES>
__global__ __launch_bounds__(256, 2) void conv3_dmap_h16(const ConvHArgs p) {
    constexpr int TN = BN / 64;
    constexpr int SLOT_BYTES = (BN / 32) * 2048;
    static_assert(BN / 32 == 4, "one weight n-tile per wave");
    extern __shared__ __attribute__((aligned(16))) char smem_raw[];
    char* patch = smem_raw;                                             // [2][256 px][64 B]
    char* wring = smem_raw + 2 * D_PATCH_BYTES;                         // [D_SLOTS][BN/32][2 KiB]
    float* sstab2 = reinterpret_cast<float*>(wring + D_SLOTS * SLOT_BYTES);   // [2 (tile parity)][BN scale | BN shift]

    // Per-lane index values are RE-DERIVED from an opaque copy of the thread id at every phase of the tile loop (refresh()):
    // left loop-invariant, hipcc hoists every address computed from them out of the loop and keeps them all alive across it
    // (256 VGPRs + 22 AGPRs and 5.7 KB of scratch per lane when forced to two blocks per CU).
    int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int wm = wave >> 1, wn = wave & 1;
    int fh = lane >> 5, frow = lane & 31;

    if (p.stagger > 0 && (int)blockIdx.x < p.first_wave) {             // see conv_f32_v2.hip
        unsigned hw;
        asm volatile("s_getreg_b32 %0, hwreg(HW_REG_HW_ID)" : "=s"(hw));
        const int slot = (hw >> 16) & 15;
        for (int i = 0; i < slot * p.stagger; ++i) __builtin_amdgcn_s_sleep(32);
    }
    if (p.prio) __builtin_amdgcn_s_setprio(2);

    const int Hp = p.Hin + 2;
    int gs = (tid & 3) ^ ((tid >> 4) & 3);                             // source granule of LDS granule (pixel (tid>>2) + 64 i, slot tid & 3)
    const unsigned short* zp = reinterpret_cast<const unsigned short*>(g_zero_page) + gs * 8;
    auto refresh = [&]() {
        int z;
        asm volatile("v_mov_b32 %0, 0" : "=v"(z));
        tid = (int)threadIdx.x + z;
        lane = tid & 63;
        fh = lane >> 5;
        frow = lane & 31;
        gs = (tid & 3) ^ ((tid >> 4) & 3);
        zp = reinterpret_cast<const unsigned short*>(g_zero_page) + gs * 8;
    };
    auto vrow = [&](int g) {
        const int n = fdiv(g, p.mg_H, p.H);
        return n * Hp + (g - n * p.H);
    };
    struct Geom { int n_tile, g0, c0, g_last, v0, PR; };
    auto geom_of = [&](int t) {
        const int nb = p.nblocks, q = nb / 8, r = nb % 8, xcd = t % 8;
        const int bid = (xcd < r ? xcd * (q + 1) : r * (q + 1) + (xcd - r) * q) + t / 8;
        const int sp = fdiv(bid, p.mg_tn, p.tiles_n);
        Geom g;
        g.n_tile = bid - sp * p.tiles_n;
        const int r_tile = fdiv(sp, p.mg_tw, p.tiles_w);
        const int w_tile = sp - r_tile * p.tiles_w;
        g.g0 = r_tile * p.TH;
        g.c0 = w_tile * p.TW;
        g.g_last = (g.g0 + p.TH < p.rows_total ? g.g0 + p.TH : p.rows_total) - 1;
        g.v0 = vrow(g.g0);
        g.PR = vrow(g.g_last) + 3 - g.v0;
        return g;
    };
    auto wsrc_of = [&](const Geom& g) { return p.wf + (size_t)(g.n_tile * (BN / 32) + wave) * p.KT * 1024 + lane * 8; };
    auto issue_w = [&](const unsigned short* wsrc, int q) {
        const int kq = q < p.KT ? q : p.KT - 1;
        const unsigned short* src = wsrc + (size_t)kq * 1024;
        char* dst = wring + q * SLOT_BYTES + wave * 2048;
        glds16(src, dst);
        glds16(src + 512, dst + 1024);
    };
    auto issue_ss = [&](const Geom& g, int par) {                      // 4 bytes per lane; waves 0-1 scale, 2-3 shift
        const int n = g.n_tile * BN + (wave & 1) * 64 + lane;
        const int ncl = n < p.Cout ? n : p.Cout - 1;
        __builtin_amdgcn_global_load_lds((gptr_t)((wave < 2 ? p.scale : p.shift) + ncl), (lptr_t)(sstab2 + par * 2 * BN + wave * 64), 4, 0, 0);
    };
    auto patch_src = [&](const Geom& g, const unsigned short* (&psrc)[D_NI]) {
#pragma unroll
        for (int i = 0; i < D_NI; ++i) {
            const int idx = (tid >> 2) + 64 * i;
            const int pr = fdiv(idx, p.mg_PC, p.PC), pc = idx - pr * p.PC;
            const int vv = g.v0 + pr;
            const int n = fdiv(vv, p.mg_Hp, Hp), yy = vv - n * Hp;
            const int hi = yy - 1, wi = g.c0 + pc - 1;
            const bool ok = (pr < g.PR) & ((unsigned)hi < (unsigned)p.Hin) & ((unsigned)wi < (unsigned)p.Win);
            const int pix = (n * p.Hin + hi) * p.Win + wi;
            psrc[i] = ok ? p.x + (size_t)pix * p.x_ld + p.x_off + gs * 8 : zp;
        }
    };
    auto issue_patch = [&](const unsigned short* const (&psrc)[D_NI]) {
        char* dst = patch + wave * 1024;
#pragma unroll
        for (int i = 0; i < D_NI; ++i) glds16(psrc[i], dst + i * 4096);
    };
    auto lane_rows = [&](const Geom& g, int (&p0)[2], int (&mpix)[2]) {
#pragma unroll
        for (int i = 0; i < 2; ++i) {
            const int pp = wm * 64 + i * 32 + ((((p.qperm >> ((frow >> 2) * 4)) & 7) << 2) | (frow & 3));
            const int r = fdiv(pp, p.mg_TW, p.TW), cc = pp - r * p.TW;
            const int gg = g.g0 + r;
            const bool ok = (pp < p.TH * p.TW) & (gg <= g.g_last) & (g.c0 + cc < p.W);
            p0[i] = ok ? (vrow(gg) - g.v0) * p.PC + cc : 0;
            mpix[i] = ok ? gg * p.W + g.c0 + cc : -1;
        }
    };

    DCtx<T, TN> c;
    c.KT = p.KT;
    c.PC = p.PC;
    int mpix[2];
    int t = blockIdx.x;
    Geom g = geom_of(t);
    c.wsrc = wsrc_of(g);
    issue_w(c.wsrc, 0);
    issue_w(c.wsrc, 1);
    issue_ss(g, 0);
    patch_src(g, c.psrc);
    issue_patch(c.psrc);
#pragma unroll
    for (int q = 2; q < D_P; ++q) issue_w(c.wsrc, q);
    lane_rows(g, c.p0, mpix);
    constexpr bool has_res = RES;
    const bool nan_chk = p.flags & YOLO_FLAG_NANCHECK;
    bool saw_nan = false;
    int par = 0;
    for (;;) {
        // weights of steps 0 and 1, scale / shift and the patch of chunk 0 have landed (and, from the second tile on, the previous
        // tile's stores have completed); steps 2 .. D_P-1 stay in flight - the same count the loop keeps
        wait_vmcnt<2 * (D_P - 2)>();
        __builtin_amdgcn_s_barrier();
        __builtin_amdgcn_sched_barrier(0);
        refresh();
        u32x4 af[2][2], bf[2][TN];
#pragma unroll
        for (int i = 0; i < 2; ++i) {
            const int px = c.p0[i];
            const int a0 = (px << 6) | ((((px >> 2) ^ fh) & 3) << 4);
            af[i][0] = *reinterpret_cast<const u32x4*>(patch + a0);
            af[i][1] = *reinterpret_cast<const u32x4*>(patch + (a0 ^ 32));
        }
#pragma unroll
        for (int j = 0; j < TN; ++j) {
            bf[0][j] = *reinterpret_cast<const u32x4*>(wring + wn * 2048 + lane * 16 + j * 4096);
            bf[1][j] = *reinterpret_cast<const u32x4*>(wring + wn * 2048 + lane * 16 + j * 4096 + 1024);
        }
        f32x16 acc[2][TN];
#pragma unroll
        for (int i = 0; i < 2; ++i)
#pragma unroll
            for (int j = 0; j < TN; ++j)
#pragma unroll
                for (int r = 0; r < 16; ++r) acc[i][j][r] = 0.f;
        int slot_w = D_P % D_SLOTS, slot_r = 1;
        const unsigned short* wp = c.wsrc + (size_t)D_P * 1024;
        __builtin_amdgcn_s_setprio(0);
        DRes rs;
        rs.ch0 = g.n_tile * BN + wn * 32 + 8 * fh;
        rs.has_res = has_res;
        u32x4 rr[2][TN][2];
#pragma unroll
        for (int i = 0; i < 2; ++i) {
            rs.rptr[i] = p.res + (size_t)(mpix[i] < 0 ? 0 : mpix[i]) * p.r_ld + p.r_off;
#pragma unroll
            for (int j = 0; j < TN; ++j)
#pragma unroll
                for (int kp = 0; kp < 2; ++kp) {
                    const u32x4 z = {0u, 0u, 0u, 0u};
                    rr[i][j][kp] = z;
                }
        }
        for (int chunk = 0; chunk + 1 < p.nchunks; ++chunk)
            d_chunk<T, BN, 0, 0, false>(p, c, chunk, patch, wring, wp, slot_w, slot_r, af, bf, acc, wave, lane, wn, fh, rs, rr);
        d_chunk<T, BN, 0, 0, true>(p, c, p.nchunks - 1, patch, wring, wp, slot_w, slot_r, af, bf, acc, wave, lane, wn, fh, rs, rr);
        if (p.prio) __builtin_amdgcn_s_setprio(2);

        // ---- the next tile's first operands leave now (see the header)
        const int tn = t + (int)gridDim.x;
        const bool has_next = tn < p.nblocks;
        Geom gn = g;
        refresh();
        if (has_next) {
            gn = geom_of(tn);
            const unsigned short* wn_src = wsrc_of(gn);
            issue_w(wn_src, 0);
            issue_w(wn_src, 1);
            issue_ss(gn, par ^ 1);
            const unsigned short* ps[D_NI];
            patch_src(gn, ps);
            issue_patch(ps);
        }
        // ---- epilogue of this tile
        refresh();
        const int ch0 = g.n_tile * BN + wn * 32 + 8 * fh;
        size_t ooff[2];
#pragma unroll
        for (int i = 0; i < 2; ++i) {
            const int m = mpix[i] < 0 ? 0 : mpix[i];
            if (p.out_mode == YOLO_OUT_NHWC) {
                ooff[i] = (size_t)m * p.y_ld + p.y_off + ch0;
            } else {                                                  // 2x nearest upsample into the concat buffer
                const int HoWo = p.Ho * p.Wo;
                const int img = m / HoWo;
                const int rem = m - img * HoWo;
                const int ho = rem / p.Wo;
                const int wo2 = rem - ho * p.Wo;
                ooff[i] = ((size_t)(img * 2 * p.Ho + 2 * ho) * (2 * p.Wo) + 2 * wo2) * p.y_ld + p.y_off + ch0;
            }
        }
        const float* sstab = sstab2 + par * 2 * BN;
        saw_nan |= d_epilogue<T, BN, ACT, RES>(p, acc, rr, sstab, mpix, ooff, ch0, wn, fh);
        if (!has_next) break;
        // ---- the rest of the next tile's prologue: weight steps 2-3 behind the stores, then the per-lane geometry
        refresh();
        t = tn;
        g = gn;
        par ^= 1;
        c.wsrc = wsrc_of(g);
#pragma unroll
        for (int q = 2; q < D_P; ++q) issue_w(c.wsrc, q);
        patch_src(g, c.psrc);
        lane_rows(g, c.p0, mpix);
    }
    if (nan_chk && saw_nan) atomicOr(p.nan_flag, 2);
}

#endif  // H16_PROBES

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

// The same conversion one 32 x 32 x taps CELL per block: the rows of a cell are contiguous runs of 32 * taps floats in the
// OIHW tensor (forward: one output channel's 32 input channels; dgrad: one output channel's 32 input channels read as the
// GEMM's n), so they are read with 16-byte loads, rounded once, parked in LDS and written out in fragment order with one
// 16-byte store per (tap, half, lane). The element-wise kernel above reads 4 bytes at a stride of taps * 4 (and a 64-bit
// divide) per element: 2 x ~100 us per fine-tune step for the forward layouts and as much again for the gradient layouts.
// Needs cin % 32 == 0 (row alignment); other items keep the element-wise kernel.
template <typename T, bool DGRAD>
__global__ __launch_bounds__(256) void pack_batch_tiled_h16(const PackBatchH b) {
    __shared__ __attribute__((aligned(16))) unsigned short tile[32][32 * 9 + 8];
    int k = 0;
    while (k + 1 < b.n && (int)blockIdx.x >= b.it[k + 1].first_block) ++k;
    const PackItemH& q = b.it[k];
    const int taps = q.ks * q.ks;
    const int chunks = q.KT / taps;
    const int cell = (int)blockIdx.x - q.first_block;
    const int nt = cell / chunks, chunk = cell - nt * chunks;
    const int run = 32 * taps;                                       // floats per row of the cell
    const int tid = threadIdx.x;
    // rows: forward = output channel a (n of the GEMM), columns (ci_local, tap); dgrad = output channel c (k of the GEMM),
    // columns (ci_local = n of the GEMM, source tap)
    {
        const int r = tid >> 3, part = tid & 7;                      // 8 threads per row
        const int row_ch = (DGRAD ? chunk : nt) * 32 + r;            // output channel of this row
        const int col0 = (DGRAD ? nt : chunk) * 32;                  // first input channel of the run
        const bool row_ok = row_ch < q.cout && col0 < q.cin;
        const float* src = q.w + ((size_t)row_ch * q.cin + col0) * taps;
        const int avail = row_ok ? ((q.cin - col0 < 32 ? q.cin - col0 : 32) * taps) : 0;   // floats of the run that exist
        for (int f = part * 4; f < run; f += 32) {
            f32x4 v = {0.f, 0.f, 0.f, 0.f};
            if (f + 3 < avail) v = *reinterpret_cast<const f32x4*>(src + f);
            else {
#pragma unroll
                for (int e = 0; e < 4; ++e) if (f + e < avail) v[e] = src[f + e];
            }
#pragma unroll
            for (int e = 0; e < 4; ++e) tile[r][f + e] = HTraits<T>::from_f32(v[e]);
        }
    }
    __syncthreads();
    for (int w = tid; w < taps * 128; w += 256) {                    // (tap, s, lane): one 16-byte store each
        const int lane = w & 63, s2 = (w >> 6) & 1, tap = w >> 7;
        const int al = lane & 31, cl = s2 * 16 + 8 * (lane >> 5);
        unsigned short h[8];
#pragma unroll
        for (int e = 0; e < 8; ++e) h[e] = DGRAD ? tile[cl + e][al * taps + (taps - 1 - tap)] : tile[al][(cl + e) * taps + tap];
        u32x4 o;
#pragma unroll
        for (int e = 0; e < 4; ++e) o[e] = (unsigned)h[2 * e] | ((unsigned)h[2 * e + 1] << 16);
        const size_t idx = (((size_t)nt * q.KT + (size_t)chunk * taps + tap) * 2 + s2) * 512 + (size_t)lane * 8;
        *reinterpret_cast<u32x4*>(q.wf + idx) = o;
    }
}

// ------------------------------------------------------------------------------ host side
static const bool g_h_stagger = !(getenv("YOLO_NO_STAGGER"));
static const bool g_h_dma = !(getenv("YOLO_NO_DMA"));
static const bool g_h_prio = !(getenv("YOLO_DMA_PRIO") && getenv("YOLO_DMA_PRIO")[0] == '0');
static const bool g_h_dma_solo = getenv("YOLO_DMA_SOLO") != nullptr;          // experiment: one conv3_dma_h16 block per CU (100 KB of LDS)
#ifdef H16_PROBES
// diagnostic library: conv3_dmap_h16 (persistent blocks) for tile 0; value = grid cap (a small one exercises the tile loop)
static const int g_h_dma_persist = getenv("YOLO_DMA_PERSIST") ? atoi(getenv("YOLO_DMA_PERSIST")) : 0;
static const int g_h_num_cus = 256;                                           // MI355X: 8 XCDs x 32 CUs; two of these blocks per CU
#else
static const int g_h_dma_persist = 0;
#endif

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
    // two passes over the items: those whose rows are 16-byte aligned runs (cin % 32 == 0) go to the tiled kernel, one cell
    // per block; the rest (the 3-channel stem) to the element-wise one
    for (int tiled = 1; tiled >= 0; --tiled) {
        int base = 0;
        while (base < n) {
            PackBatchH b;
            b.n = 0;
            int blocks = 0;
            for (; base < n && b.n < H_PACK_BATCH; ++base) {
                const int i = base;
                const bool can_tile = cin[i] % 32 == 0 && ks[i] * ks[i] <= 9;
                if (can_tile != (tiled == 1)) continue;
                PackItemH& q = b.it[b.n++];
                q.w = w[i]; q.wf = (unsigned short*)wf[i]; q.cout = cout[i]; q.cin = cin[i]; q.ks = ks[i];
                if (dgrad) {
                    const int coutp = round_up(cout[i], 32);
                    q.total = (long long)h16_frag_elems(cin[i], coutp, ks[i]);
                    q.KT = (coutp / 32) * ks[i] * ks[i];
                } else {
                    q.total = (long long)h16_frag_elems(cout[i], cin[i], ks[i]);
                    q.KT = (round_up(cin[i], 32) / 32) * ks[i] * ks[i];
                }
                if (tiled) {
                    q.nblocks = (int)(q.total / 1024 / (ks[i] * ks[i]));          // cells: n-tiles x 32-channel chunks
                } else {
                    const long long nb = (q.total + 255) / 256;
                    q.nblocks = (int)(nb < 1024 ? nb : 1024);
                }
                q.first_block = blocks;
                blocks += q.nblocks;
            }
            if (b.n == 0) continue;
            if (dtype == YOLO_BF16) {
                if (tiled) {
                    if (dgrad) hipLaunchKernelGGL((pack_batch_tiled_h16<__bf16, true>), dim3(blocks), dim3(256), 0, s, b);
                    else hipLaunchKernelGGL((pack_batch_tiled_h16<__bf16, false>), dim3(blocks), dim3(256), 0, s, b);
                } else {
                    if (dgrad) hipLaunchKernelGGL((pack_batch_h16<__bf16, true>), dim3(blocks), dim3(256), 0, s, b);
                    else hipLaunchKernelGGL((pack_batch_h16<__bf16, false>), dim3(blocks), dim3(256), 0, s, b);
                }
            } else {
                if (tiled) {
                    if (dgrad) hipLaunchKernelGGL((pack_batch_tiled_h16<_Float16, true>), dim3(blocks), dim3(256), 0, s, b);
                    else hipLaunchKernelGGL((pack_batch_tiled_h16<_Float16, false>), dim3(blocks), dim3(256), 0, s, b);
                } else {
                    if (dgrad) hipLaunchKernelGGL((pack_batch_h16<_Float16, true>), dim3(blocks), dim3(256), 0, s, b);
                    else hipLaunchKernelGGL((pack_batch_h16<_Float16, false>), dim3(blocks), dim3(256), 0, s, b);
                }
            }
            const int rc = check_launch("pack_batch_h16");
            if (rc) return rc;
        }
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

// all four parity classes of one layer in ONE launch (they were four ~6 us launches per layer and step): the classes'
// fragment streams lie back to back in `wf`; `end[cls]` = end of class cls in that concatenation
struct S2ClsEnds { long long end[4]; };
template <typename T>
__global__ void pack_dgrad_s2_cls_h16(const float* __restrict__ w, unsigned short* __restrict__ wf, int cout, int cin, S2ClsEnds ends) {
    const long long total = ends.end[3];
    for (long long g = blockIdx.x * (long long)blockDim.x + threadIdx.x; g < total; g += (long long)gridDim.x * blockDim.x) {
        const int cls = g < ends.end[0] ? 0 : (g < ends.end[1] ? 1 : (g < ends.end[2] ? 2 : 3));
        const long long i = g - (cls ? ends.end[cls - 1] : 0);
        const int ph = cls >> 1, pw = cls & 1;
        const int NT = (ph + 1) * (pw + 1);                       // taps of the class: 1, 2, 2, 4
        const int KT = (cout / 32) * NT;
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
        wf[g] = HTraits<T>::from_f32(v);
    }
}

// ---- the same gradient as ONE launch for the layers with few dx channels (C = cin <= 64, multiple of 32): the four classes
// are the column blocks of one GEMM over the dz pixels, K = 4 neighbours x cout, N = 4 classes x C:
//   dx[n, 2r+ph, 2c+pw, :] = sum over neighbours (dr <= ph, dc <= pw) of dz[n, r+dr, c+dc, :] . W[:, :, ph+1-2dr, pw+1-2dc]
// 7 of the 16 (neighbour, class) blocks are zeros (1.78 x the matrix work), which these layers can afford: the four tap-subset
// launches each read all of dz and write a quarter of dx in half-line pieces, HBM-bound at 2.5 TB/s (4 x ~97 us for the
// 64-channel layers); here dz is read once and dx written once in full 16-byte rows.
static bool s2g_ok(int cout, int cin) {
    static const bool off = getenv("YOLO_NO_S2G") != nullptr;
    return !off && (cin == 32 || cin == 64) && cout % 32 == 0 && cout >= 32;
}
static size_t s2g_frag_elems(int cout, int cin) { return s2g_ok(cout, cin) ? (size_t)(4 * cin / 32) * (4 * cout / 32) * 1024 : 0; }

// [n_tile32][kt][s][lane][e]: n = class * cin + c, kt = chunk * 4 + neighbour, k = dz channel chunk * 32 + s * 16 + 8 (lane >> 5) + e
template <typename T>
__global__ void pack_dgrad_s2g_h16(const float* __restrict__ w, unsigned short* __restrict__ wf, int cout, int cin, long long total) {
    const int KT = 4 * (cout / 32);
    for (long long i = blockIdx.x * (long long)blockDim.x + threadIdx.x; i < total; i += (long long)gridDim.x * blockDim.x) {
        const int e = (int)(i & 7);
        const int lane = (int)((i >> 3) & 63);
        const int s = (int)((i >> 9) & 1);
        const long long rest = i >> 10;
        const int kt = (int)(rest % KT);
        const int nt = (int)(rest / KT);
        const int n = nt * 32 + (lane & 31);
        const int cls = n / cin, c = n - cls * cin;
        const int ph = cls >> 1, pw = cls & 1;
        const int chunk = kt >> 2, nb = kt & 3;
        const int dr = nb >> 1, dc = nb & 1;
        const int co = chunk * 32 + s * 16 + 8 * (lane >> 5) + e;
        float v = 0.f;
        if (dr <= ph && dc <= pw && co < cout) v = w[((size_t)co * cin + c) * 9 + (ph + 1 - 2 * dr) * 3 + (pw + 1 - 2 * dc)];
        wf[i] = HTraits<T>::from_f32(v);
    }
}

size_t h16_dgrad_s2_elems(int cout, int cin) {
    size_t n = 0;
    for (int cls = 0; cls < 4; ++cls) n += cls_frag_elems(cin, cout, cls);
    return n + s2g_frag_elems(cout, cin);                   // the fused layout follows the four class streams
}

int h16_pack_dgrad_s2(const float* w_oihw, void* wf, int cout, int cin, int dtype, hipStream_t s) {
    S2ClsEnds ends;
    long long acc = 0;
    for (int cls = 0; cls < 4; ++cls) {
        if (mask_count(cls_mask(cls >> 1, cls & 1)) != ((cls >> 1) + 1) * ((cls & 1) + 1)) return fail(YOLO_ERR_ARG, "dgrad_s2: class taps");
        acc += (long long)cls_frag_elems(cin, cout, cls);
        ends.end[cls] = acc;
    }
    const int grid = (int)((acc + 255) / 256 < 8192 ? (acc + 255) / 256 : 8192);
    if (dtype == YOLO_BF16)
        hipLaunchKernelGGL(pack_dgrad_s2_cls_h16<__bf16>, dim3(grid), dim3(256), 0, s, w_oihw, (unsigned short*)wf, cout, cin, ends);
    else
        hipLaunchKernelGGL(pack_dgrad_s2_cls_h16<_Float16>, dim3(grid), dim3(256), 0, s, w_oihw, (unsigned short*)wf, cout, cin, ends);
    if (int rc = check_launch("pack_dgrad_s2_cls_h16")) return rc;
    const long long tg = (long long)s2g_frag_elems(cout, cin);
    if (tg) {
        unsigned short* wg = (unsigned short*)wf + acc;
        const int g2 = (int)((tg + 255) / 256 < 8192 ? (tg + 255) / 256 : 8192);
        if (dtype == YOLO_BF16) hipLaunchKernelGGL(pack_dgrad_s2g_h16<__bf16>, dim3(g2), dim3(256), 0, s, w_oihw, wg, cout, cin, tg);
        else hipLaunchKernelGGL(pack_dgrad_s2g_h16<_Float16>, dim3(g2), dim3(256), 0, s, w_oihw, wg, cout, cin, tg);
        return check_launch("pack_dgrad_s2g_h16");
    }
    return YOLO_OK;
}

static void fill_magics(ConvHArgs& a) {
    a.mg_H = magic_of(a.H); a.mg_TW = magic_of(a.TW); a.mg_PC = magic_of(a.PC);
    a.mg_tn = magic_of(a.tiles_n); a.mg_tw = magic_of(a.tiles_w); a.mg_Hp = magic_of(a.Hin + 2);
}

static void pick_tile_h(int Hin, int Hout, int Wout, int ks, int stride, int* th, int* tw, int* prmax, int patch_cap = H_PATCH_CAP) {
    if (ks == 1) { *th = 1; *tw = 128; *prmax = 1; return; }
    double best = -1;
    *th = 1; *tw = 1; *prmax = 3 + 2;
    for (int TW = 1; TW <= (Wout < 126 ? Wout : 126); ++TW) {
        int TH = 128 / TW;
        int pr = 0;
        while (TH >= 1) {
            const int cross = (TH + Hout - 1) / Hout;
            pr = stride * (TH - 1) + 3 + 2 * cross;
            if (pr * (stride * (TW - 1) + 3) <= patch_cap) break;
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
    const size_t lds = patch_bytes + 256 * sizeof(int);
    if constexpr (BN == 64) hipLaunchKernelGGL((conv_patch_h16_n64<T, KS, STRIDE>), dim3(a.nblocks), dim3(256), lds, s, a);
    else hipLaunchKernelGGL((conv_patch_h16<T, KS, STRIDE, BN>), dim3(a.nblocks), dim3(256), lds, s, a);
    return check_launch("conv_patch_h16");
}

template <typename T>
static int launch_dma(ConvHArgs& a, hipStream_t s) {
    constexpr int BN = 128;
    a.tiles_n = ceil_div(a.Cout, BN);
    const int tiles_r = ceil_div(a.rows_total, a.TH);
    a.nblocks = a.tiles_n * a.tiles_w * tiles_r;
    fill_magics(a);
    a.first_wave = 2 * 256;
    const long mfma_cycles = (long)a.KT * 8 * (BN / 64) / 2 * 32;
    a.stagger = g_h_stagger ? (int)((mfma_cycles + 1024) / 2048) : 0;
    a.bufmask = 1;
    a.prio = g_h_prio ? 1 : 0;
    a.mtab_off = 2 * D_PATCH_BYTES + D_SLOTS * (BN / 32) * 2048;
    size_t lds = (size_t)a.mtab_off + 256 * sizeof(int) + 2 * BN * sizeof(float);
    if (g_h_dma_solo) lds = 100 * 1024;                     // experiment: ONE block per CU (how fast is a block that has the SIMDs to itself?)
    static LdsOnce once;                                    // per device (common.h)
    if (int rc = reserve_lds(once, reinterpret_cast<const void*>(&conv3_dma_h16<T, BN>), lds, "conv3_dma_h16")) return rc;
#ifdef H16_PROBES
    // diagnostic library only (make probes): tile 9 = MFMA-shape probe, tiles 16 + bits = ablations of the K step (d_kstep)
    {
        auto go = [&](auto kern, const char* what) {
            (void)hipFuncSetAttribute(reinterpret_cast<const void*>(kern), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
            hipLaunchKernelGGL(kern, dim3(a.nblocks), dim3(256), lds, s, a);
            return check_launch(what);
        };
        switch (a.cls_ph) {
            case 9: return go(&conv3_dma_h16<__bf16, BN, 16>, "conv3_dma_h16 (probe 16)");
            case 17: return go(&conv3_dma_h16<__bf16, BN, 1>, "conv3_dma_h16 (probe 1)");
            case 18: return go(&conv3_dma_h16<__bf16, BN, 2>, "conv3_dma_h16 (probe 2)");
            case 19: return go(&conv3_dma_h16<__bf16, BN, 3>, "conv3_dma_h16 (probe 3)");
            case 20: return go(&conv3_dma_h16<__bf16, BN, 4>, "conv3_dma_h16 (probe 4)");
            case 24: return go(&conv3_dma_h16<__bf16, BN, 8>, "conv3_dma_h16 (probe 8)");
            case 28: return go(&conv3_dma_h16<__bf16, BN, 12>, "conv3_dma_h16 (probe 12)");
            case 31: return go(&conv3_dma_h16<__bf16, BN, 15>, "conv3_dma_h16 (probe 15)");
            case 21: return go(&conv3_dma_h16<__bf16, BN, 32>, "conv3_dma_h16 (probe 32)");
            case 22: return go(&conv3_dma_h16<__bf16, BN, 64>, "conv3_dma_h16 (probe 64)");
            default: break;
        }
    }
#endif
#ifdef H16_PROBES
    if (a.cls_ph == 11) {                                   // persistent blocks with next-tile prefetch (A/B)
        const size_t lds_p = (size_t)2 * D_PATCH_BYTES + D_SLOTS * (BN / 32) * 2048 + 2 * 2 * BN * sizeof(float);
        const int cap = g_h_dma_persist > 0 ? g_h_dma_persist : 2 * g_h_num_cus;
        const int grid = a.nblocks < cap ? a.nblocks : cap;
        auto go = [&](auto kern) {
            static LdsOnce once_p;                          // one per instantiation of this generic lambda
            if (int rc = reserve_lds(once_p, reinterpret_cast<const void*>(kern), lds_p, "conv3_dmap_h16")) return rc;
            hipLaunchKernelGGL(kern, dim3(grid), dim3(256), lds_p, s, a);
            return check_launch("conv3_dmap_h16");
        };
        const bool res = a.flags & YOLO_FLAG_RESIDUAL;
        YOLO_SWITCH_ACT(a.act, return res ? go(&conv3_dmap_h16<T, BN, ACT, true>) : go(&conv3_dmap_h16<T, BN, ACT, false>));
        return fail(YOLO_ERR_ARG, "conv3_dmap_h16: activation");
    }
#endif
    hipLaunchKernelGGL((conv3_dma_h16<T, BN>), dim3(a.nblocks), dim3(256), lds, s, a);
    return check_launch("conv3_dma_h16");
}

template <typename T, int GATH = 0>
static int launch_dma1(ConvHArgs& a, hipStream_t s) {
    constexpr int BN = 128;
    a.tiles_n = ceil_div(a.Cout, BN);
    a.nblocks = a.tiles_n * ceil_div(a.W, 128);
    fill_magics(a);
    a.prio = g_h_prio ? 1 : 0;
    const size_t lds = (size_t)E_SLOTS * E_SLOT_BYTES;      // 80 KiB: two blocks per CU
    static LdsOnce once;
    if (int rc = reserve_lds(once, reinterpret_cast<const void*>(&conv1_dma_h16<T, BN, GATH>), lds, "conv1_dma_h16")) return rc;
    hipLaunchKernelGGL((conv1_dma_h16<T, BN, GATH>), dim3(a.nblocks), dim3(256), lds, s, a);
    return check_launch("conv1_dma_h16");
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
    const size_t lds = (size_t)a.mtab_off + 256 * sizeof(int);
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
    a.stats = nullptr; a.stats_ld = 0;
    a.x = (const unsigned short*)dz; a.scale = nullptr; a.shift = nullptr;
    a.res = (const unsigned short*)residual; a.y = dx; a.nan_flag = nullptr;
    a.Cin = cout; a.Cout = cin;
    a.x_ld = dz_ld; a.x_off = dz_off; a.y_ld = dx_ld; a.y_off = dx_off; a.r_ld = r_ld; a.r_off = r_off;
    a.Hin = ho; a.Win = wo; a.Ho = ho; a.Wo = wo;
    const long long M = (long long)n * ho * wo;
    if (M * 4 > 0x7fffffffLL) return fail(YOLO_ERR_UNSUPPORTED, "dgrad_s2: too many pixels");
    if (s2g_ok(cout, cin) && g_h_dma && (dx_ld & 7) == 0 && (dx_off & 7) == 0 && (!residual || ((r_ld & 7) == 0 && (r_off & 7) == 0))) {
        size_t skip = 0;
        for (int cls = 0; cls < 4; ++cls) skip += cls_frag_elems(cin, cout, cls);
        a.wf = (const unsigned short*)wf + skip;
        a.scale = a.shift = reinterpret_cast<const float*>(a.wf);    // the kernel's prologue fetches a table it does not use here
        a.Cout = 4 * cin;
        a.H = cin; a.W = (int)M; a.rows_total = 1; a.TH = 1; a.TW = wo; a.PC = ho * wo;   // H: channels per class; TW / PC: divisors
        a.nchunks = cout / 32;
        a.KT = a.nchunks * 4;
        a.act = YOLO_ACT_NONE; a.out_mode = YOLO_OUT_NHWC; a.flags = residual ? YOLO_FLAG_RESIDUAL : 0;
        a.nc5 = 1;
        a.tiles_w = 1; a.first_wave = 0; a.stagger = 0; a.bufmask = 1; a.patch_cap = 128; a.mtab_off = 0;
        a.cls_ph = a.cls_pw = 0;
        if (dtype == YOLO_BF16) return launch_dma1<__bf16, 2>(a, s);
        return launch_dma1<_Float16, 2>(a, s);
    }
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

// =====================================================================================================
// conv3_ws_h16 (round 3): the 3x3 layers with <= 64 input AND output channels (32 -> 64 at 208^2, its input gradient 64 -> 32,
// the stride-2 32 -> 64 at 416 -> 208^2), weights in REGISTERS.
// conv_patch_h16_n64 gives such a layer one 128-pixel tile per block: 36 MFMAs of matrix work per wave behind a prologue of index
// arithmetic, a register-staged patch and an LDS round trip - 173 us for a layer whose bytes take ~85 us (32 -> 64 with the
// residual) and whose matrix work takes ~25. These layers are a stream: the whole filter bank is 36 KB, so
//   * a wave keeps ALL weight fragments in registers (9 taps x Cin/16 k-steps x Cout/32 n-tiles x 4 VGPRs = 144) for the lifetime
//     of a PERSISTENT workgroup (2 per CU) and walks 8 x 16-pixel output tiles;
//   * the tile's input patch with halo ((8s+1... ) x (16s+...) pixels, s = stride) arrives by LDS-DMA into a double buffer while
//     the previous tile is multiplied: one wait + ONE barrier per tile, placed between the MFMA phase and the epilogue, so the
//     stores of tile t overlap the request and the matrix work of tile t + 1;
//   * patch rows are Cin x 2 bytes with the 16-byte chunks XOR-swizzled on the DMA's source side (by (p >> 2) & 3 for 64-byte
//     rows, (p >> 1) & 7 for 128-byte rows): 16 consecutive pixels cover all banks (stride 2: two-way);
//   * operand swap as in the other DMA kernels: weights = A, pixels = B, D = [channel][pixel]; one v_permlane32_swap per register
//     pair leaves a lane with 8 consecutive channels of its pixel: 16-byte stores / residual loads.
// Halo pixels outside the image read the zero page.
constexpr int WS_TH = 8, WS_TW = 16;
struct ConvWsArgs {
    const unsigned short* x;
    const unsigned short* wf;
    const float* scale;
    const float* shift;
    const unsigned short* res;
    unsigned short* y;
    int* nan_flag;
    int N, Hin, Win, Ho, Wo;
    int x_ld, x_off, y_ld, y_off, r_ld, r_off;
    int Cout, KT, act, flags;
    int tiles_w, tiles_per_img, total_tiles;
    unsigned mg_tpi, mg_tw;
    float* stats;                         // STATS instances (train-mode forward): per-wave BatchNorm partial sums [row][2][stats_ld]
    int stats_ld;
};

// STATS = true (ACT none, no residual): raw z AND the BatchNorm partial sums of the rounded values (d_epilogue_stats). A wave
// keeps ONE running pair of sums per channel for all its tiles: per tile and 8-channel group the 32 pixel lanes of a half are
// folded by the reduce-scatter butterfly of d_epilogue_bstats (16 live values) and the result is added into the wave's private
// [2][32 NT] LDS accumulator with ds_add_f32 (one lane per address and tile: a fixed order); the accumulator is the wave's row.
template <typename T, int CIN, int NT, int STRIDE, int ACT, bool RES, bool STATS = false>
__global__ __launch_bounds__(256, 2) void conv3_ws_h16(const ConvWsArgs p) {
    typedef typename HTraits<T>::vec vec;
    constexpr int PR = STRIDE * (WS_TH - 1) + 3, PC = STRIDE * (WS_TW - 1) + 3, P = PR * PC;
    constexpr int RB = CIN * 2, CH = CIN / 8;                         // bytes and 16-byte chunks per patch pixel
    constexpr int KS16 = CIN / 16;                                     // k16 steps per tap
    constexpr int NCHUNK = P * CH, ROUNDS = (NCHUNK + 255) / 256;
    constexpr int BUF = ((P * RB + 255) / 256) * 256;
    extern __shared__ __attribute__((aligned(256))) char smem_raw[];   // [2][BUF] patches | [NT * 32] scale | [NT * 32] shift
    float* sstab = reinterpret_cast<float*>(smem_raw + 2 * BUF);
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int hl = lane >> 5, pl = lane & 31;
    float* wsum = sstab + 2 * NT * 32 + wave * (2 * NT * 32);          // STATS: this wave's [2][NT * 32] sums
    if (STATS) {
        for (int i = lane; i < 2 * NT * 32; i += 64) wsum[i] = 0.f;
    }

    // ---- the filter bank, once
    u32x4 wreg[NT][9 * KS16];
#pragma unroll
    for (int nt = 0; nt < NT; ++nt)
#pragma unroll
        for (int tap = 0; tap < 9; ++tap)
#pragma unroll
            for (int cs = 0; cs < KS16; ++cs) {
                const int kt = (cs >> 1) * 9 + tap, sh = cs & 1;
                wreg[nt][tap * KS16 + cs] = *reinterpret_cast<const u32x4*>(p.wf + ((size_t)nt * p.KT + kt) * 1024 + sh * 512 + lane * 8);
            }
    if (!STATS && tid < NT * 32) {
        const int c = tid < p.Cout ? tid : p.Cout - 1;
        sstab[tid] = tid < p.Cout ? p.scale[c] : 0.f;
        sstab[NT * 32 + tid] = tid < p.Cout ? p.shift[c] : 0.f;
    }

    auto swz = [](int pp) { return CH == 4 ? (pp >> 2) & 3 : (pp >> 1) & 7; };
    // request the patch of tile t into buffer b: ROUNDS wave-instructions of 64 x 16 bytes, lane-linear in LDS
    auto request = [&](int t, char* dst) {
        const int img = fdiv(t, p.mg_tpi, p.tiles_per_img), rem = t - img * p.tiles_per_img;
        const int th = fdiv(rem, p.mg_tw, p.tiles_w), tw = rem - th * p.tiles_w;
        const int hi0 = th * WS_TH * STRIDE - 1, wi0 = tw * WS_TW * STRIDE - 1;
        const unsigned short* zp = reinterpret_cast<const unsigned short*>(g_zero_page) + (lane & 7) * 8;
#pragma unroll
        for (int r = 0; r < ROUNDS; ++r) {
            const int idx = r * 256 + tid;
            if (idx < NCHUNK) {
                const int pp = idx / CH, sl = idx - pp * CH;
                const int pr = pp / PC, pc = pp - pr * PC;
                const int hi = hi0 + pr, wi = wi0 + pc;
                const bool ok = (unsigned)hi < (unsigned)p.Hin && (unsigned)wi < (unsigned)p.Win;
                const unsigned short* src = p.x + ((size_t)(img * p.Hin + hi) * p.Win + wi) * p.x_ld + p.x_off + ((sl ^ swz(pp)) * 8);
                glds16(ok ? src : zp, dst + (r * 256 + wave * 64) * 16);
            }
        }
    };

    // this lane's output pixel inside a tile, and its patch pixel for tap (0, 0)
    const int r_o = 2 * wave + (pl >> 4), c_o = pl & 15;
    const int p0 = (r_o * STRIDE) * PC + c_o * STRIDE;
    const int stride_t = gridDim.x;
    int t = blockIdx.x;
    if (t >= p.total_tiles) return;
    request(t, smem_raw);
    asm volatile("s_waitcnt vmcnt(0) lgkmcnt(0)" ::: "memory");
    __syncthreads();
    bool saw_nan = false;
    for (int it = 0; t < p.total_tiles; t += stride_t, ++it) {
        char* cur = smem_raw + (it & 1) * BUF;
        if (t + stride_t < p.total_tiles) request(t + stride_t, smem_raw + ((it + 1) & 1) * BUF);
        f32x16 acc[NT];
#pragma unroll
        for (int nt = 0; nt < NT; ++nt)
#pragma unroll
            for (int r = 0; r < 16; ++r) acc[nt][r] = 0.f;
#pragma unroll
        for (int tap = 0; tap < 9; ++tap) {
            const int pp = p0 + (tap / 3) * PC + (tap % 3);
            const int f = swz(pp);
            const char* row = cur + pp * RB;
#pragma unroll
            for (int cs = 0; cs < KS16; ++cs) {
                const u32x4 a = *reinterpret_cast<const u32x4*>(row + (((cs * 2 + hl) ^ f) << 4));
#pragma unroll
                for (int nt = 0; nt < NT; ++nt)
                    acc[nt] = HTraits<T>::mfma(__builtin_bit_cast(vec, wreg[nt][tap * KS16 + cs]), __builtin_bit_cast(vec, a), acc[nt]);
            }
        }
        // the next tile's patch has had the whole matrix phase to land; everybody is done reading `cur`
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        __builtin_amdgcn_s_barrier();
        asm volatile("" ::: "memory");
        // ---- epilogue (registers -> 16-byte stores), overlapping the next tile's request and matrix phase
        const int img = fdiv(t, p.mg_tpi, p.tiles_per_img), rem = t - img * p.tiles_per_img;
        const int th = fdiv(rem, p.mg_tw, p.tiles_w), tw = rem - th * p.tiles_w;
        const int ho = th * WS_TH + r_o, wo = tw * WS_TW + c_o;
        const bool live = ho < p.Ho && wo < p.Wo;
        const size_t m = ((size_t)img * p.Ho + (live ? ho : 0)) * p.Wo + (live ? wo : 0);
#pragma unroll
        for (int nt = 0; nt < NT; ++nt) {
            float v[16];
#pragma unroll
            for (int g = 0; g < 4; ++g) {
                const f32x4 sc = *reinterpret_cast<const f32x4*>(sstab + nt * 32 + 8 * g + 4 * hl);
                const f32x4 sf = *reinterpret_cast<const f32x4*>(sstab + NT * 32 + nt * 32 + 8 * g + 4 * hl);
#pragma unroll
                for (int e = 0; e < 4; ++e) v[4 * g + e] = STATS ? acc[nt][4 * g + e] : act_c<ACT>(acc[nt][4 * g + e] * sc[e] + sf[e]);
            }
#pragma unroll
            for (int kp = 0; kp < 2; ++kp) {
                float w[8];
#pragma unroll
                for (int e = 0; e < 4; ++e) {
                    const auto sw = __builtin_amdgcn_permlane32_swap(__float_as_uint(v[8 * kp + e]), __float_as_uint(v[8 * kp + 4 + e]), false, false);
                    w[e] = __uint_as_float(sw[0]);
                    w[4 + e] = __uint_as_float(sw[1]);
                }
                const int ch = nt * 32 + kp * 16 + 8 * hl;
                const bool ok = live && ch < p.Cout;
                if (RES && ok) {
                    const u32x4 r4 = *reinterpret_cast<const u32x4*>(p.res + m * p.r_ld + p.r_off + ch);
#pragma unroll
                    for (int e = 0; e < 4; ++e) {
                        w[2 * e] += HTraits<T>::to_f32((unsigned short)(r4[e] & 0xffffu));
                        w[2 * e + 1] += HTraits<T>::to_f32((unsigned short)(r4[e] >> 16));
                    }
                }
                u32x4 o;
#pragma unroll
                for (int e = 0; e < 4; ++e) {
                    saw_nan |= __builtin_isunordered(w[2 * e], w[2 * e + 1]);
                    o[e] = pack2<T>(w[2 * e], w[2 * e + 1]);
                }
                if (ok) *reinterpret_cast<u32x4*>(p.y + m * p.y_ld + p.y_off + ch) = o;
                if constexpr (STATS) {
                    const float lv = ok ? 1.f : 0.f;
                    float sq[2][8];
#pragma unroll
                    for (int e = 0; e < 4; ++e) {
                        const float r0 = HTraits<T>::to_f32((unsigned short)(o[e] & 0xffffu)) * lv;
                        const float r1 = HTraits<T>::to_f32((unsigned short)(o[e] >> 16)) * lv;
                        sq[0][2 * e] = r0; sq[0][2 * e + 1] = r1;
                        sq[1][2 * e] = r0 * r0; sq[1][2 * e + 1] = r1 * r1;
                    }
                    const bool b3 = lane & 8, b2 = lane & 4, b1 = lane & 2;
                    float l8[8];
#pragma unroll
                    for (int e = 0; e < 8; ++e) {
                        const auto sw = __builtin_amdgcn_permlane16_swap(__float_as_uint(sq[0][e]), __float_as_uint(sq[1][e]), false, false);
                        l8[e] = __uint_as_float(sw[0]) + __uint_as_float(sw[1]);
                    }
                    float l4[4];
#pragma unroll
                    for (int e = 0; e < 4; ++e) {
                        const float t0 = l8[e] + dpp_f<0x140>(l8[e]);
                        const float t1 = l8[e + 4] + dpp_f<0x140>(l8[e + 4]);
                        l4[e] = b3 ? t1 : t0;
                    }
                    float l2[2];
#pragma unroll
                    for (int e = 0; e < 2; ++e) {
                        const float t0 = l4[e] + dpp_f<0x141>(l4[e]);
                        const float t1 = l4[e + 2] + dpp_f<0x141>(l4[e + 2]);
                        l2[e] = b2 ? t1 : t0;
                    }
                    const float u0 = l2[0] + dpp_f<0x4E>(l2[0]);
                    const float u1 = l2[1] + dpp_f<0x4E>(l2[1]);
                    float l1 = b1 ? u1 : u0;
                    l1 += dpp_f<0xB1>(l1);
                    // lane: quantity (lane >> 4) & 1, channel nt * 32 + kp * 16 + 8 hl + 4 b3 + 2 b2 + b1; the odd lane of a pair is a duplicate
                    if (!(lane & 1))
                        atomicAdd(wsum + ((lane >> 4) & 1) * (NT * 32) + nt * 32 + kp * 16 + 8 * hl + (b3 ? 4 : 0) + (b2 ? 2 : 0) + (b1 ? 1 : 0), l1);
                }
            }
        }
    }
    if ((p.flags & YOLO_FLAG_NANCHECK) && saw_nan) atomicOr(p.nan_flag, 2);
    if constexpr (STATS) {
        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
        const int row = blockIdx.x * 4 + wave;
        for (int i = lane; i < 2 * NT * 32; i += 64) {
            const int qty = i / (NT * 32), c = i - qty * (NT * 32);
            p.stats[((size_t)row * 2 + qty) * p.stats_ld + c] = wsum[i];
        }
    }
}

static bool ws_eligible(const yolo_conv_desc* d, const void* residual) {
    static const bool off = getenv("YOLO_NO_CONV3_WS") != nullptr;
    if (d->tile != 14 && (off || d->tile != 0)) return false;
    if (d->ksize != 3 || d->out_mode != YOLO_OUT_NHWC || d->dtype == YOLO_F32) return false;
    const bool shape = (d->stride == 1 && ((d->cin == 32 && d->cout > 32 && d->cout <= 64) || (d->cin == 64 && d->cout <= 32))) ||
                       (d->stride == 2 && d->cin == 32 && d->cout > 32 && d->cout <= 64);
    if (!shape || d->cout % 8) return false;
    if ((d->x_ld & 7) || (d->x_off & 7) || (d->y_ld & 7) || (d->y_off & 7)) return false;
    if (residual && ((d->r_ld & 7) || (d->r_off & 7))) return false;
    if (d->stride == 2 && ((d->h & 1) || (d->w & 1))) return false;
    return true;
}

static int ws_grid(int total_tiles) { return total_tiles < 512 ? total_tiles : 512; }   // two persistent workgroups per CU

template <typename T, int CIN, int NT, int STRIDE>
static int launch_ws(ConvWsArgs& a, hipStream_t s) {
    constexpr int PR = STRIDE * (WS_TH - 1) + 3, PC = STRIDE * (WS_TW - 1) + 3;
    constexpr int BUF = ((PR * PC * CIN * 2 + 255) / 256) * 256;
    const size_t lds = 2 * (size_t)BUF + (2 + 8) * NT * 32 * sizeof(float);     // patches | scale, shift | 4 waves x [2][NT * 32] sums
    const int grid = ws_grid(a.total_tiles);
    const bool res = a.flags & YOLO_FLAG_RESIDUAL;
    auto go = [&](auto kern) -> int {
        static LdsOnce once;
        if (int rc = reserve_lds(once, reinterpret_cast<const void*>(kern), lds, "conv3_ws_h16")) return rc;
        hipLaunchKernelGGL(kern, dim3(grid), dim3(256), lds, s, a);
        return check_launch("conv3_ws_h16");
    };
    if (a.stats) return go(&conv3_ws_h16<T, CIN, NT, STRIDE, YOLO_ACT_NONE, false, true>);
    YOLO_SWITCH_ACT(a.act, return res ? go(&conv3_ws_h16<T, CIN, NT, STRIDE, ACT, true>) : go(&conv3_ws_h16<T, CIN, NT, STRIDE, ACT, false>));
    return fail(YOLO_ERR_ARG, "conv3_ws_h16: activation");
}

static int conv_ws_launch(const yolo_conv_desc* d, const void* x, const void* wf, const float* scale, const float* shift, const void* residual,
                          void* y, int32_t* nan_flag, hipStream_t s, float* stats = nullptr, int stats_ld = 0) {
    ConvWsArgs a;
    a.stats = stats; a.stats_ld = stats_ld;
    a.x = (const unsigned short*)x; a.wf = (const unsigned short*)wf; a.scale = scale; a.shift = shift;
    a.res = (const unsigned short*)residual; a.y = (unsigned short*)y; a.nan_flag = nan_flag;
    a.N = d->n; a.Hin = d->h; a.Win = d->w;
    a.Ho = (d->h + 2 - 3) / d->stride + 1; a.Wo = (d->w + 2 - 3) / d->stride + 1;
    a.x_ld = d->x_ld; a.x_off = d->x_off; a.y_ld = d->y_ld; a.y_off = d->y_off; a.r_ld = d->r_ld; a.r_off = d->r_off;
    a.Cout = d->cout; a.KT = (d->cin / 32) * 9; a.act = d->act; a.flags = d->flags;
    a.tiles_w = ceil_div(a.Wo, WS_TW);
    a.tiles_per_img = a.tiles_w * ceil_div(a.Ho, WS_TH);
    const long long total = (long long)a.tiles_per_img * d->n;
    if (total > 0x7fffffffLL || (long long)d->n * d->h * d->w > 0x7fffffffLL) return fail(YOLO_ERR_UNSUPPORTED, "conv3_ws_h16: too many pixels");
    a.total_tiles = (int)total;
    a.mg_tpi = magic_of(a.tiles_per_img); a.mg_tw = magic_of(a.tiles_w);
    if ((a.flags & YOLO_FLAG_NANCHECK) && !nan_flag) return fail(YOLO_ERR_ARG, "conv3_ws_h16: nan_flag is NULL");
    const bool bf = d->dtype == YOLO_BF16;
    if (d->stride == 2) return bf ? launch_ws<__bf16, 32, 2, 2>(a, s) : launch_ws<_Float16, 32, 2, 2>(a, s);
    if (d->cin == 32) return bf ? launch_ws<__bf16, 32, 2, 1>(a, s) : launch_ws<_Float16, 32, 2, 1>(a, s);
    return bf ? launch_ws<__bf16, 64, 1, 1>(a, s) : launch_ws<_Float16, 64, 1, 1>(a, s);
}

int conv_h16_launch(const yolo_conv_desc* d, const void* x, const void* wf, const float* scale, const float* shift,
                    const void* residual, void* y, int32_t* nan_flag, hipStream_t s) {
    return conv_h16_launch_stats(d, x, wf, scale, shift, residual, y, nan_flag, nullptr, nullptr, nullptr, s, nullptr);
}

// stats != nullptr: the launch must be one of the DMA kernels (conv3_dma_h16 / conv1_dma_h16) with an identity epilogue; it
// also writes per-wave BatchNorm partial sums. dry (rows_ld != nullptr with x == nullptr): no launch, rows_ld[0..1] = the
// number of partial rows and their channel stride, or 0 rows when this convolution has no fused-statistics kernel.
int conv_h16_launch_stats(const yolo_conv_desc* d, const void* x, const void* wf, const float* scale, const float* shift,
                          const void* residual, void* y, int32_t* nan_flag, float* stats, int* rows_ld, const size_t* stats_bytes,
                          hipStream_t s, const ConvBStats* bs) {
    const bool dry = rows_ld != nullptr && x == nullptr;
    if (rows_ld) { rows_ld[0] = 0; rows_ld[1] = 0; }
    if (d->cin % 32) return fail(YOLO_ERR_UNSUPPORTED, "conv (16-bit): cin %d must be a multiple of 32", d->cin);
    if (d->ksize == 1 && d->stride != 1) return fail(YOLO_ERR_UNSUPPORTED, "conv (16-bit): strided 1x1");
    if ((d->x_ld & 7) || (d->x_off & 7)) return fail(YOLO_ERR_ARG, "conv (16-bit): x_ld/x_off must be multiples of 8");
    ConvHArgs a;
    a.x = (const unsigned short*)x; a.wf = (const unsigned short*)wf; a.scale = scale; a.shift = shift;
    a.res = (const unsigned short*)residual; a.y = y; a.nan_flag = nan_flag;
    a.stats = stats; a.stats_ld = round_up(d->cout, 128);
    const bool want_stats = stats != nullptr || dry;
    // the DMA kernels request their scale / shift table in the prologue whatever the epilogue does with it: in statistics mode
    // (identity epilogue, table unused) hand them readable memory - the statistics buffer itself (>= cout floats)
    if (stats != nullptr) { a.scale = stats; a.shift = stats; }
    if (want_stats && (d->act != YOLO_ACT_NONE || d->out_mode != YOLO_OUT_NHWC || (!bs && (d->flags & YOLO_FLAG_RESIDUAL))))
        return dry ? YOLO_OK : fail(YOLO_ERR_ARG, "conv (16-bit): statistics need the identity epilogue (raw convolution output)");
    if (bs) {                                               // backward statistics: sums over the gradient this launch writes
        if (!want_stats) return fail(YOLO_ERR_ARG, "conv (16-bit): backward statistics without a statistics buffer");
        if (d->stride != 1) return dry ? YOLO_OK : fail(YOLO_ERR_UNSUPPORTED, "conv (16-bit): backward statistics: stride-1 input gradients only");
        if (!dry) {
            if (!bs->z || !bs->mean || !bs->scale || !bs->shift) return fail(YOLO_ERR_ARG, "conv (16-bit): backward statistics: null pointer");
            if ((bs->z_ld & 7) || (bs->z_off & 7) || d->cout % 8) return fail(YOLO_ERR_ARG, "conv (16-bit): backward statistics: z_ld / z_off / channels must be multiples of 8");
            if (bs->act != YOLO_ACT_LEAKY && bs->act != YOLO_ACT_MISH) return fail(YOLO_ERR_ARG, "conv (16-bit): backward statistics: LeakyReLU or Mish block expected");
            a.bz = (const unsigned short*)bs->z; a.bz_ld = bs->z_ld; a.bz_off = bs->z_off;
            a.bmean = bs->mean; a.bscale = bs->scale; a.bshift = bs->shift; a.bact = bs->act;
        }
    }
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
    // tile ids (16-bit): 5 / 6 = conv_patch_h16 with 64 / 128 output channels per block, 8 = conv3_dma_h16 (3x3 stride 1)
    const bool dma_ok = d->ksize == 3 && d->stride == 1 && d->cout > 64 && d->cin <= 2048 && d->cout % 8 == 0 && d->out_mode != YOLO_OUT_HEAD &&
                        (d->y_ld & 7) == 0 && (d->y_off & 7) == 0 && (!residual || ((d->r_ld & 7) == 0 && (d->r_off & 7) == 0));
    // 1x1 with >= 128 output channels and >= 4 K steps: conv1_dma_h16 (tile 8 / default); tiles 5, 6 keep conv_patch_h16
    const bool dma1_ok = d->ksize == 1 && d->stride == 1 && d->cout >= 128 && d->cin >= 128 && d->cout % 8 == 0 && d->out_mode != YOLO_OUT_HEAD &&
                         (d->y_ld & 7) == 0 && (d->y_off & 7) == 0 && (!residual || ((d->r_ld & 7) == 0 && (d->r_off & 7) == 0));
    if (d->tile == 8 && !dma_ok && !dma1_ok)
        return fail(YOLO_ERR_UNSUPPORTED, "conv (16-bit): tile 8 needs 3x3 stride 1 with more than 64 output channels, or 1x1 with >= 128 input and output channels");
    const bool use_dma = dma_ok && (d->tile >= 8 || (d->tile == 0 && g_h_dma));
    // lane quad -> pixel quad of a 32-pixel m-tile. Identity makes every patch ds_read_b128 2-way bank-conflicted with the 32x4
    // pixel tiles of 52x52 / 104x104 (its two 16-lane groups are quads {0,3,5,6} and {1,2,4,7}: 4 rows whose patch offsets collide
    // mod 16); sending even tile rows to one group and odd rows to the other removes that, and measured 1-4 % at every size
    // (profiles/r02/ab_quad_permutation.txt). Tile 10 keeps the identity map for A/B.
    if (d->tile == 14 && !ws_eligible(d, residual)) return fail(YOLO_ERR_UNSUPPORTED, "conv (16-bit): tile 14 needs a 3x3 32 -> 64 (stride 1 / 2) or 64 -> 32 (stride 1) layer, NHWC");
    if (!want_stats && ws_eligible(d, residual)) return conv_ws_launch(d, x, wf, scale, shift, residual, y, nan_flag, s);
    if (want_stats && !bs && ws_eligible(d, residual)) {            // train-mode forward of the <= 64-channel 3x3 blocks: one row per wave
        const int ho = (d->h + 2 - 3) / d->stride + 1, wo = (d->w + 2 - 3) / d->stride + 1;
        const long long total = (long long)ceil_div(wo, WS_TW) * ceil_div(ho, WS_TH) * d->n;
        if (total <= 0x7fffffffLL) {
            const int rows = 4 * ws_grid((int)total);
            if (rows_ld) { rows_ld[0] = rows; rows_ld[1] = a.stats_ld; }
            if (dry) return YOLO_OK;
            if (stats_bytes && *stats_bytes < (size_t)rows * 2 * a.stats_ld * sizeof(float)) return fail(YOLO_ERR_WORKSPACE, "conv statistics: buffer too small");
            return conv_ws_launch(d, x, wf, stats, stats, nullptr, y, nan_flag, s, stats, a.stats_ld);
        }
    }
    a.qperm = d->tile == 10 ? 0x76543210u : 0x76452310u;
    a.cls_ph = (d->tile == 0 && g_h_dma_persist) ? 11 : d->tile;
    if (dma1_ok && (d->tile == 8 || (d->tile == 0 && g_h_dma))) {
        a.H = 1; a.W = (int)M; a.rows_total = 1; a.TH = 1; a.TW = 128; a.PC = 128;
        a.nchunks = d->cin / 32;
        a.KT = a.nchunks;
        a.act = d->act; a.out_mode = d->out_mode; a.flags = d->flags;
        a.nc5 = 1;
        a.tiles_w = 1; a.first_wave = 0; a.stagger = 0; a.bufmask = 1; a.patch_cap = 128; a.mtab_off = 0;
        if (want_stats) {
            const int rows = 2 * ceil_div(a.W, 128);
            if (rows_ld) { rows_ld[0] = rows; rows_ld[1] = a.stats_ld; }
            if (dry) return YOLO_OK;
            if (stats_bytes && *stats_bytes < (size_t)rows * 2 * a.stats_ld * sizeof(float)) return fail(YOLO_ERR_WORKSPACE, "conv statistics: buffer too small");
        }
        if (d->dtype == YOLO_BF16) return launch_dma1<__bf16>(a, s);
        return launch_dma1<_Float16>(a, s);
    }
    // 3x3 stride 2 with >= 128 output channels: conv1_dma_h16 as a GEMM with gathered rows (tile 0 / 13; tiles 5, 6 keep conv_patch_h16)
    static const bool no_s2_dma = getenv("YOLO_NO_S2_DMA") != nullptr;
    const bool s2_ok = d->ksize == 3 && d->stride == 2 && d->cout >= 128 && d->cout % 8 == 0 && d->out_mode == YOLO_OUT_NHWC &&
                       (d->y_ld & 7) == 0 && (d->y_off & 7) == 0 && (!residual || ((d->r_ld & 7) == 0 && (d->r_off & 7) == 0)) &&
                       (long long)a.Ho * a.Wo < 0x7fffffffLL && d->cin * 9 / 32 >= 4;
    if (d->tile == 13 && !s2_ok) return fail(YOLO_ERR_UNSUPPORTED, "conv (16-bit): tile 13 needs 3x3 stride 2 with >= 128 output channels");
    if (s2_ok && (d->tile == 13 || (d->tile == 0 && g_h_dma && !no_s2_dma))) {
        a.H = 1; a.W = (int)M; a.rows_total = 1; a.TH = 1; a.TW = a.Wo; a.PC = a.Ho * a.Wo;      // TW / PC: divisors of the pixel index
        a.nchunks = d->cin / 32;
        a.KT = a.nchunks * 9;
        a.act = d->act; a.out_mode = d->out_mode; a.flags = d->flags;
        a.nc5 = 1;
        a.tiles_w = 1; a.first_wave = 0; a.stagger = 0; a.bufmask = 1; a.patch_cap = 128; a.mtab_off = 0;
        if (want_stats) {
            const int rows = 2 * ceil_div(a.W, 128);
            if (rows_ld) { rows_ld[0] = rows; rows_ld[1] = a.stats_ld; }
            if (dry) return YOLO_OK;
            if (stats_bytes && *stats_bytes < (size_t)rows * 2 * a.stats_ld * sizeof(float)) return fail(YOLO_ERR_WORKSPACE, "conv statistics: buffer too small");
        }
        if (d->dtype == YOLO_BF16) return launch_dma1<__bf16, 1>(a, s);
        return launch_dma1<_Float16, 1>(a, s);
    }
    if (want_stats && !use_dma) return dry ? YOLO_OK : fail(YOLO_ERR_UNSUPPORTED, "conv (16-bit): no fused-statistics kernel for this convolution");
    if (d->ksize == 1) {
        a.H = 1; a.W = (int)M; a.rows_total = 1; a.TH = 1; a.TW = 128; a.PC = 128;
    } else {
        a.H = a.Ho; a.W = a.Wo; a.rows_total = d->n * a.Ho;
        pick_tile_h(d->h, a.Ho, a.Wo, 3, d->stride, &a.TH, &a.TW, &prmax, use_dma ? D_PATCH_PIX : H_PATCH_CAP);
        a.PC = d->stride * (a.TW - 1) + 3;
    }
    if (use_dma) {
        if (prmax * a.PC > D_PATCH_PIX) return fail(YOLO_ERR_UNSUPPORTED, "conv3_dma_h16: patch of %d pixels", prmax * a.PC);
        a.patch_cap = D_PATCH_PIX;
        a.tiles_w = ceil_div(a.W, a.TW);
        a.nchunks = d->cin / 32;
        a.KT = a.nchunks * 9;
        a.act = d->act; a.out_mode = d->out_mode; a.flags = d->flags;
        a.nc5 = d->out_mode == YOLO_OUT_HEAD ? d->cout / 3 : 1;
        if (want_stats) {
            const int rows = 2 * a.tiles_w * ceil_div(a.rows_total, a.TH);
            if (rows_ld) { rows_ld[0] = rows; rows_ld[1] = a.stats_ld; }
            if (dry) return YOLO_OK;
            if (stats_bytes && *stats_bytes < (size_t)rows * 2 * a.stats_ld * sizeof(float)) return fail(YOLO_ERR_WORKSPACE, "conv statistics: buffer too small");
        }
        if (d->dtype == YOLO_BF16) return launch_dma<__bf16>(a, s);
        return launch_dma<_Float16>(a, s);
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

// ---- the network's first block on the matrix cores, 16-bit output (3 -> 32 channels, 3x3, stride 1, pad 1) ---------------
// stem_f32.hip does this layer on the vector ALUs: 432 packed FMAs per pixel whose 864 weights arrive through the scalar cache -
// 224 us at B = 32, 416^2, against ~75 us for its bytes (66 MB of fp32 NCHW input, 354 MB of 16-bit NHWC output). For a 16-bit
// output the arithmetic the reference's autocast does (model.py:80-86 under train.py:53) IS a 16-bit matrix product: input and
// weights rounded to the 16-bit type, fp32 accumulation. So: K = 27 taps padded to 32 = two v_mfma_f32_32x32x16 per 32 pixels.
//   * operand swap (cdna_hip_programming.md T21): the WEIGHTS are the A operand (M = the 32 output channels) and the pixels the
//     B operand (N = 32 consecutive pixels), so D = [channel][pixel]: a lane owns pixel (lane & 31) and channels 8g + 4h + {0..3}
//     (h = lane >> 5), and one v_permlane32_swap per register pair leaves it with 8 consecutive channels = one 16-byte store;
//   * B operand straight from global memory: lane (pixel, h) loads the 16 taps k = 16 q + 8 h + e (q = 0..1, e = 0..7) of its
//     pixel - no LDS, no barrier; the 9x overlap between neighbouring pixels is absorbed by L1 / L2 as in the vector kernel.
//     Tap k = (c * 3 + dh) * 3 + dw (the order of the packed weights, stem_pack_kernel); k >= 27 is zero on both sides;
//   * a 32-pixel tile none of whose pixels touches the image border (85 % of them at 416^2) takes loads at fixed per-lane
//     offsets with no masking at all; the others mask per tap;
//   * the weights (two A operands), scale and shift live in registers for the wave's ST_TILES tiles;
//   * every input element is the centre tap of exactly one pixel: the input NaN guard of model.py:175 rides on taps 4, 13, 22.
constexpr int ST_TILES = 8;              // 32-pixel tiles per wave
template <typename T, int ACT>
__global__ __launch_bounds__(256) void stem3x3_mfma_h16(const float* __restrict__ x, const float* __restrict__ wt,
                                                        const float* __restrict__ scale, const float* __restrict__ shift,
                                                        unsigned short* __restrict__ y, int H, int W, long long total, int y_ld,
                                                        int y_off, unsigned mg_HW, unsigned mg_W, int* nan_flag) {
    typedef typename HTraits<T>::vec vec;
    const int lane = threadIdx.x & 63, hf = lane >> 5, col = lane & 31;
    const int HW = H * W;
    // this lane's 16 taps: value offset relative to its pixel's channel-0 element, and whether the tap exists (k < 27)
    int toff[16];
    bool tval[16];
#pragma unroll
    for (int e = 0; e < 16; ++e) {
        const int k = (e >> 3) * 16 + hf * 8 + (e & 7);
        const int c = k / 9, r = k - 9 * c, dh = r / 3, dw = r - 3 * dh;
        tval[e] = k < 27;
        toff[e] = tval[e] ? c * HW + (dh - 1) * W + (dw - 1) : 0;
    }
    // A operands: weights of channel `col`, taps 16 q + 8 hf + e
    vec wa[2];
#pragma unroll
    for (int q = 0; q < 2; ++q) {
        u32x4 pk;
#pragma unroll
        for (int e2 = 0; e2 < 4; ++e2) {
            const int k0 = q * 16 + hf * 8 + 2 * e2;
            const float w0 = k0 < 27 ? wt[k0 * 32 + col] : 0.f, w1 = k0 + 1 < 27 ? wt[(k0 + 1) * 32 + col] : 0.f;
            pk[e2] = pack2<T>(w0, w1);
        }
        wa[q] = __builtin_bit_cast(vec, pk);
    }
    __shared__ __attribute__((aligned(16))) float sstab[64];        // scale[32], shift[32]: read back per tile (32 registers otherwise)
    if (threadIdx.x < 64) sstab[threadIdx.x] = threadIdx.x < 32 ? scale[threadIdx.x] : shift[threadIdx.x - 32];
    __syncthreads();
    bool bad_in = false, bad_out = false;
    const long long tile0 = ((long long)blockIdx.x * 4 + (threadIdx.x >> 6)) * ST_TILES;
    // A tile's 16 loads are issued together and nothing in `fetch` waits for them: a tap outside the image reads the pixel's own
    // element instead and is zeroed in `finish` through a bit mask (so is a tap k >= 27).
    unsigned tmask = 0;                                     // bit e: tap e exists
#pragma unroll
    for (int e = 0; e < 16; ++e) tmask |= tval[e] ? (1u << e) : 0u;
    struct Tile { float v[16]; int p; unsigned okm; bool live; };
    auto fetch = [&](int it, Tile& t) {
        const long long first = (tile0 + it) * 32;
        const long long p_raw = (first < total ? first : 0) + col;          // past the end: any valid pixel, never used
        t.live = first < total && p_raw < total;
        t.p = (int)(p_raw < total ? p_raw : total - 1);
        const int n = fdiv(t.p, mg_HW, HW), rem = t.p - n * HW;
        const int h = fdiv(rem, mg_W, W), w = rem - h * W;
        // 32-bit byte offsets from the scalar base (the launcher checks the input is below 4 GB): one VGPR per address, not two
        const unsigned pxo = (unsigned)(n * 3 * HW + rem) * 4u;              // channel 0 of this pixel
        const char* xb = reinterpret_cast<const char*>(x);
        unsigned off[16];
        t.okm = tmask;
        const bool border = h == 0 || h == H - 1 || w == 0 || w == W - 1;
        if (__ballot(border) == 0ull) {
#pragma unroll
            for (int e = 0; e < 16; ++e) off[e] = pxo + (unsigned)(toff[e] * 4);
        } else {
#pragma unroll
            for (int e = 0; e < 16; ++e) {
                const int k = (e >> 3) * 16 + hf * 8 + (e & 7);
                const int c = k / 9, r = k - 9 * c, dh = r / 3, dw = r - 3 * dh;
                const bool ok = (unsigned)(h + dh - 1) < (unsigned)H && (unsigned)(w + dw - 1) < (unsigned)W;
                off[e] = pxo + (ok ? (unsigned)(toff[e] * 4) : 0u);
                t.okm &= ok ? ~0u : ~(1u << e);
            }
        }
#pragma unroll
        for (int e = 0; e < 16; ++e) t.v[e] = *reinterpret_cast<const float*>(xb + (size_t)off[e]);
    };
    auto finish = [&](Tile& t) {
        float v[16];
#pragma unroll
        for (int e = 0; e < 16; ++e)                        // all ones or zero from bit e of the mask: v_bfe_i32 + v_and
            v[e] = __uint_as_float(__float_as_uint(t.v[e]) & (unsigned)__builtin_amdgcn_sbfe((int)t.okm, e, 1));
        // centre taps: k = 4 (hf 0, e 4), 13 (hf 1, e 5), 22 (hf 0, e 14)
        if (t.live) bad_in |= hf == 0 ? (v[4] != v[4]) || (v[14] != v[14]) : (v[5] != v[5]);
        f32x16 acc = {0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f};
#pragma unroll
        for (int q = 0; q < 2; ++q) {
            u32x4 pk;
#pragma unroll
            for (int e2 = 0; e2 < 4; ++e2) pk[e2] = pack2<T>(v[q * 8 + 2 * e2], v[q * 8 + 2 * e2 + 1]);
            acc = HTraits<T>::mfma(wa[q], __builtin_bit_cast(vec, pk), acc);
        }
        float o[16];
#pragma unroll
        for (int g4 = 0; g4 < 4; ++g4) {                    // channels 8 g4 + 4 hf + {0..3}
            const f32x4 sc4 = *reinterpret_cast<const f32x4*>(sstab + 8 * g4 + 4 * hf);
            const f32x4 sh4 = *reinterpret_cast<const f32x4*>(sstab + 32 + 8 * g4 + 4 * hf);
#pragma unroll
            for (int e = 0; e < 4; ++e) o[4 * g4 + e] = act_c<ACT>(acc[4 * g4 + e] * sc4[e] + sh4[e]);
        }
        unsigned short* dst = y + (size_t)t.p * y_ld + y_off + 8 * hf;
        bool bad = false;
#pragma unroll
        for (int kp = 0; kp < 2; ++kp) {
            float g[8];
#pragma unroll
            for (int e = 0; e < 4; ++e) {
                const auto sw = __builtin_amdgcn_permlane32_swap(__float_as_uint(o[8 * kp + e]), __float_as_uint(o[8 * kp + 4 + e]), false, false);
                g[e] = __uint_as_float(sw[0]);              // lanes 0-31: channels 16 kp + 0..7 | lanes 32-63: 16 kp + 8..15
                g[4 + e] = __uint_as_float(sw[1]);
            }
            u32x4 ov;
#pragma unroll
            for (int e = 0; e < 4; ++e) {
                bad |= __builtin_isunordered(g[2 * e], g[2 * e + 1]);
                ov[e] = pack2<T>(g[2 * e], g[2 * e + 1]);
            }
            if (t.live) *reinterpret_cast<u32x4*>(dst + 16 * kp) = ov;
        }
        bad_out |= bad && t.live;
    };
    Tile A, B;
    fetch(0, A);
#pragma unroll 1
    for (int it = 0; it < ST_TILES; it += 2) {              // tile it + 1 is requested before tile it is multiplied and stored
        fetch(it + 1, B);
        finish(A);
        fetch(it + 2, A);                                   // tile ST_TILES: fetched (a valid address), never finished
        finish(B);
    }
    if (bad_in) atomicOr(nan_flag, 1);                       // NaN in the INPUT tensor (model.py:175)
    if (bad_out) atomicOr(nan_flag, 2);
}

int stem_h16_launch(const float* x, const float* wt, const float* scale, const float* shift, void* y, int n, int h, int w, int y_ld,
                    int y_off, int act, int dtype, int* nan_flag, hipStream_t s) {
    const long long total = (long long)n * h * w;
    if (total * 12 >= (1ll << 32)) return fail(YOLO_ERR_UNSUPPORTED, "stem (16-bit): input of 4 GB or more");
    const long long waves = (total + 32 * ST_TILES - 1) / (32 * ST_TILES);
    const unsigned grid = (unsigned)((waves + 3) / 4);
    const unsigned mg_HW = magic_of(h * w), mg_W = magic_of(w);
#define YOLO_STEM_LAUNCH(T)                                                                                                       \
    YOLO_SWITCH_ACT(act, hipLaunchKernelGGL((stem3x3_mfma_h16<T, ACT>), dim3(grid), dim3(256), 0, s, x, wt, scale, shift,          \
                                            (unsigned short*)y, h, w, total, y_ld, y_off, mg_HW, mg_W, nan_flag))
    if (dtype == YOLO_BF16) { YOLO_STEM_LAUNCH(__bf16); } else { YOLO_STEM_LAUNCH(_Float16); }
#undef YOLO_STEM_LAUNCH
    return check_launch("stem3x3_mfma_h16");
}

}  // namespace yolo
