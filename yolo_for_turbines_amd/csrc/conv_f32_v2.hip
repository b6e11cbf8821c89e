// conv_f32_v2.hip — stride-1 3x3 / 1x1 fp32 convolution, "patch + fragment stream" structure.
//
// Same arithmetic and epilogues as conv_f32.hip (reference: CNNBlock.forward code/model.py:80-86,
// ResidualBlock skip :115-121, upsample+concat :189-191, head permute :145-148); different data
// movement, designed from the MI355X measurements of the v0 kernel (latency-bound at ~50-67 % of
// the f32 MFMA peak because every K step re-gathered activations from L2/HBM behind a barrier):
//
//  A (activations)  one block owns a spatial tile of TH x TW output pixels (TH*TW <= 128 MFMA rows;
//      rows are GLOBAL output rows n*H+h, so a tile may straddle images and no H-divisibility is
//      needed). Per 32-channel chunk it stages the (TH+2 [+2 per image crossed]) x (TW+2) input
//      patch — halo and zero padding included — in LDS ONCE and all 9 taps read it at shifted
//      pixel addresses: 9x fewer activation loads, one barrier per 9 K steps instead of one per step.
//  B (weights)      never touches LDS: yolo_pack_weights also emits a copy in MFMA-fragment order
//      [n_tile32][kstep][sub-step][lane][4], so a wave's B operand for 8 k-values x 32 couts is one
//      fully contiguous 1 KiB global_load_dwordx4 wave-instruction, prefetched one K step (64 MFMAs)
//      ahead into a second register set. Weights are L2/MALL resident (<= 18.9 MB per layer).
//  MFMA             v_mfma_f32_32x32x2_f32, wave tile 64 x (BN/2), 2x2 waves, block 128 x BN.
//  Occupancy        BN = 128: two patch buffers (64.5 KB LDS, 224 registers) -> 2 blocks per CU. BN = 64 can run with ONE
//      patch buffer (36.9 KB, 148 registers, one more barrier per chunk) -> 3 blocks per CU, which covers prologue /
//      epilogue better and fills 768 slots in one round at 13x13: the configuration measured fastest at 208 / 104 / 52 /
//      13 (conv_f32.hip:pick_tile).
#include "common.h"
#include <cstdlib>

namespace yolo {

typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef float f32x4 __attribute__((ext_vector_type(4)));

constexpr int V2_LD = 36;          // floats per patch pixel in LDS (32 + 4 pad)
constexpr int V2_NI = 8;           // patch pixels staged per 8-lane group -> patch <= 256 pixels
constexpr int V2_PATCH_CAP = 32 * V2_NI;

struct Conv2Args {
    const float* x;
    const float* wf;
    const float* scale;
    const float* shift;
    const float* res;
    float* y;
    int* nan_flag;
    int H, W, rows_total;      // tiling view of the output (1x1: H = 1, W = M, rows_total = 1)
    int Cin, Cout;
    int x_ld, x_off, y_ld, y_off, r_ld, r_off;
    int TH, TW, PC, patch_cap;  // tile rows/cols, patch columns, LDS pixels per buffer
    int bufmask;                // 1: two patch buffers (2 blocks/CU); 0: one buffer + an extra barrier per chunk (3 blocks/CU at BN = 64)
    int tiles_w, tiles_n, nblocks;
    int KT, nchunks;
    int act, out_mode, flags, nc5;
    int Ho, Wo;                 // real output dims (upsample / head addressing)
    int first_wave, stagger;    // blocks resident at launch; start delay (units of 64*127 cycles) per tg slot
    int prio;                   // prologue / epilogue at s_setprio 2 (A/B switch YOLO_F32_PRIO=0)
#ifdef V2_STAMPS
    unsigned long long* dbg;    // diagnostic build only (tools/v2_probe.hip): 4 s_memtime stamps per block
#endif
};


template <int KS, int TN>
struct V2Ctx {
    const float* wfrag[TN];     // per n-tile fragment stream base (+ lane*4)
    int a_off[2];               // LDS float offset of this lane's pixel for m-tile 0/1 (+4h)
    int pix[V2_NI];             // global pixel index of staged patch pixels (-1: zero)
    int KT;
};

// One K step = 32 channels of one tap = 4 sub-steps of 8 k-values (TAP is a compile-time constant:
// the caller unrolls a chunk's taps). `cur` holds this step's B fragments; the loads of the next
// step into `nxt` are issued first and have the whole step (32*TN MFMAs per wave) to land.
//
// Every global load in the loop is UNCONDITIONAL (indices clamped instead of branched around):
// hipcc's s_waitcnt insertion only counts loads it can prove were issued, so a prefetch under
// `if (kt + 1 < KT)` or a predicated halo load made it emit vmcnt(3) in front of the first MFMA of
// every step — i.e. wait for the loads just issued, a full L2 round trip (~600 cycles per 2048
// cycles of matrix work; measured 77 % pipe utilisation with one wave per SIMD).
template <int KS, int TN, int TAP>
__device__ __forceinline__ void v2_kstep(const Conv2Args& p, const V2Ctx<KS, TN>& c, int chunk, float* patch,
                                         f32x4 (&cur)[4][TN], f32x4 (&nxt)[4][TN], f32x4 (&stage)[V2_NI],
                                         f32x4 (&af)[2], f32x16 (&acc)[2][TN], int tid) {
    constexpr int TAPS = KS * KS;
    constexpr int PF_TAP = TAPS > 2 ? TAPS - 2 : 0;      // when to fetch the next chunk's patch
    const int kt = chunk * TAPS + TAP;
    const int ktn = kt + 1 < c.KT ? kt + 1 : c.KT - 1;    // clamped: the last step re-reads itself
#pragma unroll
    for (int s = 0; s < 4; ++s)
#pragma unroll
        for (int j = 0; j < TN; ++j)
            nxt[s][j] = *reinterpret_cast<const f32x4*>(c.wfrag[j] + ((size_t)ktn * 4 + s) * 256);
    if (TAP == PF_TAP) {
        const int cn = chunk + 1 < p.nchunks ? chunk + 1 : chunk;
        const int coff = p.x_off + cn * 32 + (tid & 7) * 4;
#pragma unroll
        for (int i = 0; i < V2_NI; ++i) {
            const int px = c.pix[i] < 0 ? 0 : c.pix[i];                     // always a valid address
            stage[i] = *reinterpret_cast<const f32x4*>(p.x + (size_t)px * p.x_ld + coff);
        }
    }
    // pin the prefetch loads HERE: in straight-line code the machine scheduler otherwise sinks them
    // next to their first use (lower register pressure) and the whole latency is exposed again
    __builtin_amdgcn_sched_barrier(0);
    constexpr int kh = TAP / KS, kw = TAP % KS;
    const float* Ab = patch + (chunk & p.bufmask) * (p.patch_cap * V2_LD) + (kh * p.PC + kw) * V2_LD;
    // A fragments are read from LDS one sub-step ahead of their MFMAs (`af` = sub-step 0 on entry)
    constexpr int nkh = (TAP + 1) / KS, nkw = (TAP + 1) % KS;
    const float* Ab_next = patch + (chunk & p.bufmask) * (p.patch_cap * V2_LD) + (nkh * p.PC + nkw) * V2_LD;
#pragma unroll
    for (int s = 0; s < 4; ++s) {
        f32x4 n0 = af[0], n1 = af[1];
        if (s < 3) {
            n0 = *reinterpret_cast<const f32x4*>(Ab + c.a_off[0] + (s + 1) * 8);
            n1 = *reinterpret_cast<const f32x4*>(Ab + c.a_off[1] + (s + 1) * 8);
        } else if (TAP + 1 < TAPS) {
            n0 = *reinterpret_cast<const f32x4*>(Ab_next + c.a_off[0]);
            n1 = *reinterpret_cast<const f32x4*>(Ab_next + c.a_off[1]);
        }
#pragma unroll
        for (int e = 0; e < 4; ++e)
#pragma unroll
            for (int j = 0; j < TN; ++j) {
                acc[0][j] = __builtin_amdgcn_mfma_f32_32x32x2f32(af[0][e], cur[s][j][e], acc[0][j], 0, 0, 0);
                acc[1][j] = __builtin_amdgcn_mfma_f32_32x32x2f32(af[1][e], cur[s][j][e], acc[1][j], 0, 0, 0);
            }
        af[0] = n0;
        af[1] = n1;
    }
    if (TAP == TAPS - 1) {                     // chunk boundary (the last chunk stores a dummy patch)
        if (p.bufmask == 0) __syncthreads();   // single buffer: every wave must be done reading this chunk's patch
        float* dst = patch + ((chunk + 1) & p.bufmask) * (p.patch_cap * V2_LD) + (tid >> 3) * V2_LD + (tid & 7) * 4;
#pragma unroll
        for (int i = 0; i < V2_NI; ++i) {
            f32x4 z = {0.f, 0.f, 0.f, 0.f};
            if ((tid >> 3) + 32 * i < p.patch_cap) *reinterpret_cast<f32x4*>(dst + 32 * i * V2_LD) = c.pix[i] < 0 ? z : stage[i];
        }
        __syncthreads();
        const float* An = patch + ((chunk + 1) & p.bufmask) * (p.patch_cap * V2_LD);       // tap 0 of the new chunk
        af[0] = *reinterpret_cast<const f32x4*>(An + c.a_off[0]);
        af[1] = *reinterpret_cast<const f32x4*>(An + c.a_off[1]);
    }
}

// all taps of one 32-channel chunk, B double buffer alternating (A = first step reads bA when FLIP = 0)
template <int KS, int TN, int TAP, bool FLIP>
__device__ __forceinline__ void v2_chunk(const Conv2Args& p, const V2Ctx<KS, TN>& c, int chunk, float* patch,
                                         f32x4 (&bA)[4][TN], f32x4 (&bB)[4][TN], f32x4 (&stage)[V2_NI],
                                         f32x4 (&af)[2], f32x16 (&acc)[2][TN], int tid) {
    if constexpr (TAP < KS * KS) {
        if constexpr (FLIP) v2_kstep<KS, TN, TAP>(p, c, chunk, patch, bB, bA, stage, af, acc, tid);
        else v2_kstep<KS, TN, TAP>(p, c, chunk, patch, bA, bB, stage, af, acc, tid);
        v2_chunk<KS, TN, TAP + 1, !FLIP>(p, c, chunk, patch, bA, bB, stage, af, acc, tid);
    }
}

template <int KS, int BN>
__global__ __launch_bounds__(256) void conv_patch_f32(const Conv2Args p) {
    constexpr int TN = BN / 64;          // 32-wide n tiles per wave (wave tile 64 x BN/2)
    extern __shared__ __attribute__((aligned(16))) char smem_raw[];
    float* patch = reinterpret_cast<float*>(smem_raw);      // [2][patch_cap][V2_LD]
    int* mtab = reinterpret_cast<int*>(patch + (p.bufmask + 1) * p.patch_cap * V2_LD);   // [128] MFMA row -> output pixel (or -1)

    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int wm = wave >> 1, wn = wave & 1;
    const int fh = lane >> 5, frow = lane & 31;
#ifdef V2_STAMPS
    const unsigned long long st0 = __builtin_amdgcn_s_memtime();
    const unsigned long long rt0 = __builtin_amdgcn_s_memrealtime();
#endif

    // Start stagger (speed only): the workgroups that share a CU are dispatched within ~100 cycles of
    // each other and, having identical work, stay in lock-step — their prologues and epilogues
    // coincide and the matrix pipe idles (measured: 2 x 73.7k MFMA cycles take 180k). Delaying the
    // second resident workgroup of the FIRST dispatch round by one block's MFMA time puts every
    // later epilogue/prologue under the neighbour's main loop; the offset then persists.
    if (p.stagger > 0 && (int)blockIdx.x < p.first_wave) {
        unsigned hw;
        asm volatile("s_getreg_b32 %0, hwreg(HW_REG_HW_ID)" : "=s"(hw));
        const int slot = (hw >> 16) & 15;                        // workgroup slot on this CU
        for (int i = 0; i < slot * p.stagger; ++i) __builtin_amdgcn_s_sleep(127);
    }

    // The co-resident blocks are usually in their main loops and need the issue port for a fraction of every MFMA's 64
    // cycles; this block's prologue / epilogue needs it continuously: take priority while there is no matrix work here.
    if (p.prio) __builtin_amdgcn_s_setprio(2);
    // XCD-aware, bijective block remap: blocks sharing a spatial tile (n tiles) and neighbouring
    // tiles land on the same XCD / L2 (speed only; any placement is correct).
    int bid = blockIdx.x;
    {
        const int nb = p.nblocks, q = nb / 8, r = nb % 8, xcd = bid % 8;
        bid = (xcd < r ? xcd * (q + 1) : r * (q + 1) + (xcd - r) * q) + bid / 8;
    }
    const int n_tile = bid % p.tiles_n;
    const int sp = bid / p.tiles_n;
    const int w_tile = sp % p.tiles_w;
    const int r_tile = sp / p.tiles_w;
    const int g0 = r_tile * p.TH, c0 = w_tile * p.TW;
    const int g_last = (g0 + p.TH < p.rows_total ? g0 + p.TH : p.rows_total) - 1;
    auto vrow = [&](int g) { return KS == 3 ? g + 2 * (g / p.H) : g; };
    const int v0 = vrow(g0);
    const int PR = vrow(g_last) + (KS == 3 ? 3 : 1) - v0;

    V2Ctx<KS, TN> c;
    c.KT = p.KT;
    // ---- staged patch pixels of this 8-lane group: idx = (tid >> 3) + 32 i, decoded incrementally
    //      (one integer division per thread instead of two per pixel)
    {
        const int d_pr = 32 / p.PC, d_pc = 32 - d_pr * p.PC;          // wave-uniform
        int pr = (tid >> 3) / p.PC, pc = (tid >> 3) - pr * p.PC;
        const int Hp = p.H + 2;
        const int n0i = KS == 3 ? v0 / Hp : 0;
#pragma unroll
        for (int i = 0; i < V2_NI; ++i) {
            int pix = -1;
            if (pr < PR) {
                if (KS == 3) {
                    int n = n0i, yy = v0 + pr - n0i * Hp;
                    while (yy >= Hp) { yy -= Hp; ++n; }                 // at most a few image crossings
                    const int hi = yy - 1, wi = c0 + pc - 1;
                    if ((unsigned)hi < (unsigned)p.H && (unsigned)wi < (unsigned)p.W) pix = (n * p.H + hi) * p.W + wi;
                } else {
                    const int wi = c0 + pc;
                    if (wi < p.W) pix = wi;
                }
            }
            c.pix[i] = pix;
            pr += d_pr; pc += d_pc;
            if (pc >= p.PC) { pc -= p.PC; ++pr; }
        }
    }
    // ---- MFMA A-fragment addresses (lane -> output pixel -> patch pixel)
#pragma unroll
    for (int i = 0; i < 2; ++i) {
        const int pp = wm * 64 + i * 32 + frow;
        const int r = pp / p.TW, cc = pp - r * p.TW;
        const int g = g0 + r;
        const bool ok = pp < p.TH * p.TW && g <= g_last && c0 + cc < p.W;
        c.a_off[i] = (ok ? ((vrow(g) - v0) * p.PC + cc) * V2_LD : 0) + 4 * fh;
    }
    if (tid < 128) {
        const int r = tid / p.TW, cc = tid - r * p.TW;
        const int g = g0 + r;
        mtab[tid] = (tid < p.TH * p.TW && g <= g_last && c0 + cc < p.W) ? g * p.W + c0 + cc : -1;
    }
#pragma unroll
    for (int j = 0; j < TN; ++j) {
        const int nt = n_tile * (BN / 32) + wn * TN + j;
        c.wfrag[j] = p.wf + (size_t)nt * p.KT * 1024 + lane * 4;
    }

    // folded BatchNorm scale / shift of this lane's output channel of pass j: requested here, with the first loads of the
    // block (in the epilogue they were an exposed round trip per pass, in front of which every earlier store had to drain)
    float sc[TN], sh[TN];
#pragma unroll
    for (int j = 0; j < TN; ++j) {
        const int n = n_tile * BN + wn * (BN / 2) + j * 32 + frow;
        const int ncl = n < p.Cout ? n : p.Cout - 1;           // clamped: unconditional loads
        sc[j] = p.scale[ncl];
        sh[j] = p.shift[ncl];
        if (n >= p.Cout) { sc[j] = 0.f; sh[j] = 0.f; }
    }

    f32x16 acc[2][TN];
#pragma unroll
    for (int i = 0; i < 2; ++i)
#pragma unroll
        for (int j = 0; j < TN; ++j)
#pragma unroll
            for (int r = 0; r < 16; ++r) acc[i][j][r] = 0.f;

    f32x4 bA[4][TN], bB[4][TN], stage[V2_NI];
    // prologue: B fragments of step 0, patch of chunk 0
#pragma unroll
    for (int s = 0; s < 4; ++s)
#pragma unroll
        for (int j = 0; j < TN; ++j) bA[s][j] = *reinterpret_cast<const f32x4*>(c.wfrag[j] + (size_t)s * 256);
    {
        const int coff = p.x_off + (tid & 7) * 4;
        float* dst = patch + (tid >> 3) * V2_LD + (tid & 7) * 4;
#pragma unroll
        for (int i = 0; i < V2_NI; ++i) {
            f32x4 z = {0.f, 0.f, 0.f, 0.f};
            stage[i] = c.pix[i] >= 0 ? *reinterpret_cast<const f32x4*>(p.x + (size_t)c.pix[i] * p.x_ld + coff) : z;
        }
#pragma unroll
        for (int i = 0; i < V2_NI; ++i)
            if ((tid >> 3) + 32 * i < p.patch_cap) *reinterpret_cast<f32x4*>(dst + 32 * i * V2_LD) = stage[i];
    }
    __syncthreads();
    f32x4 af[2];
    af[0] = *reinterpret_cast<const f32x4*>(patch + c.a_off[0]);
    af[1] = *reinterpret_cast<const f32x4*>(patch + c.a_off[1]);
#pragma unroll
    for (int j = 0; j < TN; ++j) asm volatile("" : "+v"(sc[j]), "+v"(sh[j]));   // pinned here: not re-loaded in the epilogue

#ifdef V2_STAMPS
    const unsigned long long st1 = __builtin_amdgcn_s_memtime();
#endif
    // chunks in pairs so the B double buffer keeps compile-time names (9 taps per chunk is odd)
    __builtin_amdgcn_s_setprio(0);
    for (int chunk = 0; chunk < p.nchunks; chunk += 2) {
        v2_chunk<KS, TN, 0, false>(p, c, chunk, patch, bA, bB, stage, af, acc, tid);
        if (chunk + 1 < p.nchunks)
            v2_chunk<KS, TN, 0, (KS * KS) % 2 == 1>(p, c, chunk + 1, patch, bA, bB, stage, af, acc, tid);
    }
    if (p.prio) __builtin_amdgcn_s_setprio(2);

    // ---------------------------------------------------------------------- epilogue
#ifdef V2_STAMPS
    asm volatile("s_nop 15\n\ts_nop 15" ::: "memory");
    const unsigned long long st2 = __builtin_amdgcn_s_memtime();
#endif
    // The 32x32 accumulator layout puts one cout per lane and pixels across registers, so direct
    // stores are 4-byte pieces (measured: ~35k cycles per block, 18 % of its life). Instead: apply
    // scale/shift/activation in registers, transpose the 128 x 64 tile through the (now idle) patch
    // LDS, and let every lane move 16 contiguous bytes of one pixel row: residual loads and stores
    // become full-line float4 accesses.
    const bool has_res = p.flags & YOLO_FLAG_RESIDUAL;
    const bool nan_chk = p.flags & YOLO_FLAG_NANCHECK;
    const int HoWo = p.Ho * p.Wo;
    constexpr int OLD = 68;                         // staging row stride (floats): 64 + 4 pad
    float* ost = patch;                             // [128][OLD] = 34,816 B <= patch region
    const bool vec_ok = (p.out_mode != YOLO_OUT_HEAD) && (p.Cout % 4 == 0);
    bool saw_nan = false;
    __syncthreads();                                // every wave is done reading the patch
    // residual rows of EVERY pass: requested before the accumulators go through LDS (they were loaded one by one, each
    // behind an s_waitcnt vmcnt(0) that also waited for the stores in front of it: 8 exposed round trips per pass), and
    // awaited before the first store is issued (one in-order counter for loads and stores)
    f32x4 rr[TN][8];
    int mrow[8];
    if (vec_ok) {
        const int c4 = tid & 15;
#pragma unroll
        for (int it = 0; it < 8; ++it) mrow[it] = mtab[(tid >> 4) + 16 * it];
#pragma unroll
        for (int j = 0; j < TN; ++j)
#pragma unroll
            for (int it = 0; it < 8; ++it) {
                const f32x4 z = {0.f, 0.f, 0.f, 0.f};
                rr[j][it] = z;
            }
        if (has_res) {
#pragma unroll
            for (int j = 0; j < TN; ++j) {
                const int n = n_tile * BN + (c4 >> 3) * (BN / 2) + j * 32 + (c4 & 7) * 4;
#pragma unroll
                for (int it = 0; it < 8; ++it) {
                    const int mc = mrow[it] < 0 ? 0 : mrow[it];
                    rr[j][it] = *reinterpret_cast<const f32x4*>(p.res + (size_t)mc * p.r_ld + p.r_off + (n < p.Cout ? n : 0));   // clamped, discarded below
                }
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
            f32x4 va[8];
#pragma unroll
            for (int it = 0; it < 8; ++it)                              // all LDS reads first: one latency, not eight
                va[it] = *reinterpret_cast<const f32x4*>(ost + ((tid >> 4) + 16 * it) * OLD + (tid & 15) * 4);
            if (has_res && j == 0) {
#pragma unroll
                for (int jj = 0; jj < TN; ++jj)
#pragma unroll
                    for (int it = 0; it < 8; ++it) asm volatile("" : "+v"(rr[jj][it]));
            }
#pragma unroll
            for (int it = 0; it < 8; ++it) {
                const int c4 = tid & 15;
                const int m = mrow[it];
                const int n = n_tile * BN + (c4 >> 3) * (BN / 2) + j * 32 + (c4 & 7) * 4;
                if (m < 0 || n >= p.Cout) continue;
                f32x4 v = va[it];
                if (has_res) v += rr[j][it];
                if (nan_chk && (v[0] != v[0] || v[1] != v[1] || v[2] != v[2] || v[3] != v[3])) saw_nan = true;
                if (p.out_mode == YOLO_OUT_NHWC) {
                    *reinterpret_cast<f32x4*>(p.y + (size_t)m * p.y_ld + p.y_off + n) = v;
                } else {                            // YOLO_OUT_UPSAMPLE2X
                    const int img = m / HoWo;
                    const int rem = m - img * HoWo;
                    const int ho = rem / p.Wo;
                    const int wo2 = rem - ho * p.Wo;
                    const int W2 = 2 * p.Wo;
                    float* d = p.y + ((size_t)(img * 2 * p.Ho + 2 * ho) * W2 + 2 * wo2) * p.y_ld + p.y_off + n;
                    *reinterpret_cast<f32x4*>(d) = v;
                    *reinterpret_cast<f32x4*>(d + p.y_ld) = v;
                    *reinterpret_cast<f32x4*>(d + (size_t)W2 * p.y_ld) = v;
                    *reinterpret_cast<f32x4*>(d + (size_t)(W2 + 1) * p.y_ld) = v;
                }
            }
        } else {                                    // head layout / odd channel counts: scalar, cout-contiguous
            for (int it = 0; it < 32; ++it) {
                const int idx = tid + 256 * it;
                const int row = idx >> 6, col = idx & 63;
                const int m = mtab[row];
                const int n = n_tile * BN + (col >> 5) * (BN / 2) + j * 32 + (col & 31);
                if (m < 0 || n >= p.Cout) continue;
                float v = ost[row * OLD + col];
                if (has_res) v += p.res[(size_t)m * p.r_ld + p.r_off + n];
                if (nan_chk && v != v) saw_nan = true;
                if (p.out_mode == YOLO_OUT_NHWC) {
                    p.y[(size_t)m * p.y_ld + p.y_off + n] = v;
                } else {
                    const int img = m / HoWo;
                    const int rem = m - img * HoWo;
                    const int ho = rem / p.Wo;
                    const int wo2 = rem - ho * p.Wo;
                    if (p.out_mode == YOLO_OUT_UPSAMPLE2X) {
                        const int W2 = 2 * p.Wo;
                        float* d = p.y + ((size_t)(img * 2 * p.Ho + 2 * ho) * W2 + 2 * wo2) * p.y_ld + p.y_off + n;
                        d[0] = v;
                        d[p.y_ld] = v;
                        d[(size_t)W2 * p.y_ld] = v;
                        d[(size_t)(W2 + 1) * p.y_ld] = v;
                    } else {
                        const int head_a = n / p.nc5, head_k = n - head_a * p.nc5;
                        p.y[((size_t)((img * 3 + head_a) * p.Ho + ho) * p.Wo + wo2) * p.nc5 + head_k] = v;
                    }
                }
            }
        }
        if (j + 1 < TN) __syncthreads();
    }
    if (nan_chk && saw_nan) atomicOr(p.nan_flag, 2);
#ifdef V2_STAMPS
    if (p.dbg && tid == 0) {
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        const unsigned long long st3 = __builtin_amdgcn_s_memtime();
        unsigned hwid;
        asm volatile("s_getreg_b32 %0, hwreg(HW_REG_HW_ID)" : "=s"(hwid));
        unsigned xcc;
        asm volatile("s_getreg_b32 %0, hwreg(HW_REG_XCC_ID)" : "=s"(xcc));
        unsigned long long* d = p.dbg + (size_t)blockIdx.x * 6;
        const unsigned long long rt1 = __builtin_amdgcn_s_memrealtime();
        d[0] = st0; d[1] = st1; d[2] = st2; d[3] = st3; d[4] = hwid; d[5] = xcc | ((rt1 - rt0) << 8);
    }
#endif
}

// fragment-order weights: [n_tile32][kt][s][lane][e] ; n = nt*32 + (lane&31),
// ci = chunk*32 + s*8 + 4*(lane>>5) + e, (chunk, tap) = divmod(kt, ks*ks)
__global__ void pack_weights_frag_f32(const float* __restrict__ w, float* __restrict__ wf, int cout, int cin, int ks,
                                      int KT, long long total) {
    const int taps = ks * ks;
    for (long long i = blockIdx.x * (long long)blockDim.x + threadIdx.x; i < total; i += (long long)gridDim.x * blockDim.x) {
        const int e = (int)(i & 3);
        const int lane = (int)((i >> 2) & 63);
        const int s = (int)((i >> 8) & 3);
        const long long rest = i >> 10;
        const int kt = (int)(rest % KT);
        const int nt = (int)(rest / KT);
        const int n = nt * 32 + (lane & 31);
        const int chunk = kt / taps, tap = kt - chunk * taps;
        const int ci = chunk * 32 + s * 8 + 4 * (lane >> 5) + e;
        wf[i] = n < cout ? w[((size_t)n * cin + ci) * taps + tap] : 0.f;
    }
}

// ------------------------------------------------------------------------------ host side
#ifdef V2_STAMPS
unsigned long long* g_v2_dbg = nullptr;
#endif
static const bool g_v2_stagger = !(getenv("YOLO_NO_STAGGER"));
static const bool g_v2_prio = !(getenv("YOLO_F32_PRIO") && getenv("YOLO_F32_PRIO")[0] == '0');
bool v2_eligible(const yolo_conv_desc* d) { return d->stride == 1 && d->cin % 32 == 0; }

size_t v2_frag_elems(int cout, int cin, int ks) {
    if (cin % 32) return 0;
    return (size_t)(round_up(cout, 128) / 32) * (cin / 32) * ks * ks * 1024;
}

int v2_pack(const float* w_oihw, float* wf, int cout, int cin, int ks, hipStream_t s) {
    const long long total = (long long)v2_frag_elems(cout, cin, ks);
    const int KT = (cin / 32) * ks * ks;
    const int grid = (int)((total + 255) / 256 < 8192 ? (total + 255) / 256 : 8192);
    hipLaunchKernelGGL(pack_weights_frag_f32, dim3(grid), dim3(256), 0, s, w_oihw, wf, cout, cin, ks, KT, total);
    return check_launch("pack_weights_frag");
}

// choose TH x TW (<= 128 pixels, patch <= V2_PATCH_CAP) maximising useful MFMA rows
static void pick_patch_tile(int H, int W, int ks, int* th, int* tw, int* prmax) {
    if (ks == 1) { *th = 1; *tw = 128; *prmax = 1; return; }
    double best = -1;
    *th = 1; *tw = 1; *prmax = 5;
    for (int TW = 1; TW <= (W < 126 ? W : 126); ++TW) {
        int TH = 128 / TW;
        while (TH >= 1) {
            const int cross = (TH + H - 1) / H;             // images a tile can straddle (upper bound)
            const int pr = TH + 2 + 2 * cross;
            if (pr * (TW + 2) <= V2_PATCH_CAP) break;
            --TH;
        }
        if (TH < 1) continue;
        const double eff = ((double)W / (ceil_div(W, TW) * TW)) * (TH * TW / 128.0);
        if (eff > best + 1e-9) {
            best = eff; *th = TH; *tw = TW;
            *prmax = TH + 2 + 2 * ((TH + H - 1) / H);
        }
    }
}

int v2_blocks(const yolo_conv_desc* d, int bn) {
    if (d->ksize == 1) return ceil_div(d->n * d->h * d->w, 128) * ceil_div(d->cout, bn);
    int th = 1, tw = 1, pr = 1;
    pick_patch_tile(d->h, d->w, 3, &th, &tw, &pr);
    return ceil_div(d->n * d->h, th) * ceil_div(d->w, tw) * ceil_div(d->cout, bn);
}

template <int KS, int BN>
static int launch_v2(Conv2Args& a, hipStream_t s) {
    a.tiles_n = ceil_div(a.Cout, BN);
    const int tiles_r = ceil_div(a.rows_total, a.TH);
    a.nblocks = a.tiles_n * a.tiles_w * tiles_r;
    // resident blocks per CU (2, or 3 with a single patch buffer) x 256 CUs are dispatched at once
    a.first_wave = (a.bufmask ? 2 : 3) * 256;
    const long mfma_cycles = (long)a.KT * 32 * (BN / 64) * 64;  // one block's matrix work per wave
    a.stagger = g_v2_stagger ? (int)((mfma_cycles + 64 * 127 / 2) / (64 * 127)) : 0;
    a.prio = g_v2_prio ? 1 : 0;
    const size_t lds = (size_t)(a.bufmask + 1) * a.patch_cap * V2_LD * sizeof(float) + 128 * sizeof(int);   // >= 128*68*4 staging
    hipLaunchKernelGGL((conv_patch_f32<KS, BN>), dim3(a.nblocks), dim3(256), lds, s, a);
    return check_launch("conv_patch_f32");
}

int conv_v2_launch(const yolo_conv_desc* d, const void* x, const float* wf, const float* scale, const float* shift,
                   const void* residual, void* y, int32_t* nan_flag, int bn, bool single_buffer, hipStream_t s) {
    Conv2Args a;
    a.x = (const float*)x; a.wf = wf; a.scale = scale; a.shift = shift; a.res = (const float*)residual;
    a.y = (float*)y; a.nan_flag = nan_flag;
    a.Cin = d->cin; a.Cout = d->cout;
    a.x_ld = d->x_ld; a.x_off = d->x_off; a.y_ld = d->y_ld; a.y_off = d->y_off; a.r_ld = d->r_ld; a.r_off = d->r_off;
    a.Ho = d->h; a.Wo = d->w;
    const long long M = (long long)d->n * d->h * d->w;
    if (M > 0x7fffffffLL) return fail(YOLO_ERR_UNSUPPORTED, "conv: N*H*W exceeds int32");
    int prmax = 1;
    if (d->ksize == 1) {
        a.H = 1; a.W = (int)M; a.rows_total = 1;
        a.TH = 1; a.TW = 128; a.PC = 128;
    } else {
        a.H = d->h; a.W = d->w; a.rows_total = d->n * d->h;
        pick_patch_tile(d->h, d->w, 3, &a.TH, &a.TW, &prmax);
        a.PC = a.TW + 2;
    }
    a.bufmask = single_buffer ? 0 : 1;
    a.patch_cap = round_up(prmax * a.PC, 32);
    if (a.patch_cap < 128) a.patch_cap = 128;              // the epilogue stages a 128 x 68 float tile in the patch region
    if (single_buffer) a.patch_cap = V2_PATCH_CAP;         // ... which then needs the whole (single) buffer: 256 px x 144 B = 36.9 KB
    if (a.patch_cap > V2_PATCH_CAP) return fail(YOLO_ERR_UNSUPPORTED, "conv v2: patch too large");
    a.tiles_w = ceil_div(a.W, a.TW);
    a.nchunks = d->cin / 32;
    a.KT = a.nchunks * d->ksize * d->ksize;
    a.act = d->act; a.out_mode = d->out_mode; a.flags = d->flags;
    a.nc5 = d->out_mode == YOLO_OUT_HEAD ? d->cout / 3 : 1;
#ifdef V2_STAMPS
    a.dbg = g_v2_dbg;
#endif
    if (d->ksize == 3) return bn == 128 ? launch_v2<3, 128>(a, s) : launch_v2<3, 64>(a, s);
    return bn == 128 ? launch_v2<1, 128>(a, s) : launch_v2<1, 64>(a, s);
}

}  // namespace yolo
