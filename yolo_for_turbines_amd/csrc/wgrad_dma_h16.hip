// wgrad_dma_h16.hip — weight gradient of the 3x3 stride-1 blocks, bf16 / f16 operands, round-3 design.
//
// Replaces the autograd backward of nn.Conv2d w.r.t. its weight (reference: code/train.py:67 under the autocast of
// train.py:53; conv definition code/model.py:60):
//     dW[co][ci][kh][kw] = sum_{n,ho,wo} dz[n,ho,wo,co] * x[n, ho+kh-1, wo+kw-1, ci]
// GEMM view: M = Cout, N = Cin per tap, K = output pixels (86k .. 1.4M at batch 32): a tiny output under a huge reduction.
//
// What round 2's kernel (wgrad_h16.hip: register-staged tiles, 2 x 256-thread blocks per CU) paid for, per 51-GFLOP layer:
// 81.5 us of kernel for ~29 us of matrix time at the loaded clock, and 75 MB of fp32 split-K partials (512 blocks x 147 KB)
// written at the end of the launch and read back by the reduce kernel (18.8 us) - traffic the algorithm does not have.
// The partial bytes of ANY split-K scheme are (accumulators resident on the chip) x 4 bytes, so the only way down is
// fewer resident accumulators per unit of matrix throughput:
//   * ONE 512-thread workgroup per CU = two K-GROUPS of four waves. Both groups own the same 64(co) x 64(ci) x 9-tap tile
//     of dW (a wave: 32 x 32 x 9 taps = 144 fp32 accumulators) and walk disjoint halves of the workgroup's pixel range;
//     at the end group 1 hands its accumulators to group 0 through LDS and ONE partial per CU leaves the chip:
//     256 x 147 KB = 37.7 MB per layer, and the reduce kernel reads half as much.
//   * Both operands arrive by LDS-DMA (global_load_lds_dwordx4) into a 3-deep ring per group: no staging registers, no
//     register-destination load in the loop, one counted s_waitcnt vmcnt + one raw s_barrier per K tile of 64 pixels
//     (36 MFMAs per wave). An LDS-DMA image is lane-linear, so the rows cannot be padded; the bank spread the transposing
//     reads need (4 pixel rows x 64 B per 32 lanes -> 4 disjoint 16-bank windows) comes from a swizzle instead: the 16-byte
//     chunk c of pixel row r lives at slot c ^ 4*bit1(r), applied to the per-lane SOURCE address of the DMA and again in
//     the fragment read (rows r and r+2 share a bank half, and always differ in bit 1).
//   * K tile = 4 x 16 output pixels; the x patch with halo (6 x 18 pixels) is staged once and each patch row's fragments
//     are read ONCE for the up to three (tile row, kh) pairs that use it: 44 transposing reads per 36 MFMAs
//     (round 2: 80).
//   * Partials stay in ACCUMULATOR order ([slice][tile][tap][wave][4 regs][lane][4]): 16-byte stores, 1 KiB contiguous per
//     wave-instruction, and the reduce kernel (wgrad_reduce_acc) adds the slices elementwise in a fixed order and only
//     then scatters the 1.2 MB result to OIHW - deterministic, no float atomics.
// Halo / out-of-range pixels and channel chunks read a zero page instead of being masked.
#include <type_traits>
#include "common.h"

namespace yolo {

typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef unsigned int u32x4 __attribute__((ext_vector_type(4)));
typedef short s16x4 __attribute__((ext_vector_type(4)));
typedef short s16x8 __attribute__((ext_vector_type(8)));
typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));
typedef _Float16 f16x8 __attribute__((ext_vector_type(8)));

constexpr int WD_TH = 4, WD_TW = 16;                    // K tile: 4 x 16 output pixels = 4 k16 steps
constexpr int WD_PC = WD_TW + 2, WD_PR = WD_TH + 2;     // x patch with halo: 6 x 18 pixels
constexpr int WD_GROWS = WD_TH * WD_TW;                 // 64 dz rows
constexpr int WD_XROWS = 128;                           // 108 patch rows, padded to 16 DMA instructions of 8 rows
constexpr int WD_ROWB = 128;                            // bytes per LDS row: 64 channels
constexpr int WD_BUF = (WD_GROWS + WD_XROWS) * WD_ROWB; // 24 KiB per ring slot
constexpr int WD_SLOTS = 3;
constexpr int WD_LDS = 2 * WD_SLOTS * WD_BUF;           // 147,456 B: also exactly one group's accumulators (9 x 4 x 4 KiB)
constexpr int WD_TILE_FLOATS = 9 * 64 * 64;             // one (slice, tile) partial
static_assert(WD_LDS == WD_TILE_FLOATS * 4, "the hand-over reuses the staging ring");

struct WgradDArgs {
    const unsigned short* dz;
    const unsigned short* x;
    float* partial;
    int N, H, W;
    int cin8, cout8;                 // channels rounded up to 8: valid 16-byte pieces of a row
    int dz_ld, dz_off, x_ld, x_off;
    int tiles_n, ntile;              // ci tiles; co tiles * ci tiles
    int th_tiles, tw_tiles, total_tiles, tiles_per_slice, nslices;
    int prio;
};

__device__ __attribute__((aligned(256))) unsigned int g_wd_zero[64];     // 256 B of zeros (one row of any slot order)

typedef const __attribute__((address_space(1))) void* wd_gptr;
typedef __attribute__((address_space(3))) void* wd_lptr;
__device__ __forceinline__ void wd_glds16(const void* g, void* l) { __builtin_amdgcn_global_load_lds((wd_gptr)g, (wd_lptr)l, 16, 0, 0); }
template <int N> __device__ __forceinline__ void wd_wait_vmcnt() { asm volatile("s_waitcnt vmcnt(%0)" ::"n"(N) : "memory"); }

template <typename T> struct WDTraits;
template <> struct WDTraits<__bf16> {
    static __device__ __forceinline__ f32x16 mfma(s16x8 a, s16x8 b, f32x16 c) {
        return __builtin_amdgcn_mfma_f32_32x32x16_bf16(__builtin_bit_cast(bf16x8, a), __builtin_bit_cast(bf16x8, b), c, 0, 0, 0);
    }
};
template <> struct WDTraits<_Float16> {
    static __device__ __forceinline__ f32x16 mfma(s16x8 a, s16x8 b, f32x16 c) {
        return __builtin_amdgcn_mfma_f32_32x32x16_f16(__builtin_bit_cast(f16x8, a), __builtin_bit_cast(f16x8, b), c, 0, 0, 0);
    }
};

// ---- transposing LDS reads in inline asm ------------------------------------------------------------------------------------
// ds_read_b64_tr_b16: 4 pixel rows x 16 channel columns of a [pixel][channel] LDS image, channel-major into the lane
// (wgrad_h16.hip, pinned by test_transposing_lds_read_semantics); two of them = the 8 consecutive k of one 32x32x16 operand.
// Why asm: behind a pending LDS-DMA hipcc puts `s_waitcnt vmcnt(0)` in front of the BUILTIN form of this read (its memory
// operand may alias the DMA's LDS write; a plain ds_read does not get that wait), which drains the 3-deep ring every K tile.
// An asm read is invisible to hipcc's counters, so its completion is counted here (cdna_hip_programming.md 5.7, form (ii)):
// LDS operations return in order, every step waits with `s_waitcnt lgkmcnt(N)`, N = the reads issued after the ones it needs,
// in a statement that names the fragment registers "+v" (no consumer can be scheduled above it).
typedef unsigned int u32x2 __attribute__((ext_vector_type(2)));
struct WFrag { u32x2 lo, hi; };
template <int OFF>
__device__ __forceinline__ void wd_read(WFrag& f, unsigned addr) {
    asm volatile("ds_read_b64_tr_b16 %0, %2 offset:%3\n\tds_read_b64_tr_b16 %1, %2 offset:%4"
                 : "=&v"(f.lo), "=&v"(f.hi) : "v"(addr), "n"(OFF), "n"(OFF + 4 * WD_ROWB));   // early clobber: the first read's
    // data may land in its destination before the second read has been issued, so neither destination may share the address register
}
template <int N> __device__ __forceinline__ void wd_wait1(WFrag& a) {
    asm volatile("s_waitcnt lgkmcnt(%2)" : "+v"(a.lo), "+v"(a.hi) : "n"(N));
}
template <int N> __device__ __forceinline__ void wd_wait2(WFrag& a, WFrag& b) {
    asm volatile("s_waitcnt lgkmcnt(%4)" : "+v"(a.lo), "+v"(a.hi), "+v"(b.lo), "+v"(b.hi) : "n"(N));
}
__device__ __forceinline__ s16x8 wd_vec(const WFrag& f) {
    const u32x4 v = {f.lo[0], f.lo[1], f.hi[0], f.hi[1]};
    return __builtin_bit_cast(s16x8, v);
}

template <int I, int N, class Fn>
__device__ __forceinline__ void wd_for(Fn&& fn) {
    if constexpr (I < N) {
        fn(std::integral_constant<int, I>{});
        wd_for<I + 1, N>(fn);
    }
}

// Fragment schedule of one K tile. The 18 x-fragments (patch row r, kw) are visited in the row order 0, 5, 1, 4, 2, 3: the rows
// that feed three MFMAs each come last, and the MFMAs of the last two fragments are DEFERRED behind the tile's barrier, where
// they cover the latency of the next tile's first reads. Fragment f+LA is requested in step f; the dz fragments a0, a3, a1, a2
// (first needed in steps 0, 3, 6, 9) are requested in the tile's prologue, interleaved with the first LA x-fragments:
//   a0 B0 a3 B1 a1 B2 a2 [B3 .. B(LA-1)].
__device__ constexpr int wd_row(int f) { constexpr int ord[6] = {0, 5, 1, 4, 2, 3}; return ord[f / 3]; }
__device__ constexpr int wd_boff(int f) { return (wd_row(f) * WD_PC + f % 3) * WD_ROWB; }       // + b_off[C & 3]
__device__ constexpr int wd_bvar(int f) { return (wd_row(f) * WD_PC + f % 3) & 3; }
// reads (2 per fragment) issued after B[f] once step f has made its own request: what `s_waitcnt lgkmcnt` may leave in flight
__device__ constexpr int wd_wait_of(int f, int LA) {
    const int bs = 17 - f < LA ? 17 - f : LA;              // x fragments f+1 .. f+LA
    const int as = f == 0 ? 3 : f == 1 ? 2 : f == 2 ? 1 : 0;   // dz fragments that sit behind B[f] in the prologue order
    const int n = 2 * (bs + as);
    return n > 15 ? 15 : n;                                // the counter field is 4 bits: waiting for more is always safe
}

template <typename T, int LA>
__global__ __launch_bounds__(512) void wgrad3_dma_h16(const WgradDArgs p) {
    constexpr int NB = LA + 1;                              // x fragments alive at once
    extern __shared__ __attribute__((aligned(16))) char smem[];
    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int kg = wave >> 2, w4 = wave & 3, wm = w4 >> 1, wn = w4 & 1;

    // ---- workgroup -> (dW tile, K slice); the tiles of one pixel range share an XCD (and its L2) when possible
    int slice, tile;
    if ((p.nslices & 7) == 0) {
        const int xcd = blockIdx.x & 7, j = blockIdx.x >> 3;
        slice = (j / p.ntile) * 8 + xcd;
        tile = j % p.ntile;
    } else {
        slice = blockIdx.x / p.ntile;
        tile = blockIdx.x % p.ntile;
    }
    const int co0 = (tile / p.tiles_n) * 64, ci0 = (tile % p.tiles_n) * 64;
    const int t0 = slice * p.tiles_per_slice;
    const int t1 = t0 + p.tiles_per_slice < p.total_tiles ? t0 + p.tiles_per_slice : p.total_tiles;
    const int niter = t1 > t0 ? (t1 - t0 + 1) >> 1 : 0;             // both groups run the same number of rounds (shared barriers)

    // ---- DMA roles: per K tile a wave moves 2 x 8 dz rows and 4 x 8 patch rows of its group (6 wave-instructions of 1 KiB)
    const int drow = lane >> 3, slot = lane & 7;
    int d_r[6], d_c[6], d_ch[6];                                    // 0-1 dz: tile row / column; 2-5 x: patch row / column; source chunk
#pragma unroll
    for (int j = 0; j < 2; ++j) {
        const int row = 8 * (2 * w4 + j) + drow;
        d_r[j] = row >> 4;
        d_c[j] = row & 15;
        d_ch[j] = (slot ^ (((row >> 1) & 1) << 2)) * 8;
    }
#pragma unroll
    for (int j = 0; j < 4; ++j) {
        const int row = 8 * (4 * w4 + j) + drow;
        d_r[2 + j] = row < WD_PR * WD_PC ? row / WD_PC - 1 : -(1 << 20);   // padding rows of the 16th instruction: never inside the image
        d_c[2 + j] = row % WD_PC - 1;
        d_ch[2 + j] = (slot ^ (((row >> 1) & 1) << 2)) * 8;
    }
    const unsigned short* zp = reinterpret_cast<const unsigned short*>(g_wd_zero) + slot * 8;
    char* const ring = smem + kg * (WD_SLOTS * WD_BUF);

    struct TileAt { int n, ho0, wo0; bool tv; };
    auto locate = [&](int it) {
        TileAt a;
        const int t = t0 + 2 * it + kg;
        a.tv = t < t1;
        const int tw = t % p.tw_tiles, rr = t / p.tw_tiles;
        const int th = rr % p.th_tiles;
        a.n = rr / p.th_tiles; a.ho0 = th * WD_TH; a.wo0 = tw * WD_TW;
        return a;
    };
    // piece j (0-1: dz rows, 2-5: patch rows) of the tile at `a` into ring slot `buf`
    auto issue1 = [&](auto J, const TileAt& a, int buf) {
        constexpr int j = decltype(J)::value;
        if constexpr (j < 2) {
            const int ho = a.ho0 + d_r[j], wo = a.wo0 + d_c[j], co = co0 + d_ch[j];
            const bool v = a.tv & (ho < p.H) & (wo < p.W) & (co < p.cout8);
            const size_t pix = (size_t)((a.n * p.H + ho) * p.W + wo);
            const unsigned short* src = v ? p.dz + pix * p.dz_ld + p.dz_off + co : zp;
            wd_glds16(src, ring + buf * WD_BUF + (2 * w4 + j) * 1024);
        } else {
            const int hi = a.ho0 + d_r[j], wi = a.wo0 + d_c[j], ci = ci0 + d_ch[j];
            const bool v = a.tv & ((unsigned)hi < (unsigned)p.H) & ((unsigned)wi < (unsigned)p.W) & (ci < p.cin8);
            const size_t pix = (size_t)((a.n * p.H + hi) * p.W + wi);
            const unsigned short* src = v ? p.x + pix * p.x_ld + p.x_off + ci : zp;
            wd_glds16(src, ring + buf * WD_BUF + WD_GROWS * WD_ROWB + (4 * w4 + j - 2) * 1024);
        }
    };

    // ---- fragment addresses. Lane (q, pp) of its 16-lane group supplies pixel row q, channel columns 4pp .. 4pp+3; the lane
    //      half selects k 0-7 / 8-15, the 16-lane group the channel half of the 32-wide tile (wgrad_h16.hip).
    const int lg = lane & 15, q = lg >> 2, pp = lg & 3, gsel = (lane >> 4) & 1;
    const int kl = 8 * (lane >> 5) + q;                              // pixel column inside a tile row (+4: second read)
    const unsigned lds0 = (unsigned)(size_t)(wd_lptr)smem + kg * (WD_SLOTS * WD_BUF);
    // dz rows kl + 16 k16 (+4): bit 1 of the row is bit 1 of kl
    const unsigned a_off = lds0 + kl * WD_ROWB + (((4 * wm + 2 * gsel + (pp >> 1)) ^ (((kl >> 1) & 1) << 2)) << 4) + 8 * (pp & 1);
    // x rows kl + C (+4), C = 18 (k16 + kh) + kw: bit 1 of the row depends on C & 3 -> four variants of the lane's column
    unsigned b_off[4];
#pragma unroll
    for (int v = 0; v < 4; ++v) {
        const int sw = (((kl & 3) + v) >> 1) & 1;
        b_off[v] = lds0 + WD_GROWS * WD_ROWB + kl * WD_ROWB + (((4 * wn + 2 * gsel + (pp >> 1)) ^ (sw << 2)) << 4) + 8 * (pp & 1);
    }

    f32x16 acc[9];
#pragma unroll
    for (int t = 0; t < 9; ++t)
#pragma unroll
        for (int r = 0; r < 16; ++r) acc[t][r] = 0.f;

    if (niter > 0) {
        const TileAt a0 = locate(0), a1 = locate(1);
        wd_for<0, 6>([&](auto J) { issue1(J, a0, 0); });
        wd_for<0, 6>([&](auto J) { issue1(J, a1, 1); });
        wd_wait_vmcnt<6>();                                          // tile 0 of this wave has landed; tile 1 may be in flight
    }
    __builtin_amdgcn_s_barrier();
    __builtin_amdgcn_sched_barrier(0);

    // deferred MFMAs of the previous tile (none yet: zero operands add nothing)
    const u32x2 z2 = {0u, 0u};
    WFrag dA[4], dB[2];
#pragma unroll
    for (int k = 0; k < 4; ++k) { dA[k].lo = z2; dA[k].hi = z2; }
    dB[0].lo = z2; dB[0].hi = z2; dB[1] = dB[0];

    if (p.prio && kg == 1) __builtin_amdgcn_s_setprio(1);            // static priority for the later-dispatched half (MI355X_MICROARCH.md, two waves per SIMD, item 4)
    int buf = 0;
    for (int it = 0; it < niter; ++it) {
        // ring slot (it + 2) % 3 was read in round it - 1: every wave has passed that round's barrier
        const int nbuf = buf == 0 ? 2 : buf - 1;
        const TileAt nt = locate(it + 2);
        const unsigned boffs = buf * WD_BUF;
        const unsigned ab = a_off + boffs;
        unsigned bb[4];
#pragma unroll
        for (int v = 0; v < 4; ++v) bb[v] = b_off[v] + boffs;
        WFrag A[4], B[NB];
        // prologue reads: a0 B0 a3 B1 a1 B2 a2 [B3 ..]
        wd_read<0>(A[0], ab);
        wd_read<wd_boff(0)>(B[0], bb[wd_bvar(0)]);
        wd_read<3 * 16 * WD_ROWB>(A[3], ab);
        wd_read<wd_boff(1)>(B[1], bb[wd_bvar(1)]);
        wd_read<1 * 16 * WD_ROWB>(A[1], ab);
        wd_read<wd_boff(2)>(B[2], bb[wd_bvar(2)]);
        wd_read<2 * 16 * WD_ROWB>(A[2], ab);
        wd_for<3, LA>([&](auto F) { constexpr int f = decltype(F)::value; wd_read<wd_boff(f)>(B[f], bb[wd_bvar(f)]); });
        // the last two fragments of the previous tile: 6 MFMAs while those travel
#pragma unroll
        for (int e = 0; e < 2; ++e)
#pragma unroll
            for (int kh = 0; kh < 3; ++kh)
                acc[kh * 3 + 1 + e] = WDTraits<T>::mfma(wd_vec(dA[3 - kh]), wd_vec(dB[e]), acc[kh * 3 + 1 + e]);
        __builtin_amdgcn_sched_barrier(0);
        wd_for<0, 18>([&](auto F) {
            constexpr int f = decltype(F)::value;
            constexpr int r = wd_row(f), kw = f % 3;
            if constexpr (f < 6) issue1(std::integral_constant<int, f>{}, nt, nbuf);     // one DMA request per step: 60+ cycles of issue each
            if constexpr (f + LA < 18) wd_read<wd_boff(f + LA)>(B[(f + LA) % NB], bb[wd_bvar(f + LA)]);
            // the dz fragment first needed in this step was requested before B[f]: covered by the same count
            if constexpr (f == 0) wd_wait2<wd_wait_of(f, LA)>(B[f % NB], A[0]);
            else if constexpr (f == 3) wd_wait2<wd_wait_of(f, LA)>(B[f % NB], A[3]);
            else if constexpr (f == 6) wd_wait2<wd_wait_of(f, LA)>(B[f % NB], A[1]);
            else if constexpr (f == 9) wd_wait2<wd_wait_of(f, LA)>(B[f % NB], A[2]);
            else wd_wait1<wd_wait_of(f, LA)>(B[f % NB]);
            if constexpr (f < 16) {
#pragma unroll
                for (int kh = 0; kh < 3; ++kh) {
                    const int k16 = r - kh;
                    if (k16 >= 0 && k16 < WD_TH) acc[kh * 3 + kw] = WDTraits<T>::mfma(wd_vec(A[k16]), wd_vec(B[f % NB]), acc[kh * 3 + kw]);
                }
            }
            __builtin_amdgcn_sched_barrier(0);
        });
        // fragments 16 and 17 (row 3, kw = 1, 2) have landed (lgkmcnt(0) above): their MFMAs run behind the barrier
        dA[1] = A[1]; dA[2] = A[2]; dA[3] = A[3];
        dB[0] = B[16 % NB]; dB[1] = B[17 % NB];
        wd_wait_vmcnt<6>();                                          // own pieces of tile it + 1 landed (tile it + 2 stays in flight)
        __builtin_amdgcn_s_barrier();                                // ... and everybody else's; everybody is done reading slot buf
        __builtin_amdgcn_sched_barrier(0);
        buf = buf == 2 ? 0 : buf + 1;
    }
#pragma unroll
    for (int e = 0; e < 2; ++e)
#pragma unroll
        for (int kh = 0; kh < 3; ++kh)
            acc[kh * 3 + 1 + e] = WDTraits<T>::mfma(wd_vec(dA[3 - kh]), wd_vec(dB[e]), acc[kh * 3 + 1 + e]);
    wd_wait_vmcnt<0>();                                              // the ring is about to carry accumulators
    __builtin_amdgcn_s_barrier();
    __builtin_amdgcn_sched_barrier(0);

    // ---- group 1 -> LDS -> group 0 -> one partial per workgroup, in accumulator order
    f32x4* hand = reinterpret_cast<f32x4*>(smem);
    const int hbase = w4 * 4 * 64 + lane;                            // + tap * 1024 + reg group * 64
    if (kg == 1) {
#pragma unroll
        for (int t = 0; t < 9; ++t)
#pragma unroll
            for (int qq = 0; qq < 4; ++qq) {
                const f32x4 v = {acc[t][4 * qq], acc[t][4 * qq + 1], acc[t][4 * qq + 2], acc[t][4 * qq + 3]};
                hand[t * 1024 + qq * 64 + hbase] = v;
            }
    }
    __syncthreads();
    if (kg == 0) {
        f32x4* out = reinterpret_cast<f32x4*>(p.partial + ((size_t)slice * p.ntile + tile) * WD_TILE_FLOATS);
#pragma unroll
        for (int t = 0; t < 9; ++t)
#pragma unroll
            for (int qq = 0; qq < 4; ++qq) {
                const f32x4 o = hand[t * 1024 + qq * 64 + hbase];
                const f32x4 v = {acc[t][4 * qq] + o[0], acc[t][4 * qq + 1] + o[1], acc[t][4 * qq + 2] + o[2], acc[t][4 * qq + 3] + o[3]};
                out[t * 1024 + qq * 64 + hbase] = v;
            }
    }
}

// dW = sum over slices of the accumulator-order partials, then OIHW. One float4 (4 consecutive co rows of one ci) per G threads;
// the G partial sums are added in a fixed order through LDS (deterministic for given (nslices, G)).
template <int G>
__global__ __launch_bounds__(256) void wgrad_reduce_acc(const float* __restrict__ partial, float* __restrict__ dw, int nslices, int ntile,
                                                        int tiles_n, int cout, int cin) {
    __shared__ f32x4 red[256];
    const size_t plane4 = (size_t)ntile * (WD_TILE_FLOATS / 4);
    const long long total = (long long)plane4 * G;
    for (long long base = blockIdx.x * 256LL; base < total; base += (long long)gridDim.x * 256) {
        const long long id = base + threadIdx.x;
        const long long i = id / G;
        const int g = (int)(id - i * G);
        const bool live = id < total;
        f32x4 s0 = {0.f, 0.f, 0.f, 0.f}, s1 = s0;
        if (live) {
            const f32x4* src = reinterpret_cast<const f32x4*>(partial) + i;
            int k = g;
            for (; k + G < nslices; k += 2 * G) {                   // two independent chains, order fixed
                s0 += src[(size_t)k * plane4];
                s1 += src[(size_t)(k + G) * plane4];
            }
            if (k < nslices) s0 += src[(size_t)k * plane4];
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
            const int ln = (int)(i & 63), qq = (int)((i >> 6) & 3), w4 = (int)((i >> 8) & 3);
            const int rest = (int)(i >> 10);
            const int tap = rest % 9, tile = rest / 9;
            const int co = (tile / tiles_n) * 64 + (w4 >> 1) * 32 + 8 * qq + 4 * (ln >> 5);
            const int ci = (tile % tiles_n) * 64 + (w4 & 1) * 32 + (ln & 31);
            if (ci < cin) {
#pragma unroll
                for (int e = 0; e < 4; ++e)
                    if (co + e < cout) dw[((size_t)(co + e) * cin + ci) * 9 + tap] = s0[e];
            }
        }
    }
}

struct WgradDPlan { int tiles_m, tiles_n, th_tiles, tw_tiles, total_tiles, nslices, tiles_per_slice; };

static WgradDPlan plan_wgrad_d(int n, int h, int w, int cin, int cout) {
    WgradDPlan q;
    q.tiles_m = ceil_div(cout, 64);
    q.tiles_n = ceil_div(cin, 64);
    q.th_tiles = ceil_div(h, WD_TH);
    q.tw_tiles = ceil_div(w, WD_TW);
    q.total_tiles = n * q.th_tiles * q.tw_tiles;
    const int ntile = q.tiles_m * q.tiles_n;
    int ns = 256 / ntile;                                   // one workgroup per CU: partial bytes = workgroups x 147 KB
    const int maxs = ceil_div(q.total_tiles, 8);            // at least 8 K tiles per workgroup
    if (ns > maxs) ns = maxs;
    if (ns < 1) ns = 1;
    if (ns >= 8) ns = ns / 8 * 8;                           // multiples of 8: XCD-aware workgroup mapping
    q.tiles_per_slice = ceil_div(q.total_tiles, ns);
    q.nslices = ns;                                         // trailing slices may be empty (they write zeros)
    return q;
}

bool wgrad_dma_eligible(int cin, int cout, int ks, int stride, int dz_ld, int dz_off, int x_ld, int x_off) {
    static const bool off = getenv("YOLO_NO_WGRAD_DMA") != nullptr;      // A/B switch: round 2's kernel
    if (off || ks != 3 || stride != 1 || cin < 32) return false;         // (the 3-channel stem stays on wgrad_patch_h16)
    if ((dz_ld & 7) || (dz_off & 7) || (x_ld & 7) || (x_off & 7)) return false;
    return dz_ld >= round_up(cout, 8) && x_ld >= round_up(cin, 8);
}

size_t wgrad_dma_workspace(int n, int h, int w, int cin, int cout) {
    const WgradDPlan q = plan_wgrad_d(n, h, w, cin, cout);
    return (size_t)q.nslices * q.tiles_m * q.tiles_n * WD_TILE_FLOATS * sizeof(float);
}

int wgrad_dma_launch(const void* dz, int dz_ld, int dz_off, const void* x, int x_ld, int x_off, float* partial, float* dw, int n, int h,
                     int w, int cin, int cout, int dtype, hipStream_t s) {
    const WgradDPlan q = plan_wgrad_d(n, h, w, cin, cout);
    if ((long long)n * h * w > 0x7fffffffLL) return fail(YOLO_ERR_UNSUPPORTED, "wgrad: too many pixels");
    if (((size_t)dz | (size_t)x | (size_t)partial) & 15) return fail(YOLO_ERR_ARG, "wgrad: operands must be 16-byte aligned");
    WgradDArgs a;
    a.dz = (const unsigned short*)dz; a.x = (const unsigned short*)x; a.partial = partial;
    a.N = n; a.H = h; a.W = w;
    a.cin8 = round_up(cin, 8); a.cout8 = round_up(cout, 8);
    a.dz_ld = dz_ld; a.dz_off = dz_off; a.x_ld = x_ld; a.x_off = x_off;
    a.tiles_n = q.tiles_n; a.ntile = q.tiles_m * q.tiles_n;
    a.th_tiles = q.th_tiles; a.tw_tiles = q.tw_tiles; a.total_tiles = q.total_tiles;
    a.tiles_per_slice = q.tiles_per_slice; a.nslices = q.nslices;
    const int grid = a.ntile * a.nslices;
    static const int la = getenv("YOLO_WGRAD_LA") ? atoi(getenv("YOLO_WGRAD_LA")) : 3;        // tuning knobs (A/B runs)
    static const int prio = getenv("YOLO_WGRAD_PRIO") ? atoi(getenv("YOLO_WGRAD_PRIO")) : 0;
    a.prio = prio;
    auto go3 = [&](auto kern) -> int {
        static LdsOnce once;                                // one per instantiation of this generic lambda
        if (int rc = reserve_lds(once, reinterpret_cast<const void*>(kern), WD_LDS, "wgrad3_dma_h16")) return rc;
        hipLaunchKernelGGL(kern, dim3(grid), dim3(512), WD_LDS, s, a);
        return YOLO_OK;
    };
    int rc3;
    if (dtype == YOLO_BF16) rc3 = la == 5 ? go3(&wgrad3_dma_h16<__bf16, 5>) : go3(&wgrad3_dma_h16<__bf16, 3>);
    else rc3 = la == 5 ? go3(&wgrad3_dma_h16<_Float16, 5>) : go3(&wgrad3_dma_h16<_Float16, 3>);
    if (rc3) return rc3;
    if (int rc = check_launch("wgrad3_dma_h16")) return rc;
    // reduce: enough threads to keep the chip busy on the small tensors of the high-resolution layers
    const long long total4 = (long long)a.ntile * (WD_TILE_FLOATS / 4);
    auto go = [&](auto kern, int G) {
        const long long nt = total4 * G;
        const int blocks = (int)((nt + 255) / 256 < 8192 ? (nt + 255) / 256 : 8192);
        hipLaunchKernelGGL(kern, dim3(blocks), dim3(256), 0, s, (const float*)partial, dw, a.nslices, a.ntile, a.tiles_n, cout, cin);
    };
    if (total4 < 32768 && a.nslices >= 32) go(wgrad_reduce_acc<16>, 16);
    else if (total4 < 131072 && a.nslices >= 8) go(wgrad_reduce_acc<4>, 4);
    else go(wgrad_reduce_acc<1>, 1);
    return check_launch("wgrad_reduce_acc");
}

}  // namespace yolo
