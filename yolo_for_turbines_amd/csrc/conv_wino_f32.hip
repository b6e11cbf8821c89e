// conv_wino_f32.hip — the fp32 3x3 stride-1 blocks (model.py:80-86 with kernel_size = 3: the second convolution of every
// residual unit model.py:115-121, the 3x3 layers of the neck and of ScalePredictionBlock model.py:140-143) by the Winograd
// minimal-filtering algorithm F(2x2, 3x3):
//
//     Y = A^T [ (G g G^T) .* (B^T d B) ] A            per 4x4 input tile d, 3x3 filter g, 2x2 output tile Y
//
// 16 multiplications per 4 outputs and (ci, co) pair instead of 36: the matrix cores do 1 / 2.25 of the direct
// convolution's work. conv_patch_f32 (the direct implicit GEMM) sits at 0.72-0.77 of the f32 matrix peak and its ceiling
// at the clock the chip holds under it is ~0.86 - the fp32 forward could only get faster by doing fewer
// multiplications. This is the algorithm the reference's own backend picks for these layers (PyTorch -> MIOpen / cuDNN
// Winograd for fp32 3x3 stride 1), exact in exact arithmetic; in fp32 the transforms add a few ulp (B^T and A^T hold only
// 0 / +-1, G only 0 / +-1/2 / 1): measured against the oracle in tests/test_gpu_parity.py, bar 1e-3 (BASELINE north_star).
//
// Three kernels:
//   wino_pack_f32    U[xi][ci/4][co_pad64][4] = (G g G^T)[xi], xi = 4a + b, once per weight version (fp64 sums, one rounding);
//   wino_xform_f32   V[xi][ci/4][tile_pad64][4] = (B^T d B)[xi] for every 4x4 input tile (stride 2, zero padding 1):
//                    a thread owns one (tile, 4 channels); 8 lanes cover the 32 channels of a 128-byte line, 8 tiles per wave:
//                    loads read full lines, stores write 128-byte runs;
//   conv_wino_f32    per workgroup 64 tiles x 64 output channels x all 16 xi: 16 independent GEMMs
//                    D_xi[co][tile] = sum_ci U_xi[co][ci] V_xi[tile][ci] on v_mfma_f32_32x32x2_f32, then the output
//                    transform A^T D A, scale / shift / activation / residual and 16-byte stores from registers.
//
// conv_wino_f32 in detail (one 256-thread workgroup per CU, one wave per SIMD):
//   * wave (wm, wn) owns 32 tiles x 32 channels x 16 xi = 16 accumulator blocks of 16 registers = 256 accumulator
//     registers (the AGPR half of a lone wave's 512). Every lane therefore holds ALL 16 xi of its (tile, 16 channels):
//     the output transform is 24 additions per output quad inside the lane, no cross-lane or LDS exchange.
//   * U is the MFMA's A operand, V its B operand: D = [channel][tile]; a lane owns one tile and four runs of four
//     consecutive channels -> float4 stores / residual loads; the four runs of a pixel are one 128-byte line.
//   * both operands by LDS-DMA (global_load_lds_dwordx4) in stages of 4 input channels: a stage is 16 xi x 64 rows x 16 B
//     for V and the same for U = 32 KiB, every wave-instruction 1 KiB contiguous on both sides (that is what the
//     [xi][ci/4][row][4] layouts are for); ring of 4 stages, two in flight beyond the one being multiplied.
//   * a lane reads its fragments with ds_read_b64: channels 2h, 2h+1 of its row (h = lane >> 5) = the k of two MFMAs;
//     a wave's read is 512 contiguous bytes (conflict-free). 32 reads feed 32 MFMAs (2,048 matrix cycles) per stage.
//   * ONE barrier per stage, placed after the 12th of the 16 xi: behind it come the DMA requests of stage i + 3 and the first
//     fragment reads of stage i + 1, then the last four xi of stage i - the barrier's skew and the LDS latency of the
//     next stage hide behind 8 MFMAs (512 cycles) instead of draining the matrix pipe once per stage.
//   * blockIdx -> (tile block, channel block) so that the channel blocks of one tile block run back to back on ONE XCD
//     (blockIdx % 8 = XCD): V is fetched into that L2 once; U (2-34 MB) is shared by everybody.
// Arithmetic: an f32 MFMA chain is a k-ordered fmaf chain per xi; results differ from the direct kernel's by the
// transforms' roundings only. An image's result does not depend on its batch (tiles are independent, the choice of
// kernel looks at cin / h / w only).
#include <type_traits>
#include "common.h"

namespace yolo {

typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef float f32x2 __attribute__((ext_vector_type(2)));

typedef const __attribute__((address_space(1))) void* wn_gptr;
typedef __attribute__((address_space(3))) void* wn_lptr;
__device__ __forceinline__ void wn_glds16(const void* g, void* l) { __builtin_amdgcn_global_load_lds((wn_gptr)g, (wn_lptr)l, 16, 0, 0); }
// the same request with the source as wave-uniform base (SGPR pair) + 32-bit lane offset, the LDS destination (wave-uniform byte
// address) through M0. Written out because inside the stage loop the compiler's strength reduction turns base + offset back
// into one 64-bit vector add per request.
__device__ __forceinline__ void wn_glds16_s(const void* base_uniform, unsigned lane_off, unsigned lds_addr_uniform) {
    asm volatile("s_mov_b32 m0, %2\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %0, %1"
                 :: "v"(lane_off), "s"(base_uniform), "s"(lds_addr_uniform) : "memory");      // (M0 is written here; the compiler re-loads it before every use of its own)
}
template <int N> __device__ __forceinline__ void wn_wait_vmcnt() { asm volatile("s_waitcnt vmcnt(%0)" ::"n"(N) : "memory"); }

// fragments of one xi: the U row (A operand) and the V row (B operand), channels 2h and 2h + 1 of the stage
template <int OFF>
__device__ __forceinline__ void wn_read2(f32x2& a, f32x2& b, unsigned ua, unsigned va) {
    asm volatile("ds_read_b64 %0, %1 offset:%2" : "=&v"(a) : "v"(ua), "n"(OFF));
    asm volatile("ds_read_b64 %0, %1 offset:%2" : "=&v"(b) : "v"(va), "n"(OFF));
}

template <int I, int N, class Fn>
__device__ __forceinline__ void wn_for(Fn&& fn) {
    if constexpr (I < N) {
        fn(std::integral_constant<int, I>{});
        wn_for<I + 1, N>(fn);
    }
}

// ------------------------------------------------------------------------------ weights: U = G g G^T
// G = [1 0 0; 1/2 1/2 1/2; 1/2 -1/2 1/2; 0 0 1]
// dgrad = 1: the input-gradient convolution's filters g'[n = ci][k = co][p][q] = w[co][ci][2 - p][2 - q] (cout = number of rows n = the
// forward's cin, cin = number of k = the forward's cout rounded up to 32: columns beyond `klim` are zero)
__global__ void wino_pack_f32(const float* __restrict__ w, float* __restrict__ U, int cout, int cin, int coutp, long long total,
                              int dgrad, int klim) {
    const int C4 = cin / 4;
    for (long long i = blockIdx.x * (long long)blockDim.x + threadIdx.x; i < total; i += (long long)gridDim.x * blockDim.x) {
        const int e = (int)(i & 3);
        const long long r0 = i >> 2;
        const int co = (int)(r0 % coutp);
        const long long r1 = r0 / coutp;
        const int c4 = (int)(r1 % C4);
        const int xi = (int)(r1 / C4);
        float out = 0.f;
        const int kk = 4 * c4 + e;
        if (co < cout && (!dgrad || kk < klim)) {
            // forward: row co, column kk of w[cout][cin]; dgrad: row co is a forward INPUT channel, column kk a forward output channel
            const float* g = dgrad ? w + ((size_t)kk * cout + co) * 9 : w + ((size_t)co * cin + kk) * 9;
            const int a = xi >> 2, b = xi & 3;
            double t[3];
#pragma unroll
            for (int q = 0; q < 3; ++q) {
                const double g0 = dgrad ? g[8 - q] : g[q], g1 = dgrad ? g[5 - q] : g[3 + q], g2 = dgrad ? g[2 - q] : g[6 + q];
                t[q] = a == 0 ? g0 : a == 1 ? 0.5 * (g0 + g1 + g2) : a == 2 ? 0.5 * (g0 - g1 + g2) : g2;
            }
            const double u = b == 0 ? t[0] : b == 1 ? 0.5 * (t[0] + t[1] + t[2]) : b == 2 ? 0.5 * (t[0] - t[1] + t[2]) : t[2];
            out = (float)u;
        }
        U[i] = out;
    }
}

// ------------------------------------------------------------------------------ input transform: V = B^T d B
struct WinoXArgs {
    const float* x;
    float* V;
    int H, W, C4;
    int x_ld, x_off;
    int th, tw, T, Tpad;
};

// B^T = [1 0 -1 0; 0 1 1 0; 0 -1 1 0; 0 1 0 -1]
__global__ __launch_bounds__(256) void wino_xform_f32(const WinoXArgs p) {
    // 8 lanes = the 8 channel quads of one pixel's 128-byte line, 8 tiles per wave: every load instruction reads 8 full
    // lines (16 bytes per lane from 64 different pixels - one tile per lane - measured 66 us at 52 x 52 for 221 MB), every
    // store instruction writes 8 runs of 128 bytes
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const int c4 = blockIdx.x * 8 + (lane & 7);
    const int t = blockIdx.y * 32 + wave * 8 + (lane >> 3);
    if (c4 >= p.C4) return;
    const bool tv = t < p.T;
    const int tt = tv ? t : 0;
    const int per = p.th * p.tw;
    const int n = tt / per, rem = tt - n * per;
    const int ty = rem / p.tw, tx = rem - ty * p.tw;
    const int r0 = 2 * ty - 1, c0 = 2 * tx - 1;
    const float* base = p.x + p.x_off + 4 * c4;
    f32x4 d[4][4];
#pragma unroll
    for (int i = 0; i < 4; ++i) {
        const int r = r0 + i;
        const bool rv = tv && r >= 0 && r < p.H;
#pragma unroll
        for (int j = 0; j < 4; ++j) {
            const int c = c0 + j;
            const bool v = rv && c >= 0 && c < p.W;
            const size_t pix = v ? ((size_t)n * p.H + r) * p.W + c : 0;
            const f32x4 ld = *reinterpret_cast<const f32x4*>(base + pix * p.x_ld);
            const f32x4 z = {0.f, 0.f, 0.f, 0.f};
            d[i][j] = v ? ld : z;
        }
    }
    // rows: r = B^T d
    f32x4 r[4][4];
#pragma unroll
    for (int j = 0; j < 4; ++j) {
        r[0][j] = d[0][j] - d[2][j];
        r[1][j] = d[1][j] + d[2][j];
        r[2][j] = d[2][j] - d[1][j];
        r[3][j] = d[1][j] - d[3][j];
    }
    float* dst = p.V + ((size_t)c4 * p.Tpad + t) * 4;
    const size_t xs = (size_t)p.C4 * p.Tpad * 4;
#pragma unroll
    for (int a = 0; a < 4; ++a) {
        const f32x4 v0 = r[a][0] - r[a][2];
        const f32x4 v1 = r[a][1] + r[a][2];
        const f32x4 v2 = r[a][2] - r[a][1];
        const f32x4 v3 = r[a][1] - r[a][3];
        *reinterpret_cast<f32x4*>(dst + (4 * a + 0) * xs) = v0;
        *reinterpret_cast<f32x4*>(dst + (4 * a + 1) * xs) = v1;
        *reinterpret_cast<f32x4*>(dst + (4 * a + 2) * xs) = v2;
        *reinterpret_cast<f32x4*>(dst + (4 * a + 3) * xs) = v3;
    }
}

// ------------------------------------------------------------------------------ 16 GEMMs + output transform + block epilogue
struct WinoArgs {
    const float* V;
    const float* U;
    const float* scale;
    const float* shift;
    const float* res;
    float* y;
    int* nan_flag;
    int T, Tpad, C4, Cout, CoutPad;
    int th, tw, H, W;
    int y_ld, y_off, r_ld, r_off;
    int flags;
    int n_mt, n_nt;
};

constexpr int WN_STAGE = 32768;          // bytes per ring stage: V [16][64][4] floats, then U [16][64][4]
constexpr int WN_SLOTS = 4;              // (a power of two: slot arithmetic by mask)
constexpr int WN_DMA = 8;                // DMA wave-instructions per wave and stage
constexpr int WN_AHEAD = 4;              // fragment reads kept in flight, in xi (2 reads each)
#ifndef WN_DMA_MODE
#define WN_DMA_MODE 0
#endif
constexpr int WN_BAR = 16 - WN_AHEAD;    // the stage's barrier sits in front of this xi (all reads of the stage are issued by then)

// PROBE (diagnostic build only, garbage results): 1 = no stage barrier / DMA wait, 2 = no DMA requests in the loop, 4 = no fragment reads;
// 8 + e ablates a part of the EPILOGUE instead: e = 0 the stores, 1 the residual requests, 3 staging and stores
template <int ACT, bool RES, int PROBE = 0>
__global__ __launch_bounds__(256, 1) void conv_wino_f32(const WinoArgs p) {
    extern __shared__ __attribute__((aligned(16))) char smem[];
    constexpr int MP = (PROBE & 8) ? 0 : PROBE;          // main-loop ablations
    constexpr int EP = (PROBE & 8) ? (PROBE & 3) : -1;   // epilogue ablation
    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int h = lane >> 5, m = lane & 31;
    const int xcd = blockIdx.x & 7, q = blockIdx.x >> 3;
    const int nt = q % p.n_nt, mt = (q / p.n_nt) * 8 + xcd;
    if (mt >= p.n_mt) return;
    const int wm = wave & 1, wn = wave >> 1;
#ifdef WN_STAMPS   // diagnostic build (make wstamps): per-workgroup phase stamps into the buffer passed as nan_flag (tools/wino_stamps.py)
    const unsigned long long st0 = __builtin_amdgcn_s_memtime();
#endif

    // ---- DMA roles: this wave moves xi = 4 wave .. 4 wave + 3 of V and of U, lane = row
    const size_t v_xi = (size_t)p.C4 * p.Tpad * 4, v_c4 = (size_t)p.Tpad * 4;
    const size_t u_xi = (size_t)p.C4 * p.CoutPad * 4, u_c4 = (size_t)p.CoutPad * 4;
    // source = wave-uniform 64-bit base + this lane's 32-bit byte offset: the form the SGPR-base addressing mode of
    // global_load_lds takes (one 32-bit VGPR per request instead of a 64-bit add per lane in front of every request)
    const char* vsrc = reinterpret_cast<const char*>(p.V + (size_t)(4 * wave) * v_xi + (size_t)mt * 256);
    const char* usrc = reinterpret_cast<const char*>(p.U + (size_t)(4 * wave) * u_xi + (size_t)nt * 256);
    const unsigned lane16 = lane * 16;
    const unsigned lds_ring = (unsigned)(size_t)(wn_lptr)smem;
    // piece k of a stage: k < 4 -> xi = 4 wave + k of V, else xi = 4 wave + k - 4 of U
    auto issue_piece = [&](auto K, int c4, int slot) {
        constexpr int k = decltype(K)::value;
        const unsigned dst = lds_ring + slot * WN_STAGE + wave * 4096;
        if constexpr (k < 4) wn_glds16_s(vsrc + (k * v_xi + c4 * v_c4) * 4, lane16, dst + k * 1024);
        else wn_glds16_s(usrc + ((k - 4) * u_xi + c4 * u_c4) * 4, lane16, dst + 16384 + (k - 4) * 1024);
    };
    auto issue = [&](int c4, int slot) { wn_for<0, WN_DMA>([&](auto K) { issue_piece(K, c4, slot); }); };

    // ---- fragment addresses: row 32 wm + m of V (B operand), row 32 wn + m of U (A operand), channels 2h, 2h + 1
    const unsigned lds0 = (unsigned)(size_t)(wn_lptr)smem;
    const unsigned vb = lds0 + (32 * wm + m) * 16 + h * 8;
    const unsigned ub = lds0 + 16384 + (32 * wn + m) * 16 + h * 8;

    f32x2 fa[16], fb[16];                 // U / V fragments of the stage being multiplied (the first WN_AHEAD also of the next)

    // folded scale / shift of this lane's 16 output channels (32 wn + 8 g + 4 h + e): requested before the ring's first pieces,
    // so that no load - and no wait for one - sits between the residual requests and the stores of the epilogue
    f32x4 sc[4], sh[4];
#pragma unroll
    for (int g = 0; g < 4; ++g) {
        const int co = nt * 64 + 32 * wn + 4 * h + 8 * g;
        const int cc = co < p.Cout ? co : 0;                          // (cout is a multiple of 4: a run is valid or not as a whole)
        sc[g] = *reinterpret_cast<const f32x4*>(p.scale + cc);
        sh[g] = *reinterpret_cast<const f32x4*>(p.shift + cc);
    }
    const int nst = p.C4;
    issue(0, 0);
    if (nst > 1) issue(1, 1);
    if (nst > 2) issue(2, 2);
    __builtin_amdgcn_sched_barrier(0);
    f32x16 acc[16];                       // zeroed while the first pieces are on their way
#pragma unroll
    for (int x = 0; x < 16; ++x)
#pragma unroll
        for (int r = 0; r < 16; ++r) acc[x][r] = 0.f;
#pragma unroll
    for (int x = 0; x < 16; ++x) asm volatile("" : "+a"(acc[x]));
    __builtin_amdgcn_sched_barrier(0);
    if (nst > 2) wn_wait_vmcnt<2 * WN_DMA>();
    else if (nst > 1) wn_wait_vmcnt<WN_DMA>();
    else wn_wait_vmcnt<0>();
    __builtin_amdgcn_s_barrier();
    __builtin_amdgcn_sched_barrier(0);
#ifdef WN_STAMPS
    const unsigned long long st1 = __builtin_amdgcn_s_memtime();
    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
#endif
    wn_for<0, WN_AHEAD>([&](auto X) { constexpr int x = decltype(X)::value; wn_read2<x * 1024>(fa[x], fb[x], ub, vb); });

    int slot = 0;
    for (int st = 0; st < nst; ++st) {
        const unsigned sb = (unsigned)slot * WN_STAGE;
        int ns = slot + 1;
        ns = ns == WN_SLOTS ? 0 : ns;
        const unsigned nb = (unsigned)ns * WN_STAGE;
        // Where the DMA requests of the ring go (WN_DMA_MODE; stage st + 3 takes the slot stage st - 1 had, free from this stage's
        // barrier on). A request costs the issuing wave 60-185 cycles (MI355X_MICROARCH.md, "LDS-DMA piece issue cost") with
        // ONE MFMA in flight behind it, and this kernel has one wave per SIMD: nobody else issues MFMAs meanwhile.
        //   0: all eight behind the first xi after the barrier;  1: one behind every second xi (12, 14, then 0 .. 10 of the next stage);
        //   2: as 1, between the two MFMAs of the xi;  3: one behind each of the eight MFMAs after the barrier
        auto dma_at = [&](auto XX, auto HH) {
            constexpr int x = decltype(XX)::value, half = decltype(HH)::value;
            if constexpr (MP & 2) return;
            if constexpr (WN_DMA_MODE == 0) {
                if constexpr (x == WN_BAR && half == 1) {
                    __builtin_amdgcn_sched_barrier(0);
                    if (st + 3 < nst) issue(st + 3, (slot + 3) & (WN_SLOTS - 1));
                    __builtin_amdgcn_sched_barrier(0);
                }
            } else if constexpr (WN_DMA_MODE == 3) {
                if constexpr (x >= WN_BAR) {
                    __builtin_amdgcn_sched_barrier(0);
                    if (st + 3 < nst) issue_piece(std::integral_constant<int, 2 * (x - WN_BAR) + half>{}, st + 3, (slot + 3) & (WN_SLOTS - 1));
                    __builtin_amdgcn_sched_barrier(0);
                }
            } else if constexpr (half == (WN_DMA_MODE == 1 ? 1 : 0)) {
                if constexpr (x == WN_BAR || x == WN_BAR + 2) {
                    __builtin_amdgcn_sched_barrier(0);
                    if (st + 3 < nst) issue_piece(std::integral_constant<int, (x - WN_BAR) / 2>{}, st + 3, (slot + 3) & (WN_SLOTS - 1));
                    __builtin_amdgcn_sched_barrier(0);
                }
                if constexpr (x < WN_BAR && x % 2 == 0) {
                    __builtin_amdgcn_sched_barrier(0);
                    if (st >= 1 && st + 2 < nst) issue_piece(std::integral_constant<int, 2 + x / 2>{}, st + 2, (slot + 2) & (WN_SLOTS - 1));
                    __builtin_amdgcn_sched_barrier(0);
                }
            }
        };
        wn_for<0, 16>([&](auto X) {
            constexpr int x = decltype(X)::value;
            static_assert(WN_AHEAD == 4, "the tail schedule below is written for four xi of look-ahead");
            if constexpr (x == WN_BAR) {
                // stage st + 1 has to be in LDS for everybody before its first fragments are read; the slot of stage st - 1
                // (= of stage st + 3) is free once everybody is here
                __builtin_amdgcn_sched_barrier(0);
                if constexpr (!(MP & 1)) {
                    if (st + 2 < nst) wn_wait_vmcnt<WN_DMA>();
                    else wn_wait_vmcnt<0>();
                    __builtin_amdgcn_s_barrier();
                }
                __builtin_amdgcn_sched_barrier(0);
            }
            if constexpr (x + WN_AHEAD < 16 && !(MP & 4)) wn_read2<(x + WN_AHEAD) * 1024>(fa[x + WN_AHEAD], fb[x + WN_AHEAD], ub + sb, vb + sb);
            // reads issued after those of xi = x and still in flight (LDS operations return in order): 2 per xi. Behind the
            // barrier: xi 12 -> 13, 14, 15; 13 -> 14, 15; 14 -> 15 and the next stage's 0, 1; 15 -> the next stage's 0 .. 3
            constexpr int after = x < WN_BAR ? 2 * WN_AHEAD : x == 12 ? 6 : x == 13 ? 4 : x == 14 ? 6 : 8;
            if constexpr (!(MP & 4)) asm volatile("s_waitcnt lgkmcnt(%2)" : "+v"(fa[x]), "+v"(fb[x]) : "n"(after));
            else asm volatile("" : "+v"(fa[x]), "+v"(fb[x]));
            acc[x] = __builtin_amdgcn_mfma_f32_32x32x2f32(fa[x][0], fb[x][0], acc[x], 0, 0, 0);
            dma_at(X, std::integral_constant<int, 0>{});
            acc[x] = __builtin_amdgcn_mfma_f32_32x32x2f32(fa[x][1], fb[x][1], acc[x], 0, 0, 0);
            dma_at(X, std::integral_constant<int, 1>{});
            if constexpr (x == WN_BAR + 1 || x == WN_BAR + 2) {
                constexpr int y = 2 * (x - WN_BAR - 1);
                __builtin_amdgcn_sched_barrier(0);
                if constexpr (!(MP & 4)) {
                    wn_read2<y * 1024>(fa[y], fb[y], ub + nb, vb + nb);
                    wn_read2<(y + 1) * 1024>(fa[y + 1], fb[y + 1], ub + nb, vb + nb);
                }
                __builtin_amdgcn_sched_barrier(0);
            }
        });
        slot = ns;
    }
    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");      // the look-ahead reads of the stage after the last
#ifdef WN_STAMPS
    asm volatile("s_nop 15\n\ts_nop 15" ::: "memory");
    const unsigned long long st2 = __builtin_amdgcn_s_memtime();
#endif

    // ------------------------------------------------------------------ output transform + epilogue through LDS
    // A^T = [1 1 1 0; 0 1 -1 -1]. From registers a lane would store 16 bytes of 16 different pixels per instruction (32-byte pieces
    // of 32 lines: measured ~20 us per workgroup for 64 KiB out + 64 KiB of residual, as much as 20 stages of matrix work). So the
    // activated tile goes through the idle ring as [4 pixels of a tile][64 tiles][64 channels + 4] and comes back with 16 lanes
    // per pixel row: stores and residual loads are 256-byte runs.
    constexpr int OLD = 68;
    float* ost = reinterpret_cast<float*>(smem);
    int* tab = reinterpret_cast<int*>(smem + 256 * OLD * 4);        // [64] first output pixel of the tile, [64] flags
#pragma unroll
    for (int g = 0; g < 4; ++g) asm volatile("" : "+v"(sc[g]), "+v"(sh[g]));      // the compiler's wait for them goes HERE: nothing is in flight
    __syncthreads();                                                  // every wave is done with the ring (no DMA is in flight)
    if (tid < 64) {
        const int t = mt * 64 + tid;
        const bool tv = t < p.T;
        const int tt = tv ? t : 0;
        const int per = p.th * p.tw;
        const int n = tt / per, rem = tt - n * per;
        const int ty = rem / p.tw, tx = rem - ty * p.tw;
        tab[tid] = (n * p.H + 2 * ty) * p.W + 2 * tx;
        tab[64 + tid] = (tv ? 1 : 0) | (2 * tx + 1 < p.W ? 2 : 0) | (2 * ty + 1 < p.H ? 4 : 0);
    }
    __syncthreads();
    // this thread's 16 rows of the staged tile: pixel pp = it / 4 of tile (tid / 16) + 16 (it % 4), channels 4 (tid % 16) ..
    const int c16 = tid & 15;
    const int co_t = nt * 64 + 4 * c16;
    const bool cv = co_t < p.Cout;
    int pix[16];
    bool pv[16];
#pragma unroll
    for (int it = 0; it < 16; ++it) {
        const int tl = (tid >> 4) + 16 * (it & 3), pp = it >> 2;
        const int fl = tab[64 + tl];
        pv[it] = cv && (fl & 1) && (!(pp & 1) || (fl & 2)) && (!(pp & 2) || (fl & 4));
        pix[it] = pv[it] ? tab[tl] + (pp & 1) + (pp >> 1) * p.W : 0;
    }
    f32x4 rr[16];
    if (RES && EP != 1) {
#pragma unroll
        for (int it = 0; it < 16; ++it)
            rr[it] = *reinterpret_cast<const f32x4*>(p.res + (size_t)pix[it] * p.r_ld + p.r_off + (cv ? co_t : 0));
    } else if (RES) {
#pragma unroll
        for (int it = 0; it < 16; ++it) rr[it] = f32x4{0.f, 0.f, 0.f, 0.f};
    }
    __builtin_amdgcn_sched_barrier(0);                                // all residual rows are requested before the transform starts
    {
        const int cob = 32 * wn + 4 * h;                              // channel inside the block: + 8 g + e
        float* dst = ost + (32 * wm + m) * OLD + cob;
#pragma unroll
        for (int g = 0; g < 4; ++g) {
            f32x4 o[4];
#pragma unroll
            for (int e = 0; e < 4; ++e) {
                const int r = 4 * g + e;
                float tr[2][4];
#pragma unroll
                for (int b = 0; b < 4; ++b) {
                    tr[0][b] = acc[b][r] + acc[4 + b][r] + acc[8 + b][r];
                    tr[1][b] = acc[4 + b][r] - acc[8 + b][r] - acc[12 + b][r];
                }
#pragma unroll
                for (int i = 0; i < 2; ++i) {
                    const float y0 = tr[i][0] + tr[i][1] + tr[i][2];
                    const float y1 = tr[i][1] - tr[i][2] - tr[i][3];
                    o[2 * i][e] = act_c<ACT>(y0 * sc[g][e] + sh[g][e]);
                    o[2 * i + 1][e] = act_c<ACT>(y1 * sc[g][e] + sh[g][e]);
                }
            }
#pragma unroll
            for (int pp = 0; pp < 4; ++pp) {
                if constexpr (EP == 3) { asm volatile("" :: "v"(o[pp])); continue; }
                *reinterpret_cast<f32x4*>(dst + pp * 64 * OLD + 8 * g) = o[pp];
            }
        }
    }
    __syncthreads();
    bool saw_nan = false;
    f32x4 va[16];
#pragma unroll
    for (int it = 0; it < 16; ++it) va[it] = *reinterpret_cast<const f32x4*>(ost + ((tid >> 4) + 16 * it) * OLD + 4 * c16);
#pragma unroll
    for (int it = 0; it < 16; ++it) {
        f32x4 v = va[it];
        if (RES) v += rr[it];
        if (pv[it]) {
            saw_nan |= (v[0] != v[0]) | (v[1] != v[1]) | (v[2] != v[2]) | (v[3] != v[3]);
            if constexpr (EP == 0 || EP == 3) asm volatile("" :: "v"(v));
            else *reinterpret_cast<f32x4*>(p.y + (size_t)pix[it] * p.y_ld + p.y_off + co_t) = v;
        }
    }
#ifdef WN_STAMPS
    if (tid == 0) {
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        const unsigned long long st3 = __builtin_amdgcn_s_memtime();
        unsigned hwid, xcc;
        asm volatile("s_getreg_b32 %0, hwreg(HW_REG_HW_ID)" : "=s"(hwid));
        asm volatile("s_getreg_b32 %0, hwreg(HW_REG_XCC_ID)" : "=s"(xcc));
        unsigned long long* d = reinterpret_cast<unsigned long long*>(p.nan_flag) + (size_t)blockIdx.x * 6;
        d[0] = st0; d[1] = st1; d[2] = st2; d[3] = st3; d[4] = hwid; d[5] = xcc;
    }
    (void)saw_nan;
#else
    if ((p.flags & YOLO_FLAG_NANCHECK) && saw_nan) atomicOr(p.nan_flag, 2);
#endif
}

// ------------------------------------------------------------------------------ the same launch in TWO PASSES over xi, two workgroups per CU
// conv_wino_f32 above needs all 512 registers of a SIMD for ONE wave, and that wave does everything of a tile in turn:
// 17 k cycles per tile outside the main loop (first-stage latency, zeroing 256 accumulators, the ~1,400 vector instructions of
// the output transform, residual requests, stores) with the matrix pipe idle - 17 % of a tile at 52 x 52, 30 % at 104 x 104.
// Here a workgroup walks the K loop twice: pass 0 accumulates xi 0..7 (rows 0, 1 of the 4 x 4 product matrix M) into 128
// accumulator registers, transforms them into a PARTIAL 2 x 2 output tile (Y = A^T M A is linear in the rows of M) that stays in
// 64 vector registers, pass 1 does the same for xi 8..15 and adds. 128 + ~110 registers per wave: TWO workgroups per CU, each
// with its own tile, ring (4 x 16 KiB) and phase - while one is in its prologue, transform or epilogue the other one's MFMAs
// have the matrix pipe. Same operand bytes per MFMA, same arithmetic up to the order of the transform's additions.
constexpr int W2_STAGE = 16384;          // bytes per ring stage: V [8][64][4] floats, then U [8][64][4]
constexpr int W2_DMA = 4;                // DMA wave-instructions per wave and stage

template <int ACT, bool RES>
__global__ __launch_bounds__(256, 2) void conv_wino2_f32(const WinoArgs p) {
    extern __shared__ __attribute__((aligned(16))) char smem[];
    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int h = lane >> 5, m = lane & 31;
    const int xcd = blockIdx.x & 7, q = blockIdx.x >> 3;
    const int nt = q % p.n_nt, mt = (q / p.n_nt) * 8 + xcd;
    if (mt >= p.n_mt) return;
    const int wm = wave & 1, wn = wave >> 1;
#ifdef WN_STAMPS
    const unsigned long long st0 = __builtin_amdgcn_s_memtime();
    unsigned long long st1 = 0;
#endif

    // ---- DMA roles: this wave moves xi = 8 pass + 2 wave, + 1 of V and of U, lane = row
    const size_t v_xi = (size_t)p.C4 * p.Tpad * 4, v_c4 = (size_t)p.Tpad * 4;
    const size_t u_xi = (size_t)p.C4 * p.CoutPad * 4, u_c4 = (size_t)p.CoutPad * 4;
    const char* v_wave = reinterpret_cast<const char*>(p.V + (size_t)(2 * wave) * v_xi + (size_t)mt * 256);
    const char* u_wave = reinterpret_cast<const char*>(p.U + (size_t)(2 * wave) * u_xi + (size_t)nt * 256);
    const unsigned lane16 = lane * 16;
    const unsigned lds0 = (unsigned)(size_t)(wn_lptr)smem;
    auto issue_piece = [&](auto K, int pass, int c4, int slot) {
        constexpr int k = decltype(K)::value;
        const unsigned dst = lds0 + slot * W2_STAGE + wave * 2048;
        if constexpr (k < 2) wn_glds16_s(v_wave + ((size_t)(8 * pass + k) * v_xi + c4 * v_c4) * 4, lane16, dst + k * 1024);
        else wn_glds16_s(u_wave + ((size_t)(8 * pass + k - 2) * u_xi + c4 * u_c4) * 4, lane16, dst + 8192 + (k - 2) * 1024);
    };
    auto issue = [&](int pass, int c4, int slot) { wn_for<0, W2_DMA>([&](auto K) { issue_piece(K, pass, c4, slot); }); };

    // ---- fragment addresses: local xi l; row 32 wm + m of V (B operand), row 32 wn + m of U (A operand), channels 2h, 2h + 1
    const unsigned vb = lds0 + (32 * wm + m) * 16 + h * 8;
    const unsigned ub = lds0 + 8192 + (32 * wn + m) * 16 + h * 8;

    // scale / shift of the 4 channels this thread STORES (4 (tid % 16) .. of the block): requested before the ring's first pieces
    const int c16 = tid & 15;
    const int co_t = nt * 64 + 4 * c16;
    const bool cv = co_t < p.Cout;
    const f32x4 sc_t = *reinterpret_cast<const f32x4*>(p.scale + (cv ? co_t : 0));
    const f32x4 sh_t = *reinterpret_cast<const f32x4*>(p.shift + (cv ? co_t : 0));
    const int nst = p.C4;
    f32x4 py[4][4];                       // the (partial) output tile: [channel run g][pixel] of this lane's tile; pass 1 adds to pass 0's
    auto run_pass = [&](auto PASS) {
        constexpr int pass = decltype(PASS)::value;
        f32x2 fa[8], fb[8];
        if constexpr (pass == 0) {           // (pass 1's first stages are requested before pass 0's transform, see below)
            issue(pass, 0, 0);
            if (nst > 1) issue(pass, 1, 1);
            if (nst > 2) issue(pass, 2, 2);
        }
        __builtin_amdgcn_sched_barrier(0);
        f32x16 acc[8];
#pragma unroll
        for (int x = 0; x < 8; ++x)
#pragma unroll
            for (int r = 0; r < 16; ++r) acc[x][r] = 0.f;
#pragma unroll
        for (int x = 0; x < 8; ++x) asm volatile("" : "+a"(acc[x]));
        __builtin_amdgcn_sched_barrier(0);
        if (nst > 2) wn_wait_vmcnt<2 * W2_DMA>();
        else if (nst > 1) wn_wait_vmcnt<W2_DMA>();
        else wn_wait_vmcnt<0>();
        __builtin_amdgcn_s_barrier();
        __builtin_amdgcn_sched_barrier(0);
#ifdef WN_STAMPS
        if constexpr (pass == 0) st1 = __builtin_amdgcn_s_memtime();
        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
#endif
        wn_read2<0>(fa[0], fb[0], ub, vb);
        wn_read2<1024>(fa[1], fb[1], ub, vb);

        int slot = 0;
        for (int st = 0; st < nst; ++st) {
            const unsigned sb = (unsigned)slot * W2_STAGE;
            const int ns = (slot + 1) & (WN_SLOTS - 1);
            const unsigned nb = (unsigned)ns * W2_STAGE;
            wn_for<0, 8>([&](auto X) {
                constexpr int x = decltype(X)::value;
                if constexpr (x == 6) {
                    // stage st + 1 has to be in LDS for everybody before its first fragments are read; the slot of stage st - 1
                    // (= of stage st + 3) is free once everybody is here
                    __builtin_amdgcn_sched_barrier(0);
                    if (st + 2 < nst) wn_wait_vmcnt<W2_DMA>();
                    else wn_wait_vmcnt<0>();
                    __builtin_amdgcn_s_barrier();
                    __builtin_amdgcn_sched_barrier(0);
                }
                if constexpr (x + 2 < 8) wn_read2<(x + 2) * 1024>(fa[x + 2], fb[x + 2], ub + sb, vb + sb);
                // reads issued after those of xi = x and still in flight: x < 6 -> x + 1, x + 2; 6 -> 7; 7 -> the next stage's 0
                constexpr int after = x < 6 ? 4 : 2;
                asm volatile("s_waitcnt lgkmcnt(%2)" : "+v"(fa[x]), "+v"(fb[x]) : "n"(after));
                acc[x] = __builtin_amdgcn_mfma_f32_32x32x2f32(fa[x][0], fb[x][0], acc[x], 0, 0, 0);
                if constexpr (x >= 6) {
                    __builtin_amdgcn_sched_barrier(0);
                    if (st + 3 < nst) issue_piece(std::integral_constant<int, 2 * (x - 6)>{}, pass, st + 3, (slot + 3) & (WN_SLOTS - 1));
                    __builtin_amdgcn_sched_barrier(0);
                }
                acc[x] = __builtin_amdgcn_mfma_f32_32x32x2f32(fa[x][1], fb[x][1], acc[x], 0, 0, 0);
                if constexpr (x >= 6) {
                    __builtin_amdgcn_sched_barrier(0);
                    if (st + 3 < nst) issue_piece(std::integral_constant<int, 2 * (x - 6) + 1>{}, pass, st + 3, (slot + 3) & (WN_SLOTS - 1));
                    wn_read2<(x - 6) * 1024>(fa[x - 6], fb[x - 6], ub + nb, vb + nb);
                    __builtin_amdgcn_sched_barrier(0);
                }
            });
            slot = ns;
        }
        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");  // the look-ahead reads of the stage after the last
        __syncthreads();                                     // everybody is done with the ring: pass 1's first stages / the staging may overwrite it
        if constexpr (pass == 0) {                           // ... and they are requested now: in flight during the transform below
            issue(1, 0, 0);
            if (nst > 1) issue(1, 1, 1);
            if (nst > 2) issue(1, 2, 2);
            __builtin_amdgcn_sched_barrier(0);
        }
        // partial output transform. A^T = [1 1 1 0; 0 1 -1 -1]: rows 0, 1 of M give t0 = M0 + M1, t1 = M1; rows 2, 3 t0 = M2, t1 = -M2 - M3
#pragma unroll
        for (int gq = 0; gq < 4; ++gq) {
#pragma unroll
            for (int e = 0; e < 4; ++e) {
                const int r = 4 * gq + e;
                float t0[4], t1[4];
#pragma unroll
                for (int b = 0; b < 4; ++b) {
                    if constexpr (pass == 0) { t0[b] = acc[b][r] + acc[4 + b][r]; t1[b] = acc[4 + b][r]; }
                    else { t0[b] = acc[b][r]; t1[b] = -acc[b][r] - acc[4 + b][r]; }
                }
                const float y0 = t0[0] + t0[1] + t0[2], y1 = t0[1] - t0[2] - t0[3];
                const float y2 = t1[0] + t1[1] + t1[2], y3 = t1[1] - t1[2] - t1[3];
                if constexpr (pass == 0) {
                    py[gq][0][e] = y0; py[gq][1][e] = y1; py[gq][2][e] = y2; py[gq][3][e] = y3;
                } else {
                    py[gq][0][e] += y0; py[gq][1][e] += y1; py[gq][2][e] += y2; py[gq][3][e] += y3;
                }
            }
        }
    };
    run_pass(std::integral_constant<int, 0>{});
    run_pass(std::integral_constant<int, 1>{});
#ifdef WN_STAMPS
    asm volatile("s_nop 15\n\ts_nop 15" ::: "memory");
    const unsigned long long st2 = __builtin_amdgcn_s_memtime();
#endif

    // ------------------------------------------------------------------ epilogue through LDS
    // The RAW output tile is staged ([pixel of the tile][tile][64 + 4] over the idle ring); the threads that store (16 lanes per
    // pixel row, 4 channels each - the same 4 for all of a thread's 16 rows) apply the folded BatchNorm, the activation and the
    // residual: their scale / shift are 8 registers requested at kernel start, all 16 residual rows are requested before the
    // first staged row is read back.
    constexpr int OLD = 68;
    float* ost = reinterpret_cast<float*>(smem);
    int* tab = reinterpret_cast<int*>(smem + 256 * OLD * 4);        // [64] first output pixel of the tile, [64] flags
    if (tid < 64) {
        const int t = mt * 64 + tid;
        const bool tv = t < p.T;
        const int tt = tv ? t : 0;
        const int per = p.th * p.tw;
        const int n = tt / per, rem = tt - n * per;
        const int ty = rem / p.tw, tx = rem - ty * p.tw;
        tab[tid] = (n * p.H + 2 * ty) * p.W + 2 * tx;
        tab[64 + tid] = (tv ? 1 : 0) | (2 * tx + 1 < p.W ? 2 : 0) | (2 * ty + 1 < p.H ? 4 : 0);
    }
    {
        float* dst = ost + (32 * wm + m) * OLD + 32 * wn + 4 * h;
#pragma unroll
        for (int gq = 0; gq < 4; ++gq)
#pragma unroll
            for (int pp = 0; pp < 4; ++pp) *reinterpret_cast<f32x4*>(dst + pp * 64 * OLD + 8 * gq) = py[gq][pp];
    }
    __syncthreads();
    // this thread's 16 rows of the staged tile: pixel pp = it / 4 of tile (tid / 16) + 16 (it % 4), channels 4 (tid % 16) ..
    bool saw_nan = false;
    int pix[16];
    bool pv[16];
#pragma unroll
    for (int it = 0; it < 16; ++it) {
        const int tl = (tid >> 4) + 16 * (it & 3), pp = it >> 2;
        const int fl = tab[64 + tl];
        pv[it] = cv && (fl & 1) && (!(pp & 1) || (fl & 2)) && (!(pp & 2) || (fl & 4));
        pix[it] = pv[it] ? tab[tl] + (pp & 1) + (pp >> 1) * p.W : 0;
    }
    f32x4 rr[16];
    if (RES) {
#pragma unroll
        for (int it = 0; it < 16; ++it)
            rr[it] = *reinterpret_cast<const f32x4*>(p.res + (size_t)pix[it] * p.r_ld + p.r_off + (cv ? co_t : 0));
    }
    __builtin_amdgcn_sched_barrier(0);
#pragma unroll
    for (int half = 0; half < 2; ++half) {
        f32x4 va[8];
#pragma unroll
        for (int i8 = 0; i8 < 8; ++i8) va[i8] = *reinterpret_cast<const f32x4*>(ost + ((tid >> 4) + 16 * (8 * half + i8)) * OLD + 4 * c16);
        // all eight values first (one run of counted waits for the residual rows), then the eight predicated stores: inside a
        // predicated block the compiler waits for EVERYTHING in flight, i.e. for the previous store's completion
#pragma unroll
        for (int i8 = 0; i8 < 8; ++i8) {
            const int it = 8 * half + i8;
#pragma unroll
            for (int e = 0; e < 4; ++e) va[i8][e] = act_c<ACT>(va[i8][e] * sc_t[e] + sh_t[e]);
            if (RES) va[i8] += rr[it];
            saw_nan |= pv[it] & ((va[i8][0] != va[i8][0]) | (va[i8][1] != va[i8][1]) | (va[i8][2] != va[i8][2]) | (va[i8][3] != va[i8][3]));
        }
#pragma unroll
        for (int i8 = 0; i8 < 8; ++i8) asm volatile("" : "+v"(va[i8]));
#pragma unroll
        for (int i8 = 0; i8 < 8; ++i8) {
            const int it = 8 * half + i8;
            if (pv[it]) *reinterpret_cast<f32x4*>(p.y + (size_t)pix[it] * p.y_ld + p.y_off + co_t) = va[i8];
        }
    }
#ifdef WN_STAMPS
    if (tid == 0) {
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        const unsigned long long st3 = __builtin_amdgcn_s_memtime();
        unsigned hwid, xcc;
        asm volatile("s_getreg_b32 %0, hwreg(HW_REG_HW_ID)" : "=s"(hwid));
        asm volatile("s_getreg_b32 %0, hwreg(HW_REG_XCC_ID)" : "=s"(xcc));
        unsigned long long* d = reinterpret_cast<unsigned long long*>(p.nan_flag) + (size_t)blockIdx.x * 6;
        d[0] = st0; d[1] = st1; d[2] = st2; d[3] = st3; d[4] = hwid; d[5] = xcc;
    }
    (void)saw_nan;
#else
    if ((p.flags & YOLO_FLAG_NANCHECK) && saw_nan) atomicOr(p.nan_flag, 2);
#endif
}

// ------------------------------------------------------------------------------ host side
static const bool g_wino_off = getenv("YOLO_NO_WINOGRAD") != nullptr;     // A/B switch: the direct kernels
// conv_wino2_f32 (two passes, two workgroups per CU) for feature maps of at most this many pixels (shape rule, never the batch):
// measured 292-297 vs 315-321 us at 13 x 13, equal at 26 x 26 / 52 x 52, 442 vs 372 us at 104 x 104. YOLO_WINO2_MAXPIX=0 switches it off.
static const long long g_wino2_maxpix = getenv("YOLO_WINO2_MAXPIX") ? atoll(getenv("YOLO_WINO2_MAXPIX")) : 256;

size_t wino_weight_elems(int cout, int cin, int ks) {
    if (ks != 3 || cin % 4) return 0;
    return (size_t)16 * (cin / 4) * round_up(cout, 64) * 4;
}

int wino_pack(const float* w_oihw, float* U, int cout, int cin, hipStream_t s) {
    const long long total = (long long)wino_weight_elems(cout, cin, 3);
    if (!total) return fail(YOLO_ERR_ARG, "wino_pack: cin %% 4");
    const int grid = (int)((total + 255) / 256 < 8192 ? (total + 255) / 256 : 8192);
    hipLaunchKernelGGL(wino_pack_f32, dim3(grid), dim3(256), 0, s, w_oihw, U, cout, cin, round_up(cout, 64), total, 0, cin);
    return check_launch("wino_pack_f32");
}

// filters of the input-gradient convolution (3x3 stride 1): dx = conv(dz, g'), rows = the forward's cin, K = the forward's cout padded to 32
int wino_pack_dgrad(const float* w_oihw, float* U, int cout, int cin, hipStream_t s) {
    const int coutp = round_up(cout, 32);
    const long long total = (long long)wino_weight_elems(cin, coutp, 3);
    if (!total) return fail(YOLO_ERR_ARG, "wino_pack_dgrad: shape");
    const int grid = (int)((total + 255) / 256 < 8192 ? (total + 255) / 256 : 8192);
    hipLaunchKernelGGL(wino_pack_f32, dim3(grid), dim3(256), 0, s, w_oihw, U, cin, coutp, round_up(cin, 64), total, 1, cout);
    return check_launch("wino_pack_f32(dgrad)");
}

// what the kernels can run at all (tile 13 forces it on anything that passes)
bool wino_supported(const yolo_conv_desc* d) {
    if (d->dtype != YOLO_F32 || d->ksize != 3 || d->stride != 1 || d->out_mode != YOLO_OUT_NHWC) return false;
    if (d->cin % 4 || d->cout % 4) return false;
    if ((d->x_ld & 3) || (d->x_off & 3) || (d->y_ld & 3) || (d->y_off & 3)) return false;
    if ((d->flags & YOLO_FLAG_RESIDUAL) && ((d->r_ld & 3) || (d->r_off & 3))) return false;
    const long long T = (long long)d->n * ((d->h + 1) / 2) * ((d->w + 1) / 2);
    if (T + 64 > 0x7fffffffLL / 4 || (long long)d->n * d->h * d->w > 0x7fffffffLL) return false;
    return true;
}

// the heuristic: where the 16 GEMMs are long enough to pay for the transform pass and the 4x larger operand stream.
// Looks at the layer's shape only, never at the batch size (an image's result may not depend on its neighbours).
bool wino_eligible(const yolo_conv_desc* d) {
    if (g_wino_off || !wino_supported(d)) return false;
    // measured at batch 32 (tools/conv_bench.py --tile 13 / 7): 64 -> 128 at 104 x 104 387 vs 433-460 us, 128 -> 256 at 52 x 52 353 vs
    // 425-470, 256 -> 512 at 26 x 26 274 vs 440-515, 512 -> 1024 at 13 x 13 313-323 vs 430; 32 -> 64 at 208 x 208 would be 8 stages per
    // workgroup behind a 709 MB transform pass: stays direct
    return d->cin >= 64 && d->cout >= 64;
}

size_t wino_workspace_bytes(const yolo_conv_desc* d) {
    if (!wino_supported(d)) return 0;
    const long long T = (long long)d->n * ((d->h + 1) / 2) * ((d->w + 1) / 2);
    return (size_t)round_up((int)T, 64) * d->cin * 16 * sizeof(float);
}

int conv_wino_launch(const yolo_conv_desc* d, const void* x, const float* U, const float* scale, const float* shift,
                     const void* residual, void* y, void* workspace, size_t workspace_bytes, int32_t* nan_flag, hipStream_t s) {
    if (!wino_supported(d)) return fail(YOLO_ERR_UNSUPPORTED, "conv winograd: needs fp32 3x3 stride 1, NHWC output, channels %% 4 == 0");
    const size_t need = wino_workspace_bytes(d);
    if (!workspace || workspace_bytes < need) return fail(YOLO_ERR_WORKSPACE, "conv winograd: workspace %zu < %zu bytes", workspace_bytes, need);
    if ((size_t)workspace & 15) return fail(YOLO_ERR_ARG, "conv winograd: workspace must be 16-byte aligned");
    const int th = (d->h + 1) / 2, tw = (d->w + 1) / 2;
    const int T = d->n * th * tw, Tpad = round_up(T, 64), C4 = d->cin / 4;
    WinoXArgs xa;
    xa.x = (const float*)x; xa.V = (float*)workspace; xa.H = d->h; xa.W = d->w; xa.C4 = C4; xa.x_ld = d->x_ld; xa.x_off = d->x_off;
    xa.th = th; xa.tw = tw; xa.T = T; xa.Tpad = Tpad;
    hipLaunchKernelGGL(wino_xform_f32, dim3(ceil_div(C4, 8), Tpad / 32), dim3(256), 0, s, xa);
    if (int rc = check_launch("wino_xform_f32")) return rc;

    WinoArgs a;
    a.V = (const float*)workspace; a.U = U; a.scale = scale; a.shift = shift; a.res = (const float*)residual; a.y = (float*)y;
    a.nan_flag = nan_flag;
    a.T = T; a.Tpad = Tpad; a.C4 = C4; a.Cout = d->cout; a.CoutPad = round_up(d->cout, 64);
    a.th = th; a.tw = tw; a.H = d->h; a.W = d->w;
    a.y_ld = d->y_ld; a.y_off = d->y_off; a.r_ld = d->r_ld; a.r_off = d->r_off; a.flags = d->flags;
    a.n_mt = Tpad / 64; a.n_nt = a.CoutPad / 64;
    const int grid = 8 * a.n_nt * ceil_div(a.n_mt, 8);
    const bool res = d->flags & YOLO_FLAG_RESIDUAL;
    if (d->tile == 14 || (d->tile != 13 && d->tile < 16 && (long long)d->h * d->w <= g_wino2_maxpix)) {       // two passes over xi, two workgroups per CU (conv_wino2_f32)
        const size_t lds2 = (size_t)256 * 68 * 4 + 512;                        // the store staging (> the 64 KiB ring)
        auto go2 = [&](auto kern) -> int {
            static LdsOnce once;
            if (int rc = reserve_lds(once, reinterpret_cast<const void*>(kern), lds2, "conv_wino2_f32")) return rc;
            hipLaunchKernelGGL(kern, dim3(grid), dim3(256), lds2, s, a);
            return check_launch("conv_wino2_f32");
        };
        YOLO_SWITCH_ACT(d->act, return res ? go2(&conv_wino2_f32<ACT, true>) : go2(&conv_wino2_f32<ACT, false>));
        return fail(YOLO_ERR_ARG, "conv winograd: activation");
    }
    const size_t lds = (size_t)WN_SLOTS * WN_STAGE;
    auto go = [&](auto kern) -> int {
        static LdsOnce once;
        if (int rc = reserve_lds(once, reinterpret_cast<const void*>(kern), lds, "conv_wino_f32")) return rc;
        hipLaunchKernelGGL(kern, dim3(grid), dim3(256), lds, s, a);
        return check_launch("conv_wino_f32");
    };
#ifdef WN_STAMPS
    switch (d->tile >= 16 ? d->tile - 16 : 0) {          // timing probes: tile 16 + bits
    case 1: return go(&conv_wino_f32<YOLO_ACT_LEAKY, true, 1>);
    case 2: return go(&conv_wino_f32<YOLO_ACT_LEAKY, true, 2>);
    case 3: return go(&conv_wino_f32<YOLO_ACT_LEAKY, true, 3>);
    case 4: return go(&conv_wino_f32<YOLO_ACT_LEAKY, true, 4>);
    case 7: return go(&conv_wino_f32<YOLO_ACT_LEAKY, true, 7>);
    case 8: return go(&conv_wino_f32<YOLO_ACT_LEAKY, true, 8>);
    case 9: return go(&conv_wino_f32<YOLO_ACT_LEAKY, true, 9>);
    case 11: return go(&conv_wino_f32<YOLO_ACT_LEAKY, true, 11>);
    default: break;
    }
#endif
    YOLO_SWITCH_ACT(d->act, return res ? go(&conv_wino_f32<ACT, true>) : go(&conv_wino_f32<ACT, false>));
    return fail(YOLO_ERR_ARG, "conv winograd: activation");
}

}  // namespace yolo
