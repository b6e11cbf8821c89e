// conv1_rs_f32.hip — the fp32 1x1 blocks (model.py:80-86 with kernel_size = 1: residual-unit entries, neck layers) with the
// WEIGHTS STATIONARY IN REGISTERS.
//
// Round 2's 1x1 kernel (conv_igemm_f32<64,64>: register-staged 64 x 64 tiles, one barrier per 32-wide K step) ran the 37 1x1
// launches of the fp32 forward at 79-86 TF = 0.54 of the f32 matrix peak, reading its input 1.9 x (one pass per 64-channel N
// tile); an LDS-DMA 128 x 128 tile did not help (three rounds of big tiles per CU). What a 1x1 layer looks like on this chip:
// v_mfma_f32_32x32x2_f32 takes ONE float per lane and operand and occupies the matrix pipe for 64 cycles - an operand stream of
// 8 bytes per lane per 64 cycles. So the operands are nowhere near a bandwidth problem; the per-tile fixed costs are. And the
// weight matrix of a 1x1 layer is small: a wave's 32 output channels x K input channels are K / 2 registers per lane
// (K = 256: 128 of the 512 a lone wave per SIMD may use).
//   * one persistent 256-thread workgroup per CU; wave w keeps the weights of ITS 32 output channels (x its K part) in
//     registers for the whole launch and walks 32-pixel tiles of the input: blockIdx -> (channel group, first tile), then
//     tile += stride. Fine-grained tiles (8k-16k matrix cycles) instead of three rounds of 33k-cycle tiles: 96 % of the last
//     round is filled at 52x52.
//   * the activations are the only stream: 32 pixels x K floats per tile by LDS-DMA (global_load_lds_dwordx4) into a
//     3-deep ring, read back ONCE per wave with ds_read_b128 (rows are XOR-swizzled by pixel & 15 on the DMA's source
//     side, so the 16 lanes of a read group hit 16 different 16-byte slots): the input is read from HBM exactly once per
//     channel group, the weights once per launch.
//   * K pairing: an MFMA consumes k = (lane >> 5) + 2 t. Lanes 0-31 take chunk g (4 consecutive channels) and lanes 32-63
//     chunk g + K/8 of their pixel's row, so one 16-byte LDS read feeds four MFMAs on both halves; the weight registers
//     are loaded in the same order. One accumulation chain of K/2 dependent MFMAs (issue interval = dependent latency = 64).
//   * weights are the MFMA's A operand, pixels its B operand: D = [channel][pixel], a lane owns one pixel and four runs of
//     four consecutive channels -> 16-byte stores / residual loads straight from registers.
//   * K = 256 / 384 / 512 (31 - 8 of the 37 launches at 80 classes): a 32-pixel tile of K = 768 / 1024 does not fit a ring in
//     160 KiB, those layers (13x13 / 26x26 neck) and the 64- / 32-channel ones stay on conv_igemm_f32.
// Same arithmetic as the other fp32 kernels (an f32 MFMA chain is a k-ordered fmaf chain); only the order of k differs.
#include <type_traits>
#include "common.h"

namespace yolo {

typedef float f32x16 __attribute__((ext_vector_type(16)));

struct Conv1RsArgs {
    const float* x;
    const float* w;                  // row-major [cout_pad128][Kpad] (the register-staged kernel's layout, ksize 1: k = ci)
    const float* scale;
    const float* shift;
    const float* res;
    float* y;
    int* nan_flag;
    int M, K, Cout, Kpad;
    int x_ld, x_off, y_ld, y_off, r_ld, r_off;
    int act, out_mode, flags;
    int Ho, Wo;
    int ngroups, tiles, tile_stride;  // channel groups; 32-pixel tiles; workgroups per channel group
};

__device__ __attribute__((aligned(256))) unsigned int g_rs_zero[64];

typedef const __attribute__((address_space(1))) void* rs_gptr;
typedef __attribute__((address_space(3))) void* rs_lptr;
__device__ __forceinline__ void rs_glds16(const void* g, void* l) { __builtin_amdgcn_global_load_lds((rs_gptr)g, (rs_lptr)l, 16, 0, 0); }
template <int N> __device__ __forceinline__ void rs_wait_vmcnt() { asm volatile("s_waitcnt vmcnt(%0)" ::"n"(N) : "memory"); }

template <int I, int N, class Fn>
__device__ __forceinline__ void rs_for(Fn&& fn) {
    if constexpr (I < N) {
        fn(std::integral_constant<int, I>{});
        rs_for<I + 1, N>(fn);
    }
}

// K = input channels (256 / 384: 3-slot ring; 512: 2 slots of 64 KiB); four channel blocks of 32 per workgroup
template <int K, int ACT, bool RES>
__global__ __launch_bounds__(256, 1) void conv1_rs_f32(const Conv1RsArgs p) {
    constexpr int WR = K / 2;                           // weight registers per lane
    constexpr int NG = K / 8;                           // 16-byte reads per lane and tile
    constexpr int ROWB = K * 4;                         // bytes per pixel row in LDS
    constexpr int TILE_B = 32 * ROWB;
    constexpr int NDMA = TILE_B / 1024 / 4;             // 1 KiB wave-instructions per wave and tile
    constexpr int SLOTS = K <= 384 ? 3 : 2;
    constexpr int AHEAD = SLOTS - 1;                    // tiles requested ahead of the one being multiplied
    static_assert(K % 128 == 0 && TILE_B % 4096 == 0, "chunk pairing needs K/8 to be a multiple of 16");
    extern __shared__ __attribute__((aligned(16))) char smem[];

    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int h = lane >> 5, m = lane & 31;
    const int grp = blockIdx.x % p.ngroups;
    const int first = blockIdx.x / p.ngroups;
    const int chb = grp * 128 + wave * 32;                           // first output channel of this wave

    // ---- the weights of this wave: W[chb + m][k], k in the pairing order of the header
    float wreg[WR];
    {
        const float* wr = p.w + (size_t)(chb + m) * p.Kpad + h * (K / 2);     // rows beyond cout are zero in the packed matrix (cout_pad128)
#pragma unroll
        for (int g = 0; g < NG; ++g) {
            const f32x4 v = *reinterpret_cast<const f32x4*>(wr + 4 * g);
            wreg[4 * g] = v[0]; wreg[4 * g + 1] = v[1]; wreg[4 * g + 2] = v[2]; wreg[4 * g + 3] = v[3];
        }
    }
    // folded scale / shift of this lane's 16 output channels: 8 q + 4 h + {0..3}, q = 0..3
    f32x4 sc[4], sh[4];
#pragma unroll
    for (int q = 0; q < 4; ++q) {
        const int c = chb + 8 * q + 4 * h;
#pragma unroll
        for (int e = 0; e < 4; ++e) {
            const int cc = c + e < p.Cout ? c + e : p.Cout - 1;
            sc[q][e] = p.scale[cc];
            sh[q][e] = p.shift[cc];
        }
    }

    // ---- DMA roles: piece j of a tile = rows / slots of wave-instruction 4 j + wave
    int d_row[NDMA], d_off[NDMA];
#pragma unroll
    for (int j = 0; j < NDMA; ++j) {
        const int flat = (4 * j + wave) * 64 + lane;                 // 16-byte slot index inside the tile image
        const int row = flat / (K / 4), slot = flat % (K / 4);
        d_row[j] = row;
        d_off[j] = (slot ^ (row & 15)) * 4;                          // source chunk of LDS slot `slot` of row `row`
    }
    auto issue = [&](int tile, int buf) {
        const bool tv = tile < p.tiles;
        char* dst = smem + buf * TILE_B + wave * 1024;
#pragma unroll
        for (int j = 0; j < NDMA; ++j) {
            int pix = tile * 32 + d_row[j];
            pix = pix < p.M ? pix : p.M - 1;                         // the last tile repeats its last pixel (discarded below)
            const float* src = tv ? p.x + (size_t)pix * p.x_ld + p.x_off + d_off[j] : reinterpret_cast<const float*>(g_rs_zero);
            rs_glds16(src, dst + j * 4096);
        }
    };
    // this lane's reads: row m, chunk g + h K/8, slot = chunk ^ (m & 15) (K/8 is a multiple of 16: the XOR touches g's low 4 bits)
    const unsigned rbase = (unsigned)(size_t)(rs_lptr)smem + m * ROWB;
    const int cbase = h * (K / 8);
    const int sw = m & 15;

    // epilogue of one tile from registers: this lane's pixel, 4 x 4 consecutive channels
    bool saw_nan = false;
    auto epilogue = [&](const f32x16& acc, int tile) {
        const int pix = tile * 32 + m;
        if (pix >= p.M || tile < 0) return;
        size_t ooff;
        if (p.out_mode == YOLO_OUT_NHWC) {
            ooff = (size_t)pix * p.y_ld + p.y_off;
        } else {                                                  // 2x nearest upsample into the concat buffer
            const int HoWo = p.Ho * p.Wo;
            const int img = pix / HoWo, rem = pix - img * HoWo;
            const int ho = rem / p.Wo, wo = rem - ho * p.Wo;
            ooff = ((size_t)(img * 2 * p.Ho + 2 * ho) * (2 * p.Wo) + 2 * wo) * p.y_ld + p.y_off;
        }
#pragma unroll
        for (int q = 0; q < 4; ++q) {
            const int c = chb + 8 * q + 4 * h;
            if (c >= p.Cout) continue;                            // cout is a multiple of 4 here (checked on the host)
            f32x4 v;
#pragma unroll
            for (int e = 0; e < 4; ++e) v[e] = act_c<ACT>(acc[4 * q + e] * sc[q][e] + sh[q][e]);
            if (RES) {
                const f32x4 r4 = *reinterpret_cast<const f32x4*>(p.res + (size_t)pix * p.r_ld + p.r_off + c);
#pragma unroll
                for (int e = 0; e < 4; ++e) v[e] += r4[e];
            }
            saw_nan |= (v[0] != v[0]) | (v[1] != v[1]) | (v[2] != v[2]) | (v[3] != v[3]);
            float* d = p.y + ooff + c;
            *reinterpret_cast<f32x4*>(d) = v;
            if (p.out_mode != YOLO_OUT_NHWC) {
                const size_t W2 = 2 * (size_t)p.Wo;
                *reinterpret_cast<f32x4*>(d + p.y_ld) = v;
                *reinterpret_cast<f32x4*>(d + W2 * p.y_ld) = v;
                *reinterpret_cast<f32x4*>(d + (W2 + 1) * p.y_ld) = v;
            }
        }
    };

    int tile = first;
    issue(tile, 0);
    if (AHEAD == 2) issue(tile + p.tile_stride, 1);
    rs_wait_vmcnt<(AHEAD - 1) * NDMA>();                             // (ordinary loads above included: one in-order counter)
    __builtin_amdgcn_s_barrier();
    __builtin_amdgcn_sched_barrier(0);
    int buf = 0;
    f32x16 prev;                                                     // accumulators of the previous tile: its epilogue runs beside this tile's MFMAs
#pragma unroll
    for (int r = 0; r < 16; ++r) prev[r] = 0.f;
    int prev_tile = -1;
    for (; tile < p.tiles; tile += p.tile_stride) {
        __builtin_amdgcn_sched_barrier(0);
        {   // ring slot (buf + AHEAD) % SLOTS was read in the previous round: every wave has passed that round's barrier
            int nb = buf + AHEAD;
            nb = nb >= SLOTS ? nb - SLOTS : nb;
            issue(tile + AHEAD * p.tile_stride, nb);
        }
        // 3 slots: everything older than the requests just made - the pieces of the NEXT tile (requested a round ago) and the
        // previous epilogue's stores - is a whole round old: this wait costs nothing and the end of the round needs none
        if (AHEAD == 2) rs_wait_vmcnt<NDMA>();
        const unsigned tb = rbase + buf * TILE_B;
        f32x16 acc;
#pragma unroll
        for (int r = 0; r < 16; ++r) acc[r] = 0.f;
        // reads in inline asm (their completion is counted here: LDS operations return in order), one 16-byte read ahead
        f32x4 frag[2];
        {
            const unsigned addr = tb + (((cbase + 0) ^ sw) << 4);
            asm volatile("ds_read_b128 %0, %1" : "=&v"(frag[0]) : "v"(addr));
        }
        rs_for<0, NG>([&](auto G) {
            constexpr int g = decltype(G)::value;
            if constexpr (g + 1 < NG) {
                const unsigned addr = tb + (((cbase + g + 1) ^ sw) << 4);
                asm volatile("ds_read_b128 %0, %1" : "=&v"(frag[(g + 1) & 1]) : "v"(addr));
                asm volatile("s_waitcnt lgkmcnt(1)" : "+v"(frag[g & 1]));
            } else {
                asm volatile("s_waitcnt lgkmcnt(0)" : "+v"(frag[g & 1]));
            }
#pragma unroll
            for (int j = 0; j < 4; ++j) acc = __builtin_amdgcn_mfma_f32_32x32x2f32(wreg[4 * g + j], frag[g & 1][j], acc, 0, 0, 0);
            // the previous tile's epilogue (~150 vector instructions, 4-16 stores) in the shadow of this tile's first MFMAs: a
            // 32x32x2 f32 MFMA holds the matrix pipe for 64 cycles and the issue port for a few
            if constexpr (g == 0) epilogue(prev, prev_tile);
        });
        prev = acc;
        prev_tile = tile;
        // 2 slots: the pieces of the next tile were requested at the top of THIS round and must have landed (the epilogue's
        // stores, issued right after them, are long done); then everybody is done reading slot buf
        if (AHEAD == 1) rs_wait_vmcnt<0>();
        __builtin_amdgcn_s_barrier();
        __builtin_amdgcn_sched_barrier(0);
        buf = buf + 1 == SLOTS ? 0 : buf + 1;
    }
    epilogue(prev, prev_tile);
    if ((p.flags & YOLO_FLAG_NANCHECK) && saw_nan) atomicOr(p.nan_flag, 2);
}

bool conv1_rs_eligible(const yolo_conv_desc* d, const void* residual) {
    static const bool off = getenv("YOLO_NO_CONV1_RS") != nullptr;   // A/B switch: round 2's kernel
    if (off || d->dtype != YOLO_F32 || d->ksize != 1 || d->stride != 1 || d->out_mode == YOLO_OUT_HEAD) return false;
    if (d->cin != 256 && d->cin != 384 && d->cin != 512) return false;          // K = 768 / 1024: a 32-pixel tile does not fit a ring
    if (d->cout % 128) return false;
    // persistent workgroups of 32-pixel tiles: below ~4 tiles per workgroup the start-up (K / 2 weight registers per lane, the
    // first ring fill) and the last, partly filled round cost more than the register-staged kernel's tiles (measured at
    // B = 32: 13x13 512->256 28.8 -> 40.3 us, 26x26 256->128 25.9 -> 35.6 us), and with K = 512 (16k-cycle tiles, 2-slot ring)
    // it measured 71 vs 67 us at 26x26: the heuristic keeps K <= 384 and feature maps of >= 2048 pixels.
    // The rule must NOT look at the batch size: this kernel adds the K products in another order than conv_patch_f32 (a lane
    // holds 4 consecutive k of its row), and an image's result may not depend on how many neighbours share its batch
    // (tests/test_gpu_fullsize.py: image 17 of 32 == the same image alone, bit for bit).
    if (d->tile != 12 && ((long long)d->h * d->w < 2048 || d->cin > 384)) return false;
    if (d->tile != 12 && residual) return false;            // with a residual row per store it measured slower (79.8 vs 74.9 us at 52 x 52)
    if ((d->x_ld & 3) || (d->x_off & 3) || (d->y_ld & 3) || (d->y_off & 3)) return false;
    if (residual && ((d->r_ld & 3) || (d->r_off & 3))) return false;
    return true;
}

template <int K>
static int launch_rs(Conv1RsArgs& a, hipStream_t s) {
    const size_t lds = (size_t)(K <= 384 ? 3 : 2) * 32 * K * 4;
    a.ngroups = ceil_div(a.Cout, 128);
    a.tiles = ceil_div(a.M, 32);
    int per = 256 / a.ngroups;                              // one workgroup per CU
    if (per < 1) per = 1;
    if (per > a.tiles) per = a.tiles;
    a.tile_stride = per;
    const int grid = per * a.ngroups;
    const bool res = a.flags & YOLO_FLAG_RESIDUAL;
    auto go = [&](auto kern) -> int {
        static LdsOnce once;
        if (int rc = reserve_lds(once, reinterpret_cast<const void*>(kern), lds, "conv1_rs_f32")) return rc;
        hipLaunchKernelGGL(kern, dim3(grid), dim3(256), lds, s, a);
        return check_launch("conv1_rs_f32");
    };
    YOLO_SWITCH_ACT(a.act, return res ? go(&conv1_rs_f32<K, ACT, true>) : go(&conv1_rs_f32<K, ACT, false>));
    return fail(YOLO_ERR_ARG, "conv1_rs_f32: activation");
}

int conv1_rs_launch(const yolo_conv_desc* d, const void* x, const void* w, const float* scale, const float* shift, const void* residual,
                    void* y, int32_t* nan_flag, hipStream_t s) {
    Conv1RsArgs a;
    a.x = (const float*)x; a.w = (const float*)w; a.scale = scale; a.shift = shift; a.res = (const float*)residual; a.y = (float*)y;
    a.nan_flag = nan_flag;
    const long long M = (long long)d->n * d->h * d->w;
    if (M > 0x7fffffffLL) return fail(YOLO_ERR_UNSUPPORTED, "conv: N*H*W exceeds int32");
    a.M = (int)M; a.K = d->cin; a.Cout = d->cout; a.Kpad = kpad_of(d->cin, 1);
    a.x_ld = d->x_ld; a.x_off = d->x_off; a.y_ld = d->y_ld; a.y_off = d->y_off; a.r_ld = d->r_ld; a.r_off = d->r_off;
    a.act = d->act; a.out_mode = d->out_mode; a.flags = d->flags; a.Ho = d->h; a.Wo = d->w;
    switch (d->cin) {
    case 256: return launch_rs<256>(a, s);
    case 384: return launch_rs<384>(a, s);
    case 512: return launch_rs<512>(a, s);
    }
    return fail(YOLO_ERR_UNSUPPORTED, "conv1_rs_f32: cin %d", d->cin);
}

}  // namespace yolo
