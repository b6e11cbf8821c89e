// wgrad_stem_h16.hip — weight gradient of the network's FIRST block (3x3, stride 1, <= 3 input channels, <= 32 output channels),
// 16-bit operands.
//
// Replaces the autograd backward of nn.Conv2d w.r.t. its weight for layers[0] (reference: code/train.py:67 under the autocast of
// train.py:53; conv definition code/model.py:60, layer list model.py:20):
//     dW[co][ci][kh][kw] = sum_{n,h,w} dz[n,h,w,co] * x[n, h+kh-1, w+kw-1, ci]
// GEMM view: M = 32 output channels, N = Cin * 9 = 27 (padded to 32), K = all N*H*W pixels (5.5 M at batch 32, 416^2): ONE
// 32 x 32 MFMA tile under a reduction over the whole batch. wgrad_patch_h16's 64 x 64 x 9-tap block tile is 1/32 full for this
// layer and pays two barriers and an exposed staging round trip per 64 pixels: 345 us for a layer whose bytes (dz: 354 MB) take
// ~70 us. Here every WAVE is an independent worker with no block-level synchronisation in its loop:
//   * K step = 16 consecutive pixels of one image row; four of them form a batch;
//   * dz operand (M x K, channels x pixels): the 16 x 32-channel rows of a step are ONE LDS-DMA instruction (64 lanes x 16 B)
//     into the wave's private slot, read back with the transposing ds_read_b64_tr_b16 (addressing of wgrad_h16.hip);
//   * x operand (K x N, pixels x (ci, tap)): the K step's input patch (3 rows x 18 pixels x 16 B) is ONE more LDS-DMA instruction
//     (54 lanes; pixels outside the image come from a zero page, so no border case is left), and lane (n, k-half) picks its 8
//     pixels of channel ci at tap (dh, dw) with eight ds_read_u16 at immediate offsets.
//     (First version: eight global two-byte gathers per lane and K step - 29 TA cycles per instruction, 217 us, and slower the
//     more waves shared a CU.)
//   * batches b + 1 and b + 2 (8 DMA instructions each) are in flight while batch b is multiplied: one counted s_waitcnt per batch.
// The four waves of a workgroup add their 32 x 32 results through LDS, the workgroups' partials are summed in a fixed order by
// stem_wgrad_reduce: deterministic, no float atomics.
#include <type_traits>
#include "common.h"

namespace yolo {

typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef unsigned int u32x4 __attribute__((ext_vector_type(4)));
typedef unsigned int u32x2 __attribute__((ext_vector_type(2)));
typedef short s16x8 __attribute__((ext_vector_type(8)));
typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));
typedef _Float16 f16x8 __attribute__((ext_vector_type(8)));

constexpr int SW_NSEG = 4;                        // K steps (16 pixels each) per batch
constexpr int SW_XLD = 8;                         // halfs per input pixel (the 16-bit input buffer of a 3-channel network)
constexpr int SW_RING = 3;                        // batches per wave in LDS: two in flight behind the one being multiplied
constexpr int SW_SEG = 2048;                      // bytes per K step: [16 px][64 B] dz rows | [3 rows][18 px][16 B] input patch (864 B)
constexpr int SW_SLOT = SW_NSEG * SW_SEG;         // one batch
constexpr int SW_WAVE_LDS = SW_RING * SW_SLOT;    // 24 KiB per wave, 96 KiB per workgroup: one workgroup per CU
constexpr int SW_VMOPS = SW_NSEG * 2;             // LDS-DMA instructions a wave issues per batch

struct StemWgArgs {
    const unsigned short* dz;
    const unsigned short* x;
    float* partial;                               // [workgroup][32 co][32 n]
    int H, W, cin;
    int dz_ld, dz_off, x_off;
    int segs_per_row, total_segs, segs_per_wave;
    unsigned mg_spr, mg_H;
};

__device__ __attribute__((aligned(256))) unsigned int g_sw_zero[64];     // 256 B of zeros: a whole fragment's 8 pixels

typedef const __attribute__((address_space(1))) void* sw_gptr;
typedef __attribute__((address_space(3))) void* sw_lptr;

__device__ __forceinline__ int sw_fdiv(int x, unsigned mg, int d) {
    if (!mg) return x;
    const unsigned q = __umulhi((unsigned)x, mg);
    return (int)(q * (unsigned)d > (unsigned)x ? q - 1 : q);
}
static unsigned sw_magic(int d) { return d <= 1 ? 0u : (unsigned)((0x100000000ULL + (unsigned)d - 1) / (unsigned)d); }

template <typename T> __device__ __forceinline__ f32x16 sw_mfma(s16x8 a, s16x8 b, f32x16 c);
template <> __device__ __forceinline__ f32x16 sw_mfma<__bf16>(s16x8 a, s16x8 b, f32x16 c) {
    return __builtin_amdgcn_mfma_f32_32x32x16_bf16(__builtin_bit_cast(bf16x8, a), __builtin_bit_cast(bf16x8, b), c, 0, 0, 0);
}
template <> __device__ __forceinline__ f32x16 sw_mfma<_Float16>(s16x8 a, s16x8 b, f32x16 c) {
    return __builtin_amdgcn_mfma_f32_32x32x16_f16(__builtin_bit_cast(f16x8, a), __builtin_bit_cast(f16x8, b), c, 0, 0, 0);
}

struct SwFrag { u32x2 lo, hi; };
// two transposing reads = the 8 consecutive k (pixels) of this lane's channel; in asm because hipcc puts s_waitcnt vmcnt(0) in
// front of the builtin form while an LDS-DMA is pending (wgrad_dma_h16.hip)
template <int OFF>
__device__ __forceinline__ void sw_read(SwFrag& f, unsigned addr) {
    asm volatile("ds_read_b64_tr_b16 %0, %2 offset:%3\n\tds_read_b64_tr_b16 %1, %2 offset:%4"
                 : "=&v"(f.lo), "=&v"(f.hi) : "v"(addr), "n"(OFF), "n"(OFF + 4 * 64));
}
// the lane's 8 pixels of one channel, 16 bytes apart in the patch: eight zero-extending 16-bit reads, packed after the wait
// (ds_read_u16_d16 / _d16_hi would pack for free, but with SRAM-ECC on - MI300 / MI355X - a d16 load clears the other half)
struct SwX { unsigned v[8]; };
template <int OFF>
__device__ __forceinline__ void sw_xread(SwX& x, unsigned addr) {
    asm volatile("ds_read_u16 %0, %8 offset:%9\n\tds_read_u16 %1, %8 offset:%10\n\tds_read_u16 %2, %8 offset:%11\n\t"
                 "ds_read_u16 %3, %8 offset:%12\n\tds_read_u16 %4, %8 offset:%13\n\tds_read_u16 %5, %8 offset:%14\n\t"
                 "ds_read_u16 %6, %8 offset:%15\n\tds_read_u16 %7, %8 offset:%16"
                 : "=&v"(x.v[0]), "=&v"(x.v[1]), "=&v"(x.v[2]), "=&v"(x.v[3]), "=&v"(x.v[4]), "=&v"(x.v[5]), "=&v"(x.v[6]), "=&v"(x.v[7])
                 : "v"(addr), "n"(OFF + 0 * 16), "n"(OFF + 1 * 16), "n"(OFF + 2 * 16), "n"(OFF + 3 * 16), "n"(OFF + 4 * 16),
                   "n"(OFF + 5 * 16), "n"(OFF + 6 * 16), "n"(OFF + 7 * 16));
}
__device__ __forceinline__ void sw_wait_x(SwX& x) {
    asm volatile("" : "+v"(x.v[0]), "+v"(x.v[1]), "+v"(x.v[2]), "+v"(x.v[3]), "+v"(x.v[4]), "+v"(x.v[5]), "+v"(x.v[6]), "+v"(x.v[7]));
}
__device__ __forceinline__ void sw_wait_f(SwFrag& a, SwFrag& b, SwFrag& c, SwFrag& d) {
    asm volatile("s_waitcnt lgkmcnt(0)" : "+v"(a.lo), "+v"(a.hi), "+v"(b.lo), "+v"(b.hi), "+v"(c.lo), "+v"(c.hi), "+v"(d.lo), "+v"(d.hi));
}
__device__ __forceinline__ s16x8 sw_pack(const SwX& x) {
    const u32x4 v = {x.v[0] | (x.v[1] << 16), x.v[2] | (x.v[3] << 16), x.v[4] | (x.v[5] << 16), x.v[6] | (x.v[7] << 16)};
    return __builtin_bit_cast(s16x8, v);
}
__device__ __forceinline__ s16x8 sw_vec(const SwFrag& f) {
    const u32x4 v = {f.lo[0], f.lo[1], f.hi[0], f.hi[1]};
    return __builtin_bit_cast(s16x8, v);
}

template <typename T>
__global__ __launch_bounds__(256) void stem_wgrad_h16(const StemWgArgs p) {
    extern __shared__ __attribute__((aligned(256))) char smem[];      // [4 waves][2 batches][SW_SLOT]; reused for the final reduction
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    char* my = smem + wave * SW_WAVE_LDS;
    const int gw = blockIdx.x * 4 + wave;
    int s0 = gw * p.segs_per_wave;
    int s1 = s0 + p.segs_per_wave < p.total_segs ? s0 + p.segs_per_wave : p.total_segs;
    if (s0 > s1) s0 = s1;
    const int nb = (s1 - s0 + SW_NSEG - 1) / SW_NSEG;

    // this lane's column n = (ci, tap) of the x operand and its k half
    const int n = lane & 31, kh = lane >> 5;
    const bool ncol = n < p.cin * 9;
    const int ci = ncol ? n / 9 : 0, tap = ncol ? n - 9 * (n / 9) : 4;
    const int dh = tap / 3, dw = tap - 3 * dh;
    // dz operand: DMA source offset of this lane inside a K step (pixel lane >> 2, 16-byte piece lane & 3) and fragment address
    const int dma_off = (lane >> 2) * p.dz_ld + (lane & 3) * 8;
    const int lg = lane & 15;
    const unsigned a_addr = (unsigned)(size_t)(sw_lptr)my + (8 * kh + (lg >> 2)) * 64 + (16 * ((lane >> 4) & 1) + 4 * (lg & 3)) * 2;
    const unsigned short* zp = reinterpret_cast<const unsigned short*>(g_sw_zero);

    // x operand: lane (n = (ci, tap), k half) reads its 8 pixels from the K step's patch [3 rows][18 columns][16 B] in LDS
    const unsigned b_addr = (unsigned)(size_t)(sw_lptr)my + 1024 + ((dh * 18 + 8 * kh + dw) * SW_XLD + ci) * 2;
    // patch DMA: lane l < 54 fetches pixel (row l / 18 - 1, column l % 18 - 1) relative to the K step; outside the image: zeros
    const int prow = lane / 18, pcol = lane - 18 * prow;
    const bool plane = lane < 54;

    auto request = [&](int b, int slot) {
#pragma unroll
        for (int g = 0; g < SW_NSEG; ++g) {
            int s = s0 + b * SW_NSEG + g;
            const bool live = s < s1;
            s = live ? s : s1 - 1;
            const int row = sw_fdiv(s, p.mg_spr, p.segs_per_row), w0 = (s - row * p.segs_per_row) * 16;      // row = img * H + h
            const int h = row - sw_fdiv(row, p.mg_H, p.H) * p.H;
            const size_t pix = (size_t)row * p.W + w0;
            char* dst = my + slot * SW_SLOT + g * SW_SEG;
            __builtin_amdgcn_global_load_lds((sw_gptr)(p.dz + pix * p.dz_ld + p.dz_off + dma_off), (sw_lptr)dst, 16, 0, 0);
            const int hh = h + prow - 1, ww = w0 + pcol - 1;
            const bool ok = live && hh >= 0 && hh < p.H && ww >= 0 && ww < p.W;
            const unsigned short* q = ok ? p.x + p.x_off + ((ptrdiff_t)pix + (prow - 1) * p.W + pcol - 1) * SW_XLD : zp;
            if (plane) __builtin_amdgcn_global_load_lds((sw_gptr)q, (sw_lptr)(dst + 1024), 16, 0, 0);
        }
    };

    f32x16 acc;
#pragma unroll
    for (int r = 0; r < 16; ++r) acc[r] = 0.f;

    auto step = [&](int b, auto SLOT) {
        constexpr int slot = decltype(SLOT)::value;
        if (b + 2 < nb) request(b + 2, (slot + 2) % SW_RING);
        // batches b + 1 and b + 2 may stay in flight (loads, stores and LDS-DMA complete in issue order)
        if (b + 2 < nb) asm volatile("s_waitcnt vmcnt(%0)" ::"n"(2 * SW_VMOPS) : "memory");
        else if (b + 1 < nb) asm volatile("s_waitcnt vmcnt(%0)" ::"n"(SW_VMOPS) : "memory");
        else asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        SwFrag f[SW_NSEG];
        SwX bx[SW_NSEG];
        sw_read<slot * SW_SLOT + 0 * SW_SEG>(f[0], a_addr);
        sw_read<slot * SW_SLOT + 1 * SW_SEG>(f[1], a_addr);
        sw_read<slot * SW_SLOT + 2 * SW_SEG>(f[2], a_addr);
        sw_read<slot * SW_SLOT + 3 * SW_SEG>(f[3], a_addr);
        sw_xread<slot * SW_SLOT + 0 * SW_SEG>(bx[0], b_addr);
        sw_xread<slot * SW_SLOT + 1 * SW_SEG>(bx[1], b_addr);
        sw_xread<slot * SW_SLOT + 2 * SW_SEG>(bx[2], b_addr);
        sw_xread<slot * SW_SLOT + 3 * SW_SEG>(bx[3], b_addr);
        sw_wait_f(f[0], f[1], f[2], f[3]);                  // (one lgkmcnt(0) for all 40 reads; the statements below name the others)
        sw_wait_x(bx[0]); sw_wait_x(bx[1]); sw_wait_x(bx[2]); sw_wait_x(bx[3]);
#pragma unroll
        for (int g = 0; g < SW_NSEG; ++g) acc = sw_mfma<T>(sw_vec(f[g]), sw_pack(bx[g]), acc);
    };

    if (nb > 0) request(0, 0);
    if (nb > 1) request(1, 1);
    for (int b = 0; b < nb; b += 3) {
        step(b, std::integral_constant<int, 0>{});
        if (b + 1 < nb) step(b + 1, std::integral_constant<int, 1>{});
        if (b + 2 < nb) step(b + 2, std::integral_constant<int, 2>{});
    }

    // ---- the four waves' tiles -> one partial per workgroup: D[co][n], co = (r & 3) + 8 (r >> 2) + 4 (lane >> 5), n = lane & 31
    asm volatile("s_waitcnt vmcnt(0) lgkmcnt(0)" ::: "memory");
    __syncthreads();
    float* red = reinterpret_cast<float*>(smem);                      // [4][16][64]
#pragma unroll
    for (int r = 0; r < 16; ++r) red[(wave * 16 + r) * 64 + lane] = acc[r];
    __syncthreads();
#pragma unroll
    for (int q = 0; q < 4; ++q) {
        const int idx = q * 256 + tid, r = idx >> 6, l = idx & 63;
        const float s = (red[(0 * 16 + r) * 64 + l] + red[(1 * 16 + r) * 64 + l]) + (red[(2 * 16 + r) * 64 + l] + red[(3 * 16 + r) * 64 + l]);
        const int co = (r & 3) + 8 * (r >> 2) + 4 * (l >> 5);
        p.partial[((size_t)blockIdx.x * 32 + co) * 32 + (l & 31)] = s;
    }
}

// dW[co][ci][kh][kw] = sum over workgroups, fixed order: grid = cout, 256 threads = 32 columns x 8 strided parts
__global__ __launch_bounds__(256) void stem_wgrad_reduce(const float* __restrict__ partial, int nblk, int cin, float* __restrict__ dw) {
    __shared__ float red[8][32];
    const int co = blockIdx.x, n = threadIdx.x & 31, part = threadIdx.x >> 5;
    float s = 0.f;
    for (int b = part; b < nblk; b += 8) s += partial[((size_t)b * 32 + co) * 32 + n];
    red[part][n] = s;
    __syncthreads();
    if (part == 0 && n < cin * 9) {
        float t = 0.f;
#pragma unroll
        for (int q = 0; q < 8; ++q) t += red[q][n];
        dw[(size_t)co * cin * 9 + n] = t;
    }
}

static int stem_wgrad_blocks(long long segs) {
    long long nb = (segs + 4 * SW_NSEG * 4 - 1) / (4 * SW_NSEG * 4);       // at least four batches per wave
    static const int cap = getenv("YOLO_STEM_WGRAD_BLOCKS") ? atoi(getenv("YOLO_STEM_WGRAD_BLOCKS")) : 256;
    if (nb > cap) nb = cap;                                               // one workgroup per CU (96 KiB of LDS each)
    return (int)(nb < 1 ? 1 : nb);
}

bool wgrad_stem_eligible(int n, int h, int w, int cin, int cout, int ksize, int stride, int dz_ld, int dz_off, int x_ld, int x_off) {
    static const bool off = getenv("YOLO_NO_STEM_WGRAD") != nullptr;
    if (off || ksize != 3 || stride != 1 || cin < 1 || cin > 3 || cout < 1 || cout > 32) return false;
    if (x_ld != SW_XLD || (x_off & 7) || dz_ld < 32 || (dz_ld & 7) || (dz_off & 7) || (w & 15) || h < 1) return false;
    return (long long)n * h * w <= 0x7fffffffLL;
}

size_t wgrad_stem_workspace(int n, int h, int w) { return (size_t)stem_wgrad_blocks((long long)n * h * (w / 16)) * 32 * 32 * sizeof(float); }

int wgrad_stem_launch(const void* dz, int dz_ld, int dz_off, const void* x, int x_off, float* workspace, float* dw_oihw, int n, int h,
                      int w, int cin, int cout, int dtype, hipStream_t s) {
    StemWgArgs a;
    a.dz = (const unsigned short*)dz; a.x = (const unsigned short*)x; a.partial = workspace;
    a.H = h; a.W = w; a.cin = cin; a.dz_ld = dz_ld; a.dz_off = dz_off; a.x_off = x_off;
    a.segs_per_row = w / 16;
    const long long segs = (long long)n * h * a.segs_per_row;
    a.total_segs = (int)segs;
    const int nblk = stem_wgrad_blocks(segs);
    a.segs_per_wave = (int)((segs + 4LL * nblk - 1) / (4LL * nblk));
    a.mg_spr = sw_magic(a.segs_per_row); a.mg_H = sw_magic(h);
    const size_t lds = 4 * SW_WAVE_LDS;
    static LdsOnce once_b, once_h;
    if (dtype == YOLO_BF16) {
        if (int rc = reserve_lds(once_b, reinterpret_cast<const void*>(&stem_wgrad_h16<__bf16>), lds, "stem_wgrad_h16")) return rc;
        hipLaunchKernelGGL(stem_wgrad_h16<__bf16>, dim3(nblk), dim3(256), lds, s, a);
    } else {
        if (int rc = reserve_lds(once_h, reinterpret_cast<const void*>(&stem_wgrad_h16<_Float16>), lds, "stem_wgrad_h16")) return rc;
        hipLaunchKernelGGL(stem_wgrad_h16<_Float16>, dim3(nblk), dim3(256), lds, s, a);
    }
    if (int rc = check_launch("stem_wgrad_h16")) return rc;
    hipLaunchKernelGGL(stem_wgrad_reduce, dim3(cout), dim3(256), 0, s, (const float*)workspace, nblk, cin, dw_oihw);
    return check_launch("stem_wgrad_reduce");
}

}  // namespace yolo
