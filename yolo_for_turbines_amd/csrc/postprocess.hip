// postprocess.hip — box decode and batched non-max suppression.   BUILD WITH -ffp-contract=off.
//
// Replaces (reference file:line): cells_to_boxes code/utils.py:86-148, non_max_suppression
// :150-191, calc_iou :38-84.
//
// NMS contract: kept set and order identical to the reference for any fp32 input. That needs
//  (1) the reference's fp32 operation order for IoU with one rounding per op (no FMA contraction,
//      IEEE division) — area and corner sums are per-box values, so they are computed once per box;
//  (2) NaN-propagating max/min (torch.max/min propagate NaN, fmaxf/fminf do not);
//  (3) a stable descending order of the objectness scores (Python sorted(reverse=True));
//  (4) thresholds: objectness compared as Python floats (double), IoU compared in fp32.
// Algorithm (all images of the batch in flight at once, no host round trip):
//  rank   — every candidate counts the candidates with a smaller 64-bit key
//           (~orderable(score) << 32 | index): keys are unique, so the count IS the stable rank.
//           O(n^2) one-instruction compares through LDS tiles, 64 lanes wide — cheaper than a
//           sort at n <= 22,743 and exactly parallel.
//  mask   — 64x64 tiles of the upper triangle: lane i computes, against 64 boxes staged in LDS,
//           the 64-bit word "j is suppressed by i if i is kept" (same class and not IoU < thr).
//  scan   — one workgroup per image walks the 64-row blocks in order: wave 0 resolves the
//           diagonal word serially with readlane broadcasts (the only sequential part of greedy
//           NMS), then all 16 waves OR the kept rows into the removed-bitmap held in LDS.
// Data is tiny (n x 24 B in, <= n x 4 B out); the bound is VALU pair work and the serial scan,
// not HBM.
#include <cstring>
#include "common.h"
#include <rocprim/rocprim.hpp>
#include <cstdlib>

namespace yolo {
typedef float f32x2 __attribute__((ext_vector_type(2)));

// ------------------------------------------------------------------------------ decode
__device__ __forceinline__ float sigmoid_f(float x) { return 1.f / (1.f + expf(-x)); }

// x / d for 0 <= x < 2^31 with m = ceil(2^32 / d) (d >= 2) or 0 (d == 1)
struct DecodeMagic { unsigned mg, mgg, m3gg; };
__device__ __forceinline__ int pp_fdiv(int x, unsigned m, int d) {
    if (!m) return x;
    const unsigned q = __umulhi((unsigned)x, m);
    return (int)(q * (unsigned)d > (unsigned)x ? q - 1 : q);
}
static unsigned pp_magic(long long d) { return d <= 1 ? 0u : (unsigned)((0x100000000ULL + (unsigned long long)d - 1) / (unsigned long long)d); }
static DecodeMagic decode_magic(int g) { return DecodeMagic{pp_magic(g), pp_magic((long long)g * g), pp_magic(3LL * g * g)}; }

// one cell's box from its (5 + nc) values at src (LDS tile or global memory): called by the four lanes q of the cell together
__device__ __forceinline__ void decode_cell(float* __restrict__ pred, long long sb, long long sa, long long sy, long long sx, long long sk,
                                            const float* __restrict__ anchors, int g, int nc, int is_pred, float* __restrict__ boxes,
                                            int n_total, int box_offset, long long cell, const float* src, long long kstride, bool staged, int q,
                                            const DecodeMagic mg) {
    // cell -> (b, a, row, col) by multiply-high (cells < 2^31, checked on the host): as four 64-bit divisions per lane this index
    // arithmetic was most of the kernel's issue time (~4.6 us per 64-cell tile)
    const int c32 = (int)cell, gg = g * g;
    const int b = pp_fdiv(c32, mg.m3gg, 3 * gg), r1 = c32 - b * 3 * gg;
    const int a = pp_fdiv(r1, mg.mgg, gg), r2 = r1 - a * gg;
    const int row = pp_fdiv(r2, mg.mg, g), col = r2 - row * g;
    float* gp = pred + b * sb + a * sa + row * sy + col * sx;
    if (!staged) src = gp;
    float cls;
    if (is_pred) {
        const int per = (nc + 3) >> 2;
        const int k0 = q * per, k1 = k0 + per < nc ? k0 + per : nc;
        bool have = k0 < k1;
        int best = k0;
        float bv = have ? src[(5 + k0) * kstride] : 0.f;
        int k = k0 + 1;
        for (; k + 8 <= k1; k += 8) {                    // 8 scores in flight: one LDS round trip per 8 instead of per score
            float v[8];
#pragma unroll
            for (int u = 0; u < 8; ++u) v[u] = src[(5 + k + u) * kstride];
#pragma unroll
            for (int u = 0; u < 8; ++u)
                if (v[u] > bv || (v[u] != v[u] && bv == bv)) { bv = v[u]; best = k + u; }
        }
        for (; k < k1; ++k) {
            const float v = src[(5 + k) * kstride];
            if (v > bv || (v != v && bv == bv)) { bv = v; best = k; }
        }
#pragma unroll
        for (int step = 1; step <= 2; step <<= 1) {      // lanes q and q + step: the later quarter wins only by the same rule
            const float ov = __shfl_down(bv, step, 4);
            const int ob = __shfl_down(best, step, 4);
            const int oh = __shfl_down((int)have, step, 4);
            if (oh && (!have || ov > bv || (ov != ov && bv == bv))) { bv = ov; best = ob; have = true; }
        }
        cls = (float)best;
    } else {
        cls = 0.f;
    }
    const float inv = (float)(1.0 / (double)g);          // `1 / grid_size` is a Python float, cast to fp32 by the multiply
    float* o = boxes + ((size_t)b * n_total + box_offset + (size_t)a * g * g + (size_t)row * g + col) * 6;
    if (is_pred) {
        // The four box values of a cell go to its four lanes (x, y: sigmoid; w, h: exp * anchor): the kernel is bound by VALU
        // issue (an expf is ~50 instructions for the whole wave however few lanes are active), and with lane 0 doing all five
        // transcendentals the other 48 lanes of the wave waited through them. Same operations per value as before.
        const float x = src[q * kstride];
        const float e = expf(q < 2 ? -x : x);
        const float v = q < 2 ? 1.f / (1.f + e) : e * anchors[2 * a + (q & 1)];
        // in-place side effect (utils.py:106-110): what cells_to_boxes does to its argument. is_pred == 2 (detect paths, where
        // the caller cannot observe the prediction tensor afterwards) leaves it alone: 16 bytes into every (5+nc)*4-byte cell
        // are a partial-line write per cell, 1.7x the algorithmic 24 bytes per box of this kernel's writes
        if (is_pred == 1) gp[q * sk] = v;
        const float add = q == 0 ? (float)col : (float)row;
        o[q] = inv * (q < 2 ? v + add : v);
        if (q == 0) {
            o[4] = sigmoid_f(src[4 * kstride]);
            o[5] = cls;
        }
    } else if (q == 0) {
        o[0] = inv * (src[0] + (float)col);
        o[1] = inv * (src[kstride] + (float)row);
        o[2] = inv * src[2 * kstride];
        o[3] = inv * src[3 * kstride];
        o[4] = src[4 * kstride];
        o[5] = src[5 * kstride];
    }
}

// 256 threads per 64 cells (b, a, row, col): FOUR lanes per cell. For this library's contiguous head layout (stride 1
// along k) the 64 x (5 + nc) floats of the block are staged through LDS with full-line 16-byte loads by all 256 threads;
// each of a cell's four lanes then finds the first maximum of a quarter of the class scores and the partial results are
// combined in class order with the same rule, which is exactly the sequential first-maximum (torch.argmax; NaN counts as
// the maximum). Lane 0 of the cell finishes it. (One lane per cell and 64-thread blocks left 6-7 waves per CU, each
// waiting on its own loads and then on 80 dependent LDS reads: 2.5 TB/s.)
__device__ __forceinline__ void decode_block(float* __restrict__ pred, long long sb, long long sa, long long sy, long long sx, long long sk,
                                             const float* __restrict__ anchors, int B, int g, int nc, int is_pred,
                                             float* __restrict__ boxes, int n_total, int box_offset, long long blk, const DecodeMagic mg) {
    extern __shared__ __attribute__((aligned(16))) float tile[];           // [64][D] when sk == 1 && cells contiguous, else unused
    const int D = 5 + nc;
    const int q = threadIdx.x & 3, cl = threadIdx.x >> 2;
    const long long cells = (long long)B * 3 * g * g;
    const long long cell0 = blk * 64;
    const long long cell = cell0 + cl;
    const bool contiguous = (sk == 1 && sx == D && sy == (long long)g * D && sa == (long long)g * g * D &&
                             sb == 3LL * g * g * D);
    const float* src;
    long long kstride;
    if (contiguous) {
        const long long first = cell0 * D;
        const long long count = (cells - cell0 < 64 ? cells - cell0 : 64) * D;
        const float* gsrc = pred + first;
        if (count == 64LL * D && ((reinterpret_cast<size_t>(gsrc) & 15) == 0)) {
            typedef float f4 __attribute__((ext_vector_type(4)));
            const int n4 = (int)(count >> 2);                                // 1,360 at 80 classes: <= 6 per thread, all in flight
            for (int i0 = threadIdx.x; i0 < n4; i0 += 256 * 6) {
                f4 r[6];
#pragma unroll
                for (int u = 0; u < 6; ++u)
                    if (i0 + u * 256 < n4) r[u] = reinterpret_cast<const f4*>(gsrc)[i0 + u * 256];
#pragma unroll
                for (int u = 0; u < 6; ++u)
                    if (i0 + u * 256 < n4) reinterpret_cast<f4*>(tile)[i0 + u * 256] = r[u];
            }
        } else {
            for (long long i = threadIdx.x; i < count; i += 256) tile[i] = gsrc[i];
        }
        __syncthreads();
        src = tile + (long long)cl * D;
        kstride = 1;
    } else {
        src = nullptr;
        kstride = sk;
    }
    if (cell >= cells) return;                                               // the four lanes of a cell leave together
    decode_cell(pred, sb, sa, sy, sx, sk, anchors, g, nc, is_pred, boxes, n_total, box_offset, cell, src, kstride, contiguous, q, mg);
}

__global__ void decode_kernel(float* __restrict__ pred, long long sb, long long sa, long long sy, long long sx, long long sk,
                              const float* __restrict__ anchors, int B, int g, int nc, int is_pred,
                              float* __restrict__ boxes, int n_total, int box_offset, const DecodeMagic mg) {
    decode_block(pred, sb, sa, sy, sx, sk, anchors, B, g, nc, is_pred, boxes, n_total, box_offset, blockIdx.x, mg);
}

// the three scales of one forward in ONE launch (demo.py:44-51 / utils.py:300-309 order): at batch 32 the three separate
// launches were launch-latency-bound (3 x ~25 us for 129 MB)
struct DecodeScale { float* pred; long long sb, sa, sy, sx, sk; const float* anchors; int g, box_offset; long long first_block; DecodeMagic mg; };
struct Decode3Args { DecodeScale sc[3]; int B, nc, n_total, is_pred; float* boxes; };

__global__ void decode3_kernel(const Decode3Args a) {
    const int k = (long long)blockIdx.x >= a.sc[2].first_block ? 2 : ((long long)blockIdx.x >= a.sc[1].first_block ? 1 : 0);
    const DecodeScale& d = a.sc[k];
    decode_block(d.pred, d.sb, d.sa, d.sy, d.sx, d.sk, d.anchors, a.B, d.g, a.nc, a.is_pred, a.boxes, a.n_total, d.box_offset,
                 (long long)blockIdx.x - d.first_block, d.mg);
}

// (Round 3 tried this launch as a STREAM: persistent workgroups, 2-4 LDS tiles, the next tiles requested by LDS-DMA while the
// current one is decoded, one barrier per tile. Slower at every depth: 3 workgroups/CU x 2 tiles 3.9 TB/s, 2 x 3 tiles 3.2,
// 1 x 4 tiles 1.9, against 4.4 TB/s for this kernel's ~7 independent workgroups per CU - a tile's decode is ~3 us of dependent
// LDS reads, compares and transcendentals per wave, so what the kernel needs is wave slots, not a deeper queue. DESIGN.md 7.4.)

// --------------------------------------------------------------------------------- NMS
struct SBox { float x1, y1, x2, y2, area, cls; float w, h; };   // 32 B, sorted order

__device__ __forceinline__ float max_nan(float a, float b) { return (a > b || a != a) ? a : b; }
__device__ __forceinline__ float min_nan(float a, float b) { return (a < b || a != a) ? a : b; }

__device__ __forceinline__ unsigned long long make_key(float score, int idx, double obj_thr) {
    if (!((double)score > obj_thr)) return ~0ull;       // filtered (also NaN): never smaller than a candidate
    score = score + 0.0f;                                // -0.0 -> +0.0 (they compare equal in Python)
    unsigned u = __float_as_uint(score);
    u = (u & 0x80000000u) ? ~u : (u | 0x80000000u);     // ascending-orderable
    return ((unsigned long long)(~u) << 32) | (unsigned)idx;   // descending score, ascending index
}

// grid (ceil(n/256), B). rank = #keys smaller than mine.
__global__ __launch_bounds__(256) void nms_rank_kernel(const float* __restrict__ boxes, int n, double obj_thr, int center,
                                                       int* __restrict__ order, SBox* __restrict__ sbox,
                                                       int* __restrict__ nvalid) {
    __shared__ unsigned long long keys[256];
    const int b = blockIdx.y;
    const float* bx = boxes + (size_t)b * n * 6;
    const int i = blockIdx.x * 256 + threadIdx.x;
    const unsigned long long mine = i < n ? make_key(bx[(size_t)i * 6 + 4], i, obj_thr) : ~0ull;
    int rank = 0;
    for (int j0 = 0; j0 < n; j0 += 256) {
        const int j = j0 + threadIdx.x;
        __syncthreads();
        keys[threadIdx.x] = j < n ? make_key(bx[(size_t)j * 6 + 4], j, obj_thr) : ~0ull;
        __syncthreads();
        const int lim = n - j0 < 256 ? n - j0 : 256;
        for (int t = 0; t < lim; ++t) rank += keys[t] < mine;
    }
    const bool valid = mine != ~0ull;
    if (valid) {
        const float* s = bx + (size_t)i * 6;
        float x = s[0], y = s[1];
        const float w = s[2], h = s[3];
        if (center) { x = x - w / 2.0f; y = y - h / 2.0f; }        // utils.py:60-64
        SBox o;
        o.x1 = x; o.y1 = y; o.x2 = x + w; o.y2 = y + h; o.area = w * h; o.cls = s[5]; o.w = w; o.h = h;
        sbox[(size_t)b * n + rank] = o;
        order[(size_t)b * n + rank] = i;
    }
    const unsigned long long bal = __ballot(valid);
    if ((threadIdx.x & 63) == 0 && bal) atomicAdd(nvalid + b, __popcll(bal));
}

// ---- ordering by ONE radix sort of the whole batch (image id in the top key bits) ----------------------------------
// The counting rank above is O(n^2) (1e8 key compares per image at n = 10,000, 5e8 at the 22,743 boxes of a 608x608
// image); the keys are unique for valid boxes (the index is part of the key), so sorting them gives the identical
// order: rocPRIM's device radix sort (the plain library piece of this file) + a key builder and a gather.
// (rocPRIM's SEGMENTED sort was tried first: with 16 segments of 10,000 keys it was slower than the counting rank.)
// A key is "valid" iff its score field is not all ones (only a NaN score could map there, and NaN is filtered), so the
// kernels after the sort read validity off the key and the one thread at the valid/filtered boundary of an image writes
// nvalid[b] (zeroed before): no counting atomics (2,500 same-address atomics per call were most of this kernel's 30 us).
__device__ __forceinline__ bool key_valid(unsigned long long k) { return ((k >> 20) & 0xffffffffull) != 0xffffffffull; }

__global__ __launch_bounds__(256) void nms_keys_kernel(const float* __restrict__ boxes, int n, double obj_thr,
                                                       unsigned long long* __restrict__ keys) {
    const int b = blockIdx.y;
    const int i = blockIdx.x * 256 + threadIdx.x;
    if (i >= n) return;
    const unsigned long long k = make_key(boxes[((size_t)b * n + i) * 6 + 4], i, obj_thr);
    // image (12 bits) | descending-score field (32 bits; all ones = filtered) | index (20 bits): ONE radix sort of the
    // whole batch leaves image b's boxes in [b n, (b + 1) n), valid ones first, in the reference's stable order
    const unsigned long long sc = k != ~0ull ? (k >> 32) : 0xffffffffull;
    keys[(size_t)b * n + i] = ((unsigned long long)b << 52) | (sc << 20) | (unsigned long long)i;
}

__global__ __launch_bounds__(256) void nms_gather_kernel(const float* __restrict__ boxes, const unsigned long long* __restrict__ sorted,
                                                         int* __restrict__ nvalid, int n, int center, int* __restrict__ order,
                                                         SBox* __restrict__ sbox) {
    const int b = blockIdx.y;
    const int r = blockIdx.x * 256 + threadIdx.x;
    if (r >= n) return;
    const unsigned long long key = sorted[(size_t)b * n + r];
    if (!key_valid(key)) return;
    if (r == n - 1 || !key_valid(sorted[(size_t)b * n + r + 1])) nvalid[b] = r + 1;
    const int i = (int)(unsigned)(key & 0xfffffull);
    const float* s = boxes + ((size_t)b * n + i) * 6;
    float x = s[0], y = s[1];
    const float w = s[2], h = s[3];
    if (center) { x = x - w / 2.0f; y = y - h / 2.0f; }            // utils.py:60-64
    SBox o;
    o.x1 = x; o.y1 = y; o.x2 = x + w; o.y2 = y + h; o.area = w * h; o.cls = s[5]; o.w = w; o.h = h;
    sbox[(size_t)b * n + r] = o;
    order[(size_t)b * n + r] = i;
}

// One 64-bit suppression word: bit jj set <=> column col0 + jj comes after row i, has the same class and !(iou < thr)
// (utils.py:38-84 + 183-186). `cols` is the staged column block in LDS, `lim` its valid length.
// TAME: every coordinate and area of the 64 rows and the columns is below 1e18 in magnitude, so nothing before the
// division can be NaN or overflow and the NaN-propagating max / min / clamp reduce to plain v_max / v_min (the sign of a
// zero they may pick differently cannot reach the comparison: it only ever yields iou = +-0); and when no row of the
// wave has a same-class column with a non-zero intersection, iou is exactly +-0 over a positive denominator and the
// IEEE division is skipped. Same bits as the general path, about half the instructions per pair.
// bare v_max / v_min: fmaxf() makes the compiler canonicalise both operands first (3 instructions instead of 1)
__device__ __forceinline__ float vmax(float a, float b) { float r; asm("v_max_f32 %0, %1, %2" : "=v"(r) : "v"(a), "v"(b)); return r; }
__device__ __forceinline__ float vmin(float a, float b) { float r; asm("v_min_f32 %0, %1, %2" : "=v"(r) : "v"(a), "v"(b)); return r; }

__device__ __forceinline__ bool tame_box(const SBox& q) {
    const float lim = 1e18f;
    return fabsf(q.x1) < lim && fabsf(q.y1) < lim && fabsf(q.x2) < lim && fabsf(q.y2) < lim && fabsf(q.area) < lim;
}

// Per-row bounds of the fast path's candidate test (below): the column must reach x2 >= lo_x, x1 <= hi_x, y2 >= lo_y, y1 <= hi_y.
// The loose form is "the boxes overlap". The tight form uses what !(iou < thr) implies for thr > 0 with positive areas: from
// iou = fl(inter / den) >= thr and den = fl(fl(fl(a1 + a2) - inter) + 1e-6) follows inter >= 0.99 (thr / (1 + thr)) a1 (every
// rounding is a factor 1 +- 2^-24; a2 >= 0), and inter = fl(iw ih) with ih <= fl(y2 - y1) of the row itself, so the width of the
// intersection must be at least T = 0.98 (thr / (1 + thr)) a1 / fl(y2 - y1) - and its height U likewise. For equal-sized boxes at
// thr = 0.45 that halves the candidates the exact pass has to divide for, and it costs nothing per pair: the four compares
// stay, only their per-row constants move. The bounds are rounded outwards by a few ulps; a pair they admit wrongly is still
// decided by the exact pass, so the result cannot change - a pair they reject is one the exact formula cannot suppress.
struct RowBounds { float lo_x, hi_x, lo_y, hi_y; };
__device__ __forceinline__ RowBounds row_bounds(const SBox& me, float thr) {
    float T = 0.f, U = 0.f;
    const float ew = me.x2 - me.x1, eh = me.y2 - me.y1;
    if (thr > 1e-6f && thr < 1e3f && me.area > 1e-30f && ew > 1e-30f && eh > 1e-30f) {
        const float kf = 0.98f * (thr / (1.0f + thr)) * me.area;
        T = kf * __builtin_amdgcn_rcpf(eh);
        U = kf * __builtin_amdgcn_rcpf(ew);
    }
    RowBounds r;
    r.lo_x = me.x1 + T; r.hi_x = me.x2 - T; r.lo_y = me.y1 + U; r.hi_y = me.y2 - U;
    r.lo_x -= fabsf(r.lo_x) * 0x1p-21f; r.hi_x += fabsf(r.hi_x) * 0x1p-21f;       // outwards: the sums above were rounded
    r.lo_y -= fabsf(r.lo_y) * 0x1p-21f; r.hi_y += fabsf(r.hi_y) * 0x1p-21f;
    return r;
}

template <bool TAME>
__device__ __forceinline__ unsigned long long suppression_word(const SBox& me, bool active, int i, const SBox* cols, int lim, int col0,
                                                              float thr, const SBox* __restrict__ gcols, const RowBounds& rbnd) {
    unsigned long long word = 0;
    if (TAME) {
        // Two passes. (1) all 64 columns, ~10 instructions per pair: CANDIDATE = same class and the boxes overlap in x and in y
        // (strictly positive width and height of the intersection). Every other same-class pair has inter = +-0 exactly, and with
        // den = (a1 + a2) + 1e-6 > 0 (checked per pair) its iou is +-0: it suppresses iff !(0 < thr), decided once per word.
        // (2) the exact IoU with the IEEE division only for the candidates, each lane walking the set bits of ITS word: with
        // 2 classes and 10,000 boxes 3 % of the pairs are candidates (~2 per word, ~7 for the busiest lane of a wave), so
        // the division work drops ~8x; the old single pass divided whenever ANY of a wave's 256 pairs intersected - always.
        const bool zero_suppresses = !(0.f < thr);        // iou == +-0 for a pair that does not intersect
        unsigned clo = 0, chi = 0, slo = 0, shi = 0;      // candidate bits / same-class-and-zero-iou bits, low and high 32 columns
        auto half = [&](int j0, unsigned& cbits, unsigned& sbits) {
#pragma unroll 8
            for (int jj = 0; jj < 32; ++jj) {             // branch-free: bitwise & / | on the predicates, all six fields read up front
                const f32x4 c = *reinterpret_cast<const f32x4*>(&cols[j0 + jj].x1);    // entries at or beyond lim are masked out below
                const f32x2 ac = *reinterpret_cast<const f32x2*>(&cols[j0 + jj].area); // area, cls
                const bool same = ac[1] == me.cls;
                const float xa = vmax(me.x1, c[0]), ya = vmax(me.y1, c[1]);
                const float xb = vmin(me.x2, c[2]), yb = vmin(me.y2, c[3]);
                const bool den_bad = !((me.area + ac[0]) + 1e-6f > 0.f);               // ((a1 + a2) - 0) + 1e-6, the pair's denominator
                const bool cand = same & (((xb > xa) & (yb > ya)) | den_bad);
                const unsigned bit = 1u << jj;
                cbits |= cand ? bit : 0u;
                sbits |= same ? bit : 0u;
            }
        };
        // the common column block: every row and column of one class and every area positive (so the denominator of a
        // non-intersecting pair is positive) - 4 min/max + 2 compares per pair and nothing else
        auto half_uni = [&](int j0, unsigned& cbits) {
#pragma unroll 8
            for (int jj = 0; jj < 32; ++jj) {
                const f32x4 c = *reinterpret_cast<const f32x4*>(&cols[j0 + jj].x1);
                const float xa = vmax(me.x1, c[0]), ya = vmax(me.y1, c[1]);
                const float xb = vmin(me.x2, c[2]), yb = vmin(me.y2, c[3]);
                cbits |= ((xb > xa) & (yb > ya)) ? (1u << jj) : 0u;
            }
        };
        const int lane_c = (int)(threadIdx.x & 63);
        const bool odd = (active && !(me.area > 0.f)) || (lane_c < lim && !(cols[lane_c].area > 0.f)) ||
                         (active && cols[0].cls != me.cls) || (lane_c < lim && cols[lane_c].cls != cols[0].cls);
        // ... and when, on top of that, every box of the block has a positive width and height, "the intersection has a positive
        // width" is just o.x1 < me.x2 && me.x1 < o.x2 (min(a2, b2) > max(a1, b1) with a2 > a1, b2 > b1), likewise for y:
        // four compares whose lane masks are ANDed on the scalar unit, and the bit goes into the word as the carry-in of
        // w + w (v_addc_co_u32): 5 vector instructions per pair instead of 8. The column's four coordinates are uniform,
        // so they come from the scalar cache (s_load of the global array) as SGPR operands - no LDS read in this loop.
        // (The counters say the kernel is VALU-issue-bound: 109 M vector instructions = 203 us of issue in a 290 us launch.)
        const bool flat = (active && !(me.x2 > me.x1 && me.y2 > me.y1)) ||
                          (lane_c < lim && !(cols[lane_c].x2 > cols[lane_c].x1 && cols[lane_c].y2 > cols[lane_c].y1));
        if (gcols != nullptr && lim == 64 && __ballot(odd || flat) == 0ull) {   // full column blocks only: constant load offsets, no address arithmetic
            auto half_fast = [&](int j0, unsigned& cbits) {
                unsigned w = 0;
#pragma unroll
                for (int g8 = 24; g8 >= 0; g8 -= 8) {                            // 8 columns' coordinates requested before they are used
                    f32x4 c8[8];
#pragma unroll
                    for (int u = 0; u < 8; ++u) c8[u] = *reinterpret_cast<const f32x4*>(&gcols[j0 + g8 + u].x1);
                    // four columns per statement, highest first. v_cmpx narrows EXEC, so four chained compares leave
                    // VCC = EXEC = the lanes where all four hold (the AND costs no instruction); EXEC is restored from a copy and
                    // the bit enters w as the carry-in of w + w: 5 vector + 1 scalar instruction per pair.
#define NMS_COL4(q) \
                    { unsigned long long sv; \
                      asm volatile("s_mov_b64 %[sv], exec\n\t" \
                                   "v_cmpx_le_f32_e32 vcc, %[a3], %[mx2]\n\tv_cmpx_ge_f32_e32 vcc, %[c3], %[mx1]\n\t" \
                                   "v_cmpx_le_f32_e32 vcc, %[b3], %[my2]\n\tv_cmpx_ge_f32_e32 vcc, %[d3], %[my1]\n\t" \
                                   "s_mov_b64 exec, %[sv]\n\tv_addc_co_u32_e32 %[w], vcc, %[w], %[w], vcc\n\t" \
                                   "v_cmpx_le_f32_e32 vcc, %[a2], %[mx2]\n\tv_cmpx_ge_f32_e32 vcc, %[c2], %[mx1]\n\t" \
                                   "v_cmpx_le_f32_e32 vcc, %[b2], %[my2]\n\tv_cmpx_ge_f32_e32 vcc, %[d2], %[my1]\n\t" \
                                   "s_mov_b64 exec, %[sv]\n\tv_addc_co_u32_e32 %[w], vcc, %[w], %[w], vcc\n\t" \
                                   "v_cmpx_le_f32_e32 vcc, %[a1], %[mx2]\n\tv_cmpx_ge_f32_e32 vcc, %[c1], %[mx1]\n\t" \
                                   "v_cmpx_le_f32_e32 vcc, %[b1], %[my2]\n\tv_cmpx_ge_f32_e32 vcc, %[d1], %[my1]\n\t" \
                                   "s_mov_b64 exec, %[sv]\n\tv_addc_co_u32_e32 %[w], vcc, %[w], %[w], vcc\n\t" \
                                   "v_cmpx_le_f32_e32 vcc, %[a0], %[mx2]\n\tv_cmpx_ge_f32_e32 vcc, %[c0], %[mx1]\n\t" \
                                   "v_cmpx_le_f32_e32 vcc, %[b0], %[my2]\n\tv_cmpx_ge_f32_e32 vcc, %[d0], %[my1]\n\t" \
                                   "s_mov_b64 exec, %[sv]\n\tv_addc_co_u32_e32 %[w], vcc, %[w], %[w], vcc" \
                                   : [w] "+v"(w), [sv] "=&s"(sv) \
                                   : [a0] "s"(c8[q][0]), [b0] "s"(c8[q][1]), [c0] "s"(c8[q][2]), [d0] "s"(c8[q][3]), \
                                     [a1] "s"(c8[q + 1][0]), [b1] "s"(c8[q + 1][1]), [c1] "s"(c8[q + 1][2]), [d1] "s"(c8[q + 1][3]), \
                                     [a2] "s"(c8[q + 2][0]), [b2] "s"(c8[q + 2][1]), [c2] "s"(c8[q + 2][2]), [d2] "s"(c8[q + 2][3]), \
                                     [a3] "s"(c8[q + 3][0]), [b3] "s"(c8[q + 3][1]), [c3] "s"(c8[q + 3][2]), [d3] "s"(c8[q + 3][3]), \
                                     [mx1] "v"(rbnd.lo_x), [my1] "v"(rbnd.lo_y), [mx2] "v"(rbnd.hi_x), [my2] "v"(rbnd.hi_y) \
                                   : "vcc"); }
                    NMS_COL4(4)
                    NMS_COL4(0)
#undef NMS_COL4
                }
                cbits = w;
            };
            half_fast(0, clo);
            half_fast(32, chi);
            slo = shi = ~0u;
        } else if (__ballot(odd) == 0ull) {
            half_uni(0, clo);
            half_uni(32, chi);
            slo = shi = ~0u;
        } else {
            half(0, clo, slo);
            half(32, chi, shi);
        }
        const unsigned long long valid = lim >= 64 ? ~0ull : ((1ull << lim) - 1ull);
        unsigned long long cw = ((((unsigned long long)chi) << 32) | clo) & valid;
        const unsigned long long sw = ((((unsigned long long)shi) << 32) | slo) & valid;
        if (!active) cw = 0ull;
        word = (active && zero_suppresses) ? (sw & ~cw) : 0ull;
        while (__ballot(cw != 0ull)) {                    // wave-uniform trip count: the busiest lane's candidates
            if (cw) {
                const int jj = __builtin_ctzll(cw);
                cw &= cw - 1ull;
                const SBox o = cols[jj];
                const float xa = vmax(me.x1, o.x1), ya = vmax(me.y1, o.y1);
                const float xb = vmin(me.x2, o.x2), yb = vmin(me.y2, o.y2);
                const float iw = vmax(xb - xa, 0.f), ih = vmax(yb - ya, 0.f);
                const float inter = iw * ih;
                const float den = ((me.area + o.area) - inter) + 1e-6f;
                const float iou = inter / den;
                if (!(iou < thr)) word |= 1ull << jj;
            }
        }
        // only columns after row i
        const int first = i + 1 - col0;                   // first admissible bit
        if (first > 0) word &= first >= 64 ? 0ull : (~0ull << first);
        return word;
    }
    for (int jj = 0; jj < lim; ++jj) {
        const SBox o = cols[jj];
        // a box of another class never suppresses (kept if `cls != top.cls` OR iou < thr): skip the column when no row of
        // this wave shares its class
        if (__ballot(active && o.cls == me.cls) == 0ull) continue;
        const float xa = max_nan(me.x1, o.x1), ya = max_nan(me.y1, o.y1);
        const float xb = min_nan(me.x2, o.x2), yb = min_nan(me.y2, o.y2);
        float iw = xb - xa, ih = yb - ya;
        iw = iw < 0.f ? 0.f : iw;                       // torch.clamp(min=0): NaN stays NaN
        ih = ih < 0.f ? 0.f : ih;
        const float inter = iw * ih;
        const float uni = (me.area + o.area) - inter;
        const float iou = inter / (uni + 1e-6f);
        const bool survive = (o.cls != me.cls) || (iou < thr);
        if (active && !survive && col0 + jj > i) word |= 1ull << jj;
    }
    return word;
}

__device__ __forceinline__ unsigned long long suppression_word(const SBox& me, bool active, int i, const SBox* cols, int lim, int col0,
                                                              float thr, const SBox* __restrict__ gcols = nullptr,
                                                              const RowBounds& rbnd = RowBounds{0.f, 0.f, 0.f, 0.f}) {
    const bool wild = (active && !tame_box(me)) || ((int)(threadIdx.x & 63) < lim && !tame_box(cols[threadIdx.x & 63]));
    return __ballot(wild) == 0ull ? suppression_word<true>(me, active, i, cols, lim, col0, thr, gcols, rbnd)
                                  : suppression_word<false>(me, active, i, cols, lim, col0, thr, gcols, rbnd);
}

// ---- class-sorted variant (large n) ---------------------------------------------------------------------------------
// A box only ever suppresses boxes of its own class (utils.py:183-186), so after a second STABLE sort by class the
// suppression matrix is block diagonal: the mask kernel visits only the column blocks whose class range overlaps the row
// block's (1/nc of the pairs for nc balanced classes) and the scan ORs only those. Greedy NMS per class in score order
// is exactly the reference's result; the kept set is then emitted in the reference's (global score) order by a slot
// array indexed with the global rank + an ordered compaction.
__device__ __forceinline__ unsigned class_bucket(float c) {   // equal float classes -> equal bucket; ascending-sortable
    return (c >= 0.0f && c < 4094.0f && c == floorf(c)) ? (unsigned)c : 4094u;
}

// key2 = image (11 bits, top bit clear) | class bucket (12; 0xfff = filtered) | global rank (20) | original index (20), sorted
// on bits 40..62. (end_bit must stay below 64 with begin_bit > 0: rocPRIM's merge path builds its mask as
// (1 << (begin + bits)) - 1, which is undefined for 64 and sorted on the LOW bits instead when tried.)
__global__ __launch_bounds__(256) void nms_keys2_kernel(const float* __restrict__ boxes, const unsigned long long* __restrict__ sorted,
                                                        int* __restrict__ nvalid, int n, unsigned long long* __restrict__ keys2) {
    const int b = blockIdx.y;
    const int r = blockIdx.x * 256 + threadIdx.x;
    if (r >= n) return;
    const unsigned long long key = sorted[(size_t)b * n + r];
    const unsigned long long i = key & 0xfffffull;
    unsigned long long c = 0xfffull;
    if (key_valid(key)) {
        c = class_bucket(boxes[((size_t)b * n + i) * 6 + 5]);
        if (r == n - 1 || !key_valid(sorted[(size_t)b * n + r + 1])) nvalid[b] = r + 1;
    }
    keys2[(size_t)b * n + r] = ((unsigned long long)b << 52) | (c << 40) | ((unsigned long long)r << 20) | i;
}

__global__ __launch_bounds__(256) void nms_gather2_kernel(const float* __restrict__ boxes, const unsigned long long* __restrict__ sorted2,
                                                          const int* __restrict__ nvalid, int n, int W, int center,
                                                          int* __restrict__ order, int* __restrict__ grank, SBox* __restrict__ sbox,
                                                          int* __restrict__ blk_lo, int* __restrict__ blk_hi) {
    const int b = blockIdx.y;
    const int q = blockIdx.x * 256 + threadIdx.x;
    const int nv = nvalid[b];
    if (q >= nv) return;
    const unsigned long long k = sorted2[(size_t)b * n + q];
    const int i = (int)(unsigned)(k & 0xfffffull);
    const float* s = boxes + ((size_t)b * n + i) * 6;
    float x = s[0], y = s[1];
    const float w = s[2], h = s[3];
    if (center) { x = x - w / 2.0f; y = y - h / 2.0f; }            // utils.py:60-64
    SBox o;
    o.x1 = x; o.y1 = y; o.x2 = x + w; o.y2 = y + h; o.area = w * h; o.cls = s[5]; o.w = w; o.h = h;
    sbox[(size_t)b * n + q] = o;
    order[(size_t)b * n + q] = i;
    grank[(size_t)b * n + q] = (int)(unsigned)((k >> 20) & 0xfffffull);
    const int c = (int)(unsigned)((k >> 40) & 0xfffull);
    if ((q & 63) == 0) blk_lo[(size_t)b * W + (q >> 6)] = c;
    if ((q & 63) == 63 || q == nv - 1) blk_hi[(size_t)b * W + (q >> 6)] = c;
}

// ---- hand-written ordering for n <= 32,768 (replaces the two rocPRIM sorts above; their ~16 merge launches were more
// than half of the batched NMS). Both orders are full sorts of UNIQUE 64-bit keys (the index / rank is part of the key):
//   order 1: make_key (score descending, index ascending)            -> global rank r of every valid box
//   order 2: class bucket | rank r | index                            -> class-major rows (a stable partition by class)
// Each is done by two launches that use the whole chip: (A) 2,048-key chunks sorted in LDS by a bitonic network, eight
// keys per thread and three strides per LDS round trip; (B) one thread per key adds up, over the other chunks of its
// image, how many keys are smaller (binary searches, all chunks of a step in flight together) — that sum plus its
// position in its own chunk IS its final position, so the merge is a scatter. No atomics; deterministic.
constexpr int SC = 2048;                 // keys per chunk
constexpr int SC_MAXCH = 128;            // chunks per image handled here (n <= 262,144: every size that still takes the class-sorted path)
constexpr int SC_GROUP = 15;             // chunks ranked at once by the merge kernels (registers); more chunks: several rounds
constexpr int SC_ROW = 80;               // 8 keys (64 B) + 16 B pad per LDS row: a lane's 8 contiguous keys never share banks with its neighbours'
__device__ __forceinline__ int sc_addr(int i) { return (i >> 3) * SC_ROW + (i & 7) * 8; }

template <int LG>            // 2^LG keys per thread, SC >> LG threads, LG strides per LDS round trip
__device__ __forceinline__ void chunk_sort_lds(char* lds, int tid) {
    constexpr int KPT = 1 << LG;
    // fully unrolled: every shift, mask and LDS offset below is an immediate
#pragma unroll
    for (int m = 1; m <= 11; ++m) {                       // bitonic merges of length 2^m; the last one ascending
#pragma unroll
        for (int p = m - 1; p >= 0; p -= LG) {            // strides 2^p .. 2^(p-LG+1) in ONE round trip
            const int lo = p >= LG - 1 ? p - (LG - 1) : 0;   // the thread's keys differ in bits lo .. lo+LG-1
            const int base = ((tid >> lo) << (lo + LG)) | (tid & ((1 << lo) - 1));
            unsigned long long r[KPT];
#pragma unroll
            for (int a = 0; a < KPT; ++a) r[a] = *reinterpret_cast<const unsigned long long*>(lds + sc_addr(base | (a << lo)));
#pragma unroll
            for (int sb = LG - 1; sb >= 0; --sb) {
                if (sb > p - lo) continue;
#pragma unroll
                for (int a = 0; a < KPT; ++a) {
                    if (a & (1 << sb)) continue;
                    const unsigned long long x = r[a], y = r[a | (1 << sb)];
                    const bool up = (((base | (a << lo)) >> m) & 1) == 0;      // direction of the length-2^m run this pair sits in
                    const bool sw = (x > y) == up;
                    r[a] = sw ? y : x;
                    r[a | (1 << sb)] = sw ? x : y;
                }
            }
#pragma unroll
            for (int a = 0; a < KPT; ++a) *reinterpret_cast<unsigned long long*>(lds + sc_addr(base | (a << lo))) = r[a];
            __syncthreads();
        }
    }
}

// (A) grid (nch, B), 256 threads. PHASE 1: keys from the boxes. PHASE 2: keys from order 1 (sorted1[b][r], r < nvalid[b]).
constexpr int SC_LG = 2;                 // chunk sort: 4 keys per thread, 512 threads (two waves per SIMD cover each other's LDS latency)
constexpr int SC_THREADS = SC >> SC_LG;
template <int PHASE>
__global__ __launch_bounds__(SC_THREADS) void nms_chunksort_kernel(const float* __restrict__ boxes, const unsigned long long* __restrict__ sorted1,
                                                            const int* __restrict__ nvalid, int n, double obj_thr,
                                                            unsigned long long* __restrict__ chunked, int* __restrict__ chunk_valid) {
    __shared__ __attribute__((aligned(16))) char lds[(SC / 8) * SC_ROW];
    const int b = blockIdx.y, c = blockIdx.x, nch = gridDim.x, tid = threadIdx.x;
    const int nv = PHASE == 2 ? nvalid[b] : 0;
#pragma unroll
    for (int e = 0; e < (1 << SC_LG); ++e) {
        const int loc = e * SC_THREADS + tid, i = c * SC + loc;
        unsigned long long key = ~0ull;
        if (PHASE == 1) {
            if (i < n) key = make_key(boxes[((size_t)b * n + i) * 6 + 4], i, obj_thr);
        } else {
            // class bucket (12 bits; 0xfff = not a candidate) | global rank (20) | original index (20): unique for every r
            key = (0xfffull << 40) | ((unsigned long long)(unsigned)i << 20);
            if (i < nv) {
                const unsigned long long idx = sorted1[(size_t)b * n + i] & 0xffffffffull;
                key = ((unsigned long long)class_bucket(boxes[((size_t)b * n + idx) * 6 + 5]) << 40) | ((unsigned long long)(unsigned)i << 20) | idx;
            }
        }
        *reinterpret_cast<unsigned long long*>(lds + sc_addr(loc)) = key;
    }
    __syncthreads();
    chunk_sort_lds<SC_LG>(lds, tid);
    unsigned long long* out = chunked + ((size_t)b * nch + c) * SC;
#pragma unroll
    for (int e = 0; e < (1 << SC_LG); ++e) {
        const int loc = e * SC_THREADS + tid;
        const unsigned long long key = *reinterpret_cast<const unsigned long long*>(lds + sc_addr(loc));
        out[loc] = key;
        if (PHASE == 1 && key != ~0ull) {                 // candidates sort first: the last one reports the count ...
            const unsigned long long nxt = loc + 1 < SC ? *reinterpret_cast<const unsigned long long*>(lds + sc_addr(loc + 1)) : ~0ull;
            if (nxt == ~0ull) chunk_valid[b * SC_MAXCH + c] = loc + 1;
        } else if (PHASE == 1 && loc == 0) {
            chunk_valid[b * SC_MAXCH + c] = 0;            // ... or the first key says there is none (no memset before the call)
        }
    }
}

// number of keys of the OTHER sorted chunks that are smaller than `key`: MAXC chunks at once (one load per chunk and step in
// flight), in rounds when the image has more than MAXC + 1 chunks (n > 32,768)
template <int MAXC>       // slot k of a round is the (first + k)-th other chunk: chunk first + k, or the one after it from the thread's own chunk on
__device__ __forceinline__ int rank_in_other_chunks(const unsigned long long* __restrict__ img, int nch, int c, unsigned long long key) {
    int total = 0;
#pragma unroll 1
    for (int first = 0; first < nch - 1; first += MAXC) {
        int pos[MAXC];
        const unsigned long long* base[MAXC];
#pragma unroll
        for (int k = 0; k < MAXC; ++k) {
            pos[k] = 0;
            const int ck = first + k + (first + k >= c ? 1 : 0);
            base[k] = img + (size_t)(ck < nch ? ck : 0) * SC;       // clamped: unconditional loads
        }
#pragma unroll 1
        for (int step = SC / 2; step >= 1; step >>= 1) {
            unsigned long long v[MAXC];
#pragma unroll
            for (int k = 0; k < MAXC; ++k) v[k] = base[k][pos[k] + step - 1];
#pragma unroll
            for (int k = 0; k < MAXC; ++k) pos[k] += v[k] < key ? step : 0;
        }
#pragma unroll
        for (int k = 0; k < MAXC; ++k) {
            // positions are in [0, SC - 1]: the last element needs its own test
            const bool live = first + k + (first + k >= c ? 1 : 0) < nch;
            const int full = pos[k] + ((live && pos[k] == SC - 1 && base[k][SC - 1] < key) ? 1 : 0);
            total += live ? full : 0;
        }
    }
    return total;
}

// (B, order 1) grid (nch * 8, B), 256 threads: one key per thread. Writes sorted1[b][rank] for the candidates and nvalid[b].
template <int MAXC>
__global__ __launch_bounds__(256) void nms_merge1_kernel(const unsigned long long* __restrict__ chunked, const int* __restrict__ chunk_valid,
                                                         int nch, int n, unsigned long long* __restrict__ sorted1, int* __restrict__ nvalid,
                                                         unsigned long long* __restrict__ row_any, int W) {
    const int b = blockIdx.y, c = blockIdx.x >> 3, ploc = (blockIdx.x & 7) * 256 + threadIdx.x;
    const unsigned long long* img = chunked + (size_t)b * nch * SC;
    if (c * SC + ploc < W) row_any[(size_t)b * W + c * SC + ploc] = 0ull;       // the mask kernel ORs into it (nch SC >= n >= W)
    if (blockIdx.x == 0 && threadIdx.x == 0) {
        int nv = 0;
        for (int k = 0; k < nch; ++k) nv += chunk_valid[b * SC_MAXCH + k];
        nvalid[b] = nv;
    }
    const unsigned long long key = img[(size_t)c * SC + ploc];
    if (key == ~0ull) return;
    const int rank = ploc + rank_in_other_chunks<MAXC>(img, nch, c, key);
    sorted1[(size_t)b * n + rank] = key;
}

// (B, order 2) same geometry; the destination row q also receives what nms_gather2_kernel wrote (box, index, rank, class range)
template <int MAXC>
__global__ __launch_bounds__(256) void nms_merge2_kernel(const float* __restrict__ boxes, const unsigned long long* __restrict__ chunked,
                                                         const int* __restrict__ nvalid, int nch, int n, int W, int center,
                                                         unsigned long long* __restrict__ sorted2, int* __restrict__ order,
                                                         int* __restrict__ grank, SBox* __restrict__ sbox, int* __restrict__ blk_lo,
                                                         int* __restrict__ blk_hi) {
    const int b = blockIdx.y, c = blockIdx.x >> 3, ploc = (blockIdx.x & 7) * 256 + threadIdx.x;
    const unsigned long long* img = chunked + (size_t)b * nch * SC;
    const unsigned long long k = img[(size_t)c * SC + ploc];
    const int cls = (int)(unsigned)((k >> 40) & 0xfffull);
    if (cls == 0xfff) return;                                   // not a candidate (they all sort behind the candidates)
    const int q = ploc + rank_in_other_chunks<MAXC>(img, nch, c, k);
    const int nv = nvalid[b];
    const int i = (int)(unsigned)(k & 0xfffffull);
    sorted2[(size_t)b * n + q] = ((unsigned long long)b << 52) | k;
    const float* s = boxes + ((size_t)b * n + i) * 6;
    float x = s[0], y = s[1];
    const float w = s[2], h = s[3];
    if (center) { x = x - w / 2.0f; y = y - h / 2.0f; }            // utils.py:60-64
    SBox o;
    o.x1 = x; o.y1 = y; o.x2 = x + w; o.y2 = y + h; o.area = w * h; o.cls = s[5]; o.w = w; o.h = h;
    sbox[(size_t)b * n + q] = o;
    order[(size_t)b * n + q] = i;
    grank[(size_t)b * n + q] = (int)(unsigned)((k >> 20) & 0xfffffull);
    if ((q & 63) == 0) blk_lo[(size_t)b * W + (q >> 6)] = cls;
    if ((q & 63) == 63 || q == nv - 1) blk_hi[(size_t)b * W + (q >> 6)] = cls;
}

// ---- the same two launches as a stand-alone ascending sort of UNIQUE 64-bit keys (none equal to ~0): the orderings of
// calc_mAP (utils.py:206,232: detections by class and descending objectness, ground truths by class and image - Python's
// stable sorts become one sort of (major | minor | original index) keys)
__global__ __launch_bounds__(SC_THREADS) void sort_chunks_kernel(const unsigned long long* __restrict__ in, int n, unsigned long long* __restrict__ chunked) {
    __shared__ __attribute__((aligned(16))) char lds[(SC / 8) * SC_ROW];
    const int c = blockIdx.x, tid = threadIdx.x;
#pragma unroll
    for (int e = 0; e < (1 << SC_LG); ++e) {
        const int loc = e * SC_THREADS + tid, i = c * SC + loc;
        *reinterpret_cast<unsigned long long*>(lds + sc_addr(loc)) = i < n ? in[i] : ~0ull;
    }
    __syncthreads();
    chunk_sort_lds<SC_LG>(lds, tid);
#pragma unroll
    for (int e = 0; e < (1 << SC_LG); ++e) {
        const int loc = e * SC_THREADS + tid;
        chunked[(size_t)c * SC + loc] = *reinterpret_cast<const unsigned long long*>(lds + sc_addr(loc));
    }
}
template <int MAXC>
__global__ __launch_bounds__(256) void sort_merge_kernel(const unsigned long long* __restrict__ chunked, int nch, unsigned long long* __restrict__ out) {
    const int c = blockIdx.x >> 3, ploc = (blockIdx.x & 7) * 256 + threadIdx.x;
    const unsigned long long key = chunked[(size_t)c * SC + ploc];
    if (key == ~0ull) return;                                   // padding of the last chunk
    out[ploc + rank_in_other_chunks<MAXC>(chunked, nch, c, key)] = key;
}

// The words of the diagonal and of the SCAN_NEAR column blocks after it are ALSO stored as near[b][rb][d][lane] (d = cb - rb):
// the scan reads exactly those for all 64 rows of a block at once, and in the mask itself they are 64 separate cache lines
// (one per row, W words apart) - an uncoalesced, HBM-latency load on the scan's serial chain; here they are 512 contiguous bytes.
constexpr int SCAN_NEAR = 3;

// grid (1, B, W), MS_WAVES waves: wave w takes row block rb against column blocks rb + w, + MS_WAVES, ... while the class ranges overlap
constexpr int MS_WAVES = 4;      // waves per workgroup = column-block stride (a 64-thread workgroup per (row block, stride) was dispatch-bound at 80
                                 // classes). Same-box A/B: 8 waves give 2 classes 540 -> 572 M boxes/s but 80 classes 1,590 -> 1,527 M; 2 waves 383 M / 1,564 M
__global__ __launch_bounds__(64 * MS_WAVES) void nms_mask_sorted_kernel(const SBox* __restrict__ sbox, const int* __restrict__ nvalid,
                                                             const int* __restrict__ blk_lo, const int* __restrict__ blk_hi, int n,
                                                             int W, float thr, unsigned long long* __restrict__ mask,
                                                             unsigned long long* __restrict__ row_any, unsigned long long* __restrict__ near) {
    // row block slowest, image in the middle: the workgroups are dispatched in that order, and a row block's work shrinks
    // along its class range, so with the image slowest the last images' longest workgroups started late and ran alone at the end
    const int rb = blockIdx.z, b = blockIdx.y;
    const int nv = nvalid[b];
    if (rb * 64 >= nv) return;
    const int nblk = (nv + 63) / 64;
    // every wave works alone (its own staging buffer, its own trip count): no workgroup barrier below, only the wave's own
    // program order (LDS operations of one wave execute in order; the wave barrier keeps the compiler from moving them).
    // (Requesting the row's box, the class bounds and the first column block before nvalid is known was tried: the compiler
    // sinks the loads below the early exit again and parks the column block in LDS - 24 -> 48 us at 80 classes.)
    __shared__ SBox cols_all[MS_WAVES][64];
    const int lane = threadIdx.x & 63, wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    SBox* cols = cols_all[wave];
    const SBox* sb = sbox + (size_t)b * n;
    const int i = rb * 64 + lane;
    const bool active = i < nv;
    const int hi_r = blk_hi[(size_t)b * W + rb];
    const int cb_first = rb + blockIdx.x * MS_WAVES + wave;
    // a wave whose first column block is already outside the row block's class range has nothing to do (with 80 classes
    // most of a workgroup's waves): it leaves before it asks for its rows' boxes
    if (cb_first >= nblk || blk_lo[(size_t)b * W + cb_first] > hi_r) return;
    const SBox me = sb[active ? i : nv - 1];
    const RowBounds rbnd = row_bounds(me, thr);
    for (int cb = cb_first; cb < nblk; cb += gridDim.x * MS_WAVES) {
        if (blk_lo[(size_t)b * W + cb] > hi_r) break;    // classes ascend: no later block can match either
        const int j = cb * 64 + lane;
        __builtin_amdgcn_wave_barrier();
        if (j < nv) cols[lane] = sb[j];
        __builtin_amdgcn_wave_barrier();
        const int lim = nv - cb * 64 < 64 ? nv - cb * 64 : 64;
        const unsigned long long word = suppression_word(me, active, i, cols, lim, cb * 64, thr, sb + cb * 64, rbnd);
        if (active) mask[((size_t)b * n + i) * W + cb] = word;
        if (active && cb - rb <= SCAN_NEAR) near[(((size_t)b * W + rb) * (SCAN_NEAR + 1) + (cb - rb)) * 64 + lane] = word;
        const unsigned long long bal = __ballot(active && word != 0ull);
        if (cb > rb && lane == 0 && bal) atomicOr(&row_any[(size_t)b * W + rb], bal);
    }
}

// kept boxes were written to slot[global rank] (-1 elsewhere): ordered compaction = the reference's output order.
// One 1,024-thread workgroup per image; a pass covers CP_R rounds of 4,096 ranks (thread: 4 consecutive ranks per round) with all
// loads issued together and TWO barriers: per-(round, wave) counts to LDS, one wave scans the CP_R x 16 counts, everybody writes.
// (The first version did a round of 1,024 ranks per two barriers: 10 us for 10,000 ranks, one of the larger pieces at 80 classes.)
constexpr int CP_R = 8;
static_assert(CP_R * 16 == 128, "the prefix below scans two values per lane");
__global__ __launch_bounds__(1024) void nms_compact_kernel(const int* __restrict__ slot, const int* __restrict__ nvalid, int n,
                                                           int* __restrict__ keep_idx) {
    __shared__ int wsum[CP_R * 16], wpre[CP_R * 16 + 1];
    const int b = blockIdx.x, tid = threadIdx.x, wave = tid >> 6, lane = tid & 63;
    const int nv = nvalid[b];
    const int* sl = slot + (size_t)b * n;
    int* out = keep_idx + (size_t)b * n;
    const unsigned long long lt = (1ull << lane) - 1ull;
    int base = 0;
    for (int g0 = 0; g0 < nv; g0 += CP_R * 4096) {
        int v[CP_R][4], before[CP_R];
#pragma unroll
        for (int r = 0; r < CP_R; ++r)
#pragma unroll
            for (int k = 0; k < 4; ++k) {
                const int g = g0 + r * 4096 + tid * 4 + k;
                v[r][k] = g < nv ? sl[g] : -1;
            }
#pragma unroll
        for (int r = 0; r < CP_R; ++r) {
            int wave_total = 0;
            before[r] = 0;
#pragma unroll
            for (int k = 0; k < 4; ++k) {
                const unsigned long long bal = __ballot(v[r][k] >= 0);
                before[r] += __popcll(bal & lt);
                wave_total += __popcll(bal);
            }
            if (lane == 0) wsum[r * 16 + wave] = wave_total;
        }
        __syncthreads();
        if (wave == 0) {                                  // exclusive prefix of the 128 counts in (round, wave) order
            const int a = wsum[lane], c = wsum[lane + 64];
            int sa = a, sc = c;
#pragma unroll
            for (int d = 1; d < 64; d <<= 1) {
                const int ta = __shfl_up(sa, d), tc = __shfl_up(sc, d);
                if (lane >= d) { sa += ta; sc += tc; }
            }
            const int tota = __shfl(sa, 63);
            wpre[lane] = sa - a;
            wpre[lane + 64] = tota + sc - c;
            if (lane == 63) wpre[CP_R * 16] = tota + sc;
        }
        __syncthreads();
#pragma unroll
        for (int r = 0; r < CP_R; ++r) {
            int off = base + wpre[r * 16 + wave] + before[r];
#pragma unroll
            for (int k = 0; k < 4; ++k)
                if (v[r][k] >= 0) out[off++] = v[r][k];
        }
        base += wpre[CP_R * 16];
        __syncthreads();                                  // the counts are reused by the next pass
    }
}

// grid (W, W, B), 64 threads. word (i, cb): bit jj set <=> j = cb*64+jj > i, same class, !(iou < thr)
__global__ __launch_bounds__(64) void nms_mask_kernel(const SBox* __restrict__ sbox, const int* __restrict__ nvalid, int n,
                                                      int W, float thr, unsigned long long* __restrict__ mask,
                                                      unsigned long long* __restrict__ row_any) {
    const int cb = blockIdx.x, rb = blockIdx.y, b = blockIdx.z;
    if (cb < rb) return;
    const int nv = nvalid[b];
    if (rb * 64 >= nv || cb * 64 >= nv) return;
    __shared__ SBox cols[64];
    const SBox* sb = sbox + (size_t)b * n;
    const int j = cb * 64 + threadIdx.x;
    if (j < nv) cols[threadIdx.x] = sb[j];
    __syncthreads();
    const int i = rb * 64 + threadIdx.x;
    const bool active = i < nv;
    const SBox me = sb[active ? i : nv - 1];
    const int lim = nv - cb * 64 < 64 ? nv - cb * 64 : 64;
    const unsigned long long word = suppression_word(me, active, i, cols, lim, cb * 64, thr);
    if (!active) return;
    mask[((size_t)b * n + i) * W + cb] = word;
    // row_any[b][rb] bit t: row rb*64+t has a suppression bit in some LATER column block (integer OR:
    // order-independent, so still deterministic)
    const unsigned long long bal = __ballot(word != 0ull);
    if (cb > rb && threadIdx.x == 0 && bal) atomicOr(&row_any[(size_t)b * W + rb], bal);
}

// one 256-thread workgroup per image. Per 64-row block: wave 0 resolves the diagonal word (the only
// sequential part of greedy NMS; skipped outright when no row of the block has a diagonal bit and none
// is already removed), then every thread owns column words and ORs in the rows that are kept AND have
// any suppression bit at all (row_any, written by the mask kernel) — with many classes most rows
// suppress nothing and cost no memory traffic.
__global__ __launch_bounds__(256) void nms_scan_kernel(const unsigned long long* __restrict__ mask, const int* __restrict__ order,
                                                       const int* __restrict__ nvalid, const unsigned long long* __restrict__ row_any,
                                                       int n, int W, int* __restrict__ keep_idx, int* __restrict__ keep_count) {
    extern __shared__ unsigned long long removed[];     // [W] + 2 words: kept broadcast, kept-and-nonzero broadcast
    const int b = blockIdx.x;
    const int nv = nvalid[b];
    const int tid = threadIdx.x;
    for (int c = tid; c < W + 2; c += 256) removed[c] = 0;
    __syncthreads();
    const unsigned long long* mk = mask + (size_t)b * n * W;
    const unsigned long long* any = row_any + (size_t)b * W;
    const int* ord = order + (size_t)b * n;
    int* out = keep_idx + (size_t)b * n;
    int count = 0;                                      // meaningful in wave 0 only
    const int nblk = (nv + 63) / 64;
    // the row block's diagonal words, row_any word and original indices are fetched ONE block ahead: the loop is a serial
    // chain of ~160 iterations per image, and a dependent global load per iteration was most of each iteration
    unsigned long long d_next = 0ull, any_next = 0ull;
    int ord_next = 0;
    if (tid < 64 && nblk > 0) {
        d_next = tid < nv ? mk[(size_t)tid * W] : 0ull;
        any_next = any[0];
        ord_next = tid < nv ? ord[tid] : 0;
    }
    for (int rb = 0; rb < nblk; ++rb) {
        const int rows = nv - rb * 64 < 64 ? nv - rb * 64 : 64;
        if (tid < 64) {
            const int i = rb * 64 + tid;
            const unsigned long long d = d_next, any_rb = any_next;
            const int ord_i = ord_next;
            if (rb + 1 < nblk) {
                const int in = i + 64;
                d_next = in < nv ? mk[(size_t)in * W + rb + 1] : 0ull;
                any_next = any[rb + 1];
                ord_next = in < nv ? ord[in] : 0;
            }
            unsigned long long rem = removed[rb];
            const unsigned long long rowmask = rows == 64 ? ~0ull : ((1ull << rows) - 1ull);
            unsigned long long kept;
            if (__ballot(d != 0ull) == 0ull) {
                kept = rowmask & ~rem;                  // nothing inside this block suppresses anything
            } else {
                // only rows that HAVE a diagonal bit can change the outcome; visit those in order
                const unsigned dlo = (unsigned)d, dhi = (unsigned)(d >> 32);
                unsigned long long cand = __ballot(d != 0ull) & ~rem;     // removed rows cannot act: dropped from the walk as it goes
                while (cand) {
                    const int t = __builtin_ctzll(cand);
                    const unsigned wl = __builtin_amdgcn_readlane(dlo, t);
                    const unsigned wh = __builtin_amdgcn_readlane(dhi, t);
                    rem |= ((unsigned long long)wh << 32) | wl;
                    cand &= cand - 1ull;
                    cand &= ~rem;
                }
                kept = rowmask & ~rem;
            }
            if (tid < rows && ((kept >> tid) & 1ull)) {
                const int pos = count + __popcll(kept & ((1ull << tid) - 1ull));
                out[pos] = ord_i;
            }
            count += __popcll(kept);
            if (tid == 0) {
                removed[W] = kept;
                removed[W + 1] = kept & any_rb;         // kept rows that suppress something in a later block
            }
        }
        __syncthreads();
        unsigned long long work = removed[W + 1];
        if (work) {
            for (int c = rb + 1 + tid; c < nblk; c += 256) {     // column blocks >= nblk were never written
                unsigned long long acc = 0;
                unsigned long long wk = work;
                while (wk) {
                    const int t = __builtin_ctzll(wk);
                    wk &= wk - 1;
                    acc |= mk[(size_t)(rb * 64 + t) * W + c];
                }
                if (acc) removed[c] |= acc;             // column c is owned by exactly this thread
            }
        }
        __syncthreads();
    }
    if (tid == 0) keep_count[b] = count;
}

// ---- scan for class-sorted rows: classes never interact, so the image's rows are cut at class boundaries into up to 16
// ranges of about equal length and ONE WAVE runs the greedy pass over each range, with no workgroup barrier inside the
// loop (the per-image chain of ~160 row blocks becomes ~10 per wave at 80 classes; with 2 classes two waves work).
// A 64-row block that straddles a cut is visited by both neighbours, each with its own row mask. Kept rows are reported
// as bits of an LDS word array (ds_or: straddling blocks) and written to slot[global rank] at the end of the kernel.
constexpr int SCAN_MLP = 8;

// ---- few classes: a TEAM of waves per class range ------------------------------------------------------------------------
// With 2 classes only 2 of an image's 16 waves had a range, and each spent its time waiting for the kept rows' mask words
// (5-6 dependent rounds of loads per 64-row block, ~2.2 us per block, 78 blocks in a chain). When at most half the waves
// have a range, every range gets G = K / ranges waves: a leader that walks the serial chain and G - 1 helpers that do the
// far pushes, talking through LDS (all waves of a workgroup are resident, so a spin on an LDS word cannot deadlock; LDS
// operations of one wave execute in order, so "data, then counter" needs no wait in between):
//   * columns rb + 1 .. rb + SCAN_NEAR of row block rb are the leader's: it loads the 64 rows' words of those columns
//     UNCONDITIONALLY two iterations ahead (no dependence on which rows are kept) and, once the kept word of block rb - d is
//     known, ORs the kept rows' words together across the wave (DPP) - no memory latency on the chain;
//   * columns beyond belong to the helpers, in S sets that take the row blocks in turn (block k -> set k % S): the leader
//     publishes kept & row_any of block rb in a ring, the helpers of that set split its rows (t % P), OR the words into the
//     team's removed[] (ds_or) and report the block done. The leader only needs block rb - SCAN_NEAR - 1 finished before it
//     resolves block rb, and it reads the progress counters and removed[rb + 1] one iteration early (counters first: if they
//     pass, the word read after them is complete), so no LDS round trip sits on the chain either.
constexpr int SCAN_RING = 8;                // > SCAN_NEAR + 1: the leader is never further ahead of a helper than that
constexpr int SCAN_HELP_MLP = 16;           // rows in flight per helper (x 2 column slices)

__device__ __forceinline__ unsigned wave_or32(unsigned v) {       // OR over the 64 lanes (all active), result wave-uniform
    v |= (unsigned)__builtin_amdgcn_update_dpp(0, (int)v, 0xB1, 0xf, 0xf, true);      // quad_perm [1,0,3,2]
    v |= (unsigned)__builtin_amdgcn_update_dpp(0, (int)v, 0x4E, 0xf, 0xf, true);      // quad_perm [2,3,0,1]
    v |= (unsigned)__builtin_amdgcn_update_dpp(0, (int)v, 0x141, 0xf, 0xf, true);     // row_half_mirror
    v |= (unsigned)__builtin_amdgcn_update_dpp(0, (int)v, 0x140, 0xf, 0xf, true);     // row_mirror: every lane holds its row's OR
    return (unsigned)__builtin_amdgcn_readlane((int)v, 0) | (unsigned)__builtin_amdgcn_readlane((int)v, 16) |
           (unsigned)__builtin_amdgcn_readlane((int)v, 32) | (unsigned)__builtin_amdgcn_readlane((int)v, 48);
}
__device__ __forceinline__ unsigned long long uniform64(unsigned long long v) {
    return ((unsigned long long)(unsigned)__builtin_amdgcn_readfirstlane((int)(unsigned)(v >> 32)) << 32) |
           (unsigned)__builtin_amdgcn_readfirstlane((int)(unsigned)v);
}

// one member of a team; returns the number of kept rows (leader) or 0 (helpers). `removed` is the team's word array, `ctrl`
// its control block (both zeroed): ring[SCAN_RING] u64, then int pub, pad, int prog[H]. `single`: the range is one class
// bucket, so every word the team touches was written by the mask kernel. Byte offsets into the image's mask are 32-bit.
__device__ __forceinline__ int scan_team(const unsigned long long* __restrict__ mk, const unsigned long long* __restrict__ any,
                                         const int* lo, const int* hi, int W, int s0, int s1, bool single, unsigned long long* removed,
                                         unsigned long long* ctrl, int member, int H, int lane, unsigned long long* keptw_b,
                                         const unsigned long long* __restrict__ near_b) {
    // relaxed workgroup-scope atomics, not volatile: the address-space inference leaves volatile accesses as flat loads
    unsigned long long* ring = ctrl;
    int* pub = (int*)(ctrl + SCAN_RING);
    int* prog = pub + 2;
    auto ld = [](const int* q) { return __hip_atomic_load(q, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP); };
    auto st = [](int* q, int v) { __hip_atomic_store(q, v, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP); };
    const int rb0 = s0 >> 6, rb1 = (s1 - 1) >> 6;
    const unsigned W8 = (unsigned)W * 8u;
    const char* mkb = reinterpret_cast<const char*>(mk);
    const int S = H >= 6 ? 3 : (H >= 2 ? 2 : 1);
    auto rows_of = [&](int rb) -> unsigned long long {   // rows of block rb inside [s0, s1)
        const int a = s0 - rb * 64 > 0 ? s0 - rb * 64 : 0, z = s1 - rb * 64 < 64 ? s1 - rb * 64 : 64;
        const unsigned long long upto = z == 64 ? ~0ull : ((1ull << z) - 1ull);
        return upto & ~((1ull << a) - 1ull);
    };
    if (member == 0) {
        // branch-free prefetch: the address is clamped into the range's rows and column blocks and the word is masked when it
        // is used (a load under a branch makes the compiler wait for vmcnt(0), i.e. for the prefetches just issued, too)
        auto word_of = [&](int rr, int c) -> unsigned long long {       // row rr * 64 + lane, column block c = rr + d, d <= SCAN_NEAR
            const int rs = rr < rb0 ? rb0 : (rr > rb1 ? rb1 : rr);      // any valid address; what is out of range is never used
            return near_b[(size_t)((rs * (SCAN_NEAR + 1) + (c - rr)) * 64 + lane)];
        };
        auto near_ok = [&](int rr, int c) -> bool { return single || lo[c] <= hi[rr > rb0 ? rr : rb0]; };   // else never written
        // two register sets, used alternately (the loop is unrolled by two): a set is consumed and then refilled for the block
        // two iterations ahead, so no in-flight load result is ever copied (a rotation n0 = n1 would wait for vmcnt(0))
        // row_any through the VECTOR memory path: a scalar load shares lgkmcnt with LDS and returns out of order, so every wait for
        // an LDS word would also wait for the s_load just issued - a global-memory latency on the chain, every iteration
        int vzero;
        asm volatile("v_mov_b32 %0, 0" : "=v"(vzero));
        struct Pre { unsigned long long d, a, n[SCAN_NEAR]; };
        auto fill = [&](Pre& p, int c) {
            p.d = word_of(c, c);
            p.a = any[(c < rb1 ? c : rb1) + vzero];
#pragma unroll
            for (int q = 0; q < SCAN_NEAR; ++q) p.n[q] = word_of(c - (q + 1), c);
        };
        Pre X, Y;
        fill(X, rb0);
        fill(Y, rb0 + 1);
        unsigned long long kp[SCAN_NEAR];              // kept words of blocks rb - 1, rb - 2, ... (0 before the range)
#pragma unroll
        for (int q = 0; q < SCAN_NEAR; ++q) kp[q] = 0ull;
        int count = 0;
        const int lane_set = lane % S;
        int needv = 0, phase = 0;                       // per lane (= helper): blocks that helper must have finished
        int pg = 0;                                     // its progress counter, read one iteration early ...
        unsigned long long rv = removed[rb0];           // ... followed by the removed word of the block
        auto iter = [&](int rb, Pre& p) {
            const int k = rb - rb0;
            const unsigned long long rowmask = rows_of(rb);
            const unsigned long long d = ((rowmask >> lane) & 1ull) ? p.d : 0ull, any_rb = uniform64(p.a);
            unsigned long long v = 0ull;
#pragma unroll
            for (int q = 0; q < SCAN_NEAR; ++q) v |= (near_ok(rb - (q + 1), rb) && ((kp[q] >> lane) & 1ull)) ? p.n[q] : 0ull;
            const unsigned long long near_rem = ((unsigned long long)wave_or32((unsigned)(v >> 32)) << 32) | wave_or32((unsigned)v);
            fill(p, rb + 2);
            const int kk = k - SCAN_NEAR - 1;           // the block whose far pushes become necessary now (its set: kk % S)
            if (kk >= 0) {
                if (lane_set == phase) needv = kk + 1;
                phase = phase + 1 == S ? 0 : phase + 1;
            }
            if (__ballot(lane < H && pg < needv) != 0ull) {          // a helper is behind: wait, then read the word again
                do {
                    __builtin_amdgcn_s_sleep(1);
                    pg = lane < H ? ld(prog + lane) : 0x7fffffff;
                } while (__ballot(lane < H && pg < needv) != 0ull);
                asm volatile("" ::: "memory");
                rv = removed[rb];
            }
            unsigned long long rem = uniform64(rv) | near_rem;
            {                                                        // next block's counters, then its word (in this order)
                pg = lane < H ? ld(prog + lane) : 0x7fffffff;
                asm volatile("" ::: "memory");
                rv = removed[rb + 1 <= rb1 ? rb + 1 : rb1];
            }
            // only rows with a diagonal bit can change the outcome, and only while they are not removed themselves: dropping
            // the removed rows from the walk after every step makes its length the number of KEPT rows with a diagonal bit
            // (clustered boxes: all 64 rows have one, two or three survive - the walk was most of the scan there)
            unsigned long long cand = __ballot(d != 0ull) & ~rem;
            const unsigned dlo = (unsigned)d, dhi = (unsigned)(d >> 32);
            while (cand) {
                const int t = __builtin_ctzll(cand);
                const unsigned wl = __builtin_amdgcn_readlane(dlo, t);
                const unsigned wh = __builtin_amdgcn_readlane(dhi, t);
                rem |= ((unsigned long long)wh << 32) | wl;
                cand &= cand - 1ull;
                cand &= ~rem;
            }
            const unsigned long long kept = rowmask & ~rem;
            const unsigned long long work = kept & any_rb;
            if (lane == 0) {
                __hip_atomic_store(ring + (k & (SCAN_RING - 1)), work, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
                asm volatile("" ::: "memory");          // the ring entry before the counter (in-order LDS: no wait needed)
                st(pub, k + 1);
                if (kept) atomicOr(&keptw_b[rb], kept);
            }
            count += __popcll(kept);
#pragma unroll
            for (int q = SCAN_NEAR - 1; q > 0; --q) kp[q] = kp[q - 1];
            kp[0] = kept;
        };
        for (int rb = rb0; rb <= rb1; rb += 2) {
            iter(rb, X);
            if (rb + 1 <= rb1) iter(rb + 1, Y);
        }
        return count;
    }
    const int h = member - 1;
    const int set = h % S, part = h / S, P = (H - set + S - 1) / S;      // this helper: part `part` of the P helpers of its set
    const unsigned long long sel = __ballot(lane % P == part);
    for (int k = set; k <= rb1 - rb0; k += S) {
        const int rb = rb0 + k;
        while (__builtin_amdgcn_readfirstlane(ld(pub)) <= k) __builtin_amdgcn_s_sleep(1);
        asm volatile("" ::: "memory");
        const unsigned long long work = uniform64(__hip_atomic_load(ring + (k & (SCAN_RING - 1)), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP)) & sel;
        if (work) {
            const int hi_r = hi[rb];
            const char* blkp = mkb + (size_t)((unsigned)(rb * 64) * W8);
            for (int cb = rb + SCAN_NEAR + 1; cb <= rb1; cb += 128) {           // two column slices (lane, lane + 64) per pass
                const int c0 = cb + lane, c1 = c0 + 64;
                const int cc0 = c0 < rb1 ? c0 : rb1, cc1 = c1 < rb1 ? c1 : rb1;   // clamped: unconditional loads, masked below
                const bool ok0 = c0 <= rb1 && (single || lo[cc0] <= hi_r);       // else never written: classes above this row block's
                const bool ok1 = c1 <= rb1 && (single || lo[cc1] <= hi_r);
                const unsigned o0 = (unsigned)cc0 * 8u, o1 = (unsigned)cc1 * 8u;
                unsigned long long acc0 = 0ull, acc1 = 0ull, wk = work;
                int tl = 0;
                if (__ballot(ok1) != 0ull) {
                    while (wk) {                        // SCAN_HELP_MLP rows' words in flight (a spent slot repeats the last row)
                        unsigned long long w0[SCAN_HELP_MLP], w1[SCAN_HELP_MLP];
#pragma unroll
                        for (int u = 0; u < SCAN_HELP_MLP; ++u) {
                            if (wk) { tl = __builtin_ctzll(wk); wk &= wk - 1ull; }
                            const char* rowp = blkp + (size_t)((unsigned)tl * W8);
                            w0[u] = *reinterpret_cast<const unsigned long long*>(rowp + o0);
                            w1[u] = *reinterpret_cast<const unsigned long long*>(rowp + o1);
                        }
#pragma unroll
                        for (int u = 0; u < SCAN_HELP_MLP; ++u) { acc0 |= w0[u]; acc1 |= w1[u]; }
                    }
                } else {
                    while (wk) {
                        unsigned long long w0[SCAN_HELP_MLP];
#pragma unroll
                        for (int u = 0; u < SCAN_HELP_MLP; ++u) {
                            if (wk) { tl = __builtin_ctzll(wk); wk &= wk - 1ull; }
                            w0[u] = *reinterpret_cast<const unsigned long long*>(blkp + (size_t)((unsigned)tl * W8) + o0);
                        }
#pragma unroll
                        for (int u = 0; u < SCAN_HELP_MLP; ++u) acc0 |= w0[u];
                    }
                }
                if (ok0 && acc0) atomicOr(&removed[c0], acc0);       // several helpers share a word
                if (ok1 && acc1) atomicOr(&removed[c1], acc1);
            }
        }
        asm volatile("" ::: "memory");                  // the ORs before the progress counter (in-order LDS)
        if (lane == 0) st(prog + h, k + 1);
    }
    return 0;
}

__global__ __launch_bounds__(1024) void nms_scan_classes_kernel(const unsigned long long* __restrict__ mask,
                                                               const unsigned long long* __restrict__ sorted2,
                                                               const int* __restrict__ nvalid, const unsigned long long* __restrict__ row_any,
                                                               const int* __restrict__ blk_lo, const int* __restrict__ blk_hi, int n, int W,
                                                               const int* __restrict__ order, const int* __restrict__ grank,
                                                               int* __restrict__ slot, int* __restrict__ keep_count,
                                                               const unsigned long long* __restrict__ near,
                                                               const unsigned long long* __restrict__ sorted1, int* __restrict__ keep_idx) {
    extern __shared__ unsigned long long lds[];         // [K][W] removed words per wave, kept[W], then lo[W], hi[W] (int), the count, cuts
    const int b = blockIdx.x, tid = threadIdx.x, wave = __builtin_amdgcn_readfirstlane(tid >> 6), lane = tid & 63, K = blockDim.x >> 6;
    const int nv = nvalid[b];
    const int nblk = (nv + 63) / 64;
    unsigned long long* removed = lds + (size_t)wave * W;
    unsigned long long* keptl = lds + (size_t)K * W;    // [W] kept rows, bit q % 64 of word q / 64 (ds_or: a block that straddles a cut has two writers)
    int* lo = (int*)(lds + (size_t)(K + 1) * W);
    int* hi = lo + W;
    int* total = hi + W;
    for (int c = tid; c < (K + 1) * W; c += blockDim.x) lds[c] = 0ull;
    for (int c = tid; c < nblk; c += blockDim.x) { lo[c] = blk_lo[(size_t)b * W + c]; hi[c] = blk_hi[(size_t)b * W + c]; }
    if (tid == 0) *total = 0;
    __syncthreads();
    // range of this wave: cuts at the first class boundary at or after k nv / K (upper bound of the bucket just before it).
    // Wave k - 1 finds cut k: the first block that STARTS above the target's bucket from the block classes in LDS, then the
    // exact row with one 64-key load of the block before it (a per-thread binary search over the keys was ~14 dependent
    // global loads, 10+ us before any wave could start).
    const unsigned long long* keys = sorted2 + (size_t)b * n;
    int* cuts = total + 1;                              // [K + 1]
    if (tid == 0) { cuts[0] = 0; cuts[K] = nv; }
    if (wave + 1 < K) {
        const int target = (int)((long long)nv * (wave + 1) / K);
        int res = 0;
        if (target > 0) {
            const int bucket = (int)(unsigned)((keys[target - 1] >> 40) & 0xfffull);      // uniform
            const int tb = target >> 6;
            int c = nblk;                               // first block after tb whose first row is above the bucket
            for (int c0 = tb + 1; c0 < nblk; c0 += 64) {
                const unsigned long long above = __ballot(c0 + lane < nblk && lo[c0 + lane] > bucket);
                if (above) { c = c0 + __builtin_ctzll(above); break; }
            }
            const int q = (c - 1) * 64 + lane;          // the boundary is inside block c - 1 or at the start of block c
            const bool in = q >= target && q < nv;
            const unsigned long long key = keys[in ? q : target - 1];
            const unsigned long long above = __ballot(in && (int)(unsigned)((key >> 40) & 0xfffull) > bucket);
            res = above ? (c - 1) * 64 + __builtin_ctzll(above) : (c * 64 < nv ? c * 64 : nv);
        }
        if (lane == 0) cuts[wave + 1] = res;
    }
    __syncthreads();
    int R = 0;                                          // ranges that hold rows
    for (int k = 0; k < K; ++k) R += cuts[k] < cuts[k + 1] ? 1 : 0;
    const int G = (R > 0 && 2 * R <= K && (unsigned long long)n * W * 8ull < (1ull << 32)) ? K / R : 1;   // teams use 32-bit byte offsets
    const unsigned long long* mk = mask + (size_t)b * n * W;
    const unsigned long long* any = row_any + (size_t)b * W;
    int s0 = 0, s1 = 0;
    int count = 0;
    if (G >= 2) {                                       // few classes: teams of G waves (see scan_team)
        const int team = wave / G, member = wave - team * G;
        if (team < R) {
            int idx = -1;
            for (int k = 0; k < K; ++k)
                if (cuts[k] < cuts[k + 1] && ++idx == team) { s0 = cuts[k]; s1 = cuts[k + 1]; }
            unsigned long long* team_removed = lds + (size_t)(team * G) * W;       // the leader's array; the first helper's holds the control block
            const bool single = ((keys[s0] >> 40) & 0xfffull) == ((keys[s1 - 1] >> 40) & 0xfffull);
            count = scan_team(mk, any, lo, hi, W, s0, s1, single, team_removed, team_removed + W, member, G - 1, lane, keptl,
                              near + (size_t)b * W * (SCAN_NEAR + 1) * 64);
        }
        s0 = s1 = 0;
    } else {
        s0 = cuts[wave]; s1 = cuts[wave + 1];
    }
    if (s0 < s1) {
        const int rb0 = s0 >> 6, rb1 = (s1 - 1) >> 6;
        auto rows_of = [&](int rb) -> unsigned long long {   // rows of block rb inside [s0, s1)
            const int a = s0 - rb * 64 > 0 ? s0 - rb * 64 : 0, z = s1 - rb * 64 < 64 ? s1 - rb * 64 : 64;
            const unsigned long long upto = z == 64 ? ~0ull : ((1ull << z) - 1ull);
            return upto & ~((1ull << a) - 1ull);
        };
        const unsigned long long* diag = near + (size_t)b * W * (SCAN_NEAR + 1) * 64 + lane;      // block rb's diagonal words: diag[rb * (NEAR + 1) * 64]
        unsigned long long d_next = ((rows_of(rb0) >> lane) & 1ull) ? diag[(size_t)rb0 * (SCAN_NEAR + 1) * 64] : 0ull;
        unsigned long long any_next = any[rb0];
        for (int rb = rb0; rb <= rb1; ++rb) {
            const unsigned long long rowmask = rows_of(rb);
            const unsigned long long d = d_next, any_rb = any_next;
            if (rb < rb1) {
                d_next = ((rows_of(rb + 1) >> lane) & 1ull) ? diag[(size_t)(rb + 1) * (SCAN_NEAR + 1) * 64] : 0ull;
                any_next = any[rb + 1];
            }
            const unsigned long long rem_v = removed[rb];   // wave-uniform: keep the serial chain on the scalar unit
            const unsigned rem_lo = (unsigned)__builtin_amdgcn_readfirstlane((unsigned)rem_v);
            const unsigned rem_hi = (unsigned)__builtin_amdgcn_readfirstlane((unsigned)(rem_v >> 32));
            unsigned long long rem = ((unsigned long long)rem_hi << 32) | rem_lo;
            // only rows with a diagonal bit can change the outcome, and only while they are not removed themselves: dropping
            // the removed rows from the walk after every step makes its length the number of KEPT rows with a diagonal bit
            // (clustered boxes: all 64 rows have one, two or three survive - the walk was most of the scan there)
            unsigned long long cand = __ballot(d != 0ull) & ~rem;
            const unsigned dlo = (unsigned)d, dhi = (unsigned)(d >> 32);
            while (cand) {
                const int t = __builtin_ctzll(cand);
                const unsigned wl = __builtin_amdgcn_readlane(dlo, t);
                const unsigned wh = __builtin_amdgcn_readlane(dhi, t);
                rem |= ((unsigned long long)wh << 32) | wl;
                cand &= cand - 1ull;
                cand &= ~rem;
            }
            const unsigned long long kept = rowmask & ~rem;
            count += __popcll(kept);
            if (lane == 0 && kept) atomicOr(&keptl[rb], kept);
            const unsigned long long work_v = kept & any_rb;  // kept rows with a bit in some later column block
            // wave-uniform by construction; said so to the compiler, which otherwise walks the row bits with ~12 VALU
            // instructions + a 64-bit multiply per row and lane. Scalar: s_ff1 / s_andn2 per row, the row base in SGPRs and
            // the lane's column as the 32-bit offset of the load.
            const unsigned long long work = ((unsigned long long)(unsigned)__builtin_amdgcn_readfirstlane((unsigned)(work_v >> 32)) << 32) |
                                            (unsigned)__builtin_amdgcn_readfirstlane((unsigned)work_v);
            if (work) {
                const int hi_r = hi[rb];
                const unsigned long long* blkp = mk + (size_t)rb * 64 * W;
                for (unsigned c = (unsigned)(rb + 1 + lane); (int)c <= rb1; c += 64) {
                    if (lo[c] > hi_r) break;            // never written: classes above this row block's
                    unsigned long long acc = 0ull, wk = work;
                    int tl = 0;
                    while (wk) {                        // SCAN_MLP rows' words in flight (a spent slot repeats the last row)
                        unsigned long long wv[SCAN_MLP];
#pragma unroll
                        for (int u = 0; u < SCAN_MLP; ++u) {
                            if (wk) { tl = __builtin_ctzll(wk); wk &= wk - 1ull; }
                            wv[u] = blkp[(size_t)tl * W + c];
                        }
#pragma unroll
                        for (int u = 0; u < SCAN_MLP; ++u) acc |= wv[u];
                    }
                    if (acc) removed[c] |= acc;         // this lane owns word c of this wave's array
                }
            }
            asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");   // the wave's own LDS writes before its next read
        }
    }
    if (lane == 0 && count) atomicAdd(total, count);
    __syncthreads();
    if (tid == 0) keep_count[b] = *total;
    // Output: the kept boxes' original indices in global score order (the reference's order).
    const int* gr = grank + (size_t)b * n;
    if (sorted1 != nullptr) {
        // order 1 is still around (chunk-sort path): mark the kept rows' GLOBAL RANKS in an LDS bitmap (ds_or instead of a
        // scattered global store per kept box), prefix the words' popcounts, and let thread g emit rank g - coalesced loads of
        // the rank's key (its low bits are the original index) and coalesced stores. Replaces the slot scatter, the slot
        // array's -1 fill and the separate compaction launch (5 + 10 us of the 115 us at 80 classes).
        unsigned long long* krank = lds;                 // the removed[] arrays are dead
        int* wpre = lo;                                  // and so are the block class bounds
        for (int c = tid; c < W; c += blockDim.x) krank[c] = 0ull;
        __syncthreads();
        for (int q0 = 0; q0 < nv; q0 += 4 * (int)blockDim.x) {
            int g[4];
#pragma unroll
            for (int u = 0; u < 4; ++u) {               // unconditional loads, four in flight
                const int q = q0 + u * (int)blockDim.x + tid;
                g[u] = gr[q < nv ? q : nv - 1];
            }
#pragma unroll
            for (int u = 0; u < 4; ++u) {
                const int q = q0 + u * (int)blockDim.x + tid;
                if (q < nv && ((keptl[q >> 6] >> (q & 63)) & 1ull)) atomicOr(&krank[g[u] >> 6], 1ull << (g[u] & 63));
            }
        }
        __syncthreads();
        if (wave == 0) {                                 // exclusive prefix of the words' popcounts: W <= 64 * 8 on this path
            const int per = (W + 63) >> 6, w0 = lane * per;
            int sum = 0;
            for (int c = 0; c < per; ++c) sum += w0 + c < W ? __popcll(krank[w0 + c]) : 0;
            int inc = sum;
#pragma unroll
            for (int d = 1; d < 64; d <<= 1) {
                const int t = __shfl_up(inc, d);
                if (lane >= d) inc += t;
            }
            int run = inc - sum;
            for (int c = 0; c < per; ++c)
                if (w0 + c < W) { wpre[w0 + c] = run; run += __popcll(krank[w0 + c]); }
        }
        __syncthreads();
        const unsigned long long* s1 = sorted1 + (size_t)b * n;
        int* out = keep_idx + (size_t)b * n;
        for (int g0 = 0; g0 < nv; g0 += 4 * (int)blockDim.x) {
            unsigned long long key[4];
#pragma unroll
            for (int u = 0; u < 4; ++u) {
                const int g = g0 + u * (int)blockDim.x + tid;
                key[u] = s1[g < nv ? g : nv - 1];
            }
#pragma unroll
            for (int u = 0; u < 4; ++u) {
                const int g = g0 + u * (int)blockDim.x + tid;
                if (g < nv) {
                    const unsigned long long w = krank[g >> 6];
                    if ((w >> (g & 63)) & 1ull) out[wpre[g >> 6] + __popcll(w & ((1ull << (g & 63)) - 1ull))] = (int)(unsigned)(key[u] & 0xffffffffull);
                }
            }
        }
        return;
    }
    // library-sort path: order 1 was overwritten; kept rows go to slot[global rank] (-1 elsewhere) and nms_compact_kernel follows
    const int* ord = order + (size_t)b * n;
    int* sl = slot + (size_t)b * n;
    for (int q0 = 0; q0 < nv; q0 += 4 * (int)blockDim.x) {
        int o[4], g[4];
#pragma unroll
        for (int u = 0; u < 4; ++u) {                   // unconditional loads, four in flight; the store is the conditional part
            const int q = q0 + u * (int)blockDim.x + tid, qc = q < nv ? q : nv - 1;
            o[u] = ord[qc];
            g[u] = gr[qc];
        }
#pragma unroll
        for (int u = 0; u < 4; ++u) {
            const int q = q0 + u * (int)blockDim.x + tid;
            if (q < nv && ((keptl[q >> 6] >> (q & 63)) & 1ull)) sl[g[u]] = o[u];
        }
    }
}

static const bool g_nms_rocprim = getenv("YOLO_NMS_ROCPRIM") != nullptr;       // A/B switch: the library sorts instead of the chunk sort + rank merge

struct NmsWs { int* nvalid; unsigned long long* row_any; int* order; SBox* sbox; unsigned long long* mask; size_t zero_bytes; size_t total;
               unsigned long long* keys_in; unsigned long long* keys_out; void* sort_tmp; size_t sort_tmp_bytes;
               int* grank; int* blk_lo; int* blk_hi; int* chunk_valid; unsigned long long* chunked;
               unsigned long long* sorted2; unsigned long long* near; };

static size_t sort_tmp_bytes(int b, int n) {
    size_t bytes = 0;
    (void)rocprim::radix_sort_keys(nullptr, bytes, (unsigned long long*)nullptr, (unsigned long long*)nullptr, (size_t)b * n, 0, 64,
                                   (hipStream_t)0);
    size_t bytes2 = 0;
    (void)rocprim::radix_sort_keys(nullptr, bytes2, (unsigned long long*)nullptr, (unsigned long long*)nullptr, (size_t)b * n, 40, 63,
                                   (hipStream_t)0);
    return bytes > bytes2 ? bytes : bytes2;
}

static NmsWs carve(void* base, int b, int n) {
    const int W = ceil_div(n > 0 ? n : 1, 64);
    size_t off = 0;
    auto take = [&](size_t bytes) { size_t o = off; off += (bytes + 255) & ~(size_t)255; return (char*)base + o; };
    NmsWs w;
    w.nvalid = (int*)take(sizeof(int) * (size_t)(b > 0 ? b : 1));
    w.row_any = (unsigned long long*)take(sizeof(unsigned long long) * (size_t)(b > 0 ? b : 1) * W);
    w.chunk_valid = (int*)take(sizeof(int) * (size_t)(b > 0 ? b : 1) * SC_MAXCH);
    w.zero_bytes = off;                                 // nvalid + row_any + chunk_valid: zeroed by one memset per call, except on the chunk-sort path (its kernels do it)
    w.order = (int*)take(sizeof(int) * (size_t)b * n);
    w.sbox = (SBox*)take(sizeof(SBox) * (size_t)b * n);
    w.mask = (unsigned long long*)take(sizeof(unsigned long long) * (size_t)b * n * W);
    w.keys_in = (unsigned long long*)take(sizeof(unsigned long long) * (size_t)b * n);
    w.keys_out = (unsigned long long*)take(sizeof(unsigned long long) * (size_t)b * n);
    w.sort_tmp_bytes = (b > 0 && n > 0) ? sort_tmp_bytes(b, n) : 0;
    w.sort_tmp = take(w.sort_tmp_bytes ? w.sort_tmp_bytes : 8);
    w.grank = (int*)take(sizeof(int) * (size_t)b * n);
    w.blk_lo = (int*)take(sizeof(int) * (size_t)(b > 0 ? b : 1) * W);
    w.blk_hi = (int*)take(sizeof(int) * (size_t)(b > 0 ? b : 1) * W);
    const bool own_sort = n <= SC * SC_MAXCH;
    w.chunked = (unsigned long long*)take(own_sort ? sizeof(unsigned long long) * (size_t)(b > 0 ? b : 1) * ceil_div(n > 0 ? n : 1, SC) * SC : 8);
    w.sorted2 = (unsigned long long*)take(own_sort ? sizeof(unsigned long long) * (size_t)(b > 0 ? b : 1) * (n > 0 ? n : 1) : 8);
    w.near = (unsigned long long*)take(sizeof(unsigned long long) * (size_t)(b > 0 ? b : 1) * W * (SCAN_NEAR + 1) * 64);
    w.total = off;
    return w;
}

}  // namespace yolo

using namespace yolo;

extern "C" {

int yolo_decode(void* pred, const int64_t* s, const float* anchors, int b, int g, int nc, int is_pred, float* boxes,
                int n_total, int box_offset, void* stream) {
    if (!pred || !s || !boxes || b <= 0 || g <= 0 || nc < 1) return fail(YOLO_ERR_ARG, "decode: bad arguments");
    if (is_pred && !anchors) return fail(YOLO_ERR_ARG, "decode: anchors required");
    if (!is_pred && nc != 1) return fail(YOLO_ERR_ARG, "decode: targets must have last dim 6");
    if (box_offset < 0 || box_offset + 3 * g * g > n_total) return fail(YOLO_ERR_ARG, "decode: box range outside n_total");
    const long long cells = (long long)b * 3 * g * g;
    if (cells > 0x7fffffffLL) return fail(YOLO_ERR_UNSUPPORTED, "decode: too many cells");
    const int D = 5 + nc;
    const size_t lds = (size_t)64 * D * sizeof(float);
    if (lds > 64 * 1024) return fail(YOLO_ERR_UNSUPPORTED, "decode: %d classes exceed the staging tile", nc);
    hipLaunchKernelGGL(decode_kernel, dim3((unsigned)((cells + 63) / 64)), dim3(256), lds, (hipStream_t)stream, (float*)pred,
                       (long long)s[0], (long long)s[1], (long long)s[2], (long long)s[3], (long long)s[4], anchors, b, g, nc,
                       is_pred, boxes, n_total, box_offset, decode_magic(g));
    return check_launch("decode");
}

int yolo_decode3(void* const* preds3, const int64_t* strides15, const float* const* anchors3, const int* grids3, int b, int nc,
                 float* boxes, int n_total, void* stream) {
    return yolo_decode3_ex(preds3, strides15, anchors3, grids3, b, nc, 1, boxes, n_total, stream);
}

int yolo_decode3_ex(void* const* preds3, const int64_t* strides15, const float* const* anchors3, const int* grids3, int b, int nc,
                    int write_back, float* boxes, int n_total, void* stream) {
    if (!preds3 || !strides15 || !anchors3 || !grids3 || !boxes || b <= 0 || nc < 1) return fail(YOLO_ERR_ARG, "decode3: bad arguments");
    Decode3Args a;
    a.B = b; a.nc = nc; a.n_total = n_total; a.boxes = boxes; a.is_pred = write_back ? 1 : 2;
    long long blocks = 0;
    int off = 0;
    for (int k = 0; k < 3; ++k) {
        const int g = grids3[k];
        if (!preds3[k] || !anchors3[k] || g <= 0) return fail(YOLO_ERR_ARG, "decode3: bad scale %d", k);
        DecodeScale& d = a.sc[k];
        d.pred = (float*)preds3[k]; d.anchors = anchors3[k]; d.g = g; d.box_offset = off; d.first_block = blocks; d.mg = decode_magic(g);
        if ((long long)b * 3 * g * g > 0x7fffffffLL) return fail(YOLO_ERR_UNSUPPORTED, "decode3: too many cells");
        d.sb = strides15[5 * k]; d.sa = strides15[5 * k + 1]; d.sy = strides15[5 * k + 2]; d.sx = strides15[5 * k + 3]; d.sk = strides15[5 * k + 4];
        off += 3 * g * g;
        blocks += ((long long)b * 3 * g * g + 63) / 64;
    }
    if (off != n_total) return fail(YOLO_ERR_ARG, "decode3: n_total %d != sum of 3 g^2 = %d", n_total, off);
    if (blocks > 0x7fffffffLL) return fail(YOLO_ERR_UNSUPPORTED, "decode3: too many cells");
    const size_t lds = (size_t)64 * (5 + nc) * sizeof(float);
    if (lds > 64 * 1024) return fail(YOLO_ERR_UNSUPPORTED, "decode: %d classes exceed the staging tile", nc);
    hipLaunchKernelGGL(decode3_kernel, dim3((unsigned)blocks), dim3(256), lds, (hipStream_t)stream, a);
    return check_launch("decode3");
}

size_t yolo_sort_u64_workspace_bytes(int n) { return n > 0 ? (size_t)ceil_div(n, SC) * SC * sizeof(unsigned long long) : 8; }

int yolo_sort_u64(const uint64_t* keys, uint64_t* sorted, int n, void* workspace, size_t workspace_bytes, void* stream) {
    if (n <= 0) return YOLO_OK;
    if (!keys || !sorted || !workspace) return fail(YOLO_ERR_ARG, "sort_u64: null pointer");
    if (n > SC * SC_MAXCH) return fail(YOLO_ERR_UNSUPPORTED, "sort_u64: n = %d exceeds %d keys", n, SC * SC_MAXCH);
    if (workspace_bytes < yolo_sort_u64_workspace_bytes(n)) return fail(YOLO_ERR_WORKSPACE, "sort_u64: workspace too small");
    hipStream_t st = (hipStream_t)stream;
    const int nch = ceil_div(n, SC);
    unsigned long long* chunked = (unsigned long long*)workspace;
    hipLaunchKernelGGL(sort_chunks_kernel, dim3(nch), dim3(SC_THREADS), 0, st, (const unsigned long long*)keys, n, chunked);
    if (int rc = check_launch("sort_chunks")) return rc;
    const dim3 gb(nch * 8);
    if (nch <= 3) hipLaunchKernelGGL(sort_merge_kernel<2>, gb, dim3(256), 0, st, chunked, nch, (unsigned long long*)sorted);
    else if (nch <= 9) hipLaunchKernelGGL(sort_merge_kernel<8>, gb, dim3(256), 0, st, chunked, nch, (unsigned long long*)sorted);
    else hipLaunchKernelGGL(sort_merge_kernel<SC_GROUP>, gb, dim3(256), 0, st, chunked, nch, (unsigned long long*)sorted);
    return check_launch("sort_merge");
}

size_t yolo_nms_workspace_bytes(int b, int n) {
    if (b <= 0 || n < 0) return 0;
    return carve(nullptr, b, n).total;
}

int yolo_nms(const float* boxes, int b, int n, double iou_threshold, double obj_threshold, int center, int32_t* keep_idx,
             int32_t* keep_count, void* workspace, size_t workspace_bytes, void* stream) {
    if (b <= 0 || n < 0 || !keep_count) return fail(YOLO_ERR_ARG, "nms: bad arguments");
    hipStream_t st = (hipStream_t)stream;
    if (n == 0) {
        if (hipMemsetAsync(keep_count, 0, sizeof(int) * (size_t)b, st) != hipSuccess) return fail(YOLO_ERR_LAUNCH, "nms: memset");
        return YOLO_OK;
    }
    if (!boxes || !keep_idx || !workspace) return fail(YOLO_ERR_ARG, "nms: null pointer");
    NmsWs w = carve(workspace, b, n);
    if (workspace_bytes < w.total) return fail(YOLO_ERR_WORKSPACE, "nms: workspace %zu < %zu bytes", workspace_bytes, w.total);
    const int W = ceil_div(n, 64);
    if ((size_t)(W + 2) * 8 > 60 * 1024) return fail(YOLO_ERR_UNSUPPORTED, "nms: n = %d too large", n);
    if (b > 65535 || W > 65535) return fail(YOLO_ERR_UNSUPPORTED, "nms: grid too large");
    int rc;
    const bool sorted_keys = n >= 2048 && n < (1 << 20) && b <= 2047;      // large n: sort the keys instead of counting
    const size_t fixed = (size_t)W * 16 + 128;              // class-range scan: kept words, lo/hi, count, cuts ...
    int K = (int)((60 * 1024 - (long long)fixed) / (long long)((size_t)W * 8));   // ... and one removed[] array per wave, in LDS
    const bool own_order = sorted_keys && K >= 1 && n <= SC * SC_MAXCH && !g_nms_rocprim;   // chunk sort + rank merge: they zero what they need
    if (!own_order && hipMemsetAsync(w.nvalid, 0, w.zero_bytes, st) != hipSuccess) return fail(YOLO_ERR_LAUNCH, "nms: memset");
    K = K > 16 ? 16 : K;
    const dim3 gn(ceil_div(n, 256), b);
    if (sorted_keys && K < 1) {                             // n > ~245,000: score order only, the plain mask and scan
        hipLaunchKernelGGL(nms_keys_kernel, gn, dim3(256), 0, st, boxes, n, obj_threshold, w.keys_in);
        rc = check_launch("nms_keys");
        if (rc) return rc;
        size_t tb = w.sort_tmp_bytes;
        if (rocprim::radix_sort_keys(w.sort_tmp, tb, w.keys_in, w.keys_out, (size_t)b * n, 0, 64, st) != hipSuccess)
            return fail(YOLO_ERR_LAUNCH, "nms: radix sort");
        hipLaunchKernelGGL(nms_gather_kernel, gn, dim3(256), 0, st, boxes, w.keys_out, w.nvalid, n, center, w.order, w.sbox);
        rc = check_launch("nms_gather");
        if (rc) return rc;
    } else if (sorted_keys) {                               // + class-sorted rows
        int* slot = (int*)w.keys_in;                        // kept boxes by global rank, -1 elsewhere (library-sort path only)
        const unsigned long long* order1 = nullptr;         // order 1 when it survives to the scan (chunk-sort path)
        if (n <= SC * SC_MAXCH && !g_nms_rocprim) {          // both orders by the chunk sort + rank merge kernels (4 launches)
            const int nch = ceil_div(n, SC);
            const dim3 ga(nch, b), gb(nch * 8, b);
            hipLaunchKernelGGL(nms_chunksort_kernel<1>, ga, dim3(SC_THREADS), 0, st, boxes, (const unsigned long long*)nullptr, (const int*)nullptr, n,
                               obj_threshold, w.chunked, w.chunk_valid);
            if (nch <= 3) hipLaunchKernelGGL(nms_merge1_kernel<2>, gb, dim3(256), 0, st, w.chunked, w.chunk_valid, nch, n, w.keys_out, w.nvalid, w.row_any, W);
            else if (nch <= 5) hipLaunchKernelGGL(nms_merge1_kernel<4>, gb, dim3(256), 0, st, w.chunked, w.chunk_valid, nch, n, w.keys_out, w.nvalid, w.row_any, W);
            else if (nch <= 9) hipLaunchKernelGGL(nms_merge1_kernel<8>, gb, dim3(256), 0, st, w.chunked, w.chunk_valid, nch, n, w.keys_out, w.nvalid, w.row_any, W);
            else hipLaunchKernelGGL(nms_merge1_kernel<SC_GROUP>, gb, dim3(256), 0, st, w.chunked, w.chunk_valid, nch, n, w.keys_out, w.nvalid, w.row_any, W);
            rc = check_launch("nms order 1");
            if (rc) return rc;
            hipLaunchKernelGGL(nms_chunksort_kernel<2>, ga, dim3(SC_THREADS), 0, st, boxes, w.keys_out, w.nvalid, n, obj_threshold, w.chunked,
                               w.chunk_valid);
            // order 2 lands in a second array: order 1 (keys_out) is still being read by the chunk sort above
            order1 = w.keys_out;
            unsigned long long* sorted2 = w.sorted2;
            if (nch <= 3) hipLaunchKernelGGL(nms_merge2_kernel<2>, gb, dim3(256), 0, st, boxes, w.chunked, w.nvalid, nch, n, W, center, sorted2, w.order, w.grank, w.sbox, w.blk_lo, w.blk_hi);
            else if (nch <= 5) hipLaunchKernelGGL(nms_merge2_kernel<4>, gb, dim3(256), 0, st, boxes, w.chunked, w.nvalid, nch, n, W, center, sorted2, w.order, w.grank, w.sbox, w.blk_lo, w.blk_hi);
            else if (nch <= 9) hipLaunchKernelGGL(nms_merge2_kernel<8>, gb, dim3(256), 0, st, boxes, w.chunked, w.nvalid, nch, n, W, center, sorted2, w.order, w.grank, w.sbox, w.blk_lo, w.blk_hi);
            else hipLaunchKernelGGL(nms_merge2_kernel<SC_GROUP>, gb, dim3(256), 0, st, boxes, w.chunked, w.nvalid, nch, n, W, center, sorted2, w.order, w.grank, w.sbox, w.blk_lo, w.blk_hi);
            rc = check_launch("nms order 2");
            if (rc) return rc;
            w.keys_out = sorted2;
        } else {
        hipLaunchKernelGGL(nms_keys_kernel, gn, dim3(256), 0, st, boxes, n, obj_threshold, w.keys_in);
        rc = check_launch("nms_keys");
        if (rc) return rc;
        size_t tb = w.sort_tmp_bytes;
        if (rocprim::radix_sort_keys(w.sort_tmp, tb, w.keys_in, w.keys_out, (size_t)b * n, 0, 64, st) != hipSuccess)
            return fail(YOLO_ERR_LAUNCH, "nms: radix sort");
        hipLaunchKernelGGL(nms_keys2_kernel, gn, dim3(256), 0, st, boxes, w.keys_out, w.nvalid, n, w.keys_in);
        rc = check_launch("nms_keys2");
        if (rc) return rc;
        tb = w.sort_tmp_bytes;
        if (rocprim::radix_sort_keys(w.sort_tmp, tb, w.keys_in, w.keys_out, (size_t)b * n, 40, 63, st) != hipSuccess)
            return fail(YOLO_ERR_LAUNCH, "nms: radix sort (class)");
        if (hipMemsetAsync(slot, 0xff, sizeof(int) * (size_t)b * n, st) != hipSuccess) return fail(YOLO_ERR_LAUNCH, "nms: memset");
        hipLaunchKernelGGL(nms_gather2_kernel, gn, dim3(256), 0, st, boxes, w.keys_out, w.nvalid, n, W, center, w.order, w.grank,
                           w.sbox, w.blk_lo, w.blk_hi);
        rc = check_launch("nms_gather2");
        if (rc) return rc;
        }
        // 4 blocks per row block: with many classes only the first 2-3 column blocks are in range and every further (empty)
        // block costs launch time (80 classes: 0.212 / 0.221 / 0.250 / 0.305 ms for 3 / 4 / 8 / 16), with 2 classes more
        // blocks help a little (1.16 / 1.12 / 1.05 / 1.02 ms)
        hipLaunchKernelGGL(nms_mask_sorted_kernel, dim3(1, b, W), dim3(64 * MS_WAVES), 0, st, w.sbox, w.nvalid, w.blk_lo, w.blk_hi, n,
                           W, (float)iou_threshold, w.mask, w.row_any, w.near);
        rc = check_launch("nms_mask_sorted");
        if (rc) return rc;
        const size_t lds = (size_t)K * W * 8 + fixed;
        hipLaunchKernelGGL(nms_scan_classes_kernel, dim3(b), dim3(64 * K), lds, st, w.mask, w.keys_out, w.nvalid, w.row_any, w.blk_lo,
                           w.blk_hi, n, W, w.order, w.grank, slot, keep_count, w.near,
                           own_order ? (const unsigned long long*)order1 : (const unsigned long long*)nullptr, keep_idx);
        rc = check_launch("nms_scan_classes");
        if (rc) return rc;
        if (own_order) return YOLO_OK;                  // the scan wrote keep_idx itself
        hipLaunchKernelGGL(nms_compact_kernel, dim3(b), dim3(1024), 0, st, slot, w.nvalid, n, keep_idx);
        return check_launch("nms_compact");
    } else {
        hipLaunchKernelGGL(nms_rank_kernel, gn, dim3(256), 0, st, boxes, n, obj_threshold, center, w.order, w.sbox, w.nvalid);
        rc = check_launch("nms_rank");
        if (rc) return rc;
    }
    hipLaunchKernelGGL(nms_mask_kernel, dim3(W, W, b), dim3(64), 0, st, w.sbox, w.nvalid, n, W, (float)iou_threshold, w.mask, w.row_any);
    rc = check_launch("nms_mask");
    if (rc) return rc;
    hipLaunchKernelGGL(nms_scan_kernel, dim3(b), dim3(256), (size_t)(W + 2) * 8, st, w.mask, w.order, w.nvalid, w.row_any, n, W,
                       keep_idx, keep_count);
    return check_launch("nms_scan");
}

}  // extern "C"
