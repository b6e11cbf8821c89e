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

namespace yolo {

// ------------------------------------------------------------------------------ decode
__device__ __forceinline__ float sigmoid_f(float x) { return 1.f / (1.f + expf(-x)); }

// One thread per cell (b, a, row, col); col is the fastest thread index so that the reference's
// permuted view (element stride 1 along col) is read coalesced. For this library's contiguous
// head layout (stride 1 along k) the 64 cells of a wave are first staged through LDS so global
// reads are full lines.
__device__ __forceinline__ void decode_block(float* __restrict__ pred, long long sb, long long sa, long long sy, long long sx, long long sk,
                                             const float* __restrict__ anchors, int B, int g, int nc, int is_pred,
                                             float* __restrict__ boxes, int n_total, int box_offset, long long blk) {
    extern __shared__ __attribute__((aligned(16))) float tile[];           // [64][D] when sk == 1 && cells contiguous, else unused
    const int D = 5 + nc;
    const long long cells = (long long)B * 3 * g * g;
    const long long cell0 = blk * 64;
    const long long cell = cell0 + threadIdx.x;
    const bool contiguous = (sk == 1 && sx == D && sy == (long long)g * D && sa == (long long)g * g * D &&
                             sb == 3LL * g * g * D);
    const float* src;
    long long kstride;
    if (contiguous) {
        const long long first = cell0 * D;
        const long long count = (cells - cell0 < 64 ? cells - cell0 : 64) * D;
        // 16-byte loads when the tile is whole (64 * D floats is a multiple of 4 and the base is 16-byte aligned):
        // 4-byte loads made this pass latency-bound at 1.1 TB/s
        const float* gsrc = pred + first;
        if (count == 64LL * D && ((reinterpret_cast<size_t>(gsrc) & 15) == 0)) {
            typedef float f4 __attribute__((ext_vector_type(4)));
            const int n4 = (int)(count >> 2);
#pragma unroll 4
            for (int i = threadIdx.x; i < n4; i += 64) reinterpret_cast<f4*>(tile)[i] = reinterpret_cast<const f4*>(gsrc)[i];
        } else {
            for (long long i = threadIdx.x; i < count; i += 64) tile[i] = gsrc[i];
        }
        __syncthreads();
        src = tile + (long long)threadIdx.x * D;
        kstride = 1;
    } else {
        src = nullptr;
        kstride = sk;
    }
    if (cell >= cells) return;
    const int col = (int)(cell % g);
    const int row = (int)((cell / g) % g);
    const int a = (int)((cell / ((long long)g * g)) % 3);
    const int b = (int)(cell / (3LL * g * g));
    float* gp = pred + b * sb + a * sa + row * sy + col * sx;
    if (!contiguous) src = gp;
    const float inv = (float)(1.0 / (double)g);          // `1 / grid_size` is a Python float, cast to fp32 by the multiply
    float p0 = src[0], p1 = src[kstride], p2 = src[2 * kstride], p3 = src[3 * kstride], p4 = src[4 * kstride];
    float cls;
    if (is_pred) {
        p0 = sigmoid_f(p0);
        p1 = sigmoid_f(p1);
        p2 = expf(p2) * anchors[2 * a];
        p3 = expf(p3) * anchors[2 * a + 1];
        p4 = sigmoid_f(p4);
        int best = 0;
        float bv = src[5 * kstride];
        for (int k = 1; k < nc; ++k) {                   // first maximum; NaN counts as maximum (torch.argmax)
            const float v = src[(5 + k) * kstride];
            if (v > bv || (v != v && bv == bv)) { bv = v; best = k; }
        }
        cls = (float)best;
        gp[0] = p0; gp[sk] = p1; gp[2 * sk] = p2; gp[3 * sk] = p3;   // in-place side effect (utils.py:106-110)
    } else {
        cls = src[5 * kstride];
    }
    float* o = boxes + ((size_t)b * n_total + box_offset + (size_t)a * g * g + (size_t)row * g + col) * 6;
    o[0] = inv * (p0 + (float)col);
    o[1] = inv * (p1 + (float)row);
    o[2] = inv * p2;
    o[3] = inv * p3;
    o[4] = p4;
    o[5] = cls;
}

__global__ void decode_kernel(float* __restrict__ pred, long long sb, long long sa, long long sy, long long sx, long long sk,
                              const float* __restrict__ anchors, int B, int g, int nc, int is_pred,
                              float* __restrict__ boxes, int n_total, int box_offset) {
    decode_block(pred, sb, sa, sy, sx, sk, anchors, B, g, nc, is_pred, boxes, n_total, box_offset, blockIdx.x);
}

// the three scales of one forward in ONE launch (demo.py:44-51 / utils.py:300-309 order): at batch 32 the three separate
// launches were launch-latency-bound (3 x ~25 us for 129 MB)
struct DecodeScale { float* pred; long long sb, sa, sy, sx, sk; const float* anchors; int g, box_offset; long long first_block; };
struct Decode3Args { DecodeScale sc[3]; int B, nc, n_total; float* boxes; };

__global__ void decode3_kernel(const Decode3Args a) {
    const int k = (long long)blockIdx.x >= a.sc[2].first_block ? 2 : ((long long)blockIdx.x >= a.sc[1].first_block ? 1 : 0);
    const DecodeScale& d = a.sc[k];
    decode_block(d.pred, d.sb, d.sa, d.sy, d.sx, d.sk, d.anchors, a.B, d.g, a.nc, 1, a.boxes, a.n_total, d.box_offset,
                 (long long)blockIdx.x - d.first_block);
}

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
__global__ __launch_bounds__(256) void nms_keys_kernel(const float* __restrict__ boxes, int n, double obj_thr,
                                                       unsigned long long* __restrict__ keys, int* __restrict__ nvalid) {
    const int b = blockIdx.y;
    const int i = blockIdx.x * 256 + threadIdx.x;
    bool valid = false;
    if (i < n) {
        const unsigned long long k = make_key(boxes[((size_t)b * n + i) * 6 + 4], i, obj_thr);
        valid = k != ~0ull;
        // image (12 bits) | descending-score field (32 bits; all ones = filtered) | index (20 bits): ONE radix sort of the
        // whole batch leaves image b's boxes in [b n, (b + 1) n), valid ones first, in the reference's stable order
        const unsigned long long sc = valid ? (k >> 32) : 0xffffffffull;
        keys[(size_t)b * n + i] = ((unsigned long long)b << 52) | (sc << 20) | (unsigned long long)i;
    }
    const unsigned long long bal = __ballot(valid);
    if ((threadIdx.x & 63) == 0 && bal) atomicAdd(nvalid + b, __popcll(bal));
}

__global__ __launch_bounds__(256) void nms_gather_kernel(const float* __restrict__ boxes, const unsigned long long* __restrict__ sorted,
                                                         const int* __restrict__ nvalid, int n, int center, int* __restrict__ order,
                                                         SBox* __restrict__ sbox) {
    const int b = blockIdx.y;
    const int r = blockIdx.x * 256 + threadIdx.x;
    if (r >= nvalid[b]) return;
    const int i = (int)(unsigned)(sorted[(size_t)b * n + r] & 0xfffffull);
    const float* s = boxes + ((size_t)b * n + i) * 6;
    float x = s[0], y = s[1];
    const float w = s[2], h = s[3];
    if (center) { x = x - w / 2.0f; y = y - h / 2.0f; }            // utils.py:60-64
    SBox o;
    o.x1 = x; o.y1 = y; o.x2 = x + w; o.y2 = y + h; o.area = w * h; o.cls = s[5]; o.w = w; o.h = h;
    sbox[(size_t)b * n + r] = o;
    order[(size_t)b * n + r] = i;
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
    if (i >= nv) return;
    const SBox me = sb[i];
    const int lim = nv - cb * 64 < 64 ? nv - cb * 64 : 64;
    unsigned long long word = 0;
    for (int jj = 0; jj < lim; ++jj) {
        const SBox o = cols[jj];
        // a box of another class never suppresses (utils.py:183-186: kept if `cls != top.cls` OR iou < thr): skip the IoU
        // when no row of this wave shares column jj's class — with 80 classes that is ~45 % of the columns
        if (__ballot(o.cls == me.cls) == 0ull) continue;
        const float xa = max_nan(me.x1, o.x1), ya = max_nan(me.y1, o.y1);
        const float xb = min_nan(me.x2, o.x2), yb = min_nan(me.y2, o.y2);
        float iw = xb - xa, ih = yb - ya;
        iw = iw < 0.f ? 0.f : iw;                       // torch.clamp(min=0): NaN stays NaN
        ih = ih < 0.f ? 0.f : ih;
        const float inter = iw * ih;
        const float uni = (me.area + o.area) - inter;
        const float iou = inter / (uni + 1e-6f);
        const bool survive = (o.cls != me.cls) || (iou < thr);
        if (!survive && cb * 64 + jj > i) word |= 1ull << jj;
    }
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
                const unsigned dlo = (unsigned)d, dhi = (unsigned)(d >> 32);
                kept = 0;
                for (int t = 0; t < rows; ++t) {
                    if (!((rem >> t) & 1ull)) {
                        kept |= 1ull << t;
                        const unsigned lo = __builtin_amdgcn_readlane(dlo, t);
                        const unsigned hi = __builtin_amdgcn_readlane(dhi, t);
                        rem |= ((unsigned long long)hi << 32) | lo;
                    }
                }
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

struct NmsWs { int* nvalid; unsigned long long* row_any; int* order; SBox* sbox; unsigned long long* mask; size_t zero_bytes; size_t total;
               unsigned long long* keys_in; unsigned long long* keys_out; void* sort_tmp; size_t sort_tmp_bytes; };

static size_t sort_tmp_bytes(int b, int n) {
    size_t bytes = 0;
    (void)rocprim::radix_sort_keys(nullptr, bytes, (unsigned long long*)nullptr, (unsigned long long*)nullptr, (size_t)b * n, 0, 64,
                                   (hipStream_t)0);
    return bytes;
}

static NmsWs carve(void* base, int b, int n) {
    const int W = ceil_div(n > 0 ? n : 1, 64);
    size_t off = 0;
    auto take = [&](size_t bytes) { size_t o = off; off += (bytes + 255) & ~(size_t)255; return (char*)base + o; };
    NmsWs w;
    w.nvalid = (int*)take(sizeof(int) * (size_t)(b > 0 ? b : 1));
    w.row_any = (unsigned long long*)take(sizeof(unsigned long long) * (size_t)(b > 0 ? b : 1) * W);
    w.zero_bytes = off;                                 // nvalid + row_any are zeroed by one memset per call
    w.order = (int*)take(sizeof(int) * (size_t)b * n);
    w.sbox = (SBox*)take(sizeof(SBox) * (size_t)b * n);
    w.mask = (unsigned long long*)take(sizeof(unsigned long long) * (size_t)b * n * W);
    w.keys_in = (unsigned long long*)take(sizeof(unsigned long long) * (size_t)b * n);
    w.keys_out = (unsigned long long*)take(sizeof(unsigned long long) * (size_t)b * n);
    w.sort_tmp_bytes = (b > 0 && n > 0) ? sort_tmp_bytes(b, n) : 0;
    w.sort_tmp = take(w.sort_tmp_bytes ? w.sort_tmp_bytes : 8);
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
    const int D = 5 + nc;
    const size_t lds = (size_t)64 * D * sizeof(float);
    if (lds > 64 * 1024) return fail(YOLO_ERR_UNSUPPORTED, "decode: %d classes exceed the staging tile", nc);
    hipLaunchKernelGGL(decode_kernel, dim3((unsigned)((cells + 63) / 64)), dim3(64), lds, (hipStream_t)stream, (float*)pred,
                       (long long)s[0], (long long)s[1], (long long)s[2], (long long)s[3], (long long)s[4], anchors, b, g, nc,
                       is_pred, boxes, n_total, box_offset);
    return check_launch("decode");
}

int yolo_decode3(void* const* preds3, const int64_t* strides15, const float* const* anchors3, const int* grids3, int b, int nc,
                 float* boxes, int n_total, void* stream) {
    if (!preds3 || !strides15 || !anchors3 || !grids3 || !boxes || b <= 0 || nc < 1) return fail(YOLO_ERR_ARG, "decode3: bad arguments");
    Decode3Args a;
    a.B = b; a.nc = nc; a.n_total = n_total; a.boxes = boxes;
    long long blocks = 0;
    int off = 0;
    for (int k = 0; k < 3; ++k) {
        const int g = grids3[k];
        if (!preds3[k] || !anchors3[k] || g <= 0) return fail(YOLO_ERR_ARG, "decode3: bad scale %d", k);
        DecodeScale& d = a.sc[k];
        d.pred = (float*)preds3[k]; d.anchors = anchors3[k]; d.g = g; d.box_offset = off; d.first_block = blocks;
        d.sb = strides15[5 * k]; d.sa = strides15[5 * k + 1]; d.sy = strides15[5 * k + 2]; d.sx = strides15[5 * k + 3]; d.sk = strides15[5 * k + 4];
        off += 3 * g * g;
        blocks += ((long long)b * 3 * g * g + 63) / 64;
    }
    if (off != n_total) return fail(YOLO_ERR_ARG, "decode3: n_total %d != sum of 3 g^2 = %d", n_total, off);
    if (blocks > 0x7fffffffLL) return fail(YOLO_ERR_UNSUPPORTED, "decode3: too many cells");
    const size_t lds = (size_t)64 * (5 + nc) * sizeof(float);
    if (lds > 64 * 1024) return fail(YOLO_ERR_UNSUPPORTED, "decode: %d classes exceed the staging tile", nc);
    hipLaunchKernelGGL(decode3_kernel, dim3((unsigned)blocks), dim3(64), lds, (hipStream_t)stream, a);
    return check_launch("decode3");
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
    if (hipMemsetAsync(w.nvalid, 0, w.zero_bytes, st) != hipSuccess) return fail(YOLO_ERR_LAUNCH, "nms: memset");
    int rc;
    if (n >= 2048 && n < (1 << 20) && b <= 4096) {        // large n: sort the keys instead of counting (same order, see above)
        hipLaunchKernelGGL(nms_keys_kernel, dim3(ceil_div(n, 256), b), dim3(256), 0, st, boxes, n, obj_threshold, w.keys_in, w.nvalid);
        rc = check_launch("nms_keys");
        if (rc) return rc;
        size_t tb = w.sort_tmp_bytes;
        if (rocprim::radix_sort_keys(w.sort_tmp, tb, w.keys_in, w.keys_out, (size_t)b * n, 0, 64, st) != hipSuccess)
            return fail(YOLO_ERR_LAUNCH, "nms: radix sort");
        hipLaunchKernelGGL(nms_gather_kernel, dim3(ceil_div(n, 256), b), dim3(256), 0, st, boxes, w.keys_out, w.nvalid, n, center, w.order,
                           w.sbox);
        rc = check_launch("nms_gather");
    } else {
        hipLaunchKernelGGL(nms_rank_kernel, dim3(ceil_div(n, 256), b), dim3(256), 0, st, boxes, n, obj_threshold, center, w.order,
                           w.sbox, w.nvalid);
        rc = check_launch("nms_rank");
    }
    if (rc) return rc;
    hipLaunchKernelGGL(nms_mask_kernel, dim3(W, W, b), dim3(64), 0, st, w.sbox, w.nvalid, n, W, (float)iou_threshold, w.mask, w.row_any);
    rc = check_launch("nms_mask");
    if (rc) return rc;
    hipLaunchKernelGGL(nms_scan_kernel, dim3(b), dim3(256), (size_t)(W + 2) * 8, st, w.mask, w.order, w.nvalid, w.row_any, n, W,
                       keep_idx, keep_count);
    return check_launch("nms_scan");
}

}  // extern "C"
