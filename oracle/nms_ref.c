/* ORACLE — TEST INFRASTRUCTURE ONLY (never linked into the product library).
 *
 * Plain-C restatement of the reference post-processing arithmetic, fp32 with one rounding
 * per operation (build with -ffp-contract=off):
 *   nms_ref     /root/reference/code/utils.py:150-191 (non_max_suppression) with the IoU of
 *               utils.py:38-84 (calc_iou) inlined.
 *   decode_ref  /root/reference/code/utils.py:86-148 (cells_to_boxes, is_pred=True) on a
 *               contiguous (B,3,g,g,5+nc) tensor.
 * Pinned by tests/golden/nms_*.npz and decode_*.npz (generated from the imported reference).
 */
#include <math.h>
#include <stdlib.h>
#include <string.h>

typedef struct { float score; int idx; } cand_t;

/* stable descending order: ties keep input order (Python sorted(..., reverse=True)) */
static int cmp_cand(const void* a, const void* b) {
    const cand_t* x = (const cand_t*)a;
    const cand_t* y = (const cand_t*)b;
    if (x->score > y->score) return -1;
    if (x->score < y->score) return 1;
    return (x->idx > y->idx) - (x->idx < y->idx);
}

static float iou_f32(const float* a, const float* b, int center) {
    float ax = a[0], ay = a[1], aw = a[2], ah = a[3];
    float bx = b[0], by = b[1], bw = b[2], bh = b[3];
    if (center) {
        ax = ax - aw / 2.0f; ay = ay - ah / 2.0f;
        bx = bx - bw / 2.0f; by = by - bh / 2.0f;
    }
    float xa = fmaxf(ax, bx), ya = fmaxf(ay, by);
    float xb = fminf(ax + aw, bx + bw), yb = fminf(ay + ah, by + bh);
    float iw = xb - xa; if (iw < 0.0f) iw = 0.0f;
    float ih = yb - ya; if (ih < 0.0f) ih = 0.0f;
    float inter = iw * ih;
    float uni = (aw * ah + bw * bh) - inter;
    return inter / (uni + 1e-6f);
}

/* boxes: n x 6 float32 [x,y,w,h,obj,cls]. Returns K, keep[0..K) = indices into boxes. */
int nms_ref(const float* boxes, int n, double iou_thr, double obj_thr, int center, int* keep) {
    cand_t* c = (cand_t*)malloc(sizeof(cand_t) * (size_t)(n > 0 ? n : 1));
    int m = 0;
    for (int i = 0; i < n; ++i)
        if ((double)boxes[6 * i + 4] > obj_thr) { c[m].score = boxes[6 * i + 4]; c[m].idx = i; ++m; }
    qsort(c, (size_t)m, sizeof(cand_t), cmp_cand);
    char* dead = (char*)calloc((size_t)(m > 0 ? m : 1), 1);
    const float thr = (float)iou_thr;          /* tensor < python-float compares in fp32 */
    int k = 0;
    for (int i = 0; i < m; ++i) {
        if (dead[i]) continue;
        const float* bi = boxes + 6 * (size_t)c[i].idx;
        keep[k++] = c[i].idx;
        for (int j = i + 1; j < m; ++j) {
            if (dead[j]) continue;
            const float* bj = boxes + 6 * (size_t)c[j].idx;
            int survive = (bj[5] != bi[5]) || (iou_f32(bi, bj, center) < thr);
            if (!survive) dead[j] = 1;
        }
    }
    free(c); free(dead);
    return k;
}

static float sigmoidf_(float x) { return 1.0f / (1.0f + expf(-x)); }

void decode_ref(const float* pred, const float* anchors, int B, int g, int nc, float* out) {
    const int d = 5 + nc;
    const float inv = (float)(1.0 / (double)g);
    for (int b = 0; b < B; ++b)
        for (int a = 0; a < 3; ++a)
            for (int r = 0; r < g; ++r)
                for (int col = 0; col < g; ++col) {
                    size_t cell = (((size_t)b * 3 + a) * g + r) * g + col;
                    const float* p = pred + cell * d;
                    float* o = out + cell * 6;
                    float sx = sigmoidf_(p[0]), sy = sigmoidf_(p[1]);
                    float w = expf(p[2]) * anchors[2 * a], h = expf(p[3]) * anchors[2 * a + 1];
                    int best = 0; float bv = p[5];
                    for (int k = 1; k < nc; ++k) {
                        float v = p[5 + k];
                        if (v > bv || (v != v && bv == bv)) { bv = v; best = k; }
                    }
                    o[0] = inv * (sx + (float)col);
                    o[1] = inv * (sy + (float)r);
                    o[2] = inv * w; o[3] = inv * h;
                    o[4] = sigmoidf_(p[4]);
                    o[5] = (float)best;
                }
}
