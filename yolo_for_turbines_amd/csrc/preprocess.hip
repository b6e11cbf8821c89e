// preprocess.hip — letterbox of one uint8 HWC image into the network's normalised CHW float input
// (the step in front of the forward in demo.predict: config.py:101-113 = albumentations LongestMaxSize -> centred
// PadIfNeeded(value 0) -> Normalize(mean 0, std 1, max 255) -> ToTensorV2; reference: code/config.py:84-113,
// code/demo.py:37-39).
//
// PARITY UNPINNED: cv2 / albumentations are not installed where this was built, so the reference transform could
// not be run to produce golden vectors. The kernel follows OpenCV's published uint8 INTER_LINEAR algorithm
// (resize.cpp: source coordinate (d + 0.5) * scale - 0.5, 11-bit fixed-point coefficients rounded to nearest even,
// horizontal pass in int32, vertical pass ((b0 * (r0 >> 4)) >> 16) + ((b1 * (r1 >> 4)) >> 16) + 2 >> 2) and
// albumentations' size / padding rules (banker's rounding of dim * scale, top/left pad = floor(diff / 2)); it is tested
// bit for bit against oracle/preprocess.py, which restates the same published rules — not against cv2 itself.
#include "common.h"

namespace yolo {

__device__ __forceinline__ int cv_round(double v) { return (int)rint(v); }          // round half to even, like cvRound

__device__ __forceinline__ void lin_coef(int d, double scale, int src_n, int* s0, int* s1, int* c0, int* c1) {
    double f = (d + 0.5) * scale - 0.5;
    int s = (int)floor(f);
    f -= s;
    if (s < 0) { s = 0; f = 0; }
    if (s >= src_n - 1) { s = src_n - 1; f = 0; }
    *s0 = s;
    *s1 = s + 1 < src_n ? s + 1 : s;
    const float ff = (float)f;
    int a1 = cv_round((double)(ff * 2048.f));
    int a0 = cv_round((double)((1.f - ff) * 2048.f));
    a0 = a0 > 32767 ? 32767 : a0;
    a1 = a1 > 32767 ? 32767 : a1;
    *c0 = a0; *c1 = a1;
}

// out: (3, S, S) fp32; one thread per output pixel
__global__ __launch_bounds__(256) void letterbox_kernel(const unsigned char* __restrict__ img, int h, int w, int nh, int nw,
                                                        int pad_top, int pad_left, int S, float* __restrict__ out) {
    const int idx = blockIdx.x * 256 + threadIdx.x;
    if (idx >= S * S) return;
    const int oy = idx / S, ox = idx - oy * S;
    const int y = oy - pad_top, x = ox - pad_left;
    float v[3] = {0.f, 0.f, 0.f};
    if ((unsigned)y < (unsigned)nh && (unsigned)x < (unsigned)nw) {
        unsigned char px[3];
        if (nh == h && nw == w) {
            for (int c = 0; c < 3; ++c) px[c] = img[((size_t)y * w + x) * 3 + c];
        } else {
            const double sx = (double)w / nw, sy = (double)h / nh;
            int x0, x1, a0, a1, y0, y1, b0, b1;
            lin_coef(x, sx, w, &x0, &x1, &a0, &a1);
            lin_coef(y, sy, h, &y0, &y1, &b0, &b1);
            for (int c = 0; c < 3; ++c) {
                const int r0 = img[((size_t)y0 * w + x0) * 3 + c] * a0 + img[((size_t)y0 * w + x1) * 3 + c] * a1;
                const int r1 = img[((size_t)y1 * w + x0) * 3 + c] * a0 + img[((size_t)y1 * w + x1) * 3 + c] * a1;
                const int t = (((b0 * (r0 >> 4)) >> 16) + ((b1 * (r1 >> 4)) >> 16) + 2) >> 2;
                px[c] = (unsigned char)(t < 0 ? 0 : (t > 255 ? 255 : t));
            }
        }
        const float inv = 1.0f / 255.0f;                     // albumentations multiplies by the fp32 reciprocal
        for (int c = 0; c < 3; ++c) v[c] = (float)px[c] * inv;
    }
    for (int c = 0; c < 3; ++c) out[(size_t)c * S * S + idx] = v[c];
}

}  // namespace yolo

using namespace yolo;

extern "C" {

/* new_hw[2] (host, out): resized size before padding; pad_tl[2] (host, out): top / left padding */
int yolo_letterbox(const unsigned char* img_hwc, int h, int w, int size, float* out_chw, int* new_hw, int* pad_tl, void* stream) {
    if (!img_hwc || !out_chw || h <= 0 || w <= 0 || size <= 0) return fail(YOLO_ERR_ARG, "letterbox: bad arguments");
    const double scale = (double)size / (double)(h > w ? h : w);
    auto py3round = [](double v) {                         // albumentations.py3round: banker's rounding
        const double r = nearbyint(v);                      // FE_TONEAREST: half to even
        return (int)r;
    };
    int nh = h, nw = w;
    if (scale != 1.0) { nh = py3round(h * scale); nw = py3round(w * scale); }
    if (nh < 1) nh = 1;
    if (nw < 1) nw = 1;
    if (nh > size || nw > size) return fail(YOLO_ERR_ARG, "letterbox: resized image exceeds the target");
    const int pad_top = (size - nh) / 2, pad_left = (size - nw) / 2;
    if (new_hw) { new_hw[0] = nh; new_hw[1] = nw; }
    if (pad_tl) { pad_tl[0] = pad_top; pad_tl[1] = pad_left; }
    hipLaunchKernelGGL(letterbox_kernel, dim3(ceil_div(size * size, 256)), dim3(256), 0, (hipStream_t)stream, img_hwc, h, w, nh, nw,
                       pad_top, pad_left, size, out_chw);
    return check_launch("letterbox");
}

}  // extern "C"
