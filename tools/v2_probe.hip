// Diagnostic build of the patch conv kernel with in-kernel s_memtime stamps (never shipped):
//   hipcc --offload-arch=gfx950 -O3 -std=c++17 tools/v2_probe.hip -o /tmp/v2_probe && /tmp/v2_probe 52 128 256 3 64
#define V2_STAMPS 1
#include "../yolo_for_turbines_amd/csrc/conv_f32_v2.hip"
#include <vector>
#include <algorithm>
#include <cstdlib>
namespace yolo {
static char g_err[512];
char* err_buf() { return g_err; }
int fail(int code, const char* fmt, ...) { va_list ap; va_start(ap, fmt); vsnprintf(g_err, sizeof(g_err), fmt, ap); va_end(ap); fprintf(stderr, "%s\n", g_err); return code; }
}
using namespace yolo;
__global__ __launch_bounds__(256) void calib(unsigned long long* out, float seed, int iters) {
    f32x16 acc[2];
    for (int a = 0; a < 2; ++a) for (int r = 0; r < 16; ++r) acc[a][r] = 0.f;
    float a0 = seed * threadIdx.x, b0 = seed + threadIdx.x;
    const unsigned long long t0 = __builtin_amdgcn_s_memtime();
    for (int i = 0; i < iters; ++i) {
        acc[0] = __builtin_amdgcn_mfma_f32_32x32x2f32(a0, b0, acc[0], 0, 0, 0);
        acc[1] = __builtin_amdgcn_mfma_f32_32x32x2f32(b0, a0, acc[1], 0, 0, 0);
    }
    asm volatile("s_nop 15\n\ts_nop 15" ::: "memory");
    const unsigned long long t1 = __builtin_amdgcn_s_memtime();
    float s = 0; for (int a = 0; a < 2; ++a) for (int r = 0; r < 16; ++r) s += acc[a][r];
    if (threadIdx.x == 0) { out[blockIdx.x * 2] = t1 - t0; out[blockIdx.x * 2 + 1] = (unsigned long long)s; }
}
int main(int argc, char** argv) {
    const int H = argc > 1 ? atoi(argv[1]) : 52, cin = argc > 2 ? atoi(argv[2]) : 128, cout = argc > 3 ? atoi(argv[3]) : 256;
    const int ks = argc > 4 ? atoi(argv[4]) : 3, bn = argc > 5 ? atoi(argv[5]) : 64, B = argc > 6 ? atoi(argv[6]) : 32;
    {
        unsigned long long* o; hipMalloc(&o, 256 * 2 * 8);
        hipLaunchKernelGGL(calib, dim3(256), dim3(256), 0, 0, o, 0.5f, 4096);
        hipDeviceSynchronize();
        unsigned long long ho[512]; hipMemcpy(ho, o, sizeof(ho), hipMemcpyDeviceToHost);
        printf("calibration: %llu memtime ticks for 8192 back-to-back MFMA 32x32x2 (expect %d cycles) -> %.3f ticks/cycle\n", ho[0], 8192 * 64, ho[0] / (8192.0 * 64));
    }
    const size_t nx = (size_t)B * H * H * cin, ny = (size_t)B * H * H * cout, nw = (size_t)cout * cin * ks * ks;
    std::vector<float> hx(nx), hw(nw);
    for (auto& v : hx) v = (rand() % 2001 - 1000) * 1e-3f;
    for (auto& v : hw) v = (rand() % 2001 - 1000) * 1e-4f;
    float *x, *w, *wf, *y, *sc, *sh; int* flag;
    hipMalloc(&x, nx * 4); hipMalloc(&w, nw * 4); hipMalloc(&y, ny * 4); hipMalloc(&sc, cout * 4); hipMalloc(&sh, cout * 4); hipMalloc(&flag, 4);
    hipMalloc(&wf, v2_frag_elems(cout, cin, ks) * 4);
    hipMemcpy(x, hx.data(), nx * 4, hipMemcpyHostToDevice); hipMemcpy(w, hw.data(), nw * 4, hipMemcpyHostToDevice);
    std::vector<float> ones(cout, 1.f); hipMemcpy(sc, ones.data(), cout * 4, hipMemcpyHostToDevice); hipMemset(sh, 0, cout * 4);
    v2_pack(w, wf, cout, cin, ks, 0);
    yolo_conv_desc d = {};
    d.n = B; d.h = H; d.w = H; d.cin = cin; d.cout = cout; d.ksize = ks; d.stride = 1; d.x_ld = cin; d.y_ld = cout; d.r_ld = cout;
    d.act = YOLO_ACT_LEAKY; d.out_mode = YOLO_OUT_NHWC; d.flags = YOLO_FLAG_NANCHECK;
    const int maxblocks = 1 << 20;
    hipMalloc(&g_v2_dbg, (size_t)maxblocks * 6 * 8); hipMemset(g_v2_dbg, 0, (size_t)maxblocks * 6 * 8);
    for (int i = 0; i < 3; ++i) conv_v2_launch(&d, x, wf, sc, sh, nullptr, y, flag, bn, 0);
    hipDeviceSynchronize();
    hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
    hipEventRecord(e0); conv_v2_launch(&d, x, wf, sc, sh, nullptr, y, flag, bn, 0); hipEventRecord(e1); hipEventSynchronize(e1);
    float ms; hipEventElapsedTime(&ms, e0, e1);
    std::vector<unsigned long long> h((size_t)maxblocks * 6);
    hipMemcpy(h.data(), g_v2_dbg, h.size() * 8, hipMemcpyDeviceToHost);
    int nb = 0; while (nb < maxblocks && h[(size_t)nb * 6 + 3]) ++nb;
    unsigned long long tmin = ~0ull, tmax = 0; double pro = 0, main_ = 0, epi = 0;
    std::vector<double> dur;
    for (int b = 0; b < nb; ++b) {
        auto* s = &h[(size_t)b * 6];
        tmin = std::min(tmin, s[0]); tmax = std::max(tmax, s[3]);
        pro += s[1] - s[0]; main_ += s[2] - s[1]; epi += s[3] - s[2]; dur.push_back(double(s[3] - s[0]));
    }
    std::sort(dur.begin(), dur.end());
    const double gflop = 2.0 * B * H * H * cout * cin * ks * ks / 1e9;
    printf("H=%d cin=%d cout=%d ks=%d bn=%d: %.1f us, %.1f TFLOP/s, blocks=%d\n", H, cin, cout, ks, bn, ms * 1e3, gflop / ms, nb);
    printf("  kernel span (memtime ticks) %.0f ; per block: prologue %.0f  mainloop %.0f  epilogue %.0f  total median %.0f p10 %.0f p90 %.0f\n",
           double(tmax - tmin), pro / nb, main_ / nb, epi / nb, dur[nb / 2], dur[nb / 10], dur[nb * 9 / 10]);
    const int KT = (cin / 32) * ks * ks;
    printf("  MFMA-cycles per wave per block = %d (KT=%d x %d MFMA x 64); ticks/us = %.1f\n", KT * 32 * (bn / 64) * 64, KT, 32 * (bn / 64), double(tmax - tmin) / (ms * 1e3));
    // concurrency: blocks per CU over time (hw id = se/cu...) -> count distinct hw ids
    std::vector<unsigned long long> ids;
    for (int b = 0; b < nb; ++b) ids.push_back((h[(size_t)b * 6 + 5] << 32) | (h[(size_t)b * 6 + 4] & 0xFFFFFF00u) >> 8);
    std::sort(ids.begin(), ids.end()); ids.erase(std::unique(ids.begin(), ids.end()), ids.end());
    printf("  distinct (xcc, hw_id>>8) = %zu\n", ids.size());
    for (int b : {0, 1, 2, 8, 9, 255, 256, 257, 264, 511, 512, 513, 520, 1000, 1001}) {
        if (b >= nb) continue;
        auto* q = &h[(size_t)b * 6];
        unsigned hw = (unsigned)q[4];
        printf("  blk %4d xcc %llu hw_id %08x wave %u simd %u pipe %u cu %u sh %u se %u tg %u  start %llu dur %llu\n", b, q[5] & 0xf, hw, hw & 15,
               (hw >> 4) & 3, (hw >> 6) & 3, (hw >> 8) & 15, (hw >> 12) & 1, (hw >> 13) & 7, (hw >> 16) & 15, q[0], q[3] - q[0]);
    }
    {
        double ratio = 0; int cnt = 0;
        for (int b = 0; b < nb; ++b) { auto* q = &h[(size_t)b * 6]; double rt = double(q[5] >> 8); if (rt > 0) { ratio += double(q[3] - q[0]) / rt; ++cnt; } }
        printf("  in-kernel clock: mean dmemtime/dmemrealtime = %.3f -> %.3f GHz (100 MHz reference)\n", ratio / cnt, ratio / cnt * 0.1);
    }
    {   // per-CU timeline (s_memtime is only comparable within one CU)
        struct CU { unsigned long long lo = ~0ull, hi = 0; double mfma_busy = 0; int n = 0; };
        std::vector<std::pair<unsigned long long, CU>> cus;
        for (int b = 0; b < nb; ++b) {
            auto* q = &h[(size_t)b * 6];
            unsigned hw = (unsigned)q[4];
            unsigned long long key = ((q[5] & 0xf) << 16) | (hw & 0xff00);        // xcc, se, sh, cu
            CU* cu = nullptr;
            for (auto& kv : cus) if (kv.first == key) cu = &kv.second;
            if (!cu) { cus.push_back({key, CU()}); cu = &cus.back().second; }
            cu->lo = std::min(cu->lo, q[0]); cu->hi = std::max(cu->hi, q[3]); cu->n++;
        }
        double span = 0, smin = 1e30, smax = 0; int nmin = 1 << 30, nmax = 0;
        for (auto& kv : cus) { double sp = double(kv.second.hi - kv.second.lo); span += sp; smin = std::min(smin, sp); smax = std::max(smax, sp); nmin = std::min(nmin, kv.second.n); nmax = std::max(nmax, kv.second.n); }
        span /= cus.size();
        const double mfma_per_block = (double)((cin / 32) * ks * ks) * 32 * (bn / 64) * 64;
        printf("  %zu CUs: blocks/CU %d..%d; span mean %.0f min %.0f max %.0f cycles; kernel %.1f us -> clock(max span) %.3f GHz; MFMA util over max span %.3f\n",
               cus.size(), nmin, nmax, span, smin, smax, ms * 1e3, smax / (ms * 1e3) / 1e3, nb * mfma_per_block / (256.0 * smax));
    }
    // timeline: active blocks in 20 slices
    for (int sl = 0; sl < 20; ++sl) {
        unsigned long long t = tmin + (tmax - tmin) * (2 * sl + 1) / 40; int act = 0, inmain = 0;
        for (int b = 0; b < nb; ++b) { auto* s = &h[(size_t)b * 6]; if (s[0] <= t && t < s[3]) { ++act; if (s[1] <= t && t < s[2]) ++inmain; } }
        printf("  t=%2d%%: resident blocks %4d, in main loop %4d\n", sl * 5 + 2, act, inmain);
    }
    return 0;
}
