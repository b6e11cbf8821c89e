// Which part of the conv inner loop costs clock? Variants of an MFMA loop with in-kernel clock
// measurement (s_memtime / s_memrealtime).  hipcc --offload-arch=gfx950 -O3 tools/clock_probe.hip
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>
typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef float f32x4 __attribute__((ext_vector_type(4)));

// MODE bit0: A frags from LDS (2 x b128 per 8 MFMA... per-wave 64-row tile), bit1: B frag from global (1 KiB / 8 MFMA / wave)
template <int MODE>
__global__ __launch_bounds__(256) void k(float* out, const float* __restrict__ wbuf, int wfloats, int iters, float seed,
                                         unsigned long long* clk) {
    __shared__ __attribute__((aligned(16))) float sm[256 * 36];
    for (int i = threadIdx.x; i < 256 * 36; i += 256) sm[i] = seed * (float)(i % 97) * 0.01f;
    __syncthreads();
    const unsigned long long t0 = __builtin_amdgcn_s_memtime(), r0 = __builtin_amdgcn_s_memrealtime();
    f32x16 acc[2];
    for (int a = 0; a < 2; ++a) for (int r = 0; r < 16; ++r) acc[a][r] = 0.f;
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    f32x4 a0 = {seed, seed * 2, seed * 3, seed * 4}, a1 = a0, b0 = {seed * 5, seed * 6, seed * 7, seed * 8};
    const float* wp = wbuf + ((size_t)(blockIdx.x % 8) * 65536 + (wave & 1) * 32768 + lane * 4) % wfloats;
    f32x4 bn = *reinterpret_cast<const f32x4*>(wp);
    for (int it = 0; it < iters; ++it) {
        if (MODE & 2) {
            b0 = bn;
            bn = *reinterpret_cast<const f32x4*>(wp + ((size_t)(it + 1) * 256) % 16384);
            __builtin_amdgcn_sched_barrier(0);
        }
        if (MODE & 1) {
            a0 = *reinterpret_cast<const f32x4*>(&sm[((lane & 31) + (it & 3) * 64) * 36 + 4 * (lane >> 5) + (it & 4) * 2]);
            a1 = *reinterpret_cast<const f32x4*>(&sm[((lane & 31) + (it & 3) * 64 + 32) * 36 + 4 * (lane >> 5) + (it & 4) * 2]);
        }
#pragma unroll
        for (int e = 0; e < 4; ++e) {
            acc[0] = __builtin_amdgcn_mfma_f32_32x32x2f32(a0[e], b0[e], acc[0], 0, 0, 0);
            acc[1] = __builtin_amdgcn_mfma_f32_32x32x2f32(a1[e], b0[e], acc[1], 0, 0, 0);
        }
    }
    float s = 0;
    for (int a = 0; a < 2; ++a) for (int r = 0; r < 16; ++r) s += acc[a][r];
    out[blockIdx.x * 256 + threadIdx.x] = s;
    if (threadIdx.x == 0) {
        clk[blockIdx.x * 2] = __builtin_amdgcn_s_memtime() - t0;
        clk[blockIdx.x * 2 + 1] = __builtin_amdgcn_s_memrealtime() - r0;
    }
}

template <int MODE>
void run(const char* name, int blocks) {
    float *out, *w; unsigned long long* clk;
    const int wfloats = 1 << 20;
    hipMalloc(&out, sizeof(float) * blocks * 256); hipMalloc(&w, wfloats * 4); hipMalloc(&clk, blocks * 16);
    std::vector<float> hw(wfloats); for (int i = 0; i < wfloats; ++i) hw[i] = (i % 1013) * 1e-3f - 0.5f;
    hipMemcpy(w, hw.data(), wfloats * 4, hipMemcpyHostToDevice);
    const int iters = 8192;
    hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
    float ms = 0;
    for (int rep = 0; rep < 6; ++rep) {
        hipEventRecord(e0);
        hipLaunchKernelGGL((k<MODE>), dim3(blocks), dim3(256), 0, 0, out, w, wfloats, iters, 0.37f, clk);
        hipEventRecord(e1); hipEventSynchronize(e1);
        hipEventElapsedTime(&ms, e0, e1);
    }
    std::vector<unsigned long long> h(blocks * 2); hipMemcpy(h.data(), clk, blocks * 16, hipMemcpyDeviceToHost);
    double ratio = 0; for (int b = 0; b < blocks; ++b) ratio += double(h[2 * b]) / double(h[2 * b + 1]);
    double flop = (double)blocks * 4 * iters * 8 * 2.0 * 32 * 32 * 2;
    printf("%-44s %8.3f ms %7.2f TFLOP/s  clock %.3f GHz  cycles/8MFMA %.1f\n", name, ms, flop / ms / 1e9, ratio / blocks * 0.1,
           double(h[0]) / iters);
    hipFree(out); hipFree(w); hipFree(clk);
}

int main() {
    for (int blocks : {256, 512}) {
        printf("blocks = %d\n", blocks);
        run<0>("regs only", blocks);
        run<1>("A from LDS (2 b128 / 8 MFMA)", blocks);
        run<2>("B from global L2 (1 KiB / 8 MFMA / wave)", blocks);
        run<3>("A from LDS + B from global", blocks);
    }
    return 0;
}
