// Standalone ceiling probe: back-to-back v_mfma_f32_32x32x2_f32 from registers (and with LDS
// fragment reads), to know what "100 %" is on THIS device before tuning the conv kernels.
//   hipcc --offload-arch=gfx950 -O3 tools/mfma_peak.hip -o /tmp/mfma_peak && /tmp/mfma_peak
#include <hip/hip_runtime.h>
#include <cstdio>
typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef float f32x4 __attribute__((ext_vector_type(4)));

template <int NACC, bool LDS>
__global__ __launch_bounds__(256) void k(float* out, int iters, float seed) {
    __shared__ __attribute__((aligned(16))) float sm[256 * 36];
    for (int i = threadIdx.x; i < 256 * 36; i += 256) sm[i] = seed * (float)(i % 97) * 0.01f;
    __syncthreads();
    f32x16 acc[NACC];
    for (int a = 0; a < NACC; ++a) for (int r = 0; r < 16; ++r) acc[a][r] = 0.f;
    const int lane = threadIdx.x & 63;
    f32x4 a0 = {seed, seed * 2, seed * 3, seed * 4}, b0 = {seed * 5, seed * 6, seed * 7, seed * 8};
    for (int it = 0; it < iters; ++it) {
        if (LDS) {
            a0 = *reinterpret_cast<const f32x4*>(&sm[((lane & 31) + (it & 7) * 32) * 36 + 4 * (lane >> 5)]);
            b0 = *reinterpret_cast<const f32x4*>(&sm[((lane & 31) + ((it + 3) & 7) * 32) * 36 + 4 * (lane >> 5) + 8]);
        }
#pragma unroll
        for (int e = 0; e < 4; ++e)
#pragma unroll
            for (int a = 0; a < NACC; ++a) acc[a] = __builtin_amdgcn_mfma_f32_32x32x2f32(a0[e], b0[e], acc[a], 0, 0, 0);
    }
    float s = 0;
    for (int a = 0; a < NACC; ++a) for (int r = 0; r < 16; ++r) s += acc[a][r];
    out[blockIdx.x * 256 + threadIdx.x] = s;
}

template <int NACC, bool LDS>
void run(const char* name, int blocks) {
    float* out; hipMalloc(&out, sizeof(float) * blocks * 256);
    const int iters = 4096;
    hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
    for (int rep = 0; rep < 3; ++rep) {
        hipEventRecord(e0);
        hipLaunchKernelGGL((k<NACC, LDS>), dim3(blocks), dim3(256), 0, 0, out, iters, 0.37f);
        hipEventRecord(e1); hipEventSynchronize(e1);
        float ms; hipEventElapsedTime(&ms, e0, e1);
        double flop = (double)blocks * 4 /*waves*/ * iters * 4 * NACC * 2.0 * 32 * 32 * 2;
        if (rep == 2) printf("%-28s blocks=%5d  %8.3f ms  %7.2f TFLOP/s\n", name, blocks, ms, flop / ms / 1e9);
    }
    hipFree(out);
}

template <int NACC, bool LDS>
void sustained(const char* name, int blocks, double seconds) {
    float* out; hipMalloc(&out, sizeof(float) * blocks * 256);
    const int iters = 4096;
    hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
    double total = 0;
    while (total < seconds) {
        hipEventRecord(e0);
        for (int r = 0; r < 40; ++r) hipLaunchKernelGGL((k<NACC, LDS>), dim3(blocks), dim3(256), 0, 0, out, iters, 0.37f);
        hipEventRecord(e1); hipEventSynchronize(e1);
        float ms; hipEventElapsedTime(&ms, e0, e1);
        double flop = 40.0 * blocks * 4 * iters * 4 * NACC * 2.0 * 32 * 32 * 2;
        total += ms / 1e3;
        printf("%-24s t=%5.2fs  %7.2f TFLOP/s\n", name, total, flop / ms / 1e9);
    }
    hipFree(out);
}

int main() {
    sustained<4, false>("sustained regs 2blk/CU", 512, 3.0);
    sustained<4, true>("sustained lds  2blk/CU", 512, 3.0);
    run<4, false>("regs, 4 acc, 1 blk/CU", 256);
    run<4, false>("regs, 4 acc, 2 blk/CU", 512);
    run<4, false>("regs, 4 acc, 8 blk/CU", 2048);
    run<2, false>("regs, 2 acc, 2 blk/CU", 512);
    run<1, false>("regs, 1 acc, 2 blk/CU", 512);
    run<1, false>("regs, 1 acc, 4 blk/CU", 1024);
    run<4, true>("lds frag, 4 acc, 1 blk/CU", 256);
    run<4, true>("lds frag, 4 acc, 2 blk/CU", 512);
    run<2, true>("lds frag, 2 acc, 4 blk/CU", 1024);
    return 0;
}
