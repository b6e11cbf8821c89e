// How many 256-thread workgroups with X bytes of dynamic LDS does one gfx950 CU really hold? (tuning probe)
// Each block spins for a fixed number of cycles; with G = 12 blocks per CU the kernel takes ceil(12 / resident) spins.
//   hipcc --offload-arch=gfx950 -O2 tools/lds_occ_probe.hip -o /tmp/lds_occ_probe && /tmp/lds_occ_probe
#include <hip/hip_runtime.h>
#include <cstdio>
__global__ __launch_bounds__(256) void spin(long long cycles, int* sink) {
    extern __shared__ int lds[];
    lds[threadIdx.x] = threadIdx.x;
    __syncthreads();
    const long long t0 = __builtin_amdgcn_s_memtime();
    while ((long long)__builtin_amdgcn_s_memtime() - t0 < cycles) {}
    if (lds[(threadIdx.x + 1) & 255] == -1) sink[0] = 1;
}
int main() {
    int* sink;
    hipMalloc(&sink, 4);
    const long long cycles = 200000;                 // ~100 us
    const int per_cu = 12, cus = 256;
    for (int kb = 24; kb <= 84; kb += 2) {
        const size_t lds = (size_t)kb * 1024;
        hipFuncSetAttribute((const void*)spin, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
        hipEvent_t a, b;
        hipEventCreate(&a); hipEventCreate(&b);
        hipLaunchKernelGGL(spin, dim3(cus * per_cu), dim3(256), lds, 0, cycles, sink);
        hipDeviceSynchronize();
        hipEventRecord(a);
        hipLaunchKernelGGL(spin, dim3(cus * per_cu), dim3(256), lds, 0, cycles, sink);
        hipEventRecord(b);
        hipEventSynchronize(b);
        float ms;
        hipEventElapsedTime(&ms, a, b);
        int occ = -1;
        hipOccupancyMaxActiveBlocksPerMultiprocessor(&occ, spin, 256, lds);
        printf("LDS %2d KB: %.3f ms  -> ~%.1f rounds (API says %d blocks/CU)\n", kb, ms, ms / 0.095, occ);
    }
    return 0;
}
