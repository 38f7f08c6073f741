// Micro-benchmark (tooling): how long is one s_memtime tick, on an idle chip and on a busy one?
#include <hip/hip_runtime.h>
#include <stdio.h>
__global__ void spin(long long ticks, float* out) {
    const long long t0 = __builtin_readcyclecounter();
    float v = threadIdx.x;
    while (__builtin_readcyclecounter() - t0 < ticks) v = v * 1.0001f + 0.5f;
    if (v == 12345.f) out[0] = v;
}
int main() {
    float* out; hipMalloc(&out, 4);
    hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
    for (int grid : {1, 64, 256, 2048}) {
        hipLaunchKernelGGL(spin, dim3(grid), dim3(256), 0, 0, 1000LL, out); hipDeviceSynchronize();
        hipEventRecord(e0);
        hipLaunchKernelGGL(spin, dim3(grid), dim3(256), 0, 0, 1000000LL, out);
        hipEventRecord(e1); hipEventSynchronize(e1);
        float ms; hipEventElapsedTime(&ms, e0, e1);
        printf("grid %5d: 1e6 ticks took %.3f ms -> %.2f ns per tick\n", grid, ms, ms * 1e6 / 1e6);
    }
    return 0;
}
