// Developer tool: sustained v_mfma_f32_32x32x2_f32 rate (register operands only) to calibrate the achievable peak
// (clock under load) on the box.  hipcc --offload-arch=gfx950 -O3 tools/mfma_peak.hip -o /tmp/mfma_peak && /tmp/mfma_peak
#include <hip/hip_runtime.h>
#include <stdio.h>
typedef float f32x16 __attribute__((ext_vector_type(16)));
template <int NACC>
__global__ __launch_bounds__(256) void peak(float* out, int iters, float a0, float b0) {
    f32x16 acc[NACC];
    for (int i = 0; i < NACC; ++i)
        for (int r = 0; r < 16; ++r) acc[i][r] = 0.f;
    float a = a0 + threadIdx.x * 1e-3f, b = b0 + threadIdx.x * 2e-3f;
    for (int it = 0; it < iters; ++it) {
#pragma unroll
        for (int i = 0; i < NACC; ++i) acc[i] = __builtin_amdgcn_mfma_f32_32x32x2f32(a, b, acc[i], 0, 0, 0);
    }
    float s = 0.f;
    for (int i = 0; i < NACC; ++i)
        for (int r = 0; r < 16; ++r) s += acc[i][r];
    out[blockIdx.x * blockDim.x + threadIdx.x] = s;
}
int main() {
    float* out;
    hipMalloc(&out, 256 * 2048 * 4);
    hipEvent_t e0, e1;
    hipEventCreate(&e0);
    hipEventCreate(&e1);
    for (int blocks_per_cu = 1; blocks_per_cu <= 2; ++blocks_per_cu) {
        int grid = 256 * blocks_per_cu, iters = 20000;
        hipLaunchKernelGGL(peak<4>, dim3(grid), dim3(256), 0, 0, out, 100, 1.f, 2.f);
        hipDeviceSynchronize();
        hipEventRecord(e0);
        hipLaunchKernelGGL(peak<4>, dim3(grid), dim3(256), 0, 0, out, iters, 1.f, 2.f);
        hipEventRecord(e1);
        hipEventSynchronize(e1);
        float ms;
        hipEventElapsedTime(&ms, e0, e1);
        double flops = (double)grid * 4 /*waves*/ * iters * 4 /*acc*/ * 4096.0;
        printf("blocks/CU %d: %.3f ms  %.1f TFLOP/s  (=> clock %.2f GHz at 64 FLOP/clk/SIMD)\n", blocks_per_cu, ms,
               flops / ms / 1e9, flops / ms / 1e9 * 1e12 / (1024.0 * 64) / 1e9);
    }
    return 0;
}
