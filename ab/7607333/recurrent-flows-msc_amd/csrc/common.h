// Shared helpers for the gfx950 kernels of librfn_hip.so.  Wave = 64 lanes everywhere.
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <stdio.h>
#include <string.h>

#define RFN_WAVE 64

extern "C" const char* rfn_last_error(void);
void rfn_set_error(const char* fmt, ...);

#define RFN_CHECK_ARG(cond, code)                                             \
    do {                                                                      \
        if (!(cond)) {                                                        \
            rfn_set_error("%s: argument check failed: %s", __func__, #cond);  \
            return (code);                                                    \
        }                                                                     \
    } while (0)

#define RFN_LAUNCH_CHECK()                                                    \
    do {                                                                      \
        hipError_t e__ = hipGetLastError();                                   \
        if (e__ != hipSuccess) {                                              \
            rfn_set_error("%s: launch failed: %s", __func__, hipGetErrorString(e__)); \
            return (int)e__;                                                  \
        }                                                                     \
    } while (0)

// Per-device scratch buffer of the split-K convolutions (grow-only, owned by the library, released at process exit;
// an outgrown buffer stays allocated because captured hipGraphs may still point into it): the K
// slices of a split convolution write their partial outputs here and a second kernel adds them in a fixed order -- no
// order-dependent float atomics in any convolution.  Returns nullptr (error set) when the buffer would have to grow
// while `s` is being captured into a hipGraph: run the same shapes eagerly once before capturing.  One buffer per device
// serves all streams: split-K convolutions on different streams of a device must not overlap in time.
float* rfn_workspace(hipStream_t s, size_t floats);

// ---- wave / block reductions (sum) ---------------------------------------------------------------
__device__ __forceinline__ float wave_sum(float v) {
#pragma unroll
    for (int off = 32; off > 0; off >>= 1) v += __shfl_xor(v, off, 64);
    return v;
}

// Sum over each 32-lane half of a wave with DPP adds (VALU rate; __shfl_xor would go through the LDS crossbar, and the
// fused activation-backward epilogue needs 128 of these per wave).  The total of lanes 0-31 lands in lanes 16-31, the
// total of lanes 32-63 in lanes 48-63.
__device__ __forceinline__ float half_wave_sum_dpp(float v) {
#define RFN_DPP_ADD(ctrl, rmask)                                                                              \
    v += __int_as_float(__builtin_amdgcn_update_dpp(0, __float_as_int(v), ctrl, rmask, 0xF, true))
    RFN_DPP_ADD(0xB1, 0xF);   // quad_perm [1,0,3,2]
    RFN_DPP_ADD(0x4E, 0xF);   // quad_perm [2,3,0,1]
    RFN_DPP_ADD(0x141, 0xF);  // row_half_mirror
    RFN_DPP_ADD(0x140, 0xF);  // row_mirror      -> every lane of a 16-lane row holds the row total
    RFN_DPP_ADD(0x142, 0xA);  // row_bcast15 into rows 1 and 3: += total of the previous row
#undef RFN_DPP_ADD
    return v;
}

// whole-wave sum on the DPP path (about a tenth of the six dependent LDS-crossbar shuffles of wave_sum): the two
// half-wave totals are read from lanes 31 and 63; the result is wave-uniform.
__device__ __forceinline__ float wave_sum_dpp(float v) {
    v = half_wave_sum_dpp(v);
    return __int_as_float(__builtin_amdgcn_readlane(__float_as_int(v), 31)) +
           __int_as_float(__builtin_amdgcn_readlane(__float_as_int(v), 63));
}

// block-wide sum for blockDim.x == 256 (4 waves); result valid in every thread.
__device__ __forceinline__ float block_sum_256(float v, float* sm /* >= 4 floats */) {
    v = wave_sum(v);
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    __syncthreads();
    if (lane == 0) sm[wave] = v;
    __syncthreads();
    return sm[0] + sm[1] + sm[2] + sm[3];
}

__device__ __forceinline__ float sigmoidf_(float x) { return 1.0f / (1.0f + __expf(-x)); }

static inline int ceil_div(long a, long b) { return (int)((a + b - 1) / b); }
static inline int next_pow2(int v) {
    int p = 1;
    while (p < v) p <<= 1;
    return p;
}
static inline int ilog2(int v) {
    int l = 0;
    while ((1 << l) < v) ++l;
    return l;
}

// Zero fill as an ordinary kernel node: hipMemsetAsync nodes inside a captured hipGraph were observed to race with the
// neighbouring kernel nodes on replay (split-K outputs picked up stale values), so the library never enqueues memsets.
static __global__ void rfn_zero_f32_kernel(float* __restrict__ p, long n) {
    for (long i = (long)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (long)gridDim.x * blockDim.x) p[i] = 0.f;
}
static inline void rfn_zero_f32(float* p, long n, hipStream_t s) {
    if (n <= 0) return;
    long blocks = (n + 1023) / 1024;
    if (blocks > 1024) blocks = 1024;
    hipLaunchKernelGGL(rfn_zero_f32_kernel, dim3((unsigned)blocks), dim3(256), 0, s, p, n);
}
