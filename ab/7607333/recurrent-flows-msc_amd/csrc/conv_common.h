// Shared pieces of the MFMA convolution kernels (fp32 and bf16x3).
#pragma once
#include "common.h"
#include "../../include/rfn_hip.h"
#include <stdlib.h>

typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef float f32x4 __attribute__((ext_vector_type(4)));

struct ConvParams {
    const float* in1;
    const float* in2;
    long in1_ns, in2_ns;
    int C1, C2;
    const float* wpk;
    float* out1;
    float* out2;
    long out1_ns, out2_ns;
    int Cout, cout_split, acc1, acc2;
    int N, H, W;
    int CoutP, Cin8;
    int ep_mode, act;
    const float* p0;
    const float* p1;
    int TWp, TH, TF, tw_shift, th_shift;
    int n_wtiles, n_htiles, n_ftiles;
    int P2, P2_shift;  // (unused by the forward kernel; kept for layout compatibility)
    int ksplit;        // gridDim.z: the K (input channel chunk) range is split over z; slice z writes its partial outputs to
                       // ws + z * ws_stride (out1's elements, then out2's, both dense) and splitk_reduce_kernel adds
                       // the slices in order (deterministic: no float atomics)
    int w_lds_off;     // float offset of the weight tile inside dynamic LDS (16-byte aligned)
    // ep_mode 4 (data-gradient conv fused with the backward of the producer's ActNorm+activation epilogue):
    const float* ybuf;  // saved forward activation y = act((u+b)*exp(l)), same shape as the output
    long ybuf_ns;
    int npl;            // split-precision planes per operand: 0 / 2 = bf16x3, 3 = bf16x6 (generic tile kernel only)
    float* part;        // [2][Cout] sums Σ gu, Σ g*y, accumulated with float atomics (caller zeroes)
    float* ws;          // split-K partial outputs (rfn_workspace), ksplit slices of ws_stride floats
    long ws_stride;
};

// out[i] = sum_z ws[z * stride + i], z ascending (the second pass of a split-K convolution); n % 4 == 0 not required
static __global__ void splitk_reduce_kernel(const float* __restrict__ ws, long stride, int ksplit, float* __restrict__ out1,
                                            long n1, float* __restrict__ out2, long n2) {
    const long n = n1 + n2;
    for (long i = (long)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (long)gridDim.x * blockDim.x) {
        float a = ws[i];
        for (int z = 1; z < ksplit; ++z) a += ws[(long)z * stride + i];
        if (i < n1) out1[i] = a;
        else out2[i - n1] = a;
    }
}
static inline void splitk_reduce(const ConvParams& p, hipStream_t s) {
    const long HW = (long)p.H * p.W, n1 = (long)p.N * p.cout_split * HW, n2 = (long)p.N * (p.Cout - p.cout_split) * HW;
    long blocks = (n1 + n2 + 255) / 256;
    if (blocks > 2048) blocks = 2048;
    hipLaunchKernelGGL(splitk_reduce_kernel, dim3((unsigned)blocks), dim3(256), 0, s, p.ws, p.ws_stride, p.ksplit, p.out1, n1,
                       p.out2, n2);
}

// Epilogue shared by the fp32 and the bf16x3 kernels (the C/D register layout of the 32x32 MFMA tile does not depend
// on the input dtype): per-channel affine + activation, bounds, output split over two tensors, accumulate / atomic.
// FASTONLY: the caller guarantees full cout tiles, one output tensor, no accumulate / split-K and ep_mode <= 3 -- only the
// straight store path is instantiated (the weight-stationary kernels have no registers to spare for the others).
template <int TCO, int TPX, int BCO, bool FASTONLY = false>
__device__ __forceinline__ void conv_epilogue(const ConvParams& p, f32x16 (&acc)[TCO][TPX], const float* ep,
                                              const int co_base, const int wco, const int kk, const int HW,
                                              const int (&pn)[TPX], const int (&ppix)[TPX],
                                              const bool (&pvalid)[TPX], const int prow = 0,
                                              float* const lds_part = nullptr, const bool lds_shared = false) {
    const int cl_base = wco * (32 * TCO) + 4 * kk;  // channel index inside the block for (a=0, r=0)
    const bool fast = FASTONLY || ((co_base + 32 * TCO <= p.Cout) && (p.cout_split == p.Cout));  // wave-uniform
    if (!FASTONLY && p.ep_mode == 4) {
        // g = this conv's result = grad wrt y = act((u+b)*exp(l)).  Emit gu = g*act'(y)*exp(l) (what the weight- and
        // data-gradient of the producer conv consume) and this wave's per-channel Σ gu (-> grad b) and Σ g*y (-> grad l),
        // so the separate elementwise+reduction pass over the 256-channel hidden tensors disappears.  Host guarantees
        // full cout tiles, a single output tensor and no split-K.
        if (co_base + 32 * TCO > p.Cout) return;  // whole wave past Cout (Cout % 64 == 0: never a partial tile)
        const float* ybase[TPX];
        float* obase[TPX];
#pragma unroll
        for (int t = 0; t < TPX; ++t) {
            const long off = (long)(co_base + 4 * kk) * HW + ppix[t];
            // masked pixels read frame 0 (always mapped) so that every load below is unconditional and batched
            ybase[t] = pvalid[t] ? p.ybuf + pn[t] * p.ybuf_ns + off : p.ybuf + (long)(co_base + 4 * kk) * HW;
            obase[t] = p.out1 + pn[t] * p.out1_ns + off;
        }
        const int l31 = threadIdx.x & 31;
#pragma unroll
        for (int a = 0; a < TCO; ++a) {
            float yv[16][TPX];
#pragma unroll
            for (int r = 0; r < 16; ++r)
#pragma unroll
                for (int t = 0; t < TPX; ++t)
                    yv[r][t] = ybase[t][(long)(a * 32 + (r & 3) + 8 * (r >> 2)) * HW];
#pragma unroll
            for (int r = 0; r < 16; ++r) {
                const int cidx = a * 32 + (r & 3) + 8 * (r >> 2);
                const float e1 = ep[BCO + cl_base + cidx];
                float sb = 0.f, sl = 0.f;
#pragma unroll
                for (int t = 0; t < TPX; ++t) {
                    const float g = acc[a][t][r];
                    const float y = yv[r][t];
                    float slope = 1.f;
                    if (p.act == 1) slope = y > 0.f ? 1.f : 0.f;
                    if (p.act == 2) slope = y > 0.f ? 1.f : 0.2f;
                    const float gu = pvalid[t] ? g * slope * e1 : 0.f;
                    if (pvalid[t]) obase[t][(long)cidx * HW] = gu;
                    sb += gu;
                    sl += pvalid[t] ? g * y : 0.f;
                }
                sb = half_wave_sum_dpp(sb);  // lanes of one 32-lane half share the channel
                sl = half_wave_sum_dpp(sl);
                if (l31 == 16) {
                    if (lds_part && lds_shared) {
                        // generic kernel: the workgroup's waves along the pixel axis share a channel -> LDS atomics; the
                        // workgroup then flushes [BCO][2] with full-width atomic instructions (a global float atomic is
                        // charged per WAVE INSTRUCTION, ~50 ns per CU, however few lanes are active: issuing them from
                        // here, two lanes at a time, cost 60-100 us per launch at the 8x8 / 4x4 levels)
                        float* dst = lds_part + (cl_base + cidx) * 2;
                        atomicAdd(dst, sb);
                        atomicAdd(dst + 1, sl);
                    } else if (lds_part) {
                        // persistent kernels: running sums of the workgroup in LDS [BCO][2] (this lane is the only
                        // owner of its channel), flushed once per workgroup instead of one row per pixel tile
                        float* dst = lds_part + (cl_base + cidx) * 2;
                        dst[0] += sb;
                        dst[1] += sl;
                    } else {
                        float* dst = p.part + co_base + cidx + 4 * kk;  // [2][Cout], accumulated
                        atomicAdd(dst, sb);
                        atomicAdd(dst + p.Cout, sl);
                    }
                }
            }
        }
        return;
    }
    if (fast) {
        float* obase[TPX];
        // (split-K: slice z stores into its own dense copy of out1 inside the workspace; out1 is dense then)
        float* const o1 = (!FASTONLY && p.ksplit > 1) ? p.ws + (long)blockIdx.z * p.ws_stride : p.out1;
#pragma unroll
        for (int t = 0; t < TPX; ++t)
            obase[t] = o1 + pn[t] * p.out1_ns + (long)(co_base + 4 * kk) * HW + ppix[t];
#pragma unroll
        for (int a = 0; a < TCO; ++a) {
#pragma unroll
            for (int r = 0; r < 16; ++r) {
                const int cidx = a * 32 + (r & 3) + 8 * (r >> 2);  // compile time
                float e0 = 0.f, e1 = 1.f;
                if (p.ep_mode != 0) {
                    e0 = blockIdx.z == 0 ? ep[cl_base + cidx] : 0.f;  // the additive term enters once per output
                    e1 = ep[BCO + cl_base + cidx];
                }
#pragma unroll
                for (int t = 0; t < TPX; ++t) {
                    float v = acc[a][t][r];
                    if (p.ep_mode != 0) v = (v + e0) * e1;
                    if (p.ep_mode == 1) {
                        if (p.act == 1) v = v > 0.f ? v : 0.f;
                        if (p.act == 2) v = v > 0.f ? v : 0.2f * v;
                    }
                    if (pvalid[t]) {
                        float* dst = obase[t] + (long)cidx * HW;
                        if (FASTONLY || p.ksplit > 1) {
                            *dst = v;
                        } else {
                            if (p.acc1) v += *dst;
                            *dst = v;
                        }
                    }
                }
            }
        }
        return;
    }
    if (FASTONLY) return;
    // general path: ragged Cout and/or output split over two tensors
#pragma unroll
    for (int a = 0; a < TCO; ++a) {
#pragma unroll
        for (int r = 0; r < 16; ++r) {
            const int cidx = a * 32 + (r & 3) + 8 * (r >> 2);
            const int co = co_base + cidx + 4 * kk;
            if (co >= p.Cout) continue;
            float e0 = 0.f, e1 = 1.f;
            if (p.ep_mode != 0) {
                e0 = blockIdx.z == 0 ? ep[cl_base + cidx] : 0.f;
                e1 = ep[BCO + cl_base + cidx];
            }
            const bool first = co < p.cout_split;
            float* obase = first ? p.out1 : p.out2;
            if (p.ksplit > 1)   // slice z of the workspace: [out1 dense | out2 dense]
                obase = p.ws + (long)blockIdx.z * p.ws_stride + (first ? 0 : (long)p.N * p.cout_split * HW);
            const long ons = first ? p.out1_ns : p.out2_ns;
            const int oc = first ? co : co - p.cout_split;
            const int accm = first ? p.acc1 : p.acc2;
#pragma unroll
            for (int t = 0; t < TPX; ++t) {
                if (!pvalid[t]) continue;
                float v = acc[a][t][r];
                if (p.ep_mode != 0) v = (v + e0) * e1;
                if (p.ep_mode == 1) {
                    if (p.act == 1) v = v > 0.f ? v : 0.f;
                    if (p.act == 2) v = v > 0.f ? v : 0.2f * v;
                }
                float* dst = obase + pn[t] * ons + (long)oc * HW + ppix[t];
                if (p.ksplit > 1) {
                    *dst = v;
                } else {
                    if (accm) v += *dst;
                    *dst = v;
                }
            }
        }
    }
}

static inline void tile_geometry(int H, int W, int BPX, int* TWp, int* TH, int* TF) {
    int tw = next_pow2(W);
    if (tw > 32) tw = 32;
    if (tw > BPX) tw = BPX;
    int th = next_pow2(H);
    if (th > BPX / tw) th = BPX / tw;
    *TWp = tw;
    *TH = th;
    *TF = BPX / (tw * th);
}

