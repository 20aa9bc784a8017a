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
    int ksplit;        // gridDim.z: the K (input channel chunk) range is split over z, partial sums meet by atomicAdd
    int w_lds_off;     // float offset of the weight tile inside dynamic LDS (16-byte aligned)
};

// Epilogue shared by the fp32 and the bf16x3 kernels (the C/D register layout of the 32x32 MFMA tile does not depend
// on the input dtype): per-channel affine + activation, bounds, output split over two tensors, accumulate / atomic.
template <int TCO, int TPX, int BCO>
__device__ __forceinline__ void conv_epilogue(const ConvParams& p, f32x16 (&acc)[TCO][TPX], const float* ep,
                                              const int co_base, const int wco, const int kk, const int HW,
                                              const int (&pn)[TPX], const int (&ppix)[TPX],
                                              const bool (&pvalid)[TPX]) {
    const int cl_base = wco * (32 * TCO) + 4 * kk;  // channel index inside the block for (a=0, r=0)
    const bool fast = (co_base + 32 * TCO <= p.Cout) && (p.cout_split == p.Cout);  // wave-uniform
    if (fast) {
        float* obase[TPX];
#pragma unroll
        for (int t = 0; t < TPX; ++t)
            obase[t] = p.out1 + pn[t] * p.out1_ns + (long)(co_base + 4 * kk) * HW + ppix[t];
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
                        if (p.ksplit > 1) {
                            atomicAdd(dst, v);
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
                    atomicAdd(dst, v);
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

