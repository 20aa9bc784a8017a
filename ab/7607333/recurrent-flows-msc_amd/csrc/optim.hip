// Adam (Kingma & Ba; torch.optim.Adam's arithmetic, RFN/trainer.py:96 of the reference builds it with defaults) over ALL
// parameter tensors of the model in one launch.  The step is memory bound: 28 bytes per parameter (p, g, m, v read; p, m,
// v written), 1.0 GB for the canonical RFN; at::native's fused multi-tensor kernel needs 35 launches of <= 4 KB argument
// tables for the 1269 tensors and reaches ~0.85 TB/s.  Here the table lives in device memory (built once while the
// gradient tensors are static, i.e. in hipGraph mode) and a flat chunk list maps workgroups to (tensor, offset).
#include "common.h"
#include "../../include/rfn_hip.h"

static_assert(sizeof(rfn_adam_entry) == 48, "rfn_adam_entry layout is part of the ABI");

constexpr int ADAM_THREADS = 256;

__global__ __launch_bounds__(ADAM_THREADS) void adam_multi_kernel(const rfn_adam_entry* __restrict__ tab,
                                                                   const int2* __restrict__ chunks, int chunk_elems,
                                                                   double lr, double beta1d, double beta2d, float eps,
                                                                   float weight_decay, int t) {
    __shared__ float s_step_size, s_bc2_sqrt;
    const int2 ck = chunks[blockIdx.x];
    const rfn_adam_entry e = tab[ck.x];
    if (threadIdx.x == 0) {
        const double step = (double)(t - e.step_offset);
        s_step_size = (float)(lr / (1.0 - pow(beta1d, step)));
        s_bc2_sqrt = (float)sqrt(1.0 - pow(beta2d, step));
    }
    __syncthreads();
    // 1 - beta in double, then rounded: 1.f - (float)0.999 is off by 1.3e-5 relative
    const float step_size = s_step_size, bc2_sqrt = s_bc2_sqrt, w1 = (float)(1.0 - beta1d), w2 = (float)(1.0 - beta2d);
    const float beta2 = (float)beta2d;
    const long base = (long)ck.y * chunk_elems;
    const long rem = e.n - base;
    const int n = (int)(rem < chunk_elems ? rem : chunk_elems);
    float* __restrict__ p = e.p + base;
    const float* __restrict__ g = e.g + base;
    float* __restrict__ m = e.m + base;
    float* __restrict__ v = e.v + base;
    auto upd = [&](float& pp, float gg, float& mm, float& vv) {
        if (weight_decay != 0.f) gg = fmaf(weight_decay, pp, gg);
        mm = fmaf(w1, gg - mm, mm);
        vv = fmaf(w2 * gg, gg, beta2 * vv);
        const float denom = sqrtf(vv) / bc2_sqrt + eps;
        pp -= step_size * (mm / denom);
    };
    const bool v4 = ((((uintptr_t)p | (uintptr_t)g | (uintptr_t)m | (uintptr_t)v) & 15) == 0);
    int i0 = 0;
    if (v4) {
        const int n4 = n >> 2;
        for (int i = threadIdx.x; i < n4; i += ADAM_THREADS) {
            float4 pp = reinterpret_cast<float4*>(p)[i];
            const float4 gg = reinterpret_cast<const float4*>(g)[i];
            float4 mm = reinterpret_cast<float4*>(m)[i];
            float4 vv = reinterpret_cast<float4*>(v)[i];
            upd(pp.x, gg.x, mm.x, vv.x);
            upd(pp.y, gg.y, mm.y, vv.y);
            upd(pp.z, gg.z, mm.z, vv.z);
            upd(pp.w, gg.w, mm.w, vv.w);
            reinterpret_cast<float4*>(p)[i] = pp;
            reinterpret_cast<float4*>(m)[i] = mm;
            reinterpret_cast<float4*>(v)[i] = vv;
        }
        i0 = n4 << 2;
    }
    for (int i = i0 + threadIdx.x; i < n; i += ADAM_THREADS) {
        float pp = p[i], mm = m[i], vv = v[i];
        upd(pp, g[i], mm, vv);
        p[i] = pp;
        m[i] = mm;
        v[i] = vv;
    }
}

extern "C" int rfn_adam_chunk_elems(void) { return 8192; }

extern "C" int rfn_adam_step_f32(const rfn_adam_entry* table, const int* chunks, int n_chunks, double lr, double beta1,
                                 double beta2, double eps, double weight_decay, int t, rfn_stream_t stream) {
    RFN_CHECK_ARG(table && chunks && n_chunks >= 0, -1);
    RFN_CHECK_ARG(beta1 >= 0. && beta1 < 1. && beta2 >= 0. && beta2 < 1. && eps >= 0., -2);
    if (n_chunks == 0) return 0;
    hipLaunchKernelGGL(adam_multi_kernel, dim3(n_chunks), dim3(ADAM_THREADS), 0, (hipStream_t)stream, table,
                       reinterpret_cast<const int2*>(chunks), rfn_adam_chunk_elems(), lr, beta1, beta2, (float)eps,
                       (float)weight_decay, t);
    RFN_LAUNCH_CHECK();
    return 0;
}
