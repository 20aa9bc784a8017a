// 3x3 convolutions on very small feature maps (H*W <= 16), batch-of-timestep sized: the per-timestep parameter nets of
// the SRNN (RFN/RFN_new.py:167-179 call prior / encoder once per frame on [B, C, 2, 2] maps at 64x64 input).
//
// On a 2x2 (or 4x4) map with padding 1 every output pixel sees every input pixel, so the convolution IS a dense layer
//     out[b][(co,po)] = bias[co] + sum_{(ci,pi)} x[b][(ci,pi)] * Weff[(ci,pi)][(co,po)],   Weff = w[co][ci][tap(po,pi)]
// and (ci,pi) / (co,po) are exactly the NCHW memory order of one sample.  With B <= 32 samples the batch is ONE MFMA
// row tile; the work is streaming Weff (a few MB, L2 / Infinity-Cache resident across the 19 timesteps) through the
// matrix cores.  Weff is packed once per optimizer step in MFMA B-fragment order, already split into bf16 (hi, lo)
// (split precision as in conv_bf16x3.hip: a*b ~= ah*bh + ah*bl + al*bh, fp32 accumulate), for the forward product and
// transposed for the data gradient; the per-timestep kernels then issue one coalesced 16-byte load per fragment.
// A general-purpose conv library pays ~35 us per such call (layout transposes + a tiled kernel + bias + activation
// launches); these kernels are bound by the weight stream.
#include "conv_common.h"

typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));

// ------------------------------------------------------------------------------------------------ packing
// packed[((tile*KS + ks)*2 + plane)*64 + lane] (16-byte units): lane -> n = tile*32 + lane%32, k = ks*16 + (lane/32)*8 + j
// transpose = 0: k = (ci,pi), n = (co,po)   (forward)        transpose = 1: k = (co,po), n = (ci,pi)   (data gradient)
__device__ __forceinline__ void smallmap_pack_item(const float* __restrict__ w, const int Cout, const int Cin, const int H,
                                                   const int W, const int transpose, bf16x8* __restrict__ packed,
                                                   const int KS, const long idx);
__global__ void smallmap_pack_kernel(const float* __restrict__ w, int Cout, int Cin, int H, int W, int transpose,
                                     bf16x8* __restrict__ packed, int KS, int NTILES) {
    const long total = (long)NTILES * KS * 64;
    for (long idx = (long)blockIdx.x * blockDim.x + threadIdx.x; idx < total; idx += (long)gridDim.x * blockDim.x)
        smallmap_pack_item(w, Cout, Cin, H, W, transpose, packed, KS, idx);
}

// one fragment pair (hi, lo) of a packed matrix: the body of both pack kernels
__device__ __forceinline__ void smallmap_pack_item(const float* __restrict__ w, const int Cout, const int Cin, const int H,
                                                   const int W, const int transpose, bf16x8* __restrict__ packed,
                                                   const int KS, const long idx) {
    const int HW = H * W;
    const int lane = (int)(idx & 63);
    const long r = idx >> 6;
    const int ks = (int)(r % KS), tile = (int)(r / KS);
    const int n = tile * 32 + (lane & 31);
    const int kbase = ks * 16 + (lane >> 5) * 8;
    bf16x8 hi, lo;
#pragma unroll
    for (int j = 0; j < 8; ++j) {
        const int k = kbase + j;
        const int cn = n / HW, pn = n - cn * HW, ck = k / HW, pk = k - ck * HW;
        const int co = transpose ? ck : cn, po = transpose ? pk : pn;
        const int ci = transpose ? cn : ck, pi = transpose ? pn : pk;
        const int dy = pi / W - po / W + 1, dx = pi % W - po % W + 1;
        float v = 0.f;
        if (co < Cout && ci < Cin && dy >= 0 && dy < 3 && dx >= 0 && dx < 3)
            v = w[((long)co * Cin + ci) * 9 + dy * 3 + dx];
        const __bf16 h = (__bf16)v;
        hi[j] = h;
        lo[j] = (__bf16)(v - (float)h);
    }
    packed[(r * 2 + 0) * 64 + lane] = hi;
    packed[(r * 2 + 1) * 64 + lane] = lo;
}

// Up to SM_PACK_MAX matrices in ONE launch (the weights change every optimizer step: the latent nets, the ConvLSTM and the
// 2x2 flow level re-pack ~60 matrices per step, each a ~10 us launch of its own until round 3).  The descriptors travel
// by value in the kernel arguments; block (x, y) works on matrix y.
#define SM_PACK_MAX 64
struct SmallmapPackDesc {   // mirrors rfn_smallmap_pack_desc in include/rfn_hip.h
    const float* w;
    float* packed;
    int Cout, Cin, H, W, transpose, pad_;
};
struct SmallmapPackTable { SmallmapPackDesc d[SM_PACK_MAX]; };
__global__ __launch_bounds__(256) void smallmap_pack_batched_kernel(const SmallmapPackTable t) {
    const SmallmapPackDesc d = t.d[blockIdx.y];
    const int HW = d.H * d.W;
    const int KS = ((d.transpose ? d.Cout : d.Cin) * HW + 15) / 16, NT = ((d.transpose ? d.Cin : d.Cout) * HW + 31) / 32;
    const long total = (long)NT * KS * 64;
    for (long idx = (long)blockIdx.x * 256 + threadIdx.x; idx < total; idx += (long)gridDim.x * 256)
        smallmap_pack_item(d.w, d.Cout, d.Cin, d.H, d.W, d.transpose, reinterpret_cast<bf16x8*>(d.packed), KS, idx);
}

static void smallmap_dims(int Cout, int Cin, int HW, int transpose, int* K, int* N, int* KS, int* NTILES) {
    *K = (transpose ? Cout : Cin) * HW;
    *N = (transpose ? Cin : Cout) * HW;
    *KS = (*K + 15) / 16;
    *NTILES = (*N + 31) / 32;
}

extern "C" long rfn_smallmap_packed_size(int Cout, int Cin, int H, int W, int transpose) {
    int K, N, KS, NT;
    smallmap_dims(Cout, Cin, H * W, transpose, &K, &N, &KS, &NT);
    return (long)NT * KS * 2 * 64 * 16;
}

extern "C" int rfn_smallmap_pack_bf16x3(const float* w, int Cout, int Cin, int H, int W, int transpose, float* packed,
                                        rfn_stream_t stream) {
    RFN_CHECK_ARG(w && packed && Cout > 0 && Cin > 0 && H > 0 && W > 0 && H * W <= 16, -1);
    RFN_CHECK_ARG(((uintptr_t)packed & 15) == 0, -2);
    int K, N, KS, NT;
    smallmap_dims(Cout, Cin, H * W, transpose, &K, &N, &KS, &NT);
    const long total = (long)NT * KS * 64;
    const int grid = (int)((total + 255) / 256 < 4096 ? (total + 255) / 256 : 4096);
    hipLaunchKernelGGL(smallmap_pack_kernel, dim3(grid), dim3(256), 0, (hipStream_t)stream, w, Cout, Cin, H, W, transpose,
                       reinterpret_cast<bf16x8*>(packed), KS, NT);
    RFN_LAUNCH_CHECK();
    return 0;
}

/* n matrices (host array of descriptors) in ceil(n / 64) launches */
extern "C" int rfn_smallmap_pack_batched_bf16x3(const void* descs_host, int n, rfn_stream_t stream) {
    RFN_CHECK_ARG(descs_host && n >= 0, -1);
    const SmallmapPackDesc* d = reinterpret_cast<const SmallmapPackDesc*>(descs_host);
    for (int i = 0; i < n; ++i) {
        RFN_CHECK_ARG(d[i].w && d[i].packed && d[i].Cout > 0 && d[i].Cin > 0 && d[i].H > 0 && d[i].W > 0 &&
                      d[i].H * d[i].W <= 16 && ((uintptr_t)d[i].packed & 15) == 0, -2);
    }
    for (int i0 = 0; i0 < n; i0 += SM_PACK_MAX) {
        const int m = n - i0 < SM_PACK_MAX ? n - i0 : SM_PACK_MAX;
        SmallmapPackTable t;
        memset(&t, 0, sizeof(t));
        long most = 0;
        for (int i = 0; i < m; ++i) {
            t.d[i] = d[i0 + i];
            int K, N, KS, NT;
            smallmap_dims(t.d[i].Cout, t.d[i].Cin, t.d[i].H * t.d[i].W, t.d[i].transpose, &K, &N, &KS, &NT);
            const long total = (long)NT * KS * 64;
            most = total > most ? total : most;
        }
        const int gx = (int)((most + 255) / 256 < 512 ? (most + 255) / 256 : 512);
        hipLaunchKernelGGL(smallmap_pack_batched_kernel, dim3(gx, m), dim3(256), 0, (hipStream_t)stream, t);
    }
    RFN_LAUNCH_CHECK();
    return 0;
}

// ------------------------------------------------------------------------------------------------ dense product
struct SmallmapParams {
    const float* a;       // [B][K]  (K = C*HW, one sample per row)
    const float* y;       // optional [B][K]: a is scaled by (y > 0 ? 1 : slope_in) -- the backward of an in-place leaky_relu
    float slope_in;
    const bf16x8* packed;
    const float* bias;    // optional, per output channel (n / HW)
    const float* add;     // optional [B][N] addend (e.g. the time-batched input projection of a recurrent layer)
    int act_out;          // 1: leaky_relu(slope_out) on the output
    float slope_out;
    float* out;           // [B][N]
    float* a_out;         // optional [B][K]: the scaled a (pre-activation gradient, kept for the weight gradient)
    int B, K, N, HW, KS;
    // convolution form (rfn_smallmap_conv_bf16x3): rows are frames of one or two NCHW tensors with frame strides, the
    // epilogue is the convolution kernels' (ep_mode 1: (v+p0)*exp(p1) then act; 2: (v+p0)*exp(3 p1); 3: v+p0), the output
    // channels may be split over two tensors and out1 may accumulate
    long a_ns, a2_ns, out_ns, out2_ns;  // row (frame) strides in floats
    const float* a2;                    // second source: columns K1 .. K-1
    int K1;
    int ep_mode, act, nsplit, acc1;     // nsplit = columns that go to out (the rest to out2)
    const float* p0;
    const float* p1;
    float* out2;
};

constexpr int SM_WAVES = 8;

__device__ __forceinline__ void smallmap_dense_body(const SmallmapParams& p, float (&red)[SM_WAVES][32][33]) {
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int l31 = lane & 31, kg = lane >> 5;
    const int tile = blockIdx.x, m0 = blockIdx.y * 32;
    const int m = m0 + l31;
    const bool mok = m < p.B;
    const float* arow = p.a + (long)(mok ? m : 0) * p.a_ns;
    const float* a2row = p.a2 ? p.a2 + (long)(mok ? m : 0) * p.a2_ns : arow;
    const float* yrow = p.y ? p.y + (long)(mok ? m : 0) * p.K : nullptr;
    float* aorow = (p.a_out && tile == 0 && mok) ? p.a_out + (long)m * p.K : nullptr;
    const bf16x8* wp = p.packed + (long)tile * p.KS * 2 * 64 + lane;

    f32x16 acc;
#pragma unroll
    for (int r = 0; r < 16; ++r) acc[r] = 0.f;

    for (int ks = wave; ks < p.KS; ks += SM_WAVES) {
        const int k = ks * 16 + kg * 8;
        const bool kok = mok && k < p.K;  // K % 8 == 0 (checked on the host): a group is all in or all out
        const int kc = kok ? k : 0;
        const float* ap = kc < p.K1 ? arow + kc : a2row + (kc - p.K1);  // K1 % 8 == 0: a group never straddles
        float4 v0 = *reinterpret_cast<const float4*>(ap);
        float4 v1 = *reinterpret_cast<const float4*>(ap + 4);
        const bf16x8 bh = wp[(long)(ks * 2 + 0) * 64];
        const bf16x8 bl = wp[(long)(ks * 2 + 1) * 64];
        float v[8] = {v0.x, v0.y, v0.z, v0.w, v1.x, v1.y, v1.z, v1.w};
        if (yrow) {
            const float* yp = yrow + (kok ? k : 0);
            const float4 y0 = *reinterpret_cast<const float4*>(yp);
            const float4 y1 = *reinterpret_cast<const float4*>(yp + 4);
            const float yy[8] = {y0.x, y0.y, y0.z, y0.w, y1.x, y1.y, y1.z, y1.w};
#pragma unroll
            for (int j = 0; j < 8; ++j) v[j] *= yy[j] > 0.f ? 1.f : p.slope_in;
        }
        if (aorow && kok) {
            *reinterpret_cast<float4*>(aorow + k) = float4{v[0], v[1], v[2], v[3]};
            *reinterpret_cast<float4*>(aorow + k + 4) = float4{v[4], v[5], v[6], v[7]};
        }
        bf16x8 ah, al;
#pragma unroll
        for (int j = 0; j < 8; ++j) {
            const float x = kok ? v[j] : 0.f;
            const __bf16 h = (__bf16)x;
            ah[j] = h;
            al[j] = (__bf16)(x - (float)h);
        }
        acc = __builtin_amdgcn_mfma_f32_32x32x16_bf16(al, bh, acc, 0, 0, 0);
        acc = __builtin_amdgcn_mfma_f32_32x32x16_bf16(ah, bl, acc, 0, 0, 0);
        acc = __builtin_amdgcn_mfma_f32_32x32x16_bf16(ah, bh, acc, 0, 0, 0);
    }
    // D[row = batch][col = n]: register r of lane l holds row (r&3) + 8*(r>>2) + 4*(l/32), col l%32
#pragma unroll
    for (int r = 0; r < 16; ++r) red[wave][(r & 3) + 8 * (r >> 2) + 4 * kg][l31] = acc[r];
    __syncthreads();
    for (int idx = tid; idx < 32 * 32; idx += 64 * SM_WAVES) {
        const int mm = idx >> 5, nn = idx & 31;
        float s = 0.f;
#pragma unroll
        for (int w = 0; w < SM_WAVES; ++w) s += red[w][mm][nn];
        const int n = tile * 32 + nn, row = m0 + mm;
        if (row < p.B && n < p.N) {
            const int c = n / p.HW;
            if (p.bias) s += p.bias[c];
            if (p.add) s += p.add[(long)row * p.N + n];
            if (p.act_out) s = s > 0.f ? s : s * p.slope_out;
            if (p.ep_mode != 0) {
                s += p.p0[c];
                if (p.ep_mode == 1) s *= expf(p.p1[c]);
                if (p.ep_mode == 2) s *= expf(3.f * p.p1[c]);
                if (p.ep_mode == 1) {
                    if (p.act == 1) s = s > 0.f ? s : 0.f;
                    if (p.act == 2) s = s > 0.f ? s : 0.2f * s;
                }
            }
            if (n < p.nsplit) {
                float* dst = p.out + (long)row * p.out_ns + n;
                if (p.acc1 & 1) s += *dst;
                *dst = s;
            } else {
                float* dst = p.out2 + (long)row * p.out2_ns + (n - p.nsplit);
                if (p.acc1 & 2) s += *dst;
                *dst = s;
            }
        }
    }
}

__global__ __launch_bounds__(64 * SM_WAVES) void smallmap_dense_kernel(const SmallmapParams p) {
    __shared__ float red[SM_WAVES][32][33];
    smallmap_dense_body(p, red);
}

// Two independent products in one launch (blockIdx.z): the encoder and the prior layer of a timestep have no data
// dependence on each other, and a launch of this kernel is mostly latency.
struct SmallmapPair {
    SmallmapParams g[2];
};
__global__ __launch_bounds__(64 * SM_WAVES) void smallmap_dense_pair_kernel(const SmallmapPair pp) {
    __shared__ float red[SM_WAVES][32][33];
    const SmallmapParams& p = pp.g[blockIdx.z];
    if ((int)blockIdx.x * 32 >= p.N || (int)blockIdx.y * 32 >= p.B) return;  // block-uniform: grid covers the larger one
    smallmap_dense_body(p, red);
}

extern "C" int rfn_smallmap_dense_bf16x3(const float* a, const float* y, float slope_in, const float* packed,
                                         const float* bias, const float* add, int act_out, float slope_out, float* out,
                                         float* a_out, int B, int K, int N, int HW, rfn_stream_t stream) {
    RFN_CHECK_ARG(a && packed && out && B >= 0 && K > 0 && N > 0 && HW > 0 && HW <= 16, -1);
    RFN_CHECK_ARG(K % 8 == 0 && K % HW == 0 && N % HW == 0, -2);
    RFN_CHECK_ARG((((uintptr_t)a | (uintptr_t)packed | (uintptr_t)(y ? y : a) | (uintptr_t)(a_out ? a_out : a)) & 15) == 0, -3);
    if (B == 0) return 0;
    SmallmapParams p;
    memset(&p, 0, sizeof(p));
    p.a = a; p.y = y; p.slope_in = slope_in; p.packed = reinterpret_cast<const bf16x8*>(packed); p.bias = bias; p.add = add;
    p.act_out = act_out; p.slope_out = slope_out; p.out = out; p.a_out = a_out;
    p.B = B; p.K = K; p.N = N; p.HW = HW; p.KS = (K + 15) / 16;
    p.a_ns = K; p.K1 = K; p.out_ns = N; p.nsplit = N;
    dim3 grid((N + 31) / 32, (B + 31) / 32);
    hipLaunchKernelGGL(smallmap_dense_kernel, grid, dim3(64 * SM_WAVES), 0, (hipStream_t)stream, p);
    RFN_LAUNCH_CHECK();
    return 0;
}

static int smallmap_fill(SmallmapParams& p, const float* a, const float* y, float slope_in, const float* packed,
                         const float* bias, const float* add, int act_out, float slope_out, float* out, float* a_out, int B,
                         int K, int N, int HW) {
    if (!(a && packed && out && B > 0 && K > 0 && N > 0 && HW > 0 && HW <= 16)) return -1;
    if (!(K % 8 == 0 && K % HW == 0 && N % HW == 0)) return -2;
    if ((((uintptr_t)a | (uintptr_t)packed | (uintptr_t)(y ? y : a) | (uintptr_t)(a_out ? a_out : a)) & 15) != 0) return -3;
    memset(&p, 0, sizeof(p));
    p.a = a; p.y = y; p.slope_in = slope_in; p.packed = reinterpret_cast<const bf16x8*>(packed); p.bias = bias; p.add = add;
    p.act_out = act_out; p.slope_out = slope_out; p.out = out; p.a_out = a_out;
    p.B = B; p.K = K; p.N = N; p.HW = HW; p.KS = (K + 15) / 16;
    p.a_ns = K; p.K1 = K; p.out_ns = N; p.nsplit = N;
    return 0;
}

// rfn_smallmap_dense_bf16x3 twice in one launch (suffix 0 / 1): two products without a data dependence, same B and HW.
extern "C" int rfn_smallmap_dense_pair_bf16x3(const float* a0, const float* y0, float slope_in0, const float* packed0,
                                              const float* bias0, const float* add0, int act_out0, float slope_out0,
                                              float* out0, float* a_out0, int K0, int N0, const float* a1, const float* y1,
                                              float slope_in1, const float* packed1, const float* bias1, const float* add1,
                                              int act_out1, float slope_out1, float* out1, float* a_out1, int K1, int N1,
                                              int B, int HW, rfn_stream_t stream) {
    if (B == 0) return 0;
    SmallmapPair pp;
    int rc = smallmap_fill(pp.g[0], a0, y0, slope_in0, packed0, bias0, add0, act_out0, slope_out0, out0, a_out0, B, K0,
                           N0, HW);
    if (!rc) rc = smallmap_fill(pp.g[1], a1, y1, slope_in1, packed1, bias1, add1, act_out1, slope_out1, out1, a_out1, B,
                                K1, N1, HW);
    if (rc) {
        rfn_set_error("rfn_smallmap_dense_pair_bf16x3: argument check failed (%d)", rc);
        return rc;
    }
    const int Nmax = N0 > N1 ? N0 : N1;
    dim3 grid((Nmax + 31) / 32, (B + 31) / 32, 2);
    hipLaunchKernelGGL(smallmap_dense_pair_kernel, grid, dim3(64 * SM_WAVES), 0, (hipStream_t)stream, pp);
    RFN_LAUNCH_CHECK();
    return 0;
}

// The same product behind the convolution entry point's interface (rfn_conv2d_fwd_bf16x3 without ks): 3x3 / pad 1 on an
// H x W <= 16 map of N frames, two-source input, the conv epilogues 0-3, output channels split at cout_split, out1
// optionally accumulated.  `packed` from rfn_smallmap_pack_bf16x3(w[Cout][C1+C2][3][3], transpose 0) -- or transpose 1
// of the forward weight for a data gradient, in which case (C1+C2) is the forward Cout and Cout the forward Cin.
extern "C" int rfn_smallmap_conv_bf16x3(const float* in1, long in1_ns, int C1, const float* in2, long in2_ns, int C2,
                                        const float* packed, float* out1, long out1_ns, float* out2, long out2_ns,
                                        int Cout, int cout_split, int acc1, int N, int H, int W, int ep_mode,
                                        const float* p0, const float* p1, int act, rfn_stream_t stream) {
    const int HW = H * W;
    RFN_CHECK_ARG(in1 && packed && out1 && C1 > 0 && C2 >= 0 && (C2 == 0 || in2) && Cout > 0 && N >= 0 && HW > 0 && HW <= 16, -1);
    RFN_CHECK_ARG(cout_split > 0 && cout_split <= Cout && (cout_split == Cout || out2), -2);
    RFN_CHECK_ARG(ep_mode >= 0 && ep_mode <= 3 && (ep_mode == 0 || p0) && ((ep_mode != 1 && ep_mode != 2) || p1), -3);
    RFN_CHECK_ARG((C1 * HW) % 8 == 0 && ((C1 + C2) * HW) % 8 == 0 && in1_ns % 4 == 0 && (C2 == 0 || in2_ns % 4 == 0), -4);
    RFN_CHECK_ARG((((uintptr_t)in1 | (uintptr_t)packed | (uintptr_t)(C2 ? in2 : in1)) & 15) == 0, -5);
    if (N == 0) return 0;
    SmallmapParams p;
    memset(&p, 0, sizeof(p));
    p.a = in1; p.a_ns = in1_ns; p.a2 = C2 ? in2 : nullptr; p.a2_ns = in2_ns; p.K1 = C1 * HW;
    p.packed = reinterpret_cast<const bf16x8*>(packed);
    p.out = out1; p.out_ns = out1_ns; p.out2 = out2; p.out2_ns = out2_ns; p.nsplit = cout_split * HW; p.acc1 = acc1;
    p.ep_mode = ep_mode; p.act = act; p.p0 = p0; p.p1 = p1;
    p.B = N; p.K = (C1 + C2) * HW; p.N = Cout * HW; p.HW = HW; p.KS = (p.K + 15) / 16;
    dim3 grid((p.N + 31) / 32, (N + 31) / 32);
    hipLaunchKernelGGL(smallmap_dense_kernel, grid, dim3(64 * SM_WAVES), 0, (hipStream_t)stream, p);
    RFN_LAUNCH_CHECK();
    return 0;
}
