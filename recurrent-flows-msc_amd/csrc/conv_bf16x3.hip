// Split-precision ("bf16x3") implicit-GEMM convolution for gfx950.
//
// fp32 MFMA (v_mfma_f32_32x32x2_f32) runs at 1/16 of the bf16 MFMA rate.  Every fp32 operand is split into two bf16
// numbers  x = hi + lo  (hi = RNE_bf16(x), lo = RNE_bf16(x - hi), 16 significant bits together) and the product is
// formed as  a*b ≈ a_hi*b_hi + a_hi*b_lo + a_lo*b_hi  with three v_mfma_f32_32x32x16_bf16 accumulating in fp32
// (the dropped a_lo*b_lo term is ≤ 2^-16 relative).  Measured on the oracle (tools/ notes in DESIGN.md): the nll /
// bits-per-dim error against an fp64 reference is ~1e-5 relative — ten times the plain fp32 error and a tenth of the
// 1e-4 budget — for 3/16 of the fp32-MFMA cycles.
//
// Same GEMM orientation, pixel tiling, halo staging, descriptors, register prefetch, split-K and epilogue as conv.hip
// (D[cout][pixel], C/D register layout is dtype independent).  What differs is the LDS image: an MFMA operand lane
// needs 8 consecutive k (channels) of one row/column in one 16-byte register quad, so activations are stored as
// [plane hi|lo][8-channel group][slot][8 x bf16] and weights (pre-split on the device by the pack kernel) as
// [iteration][plane][k-group][cout][8 x bf16]; both fragments are single ds_read_b128.
#include "conv_common.h"

typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));
typedef unsigned int u32x4 __attribute__((ext_vector_type(4)));

static inline void packed_dims_b3(int Cout_l, int Cin_l, int* CoutP, int* Cin16) {
    *CoutP = ((Cout_l + 255) / 256) * 256;
    *Cin16 = (Cin_l + 15) / 16;
}

// packed unit (16 bytes = 8 bf16) index: (((c16*T + tap)*2 + plane)*2 + g)*CoutP + co ; element j <-> cin = c16*16+g*8+j
// NPL = 2: (hi, lo) -- bf16x3;  NPL = 3: (hi, mid, lo), 24 significant bits -- "bf16x6" (six products per fp32 product)
__global__ void pack_weight_b3_kernel(const float* __restrict__ w, bf16x8* __restrict__ wpk, int Cout, int Cin, int KS,
                                      int CoutP, int Cin16, int transpose_flip, int NPL) {
    const int T = KS * KS;
    const int Co_l = transpose_flip ? Cin : Cout;
    const int Ci_l = transpose_flip ? Cout : Cin;
    const long total = (long)Cin16 * T * 2 * CoutP;  // (hi, lo) pairs of units
    for (long idx = (long)blockIdx.x * blockDim.x + threadIdx.x; idx < total; idx += (long)gridDim.x * blockDim.x) {
        const int co = (int)(idx % CoutP);
        long r = idx / CoutP;
        const int g = (int)(r & 1);
        r >>= 1;
        const int tap = (int)(r % T);
        const int c16 = (int)(r / T);
        bf16x8 hi, lo, l3;
#pragma unroll
        for (int j = 0; j < 8; ++j) {
            const int ci = c16 * 16 + g * 8 + j;
            float v = 0.f;
            if (co < Co_l && ci < Ci_l)
                v = transpose_flip ? w[((long)ci * Cin + co) * T + (T - 1 - tap)] : w[((long)co * Cin + ci) * T + tap];
            const __bf16 h = (__bf16)v;
            const float r1 = v - (float)h;
            const __bf16 m = (__bf16)r1;
            hi[j] = h;
            lo[j] = m;
            l3[j] = (__bf16)(r1 - (float)m);
        }
        const long base = ((long)(c16 * T + tap) * NPL) * 2 * CoutP;
        wpk[base + (long)(0 * 2 + g) * CoutP + co] = hi;
        wpk[base + (long)(1 * 2 + g) * CoutP + co] = lo;
        if (NPL == 3) wpk[base + (long)(2 * 2 + g) * CoutP + co] = l3;
    }
}

extern "C" long rfn_packed_weight_size_bf16x3(int Cout, int Cin, int ks) {
    // in FLOATS (4 bytes), large enough for both orientations
    int a, b, c, d;
    packed_dims_b3(Cout, Cin, &a, &b);
    packed_dims_b3(Cin, Cout, &c, &d);
    long s0 = (long)b * ks * ks * 4 * a * 4, s1 = (long)d * ks * ks * 4 * c * 4;  // units * 16 B / 4 B
    return s0 > s1 ? s0 : s1;
}
extern "C" long rfn_packed_weight_size_bf16x6(int Cout, int Cin, int ks) {
    return rfn_packed_weight_size_bf16x3(Cout, Cin, ks) / 2 * 3;  // three planes instead of two
}

static int pack_conv_weight_planes(const float* w, float* wpk, int Cout, int Cin, int ks, int transpose_flip, int NPL,
                                   rfn_stream_t stream);
extern "C" int rfn_pack_conv_weight_bf16x3(const float* w, float* wpk, int Cout, int Cin, int ks, int transpose_flip,
                                           rfn_stream_t stream) {
    return pack_conv_weight_planes(w, wpk, Cout, Cin, ks, transpose_flip, 2, stream);
}
extern "C" int rfn_pack_conv_weight_bf16x6(const float* w, float* wpk, int Cout, int Cin, int ks, int transpose_flip,
                                           rfn_stream_t stream) {
    return pack_conv_weight_planes(w, wpk, Cout, Cin, ks, transpose_flip, 3, stream);
}
static int pack_conv_weight_planes(const float* w, float* wpk, int Cout, int Cin, int ks, int transpose_flip, int NPL,
                                   rfn_stream_t stream) {
    RFN_CHECK_ARG(w && wpk && Cout > 0 && Cin > 0 && (ks == 1 || ks == 3), -1);
    RFN_CHECK_ARG(((uintptr_t)wpk & 15) == 0, -2);
    int CoutP, Cin16;
    if (!transpose_flip)
        packed_dims_b3(Cout, Cin, &CoutP, &Cin16);
    else
        packed_dims_b3(Cin, Cout, &CoutP, &Cin16);
    long total = (long)Cin16 * ks * ks * 2 * CoutP;
    int grid = (int)((total + 255) / 256 < 2048 ? (total + 255) / 256 : 2048);
    hipLaunchKernelGGL(pack_weight_b3_kernel, dim3(grid), dim3(256), 0, (hipStream_t)stream, w,
                       reinterpret_cast<bf16x8*>(wpk), Cout, Cin, ks, CoutP, Cin16, transpose_flip, NPL);
    RFN_LAUNCH_CHECK();
    return 0;
}

// ---- all weights of a model in ONE launch: descriptor table in device memory, blockIdx.y = descriptor.
// mode 0: forward conv, 1: data-gradient conv (transposed + mirrored taps), 2: tap-expanded 1x1 form of a 3x3 conv with
// tiny Cout (logical weight w'[tap*Cout + co][ci] = w[co][ci][tap], see rfn_tap_gather_f32).
struct PackDesc {  // mirrors rfn_pack_desc in include/rfn_hip.h
    const float* w;
    float* wpk;
    int Cout, Cin, ks, mode;
};
__device__ __forceinline__ void pack_weights_one(const PackDesc d);
__global__ void pack_weights_batched_b3_kernel(const PackDesc* __restrict__ descs) { pack_weights_one(descs[blockIdx.y]); }
// the same with the descriptors by value in the kernel arguments (packs queued by the host between two launches)
#define PACK_TABLE_MAX 64
struct PackTable { PackDesc d[PACK_TABLE_MAX]; };
__global__ void pack_weights_table_b3_kernel(const PackTable t) { pack_weights_one(t.d[blockIdx.y]); }
__device__ __forceinline__ void pack_weights_one(const PackDesc d) {
    const int T_src = d.ks * d.ks;
    const int NPL = (d.mode & 4) ? 3 : 2, mode = d.mode & 3;  // (mode + 4: three planes = bf16x6)
    int Co_l, Ci_l, T;
    if (mode == 0) { Co_l = d.Cout; Ci_l = d.Cin; T = T_src; }
    else if (mode == 1) { Co_l = d.Cin; Ci_l = d.Cout; T = T_src; }
    else { Co_l = T_src * d.Cout; Ci_l = d.Cin; T = 1; }
    const int CoutP = ((Co_l + 255) / 256) * 256, Cin16 = (Ci_l + 15) / 16;
    bf16x8* wpk = reinterpret_cast<bf16x8*>(d.wpk);
    const long total = (long)Cin16 * T * 2 * CoutP;
    for (long idx = (long)blockIdx.x * blockDim.x + threadIdx.x; idx < total; idx += (long)gridDim.x * blockDim.x) {
        const int co = (int)(idx % CoutP);
        long r = idx / CoutP;
        const int g = (int)(r & 1);
        r >>= 1;
        const int tap = (int)(r % T);
        const int c16 = (int)(r / T);
        bf16x8 hi, lo, l3;
#pragma unroll
        for (int j = 0; j < 8; ++j) {
            const int ci = c16 * 16 + g * 8 + j;
            float v = 0.f;
            if (co < Co_l && ci < Ci_l) {
                if (mode == 0)
                    v = d.w[((long)co * d.Cin + ci) * T_src + tap];
                else if (mode == 1)
                    v = d.w[((long)ci * d.Cin + co) * T_src + (T_src - 1 - tap)];
                else
                    v = d.w[((long)(co % d.Cout) * d.Cin + ci) * T_src + co / d.Cout];
            }
            const __bf16 h = (__bf16)v;
            const float r1 = v - (float)h;
            const __bf16 m = (__bf16)r1;
            hi[j] = h;
            lo[j] = m;
            l3[j] = (__bf16)(r1 - (float)m);
        }
        const long base = ((long)(c16 * T + tap) * NPL) * 2 * CoutP;
        wpk[base + (long)(0 * 2 + g) * CoutP + co] = hi;
        wpk[base + (long)(1 * 2 + g) * CoutP + co] = lo;
        if (NPL == 3) wpk[base + (long)(2 * 2 + g) * CoutP + co] = l3;
    }
}
extern "C" int rfn_pack_conv_weights_batched_bf16x3(const void* descs_device, int n, rfn_stream_t stream) {
    RFN_CHECK_ARG(descs_device && n >= 0, -1);
    if (n == 0) return 0;
    hipLaunchKernelGGL(pack_weights_batched_b3_kernel, dim3(96, n), dim3(256), 0, (hipStream_t)stream,
                       reinterpret_cast<const PackDesc*>(descs_device));
    RFN_LAUNCH_CHECK();
    return 0;
}
/* n descriptors in HOST memory: ceil(n / 64) launches, the table travels in the kernel arguments */
extern "C" int rfn_pack_conv_weights_hostdescs_bf16x3(const void* descs_host, int n, rfn_stream_t stream) {
    RFN_CHECK_ARG(descs_host && n >= 0, -1);
    const PackDesc* d = reinterpret_cast<const PackDesc*>(descs_host);
    for (int i = 0; i < n; ++i)
        RFN_CHECK_ARG(d[i].w && d[i].wpk && d[i].Cout > 0 && d[i].Cin > 0 && (d[i].ks == 1 || d[i].ks == 3) &&
                      ((uintptr_t)d[i].wpk & 15) == 0, -2);
    for (int i0 = 0; i0 < n; i0 += PACK_TABLE_MAX) {
        const int m = n - i0 < PACK_TABLE_MAX ? n - i0 : PACK_TABLE_MAX;
        PackTable t;
        memset(&t, 0, sizeof(t));
        long most = 0;   // blocks per matrix follow the LARGEST one of the launch (a 2 M-element weight on 24 blocks is 0.1 ms)
        for (int i = 0; i < m; ++i) {
            t.d[i] = d[i0 + i];
            const int T_src = t.d[i].ks * t.d[i].ks, mode = t.d[i].mode & 3;
            const int Co_l = mode == 0 ? t.d[i].Cout : (mode == 1 ? t.d[i].Cin : T_src * t.d[i].Cout);
            const int Ci_l = mode == 1 ? t.d[i].Cout : t.d[i].Cin, T = mode == 2 ? 1 : T_src;
            const long total = (long)((Ci_l + 15) / 16) * T * 2 * (((Co_l + 255) / 256) * 256);
            most = total > most ? total : most;
        }
        int gx = (int)((most + 255) / 256);
        gx = gx < 24 ? 24 : (gx > 1024 ? 1024 : gx);
        while ((long)gx * m > 16384) gx = (gx + 1) / 2;   // (many matrices: the launch stays a few thousand blocks)
        hipLaunchKernelGGL(pack_weights_table_b3_kernel, dim3(gx, m), dim3(256), 0, (hipStream_t)stream, t);
    }
    RFN_LAUNCH_CHECK();
    return 0;
}

// NPL = 2: two bf16 pieces per operand, three products (bf16x3); NPL = 3: three pieces, six products ("bf16x6", 24
// significant bits: the fp32-grade arithmetic of the forward convolutions at the levels the fused kernel does not take)
template <int KS, int WCO, int WPX, int TCO, int TPX, int KC, int NPL = 2>
__global__ __launch_bounds__(256) void conv_b3_kernel(const ConvParams p) {
    constexpr int T = KS * KS, PAD = KS / 2;
    constexpr int BCO = 32 * TCO * WCO, BPX = 32 * TPX * WPX;
    constexpr int NG = KC / 8;    // 8-channel groups per chunk
    constexpr int NS = KC / 16;   // k16 steps per chunk and tap
    static_assert(WCO * WPX == 4 && KC % 16 == 0, "4 waves, whole k16 steps");
    extern __shared__ __attribute__((aligned(16))) unsigned char lds_raw[];
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int wco = wave / WPX, wpx = wave % WPX;
    const int l31 = lane & 31, kk = lane >> 5;
    const int HW = p.H * p.W;
    const int Cin = p.C1 + p.C2;

    int pt = blockIdx.x;
    const int wt = pt % p.n_wtiles;
    pt /= p.n_wtiles;
    const int ht = pt % p.n_htiles;
    const int ft = pt / p.n_htiles;
    const int x0 = wt * p.TWp, y0 = ht * p.TH, f0 = ft * p.TF;
    const int RW = p.TWp + 2 * PAD, RH = p.TH + 2 * PAD;
    const int FRM = RH * RW, IMG = p.TF * FRM;
    const int co_base = blockIdx.y * BCO + wco * (32 * TCO);

    // LDS: Bs[plane][NG][IMG] units | Ws[NS*T][plane][2][BCO] units | ep[2][BCO] floats
    bf16x8* Bs = reinterpret_cast<bf16x8*>(lds_raw);
    bf16x8* Ws = Bs + NPL * NG * IMG;
    float* ep = reinterpret_cast<float*>(Ws + NS * T * NPL * 2 * BCO);

    int lds_off[TPX], pn[TPX], ppix[TPX];
    bool pvalid[TPX];
#pragma unroll
    for (int t = 0; t < TPX; ++t) {
        const int q = (wpx * TPX + t) * 32 + l31;
        const int col = q & (p.TWp - 1);
        const int row = (q >> p.tw_shift) & (p.TH - 1);
        const int f = q >> (p.tw_shift + p.th_shift);
        lds_off[t] = f * FRM + row * RW + col;
        pvalid[t] = (x0 + col < p.W) && (y0 + row < p.H) && (f0 + f < p.N);
        pn[t] = f0 + f;
        ppix[t] = (y0 + row) * p.W + x0 + col;
    }

    f32x16 acc[TCO][TPX];
#pragma unroll
    for (int a = 0; a < TCO; ++a)
#pragma unroll
        for (int t = 0; t < TPX; ++t)
#pragma unroll
            for (int r = 0; r < 16; ++r) acc[a][t][r] = 0.f;
    constexpr bool SINGLE = TCO * TPX == 1;
    f32x16 accx[2];
    if (SINGLE) {
#pragma unroll
        for (int r = 0; r < 16; ++r) accx[0][r] = accx[1][r] = 0.f;
    }

    // ---- activation staging descriptors (as in conv.hip): thread -> image slot(s), 8-channel group phase
    constexpr int NPOS = (KS == 1) ? 1 : (BPX >= 256 ? 4 : (BPX >= 128 ? 2 : 1));
    constexpr int GROUPS = (KS == 1 && BPX < 256) ? (BPX >= 64 ? 256 / BPX : 4) : 1;
    constexpr int P2 = 256 / GROUPS;
    constexpr int NGT = (NG + GROUPS - 1) / GROUPS;  // channel groups per thread
    const int slot = tid & (P2 - 1);
    const int phase = __builtin_amdgcn_readfirstlane(tid / P2);
    int soff1[NPOS], soff2[NPOS];
    bool sok[NPOS], sin[NPOS];
#pragma unroll
    for (int j = 0; j < NPOS; ++j) {
        const int r = slot + P2 * j;
        sin[j] = r < IMG;
        const int f = r / FRM;
        const int rr = r - f * FRM;
        const int yy = rr / RW;
        const int xx = rr - yy * RW;
        const int gy = y0 + yy - PAD, gx = x0 + xx - PAD;
        sok[j] = sin[j] && (f0 + f < p.N) && gy >= 0 && gy < p.H && gx >= 0 && gx < p.W;
        soff1[j] = sok[j] ? (int)(f * p.in1_ns) + gy * p.W + gx : 0;
        soff2[j] = sok[j] ? (int)(f * p.in2_ns) + gy * p.W + gx : 0;
    }
    const float* in1b = p.in1 + (long)f0 * p.in1_ns;
    const float* in2b = p.in2 ? p.in2 + (long)f0 * p.in2_ns : p.in1;
    float stg[NGT][NPOS][8];
    auto prefetch = [&](int chunk) {
#pragma unroll
        for (int i = 0; i < NGT; ++i) {
            const int gi = phase + GROUPS * i;  // scalar: 8-channel group inside the chunk
#pragma unroll
            for (int c = 0; c < 8; ++c) {
                const int ch = chunk * KC + gi * 8 + c;
                const bool chv = gi < NG && ch < Cin;
                const int chc = chv ? ch : 0;
                const bool first = chc < p.C1;
                const float* src = first ? in1b + (long)chc * HW : in2b + (long)(chc - p.C1) * HW;
#pragma unroll
                for (int j = 0; j < NPOS; ++j) {
                    const float v = src[first ? soff1[j] : soff2[j]];
                    stg[i][j][c] = (sok[j] && chv) ? v : 0.f;
                }
            }
        }
    };
    auto commit = [&]() {  // split to (hi, lo) and store the two 16-byte units
#pragma unroll
        for (int i = 0; i < NGT; ++i) {
            const int gi = phase + GROUPS * i;
            if (gi < NG) {
#pragma unroll
                for (int j = 0; j < NPOS; ++j) {
                    if (sin[j]) {
                        bf16x8 hi, lo, l3;
#pragma unroll
                        for (int c = 0; c < 8; ++c) {
                            const float v = stg[i][j][c];
                            const __bf16 h = (__bf16)v;
                            const float r1 = v - (float)h;
                            const __bf16 m = (__bf16)r1;
                            hi[c] = h;
                            lo[c] = m;
                            if (NPL == 3) l3[c] = (__bf16)(r1 - (float)m);
                        }
                        const int r = slot + P2 * j;
                        Bs[(0 * NG + gi) * IMG + r] = hi;
                        Bs[(1 * NG + gi) * IMG + r] = lo;
                        if (NPL == 3) Bs[(2 * NG + gi) * IMG + r] = l3;
                    }
                }
            }
        }
    };

    // ---- weight tile of a chunk: verbatim copy of NS*T*4 runs of BCO units from the packed buffer
    const int total_it = p.Cin8 * T;  // Cin8 holds Cin16 for this kernel
    const u32x4* wp4 = reinterpret_cast<const u32x4*>(p.wpk);
    constexpr int WRUNS = NS * T * NPL * 2;
    constexpr int WPT = (WRUNS * BCO + 255) / 256;
    u32x4* Ws4 = reinterpret_cast<u32x4*>(Ws);
    const long wblk = (long)blockIdx.y * BCO;
    u32x4 wstg[WPT];
    auto wprefetch = [&](int chunk) {
        const int it0 = chunk * NS * T;
#pragma unroll
        for (int j = 0; j < WPT; ++j) {
            const int e = tid + 256 * j;
            const int run = e / BCO, col = e % BCO;
            const int itg = it0 + run / (NPL * 2);
            const bool ok = (WRUNS * BCO % 256 == 0 || e < WRUNS * BCO) && itg < total_it;
            const u32x4 v = wp4[((long)(ok ? itg : 0) * (NPL * 2) + run % (NPL * 2)) * p.CoutP + wblk + col];
            wstg[j] = ok ? v : u32x4{0u, 0u, 0u, 0u};
        }
    };
    auto wcommit = [&]() {
#pragma unroll
        for (int j = 0; j < WPT; ++j) {
            const int e = tid + 256 * j;
            if (WRUNS * BCO % 256 == 0 || e < WRUNS * BCO) Ws4[e] = wstg[j];
        }
    };

    if (p.ep_mode != 0) {
        for (int c = tid; c < BCO; c += 256) {
            const int co = blockIdx.y * BCO + c;
            float e0 = 0.f, e1 = 1.f;
            if (co < p.Cout) {
                if (p.ep_mode != 4) e0 = p.p0[co];
                if (p.ep_mode == 1 || p.ep_mode == 4) e1 = expf(p.p1[co]);
                if (p.ep_mode == 2) e1 = expf(3.f * p.p1[co]);
            }
            ep[c] = e0;
            ep[BCO + c] = e1;
        }
    }

    const int nchunks_all = (p.Cin8 * 16 + KC - 1) / KC;
    const int cps = (nchunks_all + p.ksplit - 1) / p.ksplit;
    const int chunk0 = blockIdx.z * cps;
    const int nchunks = chunk0 + cps < nchunks_all ? chunk0 + cps : nchunks_all;
    if (chunk0 < nchunks) {
        prefetch(chunk0);
        wprefetch(chunk0);
    }
    const bf16x8* wa = Ws + kk * BCO + wco * (32 * TCO) + l31;  // + ((itl*2 + plane)*2)*BCO + a*32
    for (int chunk = chunk0; chunk < nchunks; ++chunk) {
        __syncthreads();
        commit();
        wcommit();
        __syncthreads();
        if (chunk + 1 < nchunks) {
            prefetch(chunk + 1);
            wprefetch(chunk + 1);
        }
#pragma unroll
        for (int s = 0; s < NS; ++s) {
#pragma unroll
            for (int tap = 0; tap < T; ++tap) {
                const int itl = s * T + tap;
                const int tapoff = (tap / KS) * RW + (tap % KS);
                bf16x8 ah[TCO], al[TCO], a3[TCO], bh[TPX], bl[TPX], b3[TPX];
#pragma unroll
                for (int a = 0; a < TCO; ++a) {
                    ah[a] = wa[((itl * NPL + 0) * 2) * BCO + a * 32];
                    al[a] = wa[((itl * NPL + 1) * 2) * BCO + a * 32];
                    if (NPL == 3) a3[a] = wa[((itl * NPL + 2) * 2) * BCO + a * 32];
                }
                const bf16x8* bb = Bs + (2 * s + kk) * IMG + tapoff;
#pragma unroll
                for (int t = 0; t < TPX; ++t) {
                    bh[t] = bb[lds_off[t]];
                    bl[t] = bb[NG * IMG + lds_off[t]];
                    if (NPL == 3) b3[t] = bb[2 * NG * IMG + lds_off[t]];
                }
#pragma unroll
                for (int a = 0; a < TCO; ++a)
#pragma unroll
                    for (int t = 0; t < TPX; ++t) {
                        if (NPL == 3) {
                            // six products, smallest first: mid*mid, hi*lo3, lo3*hi, hi*mid, mid*hi, hi*hi
                            if (SINGLE) {
                                accx[0] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(al[a], bl[t], accx[0], 0, 0, 0);
                                accx[1] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(ah[a], b3[t], accx[1], 0, 0, 0);
                                acc[a][t] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a3[a], bh[t], acc[a][t], 0, 0, 0);
                                accx[0] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(ah[a], bl[t], accx[0], 0, 0, 0);
                                accx[1] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(al[a], bh[t], accx[1], 0, 0, 0);
                                acc[a][t] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(ah[a], bh[t], acc[a][t], 0, 0, 0);
                            } else {
                                acc[a][t] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(al[a], bl[t], acc[a][t], 0, 0, 0);
                                acc[a][t] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(ah[a], b3[t], acc[a][t], 0, 0, 0);
                                acc[a][t] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a3[a], bh[t], acc[a][t], 0, 0, 0);
                                acc[a][t] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(ah[a], bl[t], acc[a][t], 0, 0, 0);
                                acc[a][t] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(al[a], bh[t], acc[a][t], 0, 0, 0);
                                acc[a][t] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(ah[a], bh[t], acc[a][t], 0, 0, 0);
                            }
                        } else if (SINGLE) {
                            // one accumulator tile per wave: three independent chains instead of one of 3*T*NS
                            // dependent MFMAs per chunk (each waits for the previous result)
                            acc[a][t] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(ah[a], bh[t], acc[a][t], 0, 0, 0);
                            accx[0] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(al[a], bh[t], accx[0], 0, 0, 0);
                            accx[1] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(ah[a], bl[t], accx[1], 0, 0, 0);
                        } else {
                            acc[a][t] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(al[a], bh[t], acc[a][t], 0, 0, 0);
                            acc[a][t] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(ah[a], bl[t], acc[a][t], 0, 0, 0);
                            acc[a][t] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(ah[a], bh[t], acc[a][t], 0, 0, 0);
                        }
                    }
            }
        }
    }
    if (SINGLE) {
#pragma unroll
        for (int r = 0; r < 16; ++r) acc[0][0][r] += accx[0][r] + accx[1][r];
    }

    float* psum = ep + 2 * BCO;  // [BCO][2] sums of the fused activation backward (ep_mode 4)
    if (p.ep_mode == 4)
        for (int c = tid; c < 2 * BCO; c += 256) psum[c] = 0.f;
    __syncthreads();
    conv_epilogue<TCO, TPX, BCO>(p, acc, ep, co_base, wco, kk, HW, pn, ppix, pvalid, blockIdx.x * WPX + wpx,
                                 p.ep_mode == 4 ? psum : nullptr, true);
    if (p.ep_mode == 4) {
        __syncthreads();
        for (int c = tid; c < 2 * BCO; c += 256) {
            const int co = blockIdx.y * BCO + (c >> 1);
            if (co < p.Cout) atomicAdd(&p.part[(long)(c & 1) * p.Cout + co], psum[c]);
        }
    }
}

template <int KS, int WCO, int WPX, int TCO, int TPX, int KC, int NPL = 2>
static int launch_conv_b3(ConvParams& p, hipStream_t s) {
    constexpr int BCO = 32 * TCO * WCO, BPX = 32 * TPX * WPX, PAD = KS / 2;
    tile_geometry(p.H, p.W, BPX, &p.TWp, &p.TH, &p.TF);
    p.tw_shift = ilog2(p.TWp);
    p.th_shift = ilog2(p.TH);
    p.n_wtiles = ceil_div(p.W, p.TWp);
    p.n_htiles = ceil_div(p.H, p.TH);
    p.n_ftiles = ceil_div(p.N, p.TF);
    const int IMG = p.TF * (p.TH + 2 * PAD) * (p.TWp + 2 * PAD);
    constexpr int NPOS = (KS == 1) ? 1 : (BPX >= 256 ? 4 : (BPX >= 128 ? 2 : 1));
    constexpr int P2 = (KS == 1 && BPX < 256) ? (BPX >= 64 ? BPX : 64) : 256;
    if (IMG > P2 * NPOS) {
        rfn_set_error("conv2d(bf16x3): map %dx%d needs %d LDS slots (> %d supported)", p.H, p.W, IMG, P2 * NPOS);
        return -7;
    }
    size_t lds = (size_t)NPL * (KC / 8) * IMG * 16 + (size_t)(KC / 16) * KS * KS * NPL * 2 * BCO * 16 + 4 * BCO * 4;
    auto kern = conv_b3_kernel<KS, WCO, WPX, TCO, TPX, KC, NPL>;
    if (lds > 65536) (void)hipFuncSetAttribute((const void*)kern, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
    dim3 grid(p.n_wtiles * p.n_htiles * p.n_ftiles, ceil_div(p.Cout, BCO));
    p.ksplit = 1;
    const int wgs = grid.x * grid.y;
    const int nchunks = (p.Cin8 * 16 + KC - 1) / KC;
    const int HW = p.H * p.W;
    const bool dense = p.out1_ns == (long)p.cout_split * HW &&
                       (p.cout_split == p.Cout || p.out2_ns == (long)(p.Cout - p.cout_split) * HW);
    if (wgs < 128 && nchunks >= 8 && p.ep_mode != 1 && p.ep_mode != 4 && !p.acc1 && !p.acc2 && dense) {
        int ks_ = 512 / wgs;
        if (ks_ > nchunks / 2) ks_ = nchunks / 2;
        if (ks_ > 1) {
            p.ksplit = ks_;
            p.ws_stride = (long)p.N * p.Cout * HW;
            p.ws = rfn_workspace(s, (size_t)ks_ * (size_t)p.ws_stride);
            if (!p.ws) return -8;
        }
    }
    grid.z = p.ksplit;
    hipLaunchKernelGGL(kern, grid, dim3(256), lds, s, p);
    if (p.ksplit > 1) splitk_reduce(p, s);   // slices added in order: deterministic
    return 0;
}

static int dispatch_conv_b3(ConvParams& p, int ks, hipStream_t s);

// number of partial-sum rows rfn_conv2d_dgrad_act_bf16x3 writes for (N,H,W,ks,Cout): pixel tiles x waves along pixels
static bool conv1x1_ws_eligible(int ks, int Cin, int C2, int Cout, long npix);
static bool conv3x3_ws_rows(int N, int H, int W, int ks, int Cout, int Cin, int* rows);
extern "C" int rfn_conv2d_dgrad_act_rows_bf16x3(int N, int H, int W, int ks, int Cout, int Cin) {
    if (conv1x1_ws_eligible(ks, Cin, 0, Cout, (long)N * H * W)) {
        const long nt = ((long)N * H * W + 31) / 32;
        return (int)(nt < 256 ? nt : 256);  // one row per persistent workgroup
    }
    int ws3_rows = 0;
    if (conv3x3_ws_rows(N, H, W, ks, Cout, Cin, &ws3_rows)) return ws3_rows;
    ConvParams p;
    memset(&p, 0, sizeof(p));
    p.N = N; p.H = H; p.W = W; p.Cout = Cout;
    const bool few_px = (long)N * H * W * ((Cout + 127) / 128) < 256L * 128;
    int BPX, WPX;
    if (ks == 3) { BPX = few_px ? 64 : 128; WPX = 2; }
    else if (few_px || Cout <= 64) { BPX = 64; WPX = 2; }
    else if (Cout <= 128) { BPX = 128; WPX = 2; }
    else { BPX = 64; WPX = 1; }
    if (Cout <= 32) { BPX = 128; WPX = 4; }
    int TWp, TH, TF;
    tile_geometry(H, W, BPX, &TWp, &TH, &TF);
    return ceil_div(W, TWp) * ceil_div(H, TH) * ceil_div(N, TF) * WPX;
}

// data-gradient conv fused with the backward of the producer's Conv2dNorm epilogue (ActNorm + activation):
//   g  = conv(gin, wpk)                      (wpk packed with transpose_flip = 1)
//   gu = g * act'(y) * exp(logs[c])          -> out        (y = saved forward activation, same shape as out)
//   part[0][c] += Σ_pixels gu ,  part[1][c] += Σ g*y     (float atomics; the caller zeroes part [2][Cout])
// Cout must be a multiple of 64.
extern "C" int rfn_conv2d_dgrad_act_bf16x3(const float* gin, long gin_ns, int Cin, const float* wpk, const float* y,
                                           long y_ns, const float* logs, int act, float* out, long out_ns, float* part,
                                           int Cout, int N, int H, int W, int ks, rfn_stream_t stream) {
    RFN_CHECK_ARG(gin && wpk && y && logs && out && part && Cin > 0 && Cout > 0 && Cout % 64 == 0, -1);
    RFN_CHECK_ARG((ks == 1 || ks == 3) && N >= 0 && H > 0 && W > 0 && ((uintptr_t)wpk & 15) == 0, -2);
    if (N == 0) return 0;
    ConvParams p;
    memset(&p, 0, sizeof(p));
    p.in1 = gin; p.in1_ns = gin_ns; p.C1 = Cin; p.wpk = wpk; p.out1 = out; p.out1_ns = out_ns;
    p.Cout = Cout; p.cout_split = Cout; p.N = N; p.H = H; p.W = W;
    packed_dims_b3(Cout, Cin, &p.CoutP, &p.Cin8);
    p.ep_mode = 4; p.act = act; p.p1 = logs; p.ybuf = y; p.ybuf_ns = y_ns; p.part = part;
    int rc = dispatch_conv_b3(p, ks, (hipStream_t)stream);
    if (rc) return rc;
    RFN_LAUNCH_CHECK();
    return 0;
}

// shared body of rfn_conv2d_fwd_bf16x3 (npl = 2) and rfn_conv2d_fwd_bf16x6 (npl = 3)
static int conv2d_fwd_split(const float* in1, long in1_ns, int C1, const float* in2, long in2_ns, int C2,
                            const float* wpk, float* out1, long out1_ns, float* out2, long out2_ns, int Cout,
                            int cout_split, int acc1, int acc2, int N, int H, int W, int ks, int ep_mode,
                            const float* p0, const float* p1, int act, int npl, rfn_stream_t stream) {
    RFN_CHECK_ARG(in1 && wpk && out1 && C1 > 0 && C2 >= 0 && Cout > 0 && N >= 0 && H > 0 && W > 0, -1);
    RFN_CHECK_ARG(ks == 1 || ks == 3, -2);
    RFN_CHECK_ARG(C2 == 0 || in2, -3);
    RFN_CHECK_ARG(cout_split >= 0 && cout_split <= Cout && (cout_split == Cout || out2), -4);
    RFN_CHECK_ARG(ep_mode >= 0 && ep_mode <= 3 && (ep_mode == 0 || p0) && ((ep_mode != 1 && ep_mode != 2) || p1), -5);
    RFN_CHECK_ARG(((uintptr_t)wpk & 15) == 0, -6);
    if (N == 0) return 0;
    ConvParams p;
    memset(&p, 0, sizeof(p));
    p.in1 = in1; p.in2 = in2; p.in1_ns = in1_ns; p.in2_ns = in2_ns; p.C1 = C1; p.C2 = C2;
    p.wpk = wpk; p.out1 = out1; p.out2 = out2; p.out1_ns = out1_ns; p.out2_ns = out2_ns;
    p.Cout = Cout; p.cout_split = cout_split; p.acc1 = acc1; p.acc2 = acc2;
    p.N = N; p.H = H; p.W = W;
    packed_dims_b3(Cout, C1 + C2, &p.CoutP, &p.Cin8);
    p.ep_mode = ep_mode; p.act = act; p.p0 = p0; p.p1 = p1;
    p.npl = npl;
    int rc = dispatch_conv_b3(p, ks, (hipStream_t)stream);
    if (rc) return rc;
    RFN_LAUNCH_CHECK();
    return 0;
}
extern "C" int rfn_conv2d_fwd_bf16x3(const float* in1, long in1_ns, int C1, const float* in2, long in2_ns, int C2,
                                     const float* wpk, float* out1, long out1_ns, float* out2, long out2_ns, int Cout,
                                     int cout_split, int acc1, int acc2, int N, int H, int W, int ks, int ep_mode,
                                     const float* p0, const float* p1, int act, rfn_stream_t stream) {
    return conv2d_fwd_split(in1, in1_ns, C1, in2, in2_ns, C2, wpk, out1, out1_ns, out2, out2_ns, Cout, cout_split, acc1,
                            acc2, N, H, W, ks, ep_mode, p0, p1, act, 2, stream);
}
// same convolution on three bf16 pieces per operand and six MFMAs per product (24 significant bits, fp32-grade): wpk from
// rfn_pack_conv_weight_bf16x6
extern "C" int rfn_conv2d_fwd_bf16x6(const float* in1, long in1_ns, int C1, const float* in2, long in2_ns, int C2,
                                     const float* wpk, float* out1, long out1_ns, float* out2, long out2_ns, int Cout,
                                     int cout_split, int acc1, int acc2, int N, int H, int W, int ks, int ep_mode,
                                     const float* p0, const float* p1, int act, rfn_stream_t stream) {
    return conv2d_fwd_split(in1, in1_ns, C1, in2, in2_ns, C2, wpk, out1, out1_ns, out2, out2_ns, Cout, cout_split, acc1,
                            acc2, N, H, W, ks, ep_mode, p0, p1, act, 3, stream);
}


// ------------------------------------------------------------------------------------------------ weight-stationary 1x1
// The flow's hidden 1x1 convolution (Hd -> Hd = 256 -> 256, forward and data gradient, 60 launches per step on up to
// 622 592 pixels) with the weights held in REGISTERS: 8 waves, wave w owns output channels [32w, 32w+32) and keeps their
// 32 x 256 (hi, lo) slab in 128 VGPRs for the whole launch, so the only LDS traffic is the activation tile (staged once,
// read by all 8 waves) and nothing but activations is streamed.  One persistent workgroup per CU sweeps 64-pixel
// tiles; the loads of tile t+1 are in flight while tile t is multiplied and stored (double-buffered LDS, one barrier
// per tile).  Per tile and CU: 96 MFMAs per wave against 64 KB read + 64 KB written -> the kernel sits on the HBM roof
// instead of the issue / LDS / barrier overheads of the generic tile kernel above (which re-stages the 256 x 32
// weight chunk for every 64 pixels).
template <int NSTEPS>
__global__ __launch_bounds__(512) void conv1x1_ws_kernel(const ConvParams p, const int n_tiles, const int hw_shift) {
    constexpr int NG = NSTEPS * 2;  // 8-channel units per pixel
    constexpr int TP = 32;          // pixels per tile
    extern __shared__ __attribute__((aligned(16))) unsigned char lds_raw[];
    bf16x8* Bs = reinterpret_cast<bf16x8*>(lds_raw);                       // [2 buffers][plane][NG][TP]
    float* ep = reinterpret_cast<float*>(Bs + 2 * 2 * NG * TP);            // [2][256]
    float* psum = ep + 2 * 256;                                            // [256][2] running sums of ep_mode 4
    for (int c = threadIdx.x; c < 512; c += 512) psum[c] = 0.f;
    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int l31 = lane & 31, kk = lane >> 5;
    const int HW = p.H * p.W;
    const long total = (long)p.N * HW;
    const int co_base = blockIdx.y * 256 + wave * 32;

    // weights -> registers (packed unit index ((s*2 + plane)*2 + kk)*CoutP + co, see pack_weight_b3_kernel)
    bf16x8 wh[NSTEPS], wl[NSTEPS];
    {
        const bf16x8* wp = reinterpret_cast<const bf16x8*>(p.wpk) + co_base + l31;
#pragma unroll
        for (int s = 0; s < NSTEPS; ++s) {
            const bool ok = s < p.Cin8;  // Cin8 holds the number of 16-channel steps
            const int sc = ok ? s : 0;
            const bf16x8 h = wp[(long)((sc * 2 + 0) * 2 + kk) * p.CoutP];
            const bf16x8 l = wp[(long)((sc * 2 + 1) * 2 + kk) * p.CoutP];
            const bf16x8 z = {0, 0, 0, 0, 0, 0, 0, 0};
            wh[s] = ok ? h : z;
            wl[s] = ok ? l : z;
        }
    }
    if (p.ep_mode != 0) {
        for (int c = tid; c < 256; c += 512) {
            const int co = blockIdx.y * 256 + c;
            float e0 = 0.f, e1 = 1.f;
            if (co < p.Cout) {
                if (p.ep_mode != 4) e0 = p.p0[co];
                if (p.ep_mode == 1 || p.ep_mode == 4) e1 = expf(p.p1[co]);
                if (p.ep_mode == 2) e1 = expf(3.f * p.p1[co]);
            }
            ep[c] = e0;
            ep[256 + c] = e1;
        }
    }

    // Staging role: pixel l31 of the tile, channel units 2*wave + kk + 16*i (i < NG/16).  Loads go through one buffer
    // descriptor: a per-lane 32-bit byte offset (frame, pixel, kk) plus a scalar channel offset, so the loads of a tile
    // need no per-load 64-bit address registers (they would not fit next to the weights).
    constexpr int NU = NG / 16;  // units per thread and tile
    const int Cin = p.C1;
    const unsigned long in_bytes = (unsigned long)p.N * (unsigned long)p.in1_ns * 4ul;
    const auto rsrc = __builtin_amdgcn_make_buffer_rsrc(const_cast<float*>(p.in1), 0,
                                                        in_bytes > 0xFFFFFFFFul ? 0xFFFFFFFFu : (unsigned)in_bytes, 0x00020000);
    float stg[NU][8];
    bool sval = false;
    auto split_q = [&](long q, int& n, int& pix) {
        if (hw_shift >= 0) {
            n = (int)(q >> hw_shift);
            pix = (int)(q & (HW - 1));
        } else {
            n = (int)(q / HW);
            pix = (int)(q - (long)n * HW);
        }
    };
    auto prefetch = [&](int tile) {
        const long q = (long)tile * TP + l31;
        sval = tile < n_tiles && q < total;
        int n, pix;
        split_q(sval ? q : 0, n, pix);
        const unsigned voff = (unsigned)(((long)n * p.in1_ns + pix + (long)kk * 8 * HW) * 4);
#pragma unroll
        for (int i = 0; i < NU; ++i)
#pragma unroll
            for (int c = 0; c < 8; ++c) {
                const int ch0 = (2 * wave + 16 * i) * 8 + c;  // scalar; + 8*kk per lane (in voff)
                const int soff = (ch0 < Cin ? ch0 : 0) * HW * 4;  // kk = 1 lanes past Cin: in-bounds or zero (descriptor), masked in commit
                stg[i][c] = __uint_as_float(__builtin_amdgcn_raw_buffer_load_b32(rsrc, voff, soff, 0));
            }
    };
    auto commit = [&](int buf) {
        bf16x8* dst = Bs + (long)buf * 2 * NG * TP + l31;
#pragma unroll
        for (int i = 0; i < NU; ++i) {
            const int g = 2 * wave + kk + 16 * i;
            bf16x8 hi, lo;
#pragma unroll
            for (int c = 0; c < 8; ++c) {
                const float v = (sval && g * 8 + c < Cin) ? stg[i][c] : 0.f;
                const __bf16 h = (__bf16)v;
                hi[c] = h;
                lo[c] = (__bf16)(v - (float)h);
            }
            dst[(0 * NG + g) * TP] = hi;
            dst[(1 * NG + g) * TP] = lo;
        }
    };

    int tile = blockIdx.x, buf = 0;
    prefetch(tile);
    for (; tile < n_tiles; tile += gridDim.x, buf ^= 1) {
        commit(buf);
        prefetch(tile + gridDim.x);         // unconditional (clamped): in flight during the MFMAs and stores below
        __builtin_amdgcn_sched_barrier(0);
        __syncthreads();
        const bf16x8* bb = Bs + (long)buf * 2 * NG * TP + kk * TP + l31;
        f32x16 acc[1][1];
#pragma unroll
        for (int r = 0; r < 16; ++r) acc[0][0][r] = 0.f;
#pragma unroll
        for (int s = 0; s < NSTEPS; ++s) {
            const bf16x8 bh = bb[(0 * NG + 2 * s) * TP];
            const bf16x8 bl = bb[(1 * NG + 2 * s) * TP];
            acc[0][0] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(wl[s], bh, acc[0][0], 0, 0, 0);
            acc[0][0] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(wh[s], bl, acc[0][0], 0, 0, 0);
            acc[0][0] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(wh[s], bh, acc[0][0], 0, 0, 0);
        }
        int pn[1], ppix[1];
        bool pvalid[1];
        const long q = (long)tile * TP + l31;
        pvalid[0] = q < total;
        split_q(pvalid[0] ? q : 0, pn[0], ppix[0]);
        conv_epilogue<1, 1, 256>(p, acc, ep, co_base, wave, kk, HW, pn, ppix, pvalid, tile, psum);
    }
    if (p.ep_mode == 4) {  // the workgroup's partial sums: one atomic per (sum, channel)
        __syncthreads();
        for (int c = tid; c < 512; c += 512)  // psum is [256][2]; part rows are [2][Cout]
            atomicAdd(&p.part[(long)(c & 1) * p.Cout + blockIdx.y * 256 + (c >> 1)], psum[c]);
    }
}


// ------------------------------------------------------------------------------------------------ weight-stationary 3x3
// The coupling net's FIRST convolution at the shallow levels (z1 | condition -> Hd: 18 -> 256 on 622 592 pixels at level
// 0): K = 9 * Cin is tiny, the output (637 MB) is everything.  The generic kernel pads every tap to 16-channel chunks
// (18 -> 32: 44 % wasted MFMAs), re-stages a weight chunk per 128 pixels and runs four cout blocks per pixel tile.
// Here the K dimension is the dense list of (tap, 8-channel unit) pairs (27 units for Cin <= 24), the 256 x K weights
// live in registers (8 waves x 32 output channels), and one persistent workgroup per CU walks 32-pixel tiles whose
// haloed input image (<= 102 positions x NG units) is double-buffered in LDS.
template <int NG, int PT, bool FASTEP = true>
__global__ __launch_bounds__(512) void conv3x3_ws_kernel(const ConvParams p, const int n_tiles, const int tw_shift,
                                                         const int tpf_shift, const int wt_shift) {
    constexpr int NU = 9 * NG, NSTEPS = (NU + 1) / 2;
    extern __shared__ __attribute__((aligned(16))) unsigned char lds_raw[];
    const int TW = 1 << tw_shift, TH = (32 * PT) >> tw_shift;  // PT independent 32-pixel accumulator tiles per step
    const int IMGW = TW + 2, IMG = IMGW * (TH + 2);
    const int IMGP = (IMG + 3) & ~3;
    bf16x8* Bs = reinterpret_cast<bf16x8*>(lds_raw);                        // [2 buffers][plane][NG][IMGP]
    float* ep = reinterpret_cast<float*>(Bs + 2 * 2 * NG * IMGP);           // [2][256]
    float* psum = ep + 2 * 256;                                             // [256][2] running sums of ep_mode 4
    for (int c = threadIdx.x; c < 512; c += 512) psum[c] = 0.f;
    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int l31 = lane & 31, kk = lane >> 5;
    const int HW = p.H * p.W, Cin = p.C1 + p.C2;
    const int co_base = blockIdx.y * 256 + wave * 32;

    // weights -> registers: k-step s covers units 2s (lanes 0-31) and 2s+1 (lanes 32-63); unit u = (tap u / NG, group
    // u % NG) sits in the packed buffer at (((c16*9 + tap)*2 + plane)*2 + g)*CoutP + co with c16 = group/2, g = group%2
    bf16x8 wh[NSTEPS], wl[NSTEPS];
    {
        const bf16x8* wp = reinterpret_cast<const bf16x8*>(p.wpk) + co_base + l31;
        const bf16x8 z = {0, 0, 0, 0, 0, 0, 0, 0};
#pragma unroll
        for (int s = 0; s < NSTEPS; ++s) {
            const int u = 2 * s + kk;
            const bool ok = u < NU && (u % NG) / 2 < p.Cin8;
            const int uc = ok ? u : 0;
            const int tap = uc / NG, grp = uc % NG;
            const long base = (long)(((grp / 2) * 9 + tap) * 2) * 2 + (grp & 1);
            const bf16x8 h = wp[(base + 0) * p.CoutP];
            const bf16x8 l = wp[(base + 2) * p.CoutP];
            wh[s] = ok ? h : z;
            wl[s] = ok ? l : z;
        }
    }
    if (p.ep_mode != 0) {
        for (int c = tid; c < 256; c += 512) {
            const int co = blockIdx.y * 256 + c;
            float e0 = 0.f, e1 = 1.f;
            if (co < p.Cout) {
                if (p.ep_mode != 4) e0 = p.p0[co];
                if (p.ep_mode == 1 || p.ep_mode == 4) e1 = expf(p.p1[co]);
                if (p.ep_mode == 2) e1 = expf(3.f * p.p1[co]);
            }
            ep[c] = e0;
            ep[256 + c] = e1;
        }
    }

    // staging role: one (unit, image position) pair per thread (NG * IMG <= 512)
    const bool srole = tid < NG * IMG;
    const int sg = srole ? tid / IMG : 0, spos = srole ? tid - sg * IMG : 0;
    const int syy = spos / IMGW, sxx = spos - syy * IMGW;
    float stg[8];
    bool sval = false;
    auto tile_origin = [&](int tile, int& n, int& y0, int& x0) {
        n = tile >> tpf_shift;
        const int r = tile & ((1 << tpf_shift) - 1);
        y0 = (r >> wt_shift) * TH;
        x0 = (r & ((1 << wt_shift) - 1)) * TW;
    };
    auto prefetch = [&](int tile) {
        int n, y0, x0;
        tile_origin(tile < n_tiles ? tile : 0, n, y0, x0);
        const int gy = y0 + syy - 1, gx = x0 + sxx - 1;
        sval = srole && tile < n_tiles && gy >= 0 && gy < p.H && gx >= 0 && gx < p.W;
        const long off = sval ? (long)gy * p.W + gx : 0;
#pragma unroll
        for (int c = 0; c < 8; ++c) {
            const int ch = sg * 8 + c;
            const int chc = ch < Cin ? ch : 0;
            const float* src = chc < p.C1 ? p.in1 + (long)n * p.in1_ns + (long)chc * HW
                                          : p.in2 + (long)n * p.in2_ns + (long)(chc - p.C1) * HW;
            stg[c] = src[off];
        }
    };
    auto commit = [&](int buf) {
        if (!srole) return;
        bf16x8 hi, lo;
#pragma unroll
        for (int c = 0; c < 8; ++c) {
            const float v = (sval && sg * 8 + c < Cin) ? stg[c] : 0.f;
            const __bf16 h = (__bf16)v;
            hi[c] = h;
            lo[c] = (__bf16)(v - (float)h);
        }
        bf16x8* dst = Bs + (long)buf * 2 * NG * IMGP + spos;
        dst[(0 * NG + sg) * IMGP] = hi;
        dst[(1 * NG + sg) * IMGP] = lo;
    };

    // this lane's pixels inside the tile and their image positions (centre tap)
    int pcol[PT], prow[PT], pbase[PT];
#pragma unroll
    for (int t = 0; t < PT; ++t) {
        const int px = t * 32 + l31;
        pcol[t] = px & (TW - 1);
        prow[t] = px >> tw_shift;
        pbase[t] = (prow[t] + 1) * IMGW + pcol[t] + 1;
    }
    int tile = blockIdx.x, buf = 0;
    prefetch(tile);
    for (; tile < n_tiles; tile += gridDim.x, buf ^= 1) {
        commit(buf);
        prefetch(tile + gridDim.x);
        __builtin_amdgcn_sched_barrier(0);
        __syncthreads();
        const bf16x8* bb = Bs + (long)buf * 2 * NG * IMGP;
        f32x16 acc[1][PT];
#pragma unroll
        for (int t = 0; t < PT; ++t)
#pragma unroll
            for (int r = 0; r < 16; ++r) acc[0][t][r] = 0.f;
#pragma unroll
        for (int s = 0; s < NSTEPS; ++s) {
            // compile-time unit pair of this k-step; the padding unit past 9*NG reads unit 0 (its weights are zero)
            const int uA = 2 * s, uB = (2 * s + 1 < NU) ? 2 * s + 1 : 0;
            const int offA = (uA % NG) * IMGP + (uA / NG / 3 - 1) * IMGW + (uA / NG % 3 - 1);
            const int offB = (uB % NG) * IMGP + (uB / NG / 3 - 1) * IMGW + (uB / NG % 3 - 1);
            const int off = kk ? offB : offA;
#pragma unroll
            for (int t = 0; t < PT; ++t) {
                const bf16x8 bh = bb[pbase[t] + off];
                const bf16x8 bl = bb[NG * IMGP + pbase[t] + off];
                acc[0][t] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(wl[s], bh, acc[0][t], 0, 0, 0);
                acc[0][t] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(wh[s], bl, acc[0][t], 0, 0, 0);
                acc[0][t] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(wh[s], bh, acc[0][t], 0, 0, 0);
            }
        }
        int n, y0, x0;
        tile_origin(tile, n, y0, x0);
        int pn[PT], ppix[PT];
        bool pvalid[PT];
#pragma unroll
        for (int t = 0; t < PT; ++t) {
            pn[t] = n;
            ppix[t] = (y0 + prow[t]) * p.W + x0 + pcol[t];
            pvalid[t] = true;
        }
        conv_epilogue<1, PT, 256, FASTEP>(p, acc, ep, co_base, wave, kk, HW, pn, ppix, pvalid, tile, psum);
    }
    if (!FASTEP && p.ep_mode == 4) {  // the workgroup's partial sums: one atomic per (sum, channel)
        __syncthreads();
        for (int c = tid; c < 512; c += 512)  // psum is [256][2]; part rows are [2][Cout]
            atomicAdd(&p.part[(long)(c & 1) * p.Cout + blockIdx.y * 256 + (c >> 1)], psum[c]);
    }
}

static bool conv3x3_ws_eligible(const ConvParams& p, int ks) {
    static const bool off = getenv("RFN_CONV_WS") && atoi(getenv("RFN_CONV_WS")) == 0;
    const int Cin = p.C1 + p.C2;
    const bool pow2 = (p.H & (p.H - 1)) == 0 && (p.W & (p.W - 1)) == 0;
    const bool shape = !off && ks == 3 && p.Cout % 256 == 0 && pow2 && p.W >= 8 && (long)p.H * p.W >= 64 &&
                       (long)p.N * p.H * p.W >= 64L * 256 && p.cout_split == p.Cout && !p.acc1;
    if (p.ep_mode == 4) return shape && Cin <= 8 && p.C2 == 0;  // fused activation backward: 9-unit variant only
    return shape && Cin <= 40 && p.ep_mode >= 0 && p.ep_mode <= 3;
}
// rows of partial sums the ep_mode-4 variant writes: one per 64-pixel tile
static bool conv3x3_ws_rows(int N, int H, int W, int ks, int Cout, int Cin, int* rows) {
    ConvParams p;
    memset(&p, 0, sizeof(p));
    p.N = N; p.H = H; p.W = W; p.Cout = Cout; p.cout_split = Cout; p.C1 = Cin; p.ep_mode = 4;
    if (!conv3x3_ws_eligible(p, ks)) return false;
    const long nt = (long)N * H * W / 64;
    *rows = (int)(nt < 256 ? nt : 256);  // one row per persistent workgroup
    return true;
}

template <int NG, int PT, bool FASTEP = true>
static int launch_conv3x3_ws_t(ConvParams& p, hipStream_t s) {
    const int TW = p.W < 32 ? p.W : 32, TH = 32 * PT / TW;
    const int tw_shift = ilog2(TW);
    const int wt = p.W / TW, ht = p.H / TH;  // tiles per row / column of a frame
    const int wt_shift = ilog2(wt), tpf_shift = ilog2(wt * ht);
    const int n_tiles = p.N * wt * ht;
    const int IMG = (TW + 2) * (TH + 2), IMGP = (IMG + 3) & ~3;
    if (NG * IMG > 512) {
        rfn_set_error("conv3x3_ws: %d staging items", NG * IMG);
        return -8;
    }
    p.ksplit = 1;
    const size_t lds = (size_t)2 * 2 * NG * IMGP * 16 + 4 * 256 * 4;
    auto kern = conv3x3_ws_kernel<NG, PT, FASTEP>;
    dim3 grid(n_tiles < 256 ? n_tiles : 256, p.Cout / 256);
    hipLaunchKernelGGL(kern, grid, dim3(512), lds, s, p, n_tiles, tw_shift, tpf_shift, wt_shift);
    return 0;
}
static int launch_conv3x3_ws(ConvParams& p, hipStream_t s) {
    // Cin <= 24: 27 units (112 weight registers), two accumulator tiles per step; Cin <= 40: 45 units (184), one
    if (p.ep_mode == 4) return launch_conv3x3_ws_t<1, 2, false>(p, s);
    return p.C1 + p.C2 <= 24 ? launch_conv3x3_ws_t<3, 2>(p, s) : launch_conv3x3_ws_t<5, 1>(p, s);
}

static bool conv1x1_ws_eligible(int ks, int Cin, int C2, int Cout, long npix) {
    static const bool off = getenv("RFN_CONV_WS") && atoi(getenv("RFN_CONV_WS")) == 0;
    return !off && ks == 1 && C2 == 0 && Cout % 256 == 0 && Cin > 128 && Cin <= 256 && npix >= 64L * 256;
}

static int launch_conv1x1_ws(ConvParams& p, hipStream_t s) {
    const int HW = p.H * p.W;
    const long total = (long)p.N * HW;
    const int n_tiles = (int)((total + 31) / 32);
    int hw_shift = -1;
    if ((HW & (HW - 1)) == 0) hw_shift = ilog2(HW);
    p.ksplit = 1;
    const size_t lds = (size_t)2 * 2 * 32 * 32 * 16 + 4 * 256 * 4;
    auto kern = conv1x1_ws_kernel<16>;
    (void)hipFuncSetAttribute((const void*)kern, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
    dim3 grid(n_tiles < 256 ? n_tiles : 256, p.Cout / 256);
    hipLaunchKernelGGL(kern, grid, dim3(512), lds, s, p, n_tiles, hw_shift);
    return 0;
}

static int dispatch_conv_b3(ConvParams& p, int ks, hipStream_t s) {
    const int Cout = p.Cout, N = p.N, H = p.H, W = p.W;
    const bool few_px = (long)N * H * W * ((Cout + 127) / 128) < 256L * 128;
    int rc;
    if (p.npl == 3) {  // bf16x6: the generic tile kernel only (forward convolutions of the middle flow levels)
        if (ks == 3) {
            if (Cout <= 32)
                rc = launch_conv_b3<3, 1, 4, 1, 1, 16, 3>(p, s);
            else if (few_px)
                rc = launch_conv_b3<3, 2, 2, 1, 1, 16, 3>(p, s);
            else
                rc = launch_conv_b3<3, 2, 2, 1, 2, 16, 3>(p, s);
        } else {
            if (Cout <= 32)
                rc = launch_conv_b3<1, 1, 4, 1, 1, 32, 3>(p, s);
            else if (few_px || Cout <= 64)
                rc = launch_conv_b3<1, 2, 2, 1, 1, 32, 3>(p, s);
            else if (Cout <= 128)
                rc = launch_conv_b3<1, 2, 2, 2, 2, 32, 3>(p, s);
            else
                rc = launch_conv_b3<1, 4, 1, 2, 2, 32, 3>(p, s);
        }
        return rc;
    }
    if (conv1x1_ws_eligible(ks, p.C1 + p.C2, p.C2, Cout, (long)N * H * W)) return launch_conv1x1_ws(p, s);
    if (conv3x3_ws_eligible(p, ks)) return launch_conv3x3_ws(p, s);
    if (ks == 3) {
        if (Cout <= 32)
            rc = launch_conv_b3<3, 1, 4, 1, 1, 16>(p, s);   // 32 co x 128 px
        else if (few_px)
            rc = launch_conv_b3<3, 2, 2, 1, 1, 16>(p, s);   // 64 co x 64 px (32-channel chunks measured 1.6x slower)
        else
            rc = launch_conv_b3<3, 2, 2, 1, 2, 16>(p, s);   // 64 co x 128 px
    } else {
        if (Cout <= 32)
            rc = launch_conv_b3<1, 1, 4, 1, 1, 32>(p, s);
        else if (few_px || Cout <= 64)
            rc = launch_conv_b3<1, 2, 2, 1, 1, 32>(p, s);   // 64 co x 64 px
        else if (Cout <= 128)
            rc = launch_conv_b3<1, 2, 2, 2, 2, 32>(p, s);   // 128 co x 128 px
        else
            rc = launch_conv_b3<1, 4, 1, 2, 2, 32>(p, s);   // 256 co x 64 px: the input tile is read once for 256 couts
    }
    return rc;
}
