// Implicit-GEMM convolution (forward / data-gradient / weight-gradient) on the fp32 matrix cores of gfx950.
//
// Orientation (chosen for NCHW): D[cout][pixel] += W[cout][k] * X[k][pixel] with v_mfma_f32_32x32x2_f32, i.e.
//   A operand (32 rows)  = weights      A[i = cout][k]      lane l holds (i = l&31, k = l>>5)
//   B operand (32 cols)  = activations  B[k][j = pixel]     lane l holds (j = l&31, k = l>>5)
//   D: col = lane&31 = pixel, row = (r&3) + 8*(r>>2) + 4*(lane>>5) = cout   -> a register's 32 lanes are 32
//   consecutive pixels of one channel plane = one coalesced 128-byte store in NCHW.
// A pixel tile is TF frames x TH rows x TWp cols (powers of two, TF*TH*TWp = BPX) so deep flow levels (2x2, 4x4
// maps) fill a tile with many frames.  The input tile (with its 3x3 halo, zero filled outside the image) is staged
// in LDS as [channel][frame][row+2][col+2]: a tap shift is a constant LDS offset and B-fragment reads are
// lane-consecutive.  Weights are pre-packed (rfn_pack_conv_weight_f32) so that one 16-byte load per lane yields the
// A fragments of four consecutive k-steps; they stream straight from L2 into registers, prefetched one iteration ahead.
#include "conv_common.h"

// packed weight index: (((g8*T + tap)*2 + kk)*CoutP + co)*4 + ks   <->  cin = g8*8 + 2*ks + kk
__global__ void pack_weight_kernel(const float* __restrict__ w, float* __restrict__ wpk, int Cout, int Cin, int KS,
                                   int CoutP, int Cin8, int transpose_flip) {
    // logical conv described by the packed buffer: Co_l outputs, Ci_l inputs
    const int T = KS * KS;
    const int Co_l = transpose_flip ? Cin : Cout;
    const int Ci_l = transpose_flip ? Cout : Cin;
    const long total = (long)Cin8 * T * 2 * CoutP * 4;
    for (long idx = (long)blockIdx.x * blockDim.x + threadIdx.x; idx < total; idx += (long)gridDim.x * blockDim.x) {
        int ks = (int)(idx & 3);
        long r = idx >> 2;
        int co = (int)(r % CoutP);
        r /= CoutP;
        int kk = (int)(r & 1);
        r >>= 1;
        int tap = (int)(r % T);
        int g8 = (int)(r / T);
        int ci = g8 * 8 + 2 * ks + kk;
        float v = 0.f;
        if (co < Co_l && ci < Ci_l) {
            if (!transpose_flip)
                v = w[((long)co * Cin + ci) * T + tap];
            else
                v = w[((long)ci * Cin + co) * T + (T - 1 - tap)];  // w[cout=ci_l][cin=co_l][mirrored tap]
        }
        wpk[idx] = v;
    }
}

// CoutP is padded to the cout block of the kernel configuration chosen for this Cout (see rfn_conv2d_fwd_f32), so
// every A-fragment load of a launched block stays inside the packed buffer.
static inline void packed_dims(int Cout_l, int Cin_l, int* CoutP, int* Cin8) {
    *CoutP = Cout_l <= 32 ? 32 : (Cout_l <= 64 ? 64 : ((Cout_l + 255) / 256) * 256);
    *Cin8 = (Cin_l + 7) / 8;
}

extern "C" long rfn_packed_weight_size(int Cout, int Cin, int ks) {
    // large enough for both orientations
    int a, b, c, d;
    packed_dims(Cout, Cin, &a, &b);
    packed_dims(Cin, Cout, &c, &d);
    long s0 = (long)b * ks * ks * 2 * a * 4, s1 = (long)d * ks * ks * 2 * c * 4;
    return s0 > s1 ? s0 : s1;
}

extern "C" int rfn_pack_conv_weight_f32(const float* w, float* wpk, int Cout, int Cin, int ks, int transpose_flip,
                                        rfn_stream_t stream) {
    RFN_CHECK_ARG(w && wpk && Cout > 0 && Cin > 0 && (ks == 1 || ks == 3), -1);
    int CoutP, Cin8;
    if (!transpose_flip)
        packed_dims(Cout, Cin, &CoutP, &Cin8);
    else
        packed_dims(Cin, Cout, &CoutP, &Cin8);
    long total = (long)Cin8 * ks * ks * 2 * CoutP * 4;
    int grid = (int)((total + 255) / 256 < 2048 ? (total + 255) / 256 : 2048);
    hipLaunchKernelGGL(pack_weight_kernel, dim3(grid), dim3(256), 0, (hipStream_t)stream, w, wpk, Cout, Cin, ks, CoutP,
                       Cin8, transpose_flip);
    RFN_LAUNCH_CHECK();
    return 0;
}

// ------------------------------------------------------------------------------------------------ forward / dgrad
template <int KS, int WCO, int WPX, int TCO, int TPX, int KC>
#ifndef RFN_CONV_WAVES
#define RFN_CONV_WAVES 1
#endif
__global__ __launch_bounds__(256, RFN_CONV_WAVES) void conv_mfma_kernel(const ConvParams p) {
    constexpr int T = KS * KS, PAD = KS / 2;
    constexpr int BCO = 32 * TCO * WCO;
    static_assert(WCO * WPX == 4, "4 waves");
    extern __shared__ float lds[];  // [KC][IMG]
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
    const int FRM = RH * RW;       // one frame's padded image
    const int IMG = p.TF * FRM;    // per channel
    const int co_base = blockIdx.y * BCO + wco * (32 * TCO);

    int lds_off[TPX], pn[TPX], ppix[TPX];
    bool pvalid[TPX];
#pragma unroll
    for (int t = 0; t < TPX; ++t) {
        int q = (wpx * TPX + t) * 32 + l31;
        int col = q & (p.TWp - 1);
        int row = (q >> p.tw_shift) & (p.TH - 1);
        int f = q >> (p.tw_shift + p.th_shift);
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

    // ---- staging descriptors, computed ONCE per workgroup (the integer divisions live here, not in the K loop):
    // thread -> image slot(s) r = slot + P2*j and channel phase; per chunk it then only adds a channel offset.
    // P2 = slots per pass (compile time): a 1x1 tile of 128 pixels is covered by half the block, so the two halves
    // take alternate channels (GROUPS = 2); every other case has IMG > 128 and uses all 256 threads as slots.
    constexpr int BPX = 32 * TPX * WPX;
    constexpr int NPOS = (KS == 1) ? 1 : (BPX >= 256 ? 4 : 2);
    // (P2 >= 64 keeps `phase` wave-uniform)
    constexpr int GROUPS = (KS == 1 && BPX < 256) ? (BPX >= 64 ? 256 / BPX : 4) : 1;
    constexpr int P2 = 256 / GROUPS;
    const int slot = tid & (P2 - 1);
    const int phase = __builtin_amdgcn_readfirstlane(tid / P2);  // wave-uniform: keeps channel math scalar
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
        // invalid slots read element 0 of the channel plane (always mapped) and are zeroed after the load, so the
        // loads below are unconditional and the compiler can keep all of them in flight (no per-load branch/wait)
        soff1[j] = sok[j] ? (int)(f * p.in1_ns) + gy * p.W + gx : 0;
        soff2[j] = sok[j] ? (int)(f * p.in2_ns) + gy * p.W + gx : 0;
    }
    const float* in1b = p.in1 + (long)f0 * p.in1_ns;
    const float* in2b = p.in2 ? p.in2 + (long)f0 * p.in2_ns : p.in1;
    float stg[KC / GROUPS][NPOS];
    auto prefetch = [&](int chunk) {
#pragma unroll
        for (int i = 0; i < KC / GROUPS; ++i) {
            const int ch = chunk * KC + phase + GROUPS * i;  // scalar
            const bool chv = ch < Cin;
            const int chc = chv ? ch : 0;
            const bool first = chc < p.C1;
            const float* src = first ? in1b + (long)chc * HW : in2b + (long)(chc - p.C1) * HW;
#pragma unroll
            for (int j = 0; j < NPOS; ++j) {
                const float v = src[first ? soff1[j] : soff2[j]];
                stg[i][j] = (sok[j] && chv) ? v : 0.f;
            }
        }
    };
    auto commit = [&]() {
#pragma unroll
        for (int i = 0; i < KC / GROUPS; ++i) {
            const int c = phase + GROUPS * i;
#pragma unroll
            for (int j = 0; j < NPOS; ++j)
                if (sin[j]) lds[c * IMG + slot + P2 * j] = stg[i][j];
        }
    };

    // ---- weights of one K chunk also go through LDS.  (Loading A fragments straight from L2 inside the MFMA loop
    // looked free, but vmcnt retires in order: every such load queued behind the HBM prefetch of the next activation
    // chunk and stalled the wave for a full HBM round trip per chunk.)  The packed layout is linear per (iteration,
    // k-parity): runs of BCO float4, copied verbatim -> Ws4[(it_local*2 + kk)*BCO + cout_local].
    const int total_it = p.Cin8 * T;
    const f32x4* wp4 = reinterpret_cast<const f32x4*>(p.wpk);
    constexpr int ITC = (KC / 8) * T;               // iterations (8-channel group x tap) per chunk
    constexpr int WRUNS = ITC * 2;                  // runs of BCO float4 per chunk
    constexpr int WPT = (WRUNS * BCO + 255) / 256;  // float4 per thread per chunk
    f32x4* Ws4 = reinterpret_cast<f32x4*>(lds + p.w_lds_off);
    const long wblk = (long)blockIdx.y * BCO;
    f32x4 wstg[WPT];
    auto wprefetch = [&](int chunk) {
        const int it0 = chunk * ITC;
#pragma unroll
        for (int j = 0; j < WPT; ++j) {
            const int e = tid + 256 * j;
            const int run = e / BCO, col = e % BCO;  // BCO is a power of two
            const int itg = it0 + (run >> 1);
            const bool ok = (WRUNS * BCO % 256 == 0 || e < WRUNS * BCO) && itg < total_it;
            const f32x4 v = wp4[((long)(ok ? itg : 0) * 2 + (run & 1)) * p.CoutP + wblk + col];
            wstg[j] = ok ? v : f32x4{0.f, 0.f, 0.f, 0.f};
        }
    };
    auto wcommit = [&]() {
#pragma unroll
        for (int j = 0; j < WPT; ++j) {
            const int e = tid + 256 * j;
            if (WRUNS * BCO % 256 == 0 || e < WRUNS * BCO) Ws4[e] = wstg[j];
        }
    };

    // epilogue parameters of this block's BCO channels -> LDS (read after the K loop; its barriers order the write)
    float* ep = lds + KC * IMG;  // [2][BCO]
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

    const int nchunks_all = (p.Cin8 * 8 + KC - 1) / KC;
    const int cps = (nchunks_all + p.ksplit - 1) / p.ksplit;  // chunks per K split
    const int chunk0 = blockIdx.z * cps;
    const int nchunks = chunk0 + cps < nchunks_all ? chunk0 + cps : nchunks_all;
    if (chunk0 < nchunks) {
        prefetch(chunk0);
        wprefetch(chunk0);
    }
    const f32x4* wa = Ws4 + kk * BCO + wco * (32 * TCO) + l31;  // + it_local*2*BCO + a*32
    for (int chunk = chunk0; chunk < nchunks; ++chunk) {
        __syncthreads();  // every wave is done reading the previous chunk
        commit();
        wcommit();
        __syncthreads();
        if (chunk + 1 < nchunks) {  // global loads of the next chunk stay in flight under the MFMAs below
            prefetch(chunk + 1);
            wprefetch(chunk + 1);
        }
#pragma unroll
        for (int s8 = 0; s8 < KC / 8; ++s8) {
#pragma unroll
            for (int tap = 0; tap < T; ++tap) {
                const int itl = s8 * T + tap;
                f32x4 af[TCO];
#pragma unroll
                for (int a = 0; a < TCO; ++a) af[a] = wa[itl * 2 * BCO + a * 32];
                const int tapoff = (tap / KS) * RW + (tap % KS);
                const float* lbase = lds + (s8 * 8 + kk) * IMG + tapoff;
                // four k-steps of the 8-channel group, branch free.  Ragged Cin tails and groups past Cin multiply
                // zero weights by zero-filled LDS rows.
#pragma unroll
                for (int ks = 0; ks < 4; ++ks) {
                    float b[TPX];
#pragma unroll
                    for (int t = 0; t < TPX; ++t) b[t] = lbase[2 * ks * IMG + lds_off[t]];
#pragma unroll
                    for (int a = 0; a < TCO; ++a)
#pragma unroll
                        for (int t = 0; t < TPX; ++t)
                            acc[a][t] = __builtin_amdgcn_mfma_f32_32x32x2f32(af[a][ks], b[t], acc[a][t], 0, 0, 0);
                }
            }
        }
    }

    // ---- epilogue
    __syncthreads();  // ep[] visible even when the K loop ran zero chunks
    conv_epilogue<TCO, TPX, BCO>(p, acc, ep, co_base, wco, kk, HW, pn, ppix, pvalid);
}

template <int KS, int WCO, int WPX, int TCO, int TPX, int KC>
static int launch_conv(ConvParams& p, hipStream_t s) {
    constexpr int BCO = 32 * TCO * WCO, BPX = 32 * TPX * WPX, PAD = KS / 2;
    tile_geometry(p.H, p.W, BPX, &p.TWp, &p.TH, &p.TF);
    p.tw_shift = ilog2(p.TWp);
    p.th_shift = ilog2(p.TH);
    p.n_wtiles = ceil_div(p.W, p.TWp);
    p.n_htiles = ceil_div(p.H, p.TH);
    p.n_ftiles = ceil_div(p.N, p.TF);
    const int IMG = p.TF * (p.TH + 2 * PAD) * (p.TWp + 2 * PAD);
    constexpr int NPOS = (KS == 1) ? 1 : (BPX >= 256 ? 4 : 2);
    constexpr int P2 = (KS == 1 && BPX < 256) ? (BPX >= 64 ? BPX : 64) : 256;
    if (IMG > P2 * NPOS) {
        rfn_set_error("conv2d: map %dx%d needs an LDS image of %d slots (> %d supported)", p.H, p.W, IMG, P2 * NPOS);
        return -7;
    }
    p.w_lds_off = ((KC * IMG + 2 * BCO + 3) / 4) * 4;
    size_t lds = ((size_t)p.w_lds_off + (size_t)(KC / 8) * KS * KS * 2 * BCO * 4) * 4;
    auto kern = conv_mfma_kernel<KS, WCO, WPX, TCO, TPX, KC>;
    if (lds > 65536) (void)hipFuncSetAttribute((const void*)kern, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
    dim3 grid(p.n_wtiles * p.n_htiles * p.n_ftiles, ceil_div(p.Cout, BCO));
    // Split K over gridDim.z when the (pixel, cout) grid alone cannot fill 256 CUs (ConvLSTM: 128 pixels x 800
    // couts x K = 6408).  Needs a linear epilogue, dense zero-initialisable outputs and no accumulate flag.
    p.ksplit = 1;
    const int wgs = grid.x * grid.y;
    const int nchunks = (p.Cin8 * 8 + KC - 1) / KC;
    const int HW = p.H * p.W;
    const bool dense = p.out1_ns == (long)p.cout_split * HW &&
                       (p.cout_split == p.Cout || p.out2_ns == (long)(p.Cout - p.cout_split) * HW);
    if (wgs < 128 && nchunks >= 8 && p.ep_mode != 1 && !p.acc1 && !p.acc2 && dense) {
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

extern "C" int rfn_conv2d_fwd_f32(const float* in1, long in1_ns, int C1, const float* in2, long in2_ns, int C2,
                                  const float* wpk, float* out1, long out1_ns, float* out2, long out2_ns, int Cout,
                                  int cout_split, int acc1, int acc2, int N, int H, int W, int ks, int ep_mode,
                                  const float* p0, const float* p1, int act, rfn_stream_t stream) {
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
    packed_dims(Cout, C1 + C2, &p.CoutP, &p.Cin8);
    p.ep_mode = ep_mode; p.act = act; p.p0 = p0; p.p1 = p1;
    hipStream_t s = (hipStream_t)stream;
    int rc;
    // few pixels in total (deep flow levels, ConvLSTM): 32-pixel tiles so that enough workgroups exist
    const bool few_px = Cout > 64 && (long)N * H * W * ((Cout + 127) / 128) < 256L * 128;
    if (ks == 3) {
        if (Cout <= 32)
            rc = launch_conv<3, 1, 4, 1, 2, 8>(p, s);
        else if (Cout <= 64)
            rc = launch_conv<3, 1, 4, 2, 1, 8>(p, s);
        else if (few_px)
            rc = launch_conv<3, 4, 1, 1, 1, 8>(p, s);
        else
            rc = launch_conv<3, 2, 2, 2, 2, 8>(p, s);
    } else {
        if (Cout <= 32)
            rc = launch_conv<1, 1, 4, 1, 2, 32>(p, s);
        else if (Cout <= 64)
            rc = launch_conv<1, 1, 4, 2, 1, 32>(p, s);
        else if (few_px)
            rc = launch_conv<1, 4, 1, 1, 1, 32>(p, s);
        else {
            static int variant = getenv("RFN_CONV_VARIANT") ? atoi(getenv("RFN_CONV_VARIANT")) : 0;
            switch (Cout <= 128 ? 6 : variant) {  // <= 128 couts: one 128-row block, not a half-empty 256-row one
                case 1: rc = launch_conv<1, 2, 2, 2, 2, 64>(p, s); break;
                case 2: rc = launch_conv<1, 2, 2, 2, 4, 32>(p, s); break;
                case 3: rc = launch_conv<1, 4, 1, 2, 2, 32>(p, s); break;
                case 4: rc = launch_conv<1, 2, 2, 4, 2, 32>(p, s); break;
                case 5: rc = launch_conv<1, 2, 2, 2, 2, 16>(p, s); break;
                case 6: rc = launch_conv<1, 2, 2, 2, 2, 32>(p, s); break;
                default: rc = launch_conv<1, 4, 1, 2, 2, 32>(p, s);
            }
        }
    }
    if (rc) return rc;
    RFN_LAUNCH_CHECK();
    return 0;
}

// ------------------------------------------------------------------------------------------------ weight gradient
// gwt[tap][co][ci] += Σ_pixels g[co][px] * x[ci][px + tap]      (tap-major so the final float atomics of one
// register are 32 consecutive floats = full-rate 128-byte atomic segments).
struct WgradParams {
    const float* in1;
    const float* in2;
    long in1_ns, in2_ns;
    int C1, C2;
    const float* g;
    long g_ns;
    int Cout;
    float* gwt;
    int N, H, W;
    int TWp, TH, TF, tw_shift, th_shift;
    int n_wtiles, n_htiles, n_ftiles, n_pix_tiles;
    int P2, P2_shift;
};

template <int KS, int WCO, int WCI, int TCO, int TCI, int BPX>
__global__ __launch_bounds__(256) void wgrad_mfma_kernel(const WgradParams p) {
    constexpr int T = KS * KS, PAD = KS / 2;
    constexpr int BCO = 32 * TCO * WCO, BCI = 32 * TCI * WCI;
    constexpr int GSTR = BPX + 1;
    static_assert(WCO * WCI == 4, "4 waves");
    extern __shared__ float lds[];
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int wco = wave / WCI, wci = wave % WCI;
    const int l31 = lane & 31, kk = lane >> 5;
    const int HW = p.H * p.W, Cin = p.C1 + p.C2;
    const int RW = p.TWp + 2 * PAD, RH = p.TH + 2 * PAD, FRM = RH * RW, IMG = p.TF * FRM;
    const int XSTR = IMG | 1;
    float* Gs = lds;               // [BCO][GSTR]
    float* Xs = lds + BCO * GSTR;  // [BCI][XSTR]
    const int co0 = blockIdx.z * BCO, ci0 = blockIdx.y * BCI;

    f32x16 acc[TCO][TCI][T];
#pragma unroll
    for (int a = 0; a < TCO; ++a)
#pragma unroll
        for (int b = 0; b < TCI; ++b)
#pragma unroll
            for (int t = 0; t < T; ++t)
#pragma unroll
                for (int r = 0; r < 16; ++r) acc[a][b][t][r] = 0.f;

    // ---- staging descriptors (tile independent part computed once; see conv_mfma_kernel)
    constexpr int GGRP = 256 / BPX;                      // G: thread -> pixel slot q, channel phase
    const int gq = tid & (BPX - 1);
    const int gphase = __builtin_amdgcn_readfirstlane(tid / BPX);
    const int gcol = gq & (p.TWp - 1), grow = (gq >> p.tw_shift) & (p.TH - 1), gf = gq >> (p.tw_shift + p.th_shift);
    constexpr int XNPOS = (KS == 1) ? 1 : 2;             // X: image slot(s) r = slot + P2*j, channel phase
    const int xslot = tid & (p.P2 - 1), xgroups = 256 >> p.P2_shift;
    const int xphase = __builtin_amdgcn_readfirstlane(tid >> p.P2_shift);
    int xf[XNPOS], xyy[XNPOS], xxx[XNPOS];
    bool xin[XNPOS];
#pragma unroll
    for (int j = 0; j < XNPOS; ++j) {
        const int r = xslot + p.P2 * j;
        xin[j] = r < IMG;
        xf[j] = r / FRM;
        const int rr = r - xf[j] * FRM;
        xyy[j] = rr / RW;
        xxx[j] = rr - xyy[j] * RW;
    }

    for (int ptile = blockIdx.x; ptile < p.n_pix_tiles; ptile += gridDim.x) {
        int pt = ptile;
        const int wt = pt % p.n_wtiles;
        pt /= p.n_wtiles;
        const int ht = pt % p.n_htiles;
        const int ft = pt / p.n_htiles;
        const int x0 = wt * p.TWp, y0 = ht * p.TH, f0 = ft * p.TF;
        __syncthreads();
        {   // G tile [BCO][BPX]: batches of 8 unconditional loads (invalid -> element 0, zeroed afterwards)
            const bool ok = (f0 + gf < p.N) && (y0 + grow < p.H) && (x0 + gcol < p.W);
            const float* gsrc = p.g + (long)f0 * p.g_ns + (ok ? (long)gf * p.g_ns + (y0 + grow) * p.W + x0 + gcol : 0);
            constexpr int GB = 32;  // loads in flight per thread per batch
            for (int cl0 = gphase; cl0 < BCO; cl0 += GGRP * GB) {
                float v[GB];
#pragma unroll
                for (int u = 0; u < GB; ++u) {
                    const int co = co0 + cl0 + u * GGRP;
                    v[u] = gsrc[(long)(co < p.Cout ? co : 0) * HW];
                }
#pragma unroll
                for (int u = 0; u < GB; ++u) {
                    const int cl = cl0 + u * GGRP;
                    if (cl < BCO) Gs[cl * GSTR + gq] = (ok && co0 + cl < p.Cout) ? v[u] : 0.f;
                }
            }
        }
        {   // X tile [BCI][IMG] with halo, same batching
            const float* in1b = p.in1 + (long)f0 * p.in1_ns;
            const float* in2b = p.in2 ? p.in2 + (long)f0 * p.in2_ns : p.in1;
#pragma unroll
            for (int j = 0; j < XNPOS; ++j) {
                if (!xin[j]) continue;
                const int gy = y0 + xyy[j] - PAD, gx = x0 + xxx[j] - PAD;
                const bool ok = (f0 + xf[j] < p.N) && gy >= 0 && gy < p.H && gx >= 0 && gx < p.W;
                const int off1 = ok ? (int)(xf[j] * p.in1_ns) + gy * p.W + gx : 0;
                const int off2 = ok ? (int)(xf[j] * p.in2_ns) + gy * p.W + gx : 0;
                float* xdst = Xs + xslot + p.P2 * j;
                constexpr int XB = 32;
                for (int cl0 = xphase; cl0 < BCI; cl0 += xgroups * XB) {
                    float v[XB];
#pragma unroll
                    for (int u = 0; u < XB; ++u) {
                        const int ch = ci0 + cl0 + u * xgroups;
                        const int chc = ch < Cin ? ch : 0;
                        v[u] = chc < p.C1 ? in1b[(long)chc * HW + off1] : in2b[(long)(chc - p.C1) * HW + off2];
                    }
#pragma unroll
                    for (int u = 0; u < XB; ++u) {
                        const int cl = cl0 + u * xgroups;
                        if (cl < BCI) xdst[cl * XSTR] = (ok && ci0 + cl < Cin) ? v[u] : 0.f;
                    }
                }
            }
        }
        __syncthreads();
        const float* ga = Gs + (wco * TCO * 32 + l31) * GSTR;
        const float* xb = Xs + (wci * TCI * 32 + l31) * XSTR;
#pragma unroll 4
        for (int k2 = 0; k2 < BPX / 2; ++k2) {
            const int q = 2 * k2 + kk;
            const int col = q & (p.TWp - 1), row = (q >> p.tw_shift) & (p.TH - 1), f = q >> (p.tw_shift + p.th_shift);
            const int ioff = f * FRM + row * RW + col;
            float av[TCO];
#pragma unroll
            for (int a = 0; a < TCO; ++a) av[a] = ga[a * 32 * GSTR + q];
#pragma unroll
            for (int t = 0; t < T; ++t) {
                const int tapoff = (t / KS) * RW + (t % KS);
                float bv[TCI];
#pragma unroll
                for (int b = 0; b < TCI; ++b) bv[b] = xb[b * 32 * XSTR + ioff + tapoff];
#pragma unroll
                for (int a = 0; a < TCO; ++a)
#pragma unroll
                    for (int b = 0; b < TCI; ++b)
                        acc[a][b][t] = __builtin_amdgcn_mfma_f32_32x32x2f32(av[a], bv[b], acc[a][b][t], 0, 0, 0);
            }
        }
    }
    // D[i = co][j = ci]: col = lane&31 = ci, row = (r&3)+8*(r>>2)+4*kk = co
#pragma unroll
    for (int a = 0; a < TCO; ++a)
#pragma unroll
        for (int b = 0; b < TCI; ++b) {
            const int ci = ci0 + (wci * TCI + b) * 32 + l31;
#pragma unroll
            for (int t = 0; t < T; ++t)
#pragma unroll
                for (int r = 0; r < 16; ++r) {
                    const int co = co0 + (wco * TCO + a) * 32 + (r & 3) + 8 * (r >> 2) + 4 * kk;
                    if (co < p.Cout && ci < Cin) atomicAdd(&p.gwt[((long)t * p.Cout + co) * Cin + ci], acc[a][b][t][r]);
                }
        }
}

template <int KS, int WCO, int WCI, int TCO, int TCI, int BPX>
static void launch_wgrad_bpx(WgradParams& p, hipStream_t s) {
    constexpr int BCO = 32 * TCO * WCO, BCI = 32 * TCI * WCI, PAD = KS / 2;
    tile_geometry(p.H, p.W, BPX, &p.TWp, &p.TH, &p.TF);
    p.tw_shift = ilog2(p.TWp);
    p.th_shift = ilog2(p.TH);
    p.n_wtiles = ceil_div(p.W, p.TWp);
    p.n_htiles = ceil_div(p.H, p.TH);
    p.n_ftiles = ceil_div(p.N, p.TF);
    p.n_pix_tiles = p.n_wtiles * p.n_htiles * p.n_ftiles;
    int IMG = p.TF * (p.TH + 2 * PAD) * (p.TWp + 2 * PAD);
    p.P2 = next_pow2(IMG) < 256 ? next_pow2(IMG) : 256;
    p.P2_shift = ilog2(p.P2);
    size_t lds = ((size_t)BCO * (BPX + 1) + (size_t)BCI * (IMG | 1)) * 4;
    int tiles = ceil_div(p.Cout, BCO) * ceil_div(p.C1 + p.C2, BCI);
    int S = 1024 / tiles;
    if (S < 1) S = 1;
    if (S > p.n_pix_tiles) S = p.n_pix_tiles;
    auto kern = wgrad_mfma_kernel<KS, WCO, WCI, TCO, TCI, BPX>;
    if (lds > 65536) (void)hipFuncSetAttribute((const void*)kern, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
    dim3 grid(S, ceil_div(p.C1 + p.C2, BCI), ceil_div(p.Cout, BCO));
    hipLaunchKernelGGL(kern, grid, dim3(256), lds, s, p);
}

// tiny maps (<= 4x4) use 64-pixel tiles: their zero-padded LDS image is 2.25-4x the tile, and 128 would not fit.
template <int KS, int WCO, int WCI, int TCO, int TCI>
static void launch_wgrad(WgradParams& p, hipStream_t s) {
    // 64-pixel stages everywhere: the G+X stage then takes <= ~66 KB of LDS, so 2-3 workgroups share a CU and one's
    // staging overlaps another's MFMAs (128-pixel stages left a single workgroup per CU: 57 -> 70 TFLOP/s on conv2)
    static int use128 = getenv("RFN_WGRAD_BPX128") ? atoi(getenv("RFN_WGRAD_BPX128")) : 0;
    if (p.H * p.W <= 16 || !use128)
        launch_wgrad_bpx<KS, WCO, WCI, TCO, TCI, 64>(p, s);
    else
        launch_wgrad_bpx<KS, WCO, WCI, TCO, TCI, 128>(p, s);
}

extern "C" int rfn_conv2d_wgrad_f32(const float* in1, long in1_ns, int C1, const float* in2, long in2_ns, int C2,
                                    const float* g, long g_ns, int Cout, float* gwt, int N, int H, int W, int ks,
                                    rfn_stream_t stream) {
    RFN_CHECK_ARG(in1 && g && gwt && C1 > 0 && C2 >= 0 && Cout > 0 && N >= 0 && H > 0 && W > 0, -1);
    RFN_CHECK_ARG(ks == 1 || ks == 3, -2);
    RFN_CHECK_ARG(C2 == 0 || in2, -3);
    if (N == 0) return 0;
    WgradParams p;
    memset(&p, 0, sizeof(p));
    p.in1 = in1; p.in2 = in2; p.in1_ns = in1_ns; p.in2_ns = in2_ns; p.C1 = C1; p.C2 = C2;
    p.g = g; p.g_ns = g_ns; p.Cout = Cout; p.gwt = gwt; p.N = N; p.H = H; p.W = W;
    hipStream_t s = (hipStream_t)stream;
    const int Cin = C1 + C2;
    if (ks == 3) {
        if (Cin <= 32)
            launch_wgrad<3, 4, 1, 1, 1>(p, s);   // 128 co x 32 ci
        else if (Cout <= 32)
            launch_wgrad<3, 1, 4, 1, 1>(p, s);   // 32 co x 128 ci
        else
            launch_wgrad<3, 2, 2, 1, 1>(p, s);   // 64 x 64
    } else {
        if (Cin <= 32)
            launch_wgrad<1, 4, 1, 2, 1>(p, s);   // 256 co x 32 ci
        else if (Cout <= 32)
            launch_wgrad<1, 1, 4, 1, 2>(p, s);   // 32 co x 256 ci
        else if (Cout <= 64)
            launch_wgrad_bpx<1, 1, 4, 2, 2, 64>(p, s);  // 64 co x 256 ci (tap-expanded conv3 at level 0: 36 rows)
        else
            launch_wgrad<1, 2, 2, 2, 2>(p, s);   // 128 x 128
    }
    RFN_LAUNCH_CHECK();
    return 0;
}

// gw[co][ci][tap] (+)= gwt[tap][co][ci]
__global__ void wgrad_finish_kernel(const float* __restrict__ gwt, float* __restrict__ gw, int Cout, int Cin, int T,
                                    int accumulate) {
    const long total = (long)Cout * Cin * T;
    for (long idx = (long)blockIdx.x * blockDim.x + threadIdx.x; idx < total; idx += (long)gridDim.x * blockDim.x) {
        int t = (int)(idx % T);
        long r = idx / T;  // co*Cin + ci
        float v = gwt[(long)t * Cout * Cin + r];
        gw[idx] = accumulate ? gw[idx] + v : v;
    }
}
extern "C" int rfn_wgrad_finish_f32(const float* gwt, float* gw, int Cout, int Cin, int ks, int accumulate,
                                    rfn_stream_t stream) {
    RFN_CHECK_ARG(gwt && gw && Cout > 0 && Cin > 0 && (ks == 1 || ks == 3), -1);
    long total = (long)Cout * Cin * ks * ks;
    int grid = (int)((total + 255) / 256 < 2048 ? (total + 255) / 256 : 2048);
    hipLaunchKernelGGL(wgrad_finish_kernel, dim3(grid), dim3(256), 0, (hipStream_t)stream, gwt, gw, Cout, Cin, ks * ks,
                       accumulate);
    RFN_LAUNCH_CHECK();
    return 0;
}


// ------------------------------------------------------------------------------------------------ few input channels
// 3x3 / pad 1 convolution of an image with 1 .. 4 channels into 16 or 32 feature maps (the first convolution of the
// frame extractor, Utils/modules.py VGG block on the 64x64 frames): 9 Cin products per output value.  As an MFMA problem
// it is a K = 9 .. 36 contraction padded to 16 / 48 with a 32-row tile around 16 outputs (0.32 ms forward, 0.35 ms for
// the weight gradient at 640 frames, both at a tenth of the bytes' time); as plain fp32 FMAs it is exact and at the HBM
// rate: one thread per pixel, the weights as 16-byte broadcast reads from LDS ([ci][tap][co]).
template <int CO>
__global__ __launch_bounds__(256) void conv3x3_fewcin_fwd_kernel(const float* __restrict__ in, long in_ns, int Cin,
                                                                 const float* __restrict__ w, float* __restrict__ out,
                                                                 long out_ns, int N, int H, int W) {
    __shared__ __attribute__((aligned(16))) float ws[4 * 9 * CO];
    for (int e = threadIdx.x; e < Cin * 9 * CO; e += 256) {
        const int co = e % CO, r = e / CO, tap = r % 9, ci = r / 9;
        ws[e] = w[((long)co * Cin + ci) * 9 + tap];
    }
    __syncthreads();
    const long HW = (long)H * W, total = (long)N * HW;
    for (long q = (long)blockIdx.x * 256 + threadIdx.x; q < total; q += (long)gridDim.x * 256) {
        const long n = q / HW;
        const int p = (int)(q - n * HW), y = p / W, x = p - y * W;
        float acc[CO];
#pragma unroll
        for (int c = 0; c < CO; ++c) acc[c] = 0.f;
        for (int ci = 0; ci < Cin; ++ci) {
            const float* src = in + n * in_ns + (long)ci * HW;
#pragma unroll
            for (int t = 0; t < 9; ++t) {
                const int yy = y + t / 3 - 1, xx = x + t % 3 - 1;
                const float v = (yy >= 0 && yy < H && xx >= 0 && xx < W) ? src[(long)yy * W + xx] : 0.f;
                const f32x4* w4 = reinterpret_cast<const f32x4*>(ws + (ci * 9 + t) * CO);
#pragma unroll
                for (int c4 = 0; c4 < CO / 4; ++c4) {
                    const f32x4 wv = w4[c4];
                    acc[4 * c4 + 0] = fmaf(wv[0], v, acc[4 * c4 + 0]);
                    acc[4 * c4 + 1] = fmaf(wv[1], v, acc[4 * c4 + 1]);
                    acc[4 * c4 + 2] = fmaf(wv[2], v, acc[4 * c4 + 2]);
                    acc[4 * c4 + 3] = fmaf(wv[3], v, acc[4 * c4 + 3]);
                }
            }
        }
        float* dst = out + n * out_ns + p;
#pragma unroll
        for (int c = 0; c < CO; ++c) dst[(long)c * HW] = acc[c];
    }
}

extern "C" int rfn_conv3x3_fewcin_supported(int Cin, int Cout) { return Cin >= 1 && Cin <= 4 && (Cout == 16 || Cout == 32); }

extern "C" int rfn_conv3x3_fewcin_fwd_f32(const float* in, long in_ns, int Cin, const float* w, float* out, long out_ns,
                                          int Cout, int N, int H, int W, rfn_stream_t stream) {
    RFN_CHECK_ARG(in && w && out && N >= 0 && H > 0 && W > 0, -1);
    RFN_CHECK_ARG(rfn_conv3x3_fewcin_supported(Cin, Cout), -2);
    if (N == 0) return 0;
    const long tot = (long)N * H * W;
    const int grid = (int)((tot + 255) / 256 < 8192 ? (tot + 255) / 256 : 8192);
    if (Cout == 16)
        hipLaunchKernelGGL(conv3x3_fewcin_fwd_kernel<16>, dim3(grid), dim3(256), 0, (hipStream_t)stream, in, in_ns, Cin, w, out,
                           out_ns, N, H, W);
    else
        hipLaunchKernelGGL(conv3x3_fewcin_fwd_kernel<32>, dim3(grid), dim3(256), 0, (hipStream_t)stream, in, in_ns, Cin, w, out,
                           out_ns, N, H, W);
    RFN_LAUNCH_CHECK();
    return 0;
}

// Weight gradient of the same convolution for ONE input channel and 16 outputs: gw[co][tap] = sum over frames and pixels
// of g[co][px] * in[px + tap] -- 144 running sums per thread over a grid-stride sweep, wave sums by DPP, one LDS row per
// wave, 144 float atomics per workgroup (gw zeroed by the caller).
__global__ __launch_bounds__(256) void conv3x3_c1_wgrad16_kernel(const float* __restrict__ in, long in_ns,
                                                                 const float* __restrict__ g, long g_ns,
                                                                 float* __restrict__ gw, int N, int H, int W) {
    __shared__ float red[4][144];
    float acc[16][9];
#pragma unroll
    for (int c = 0; c < 16; ++c)
#pragma unroll
        for (int t = 0; t < 9; ++t) acc[c][t] = 0.f;
    const long HW = (long)H * W, total = (long)N * HW;
    for (long q = (long)blockIdx.x * 256 + threadIdx.x; q < total; q += (long)gridDim.x * 256) {
        const long n = q / HW;
        const int p = (int)(q - n * HW), y = p / W, x = p - y * W;
        const float* src = in + n * in_ns;
        float v[9];
#pragma unroll
        for (int t = 0; t < 9; ++t) {
            const int yy = y + t / 3 - 1, xx = x + t % 3 - 1;
            v[t] = (yy >= 0 && yy < H && xx >= 0 && xx < W) ? src[(long)yy * W + xx] : 0.f;
        }
        const float* gp = g + n * g_ns + p;
#pragma unroll
        for (int c = 0; c < 16; ++c) {
            const float gv = gp[(long)c * HW];
#pragma unroll
            for (int t = 0; t < 9; ++t) acc[c][t] = fmaf(gv, v[t], acc[c][t]);
        }
    }
    const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63;
#pragma unroll
    for (int c = 0; c < 16; ++c)
#pragma unroll
        for (int t = 0; t < 9; ++t) {
            const float sum = wave_sum_dpp(acc[c][t]);
            if (lane == 0) red[wave][c * 9 + t] = sum;
        }
    __syncthreads();
    if (threadIdx.x < 144)
        atomicAdd(&gw[threadIdx.x], (red[0][threadIdx.x] + red[1][threadIdx.x]) + (red[2][threadIdx.x] + red[3][threadIdx.x]));
}

extern "C" int rfn_conv3x3_c1_wgrad16_f32(const float* in, long in_ns, const float* g, long g_ns, float* gw, int N, int H,
                                          int W, rfn_stream_t stream) {
    RFN_CHECK_ARG(in && g && gw && N >= 0 && H > 0 && W > 0, -1);
    if (N == 0) return 0;
    const long tot = (long)N * H * W;
    const int grid = (int)((tot + 255) / 256 < 1024 ? (tot + 255) / 256 : 1024);
    hipLaunchKernelGGL(conv3x3_c1_wgrad16_kernel, dim3(grid), dim3(256), 0, (hipStream_t)stream, in, in_ns, g, g_ns, gw, N, H, W);
    RFN_LAUNCH_CHECK();
    return 0;
}
