// 3x3 convolution with FEW output channels (<= 64) and many input channels on 32x32 / 16x16 maps, bf16x3 arithmetic:
// the data gradient of the first coupling-net convolution at the two finest flow levels (256 -> 18 / 36 channels,
// Flow/glow_modules.py:232-238 backwards), where the generic implicit-GEMM kernel spends its time re-reading shifted
// input fragments from LDS for a single 32-row output tile.
//
// GEMM orientation D[cout][pixel] as everywhere else; K = (input-channel chunk of 16) x (9 taps).
//   * a block owns a band of 16 rows of one frame (4 waves x R = 4 rows); a wave owns its 4 rows as TPX tiles of 32
//     pixels (one row at W = 32, two rows at W = 16) and ALL output channels (MT tiles of 32 rows);
//   * per 16-channel chunk the haloed band (18 rows x (W+2) columns) is staged in LDS pre-split into bf16 hi / lo as
//     [plane][8-channel group][position][8 x bf16]: an MFMA B fragment of any tap is one ds_read_b128 at a shifted
//     position.  A fragment of input row i serves the output rows i-1, i, i+1 (taps dy = +1, 0, -1), so a wave reads
//     (rows + 2) x 3 fragments per chunk instead of rows x 9;
//   * the weights of a chunk (9 taps x MT tiles, hi and lo) arrive by LDS-DMA straight from the pack buffer the generic
//     kernel uses (rfn_pack_conv_weight_bf16x3 layout), already in fragment order;
//   * three-stage software pipeline over the chunks: while the MFMAs of chunk c run, the registers loaded during chunk
//     c-1 (chunk c+1's image: 8 channels x 4 pixels per thread, dwordx4 buffer loads whose out-of-frame lanes read
//     zeros) are split and written to the other LDS buffer and chunk c+2 is being loaded; the pieces are spread evenly
//     over the chunk's (dx, input row) steps and the compiler interleaves them with the MFMAs.
// Measured (N = 608): 256 -> 18 at 32x32 0.36 ms (1.9 TB/s algorithmic) against 0.40-0.48 ms of the generic kernel;
// the loop is bound by instruction issue (each vector-memory wave instruction costs ~100 cycles beside the MFMAs:
// MI355X_MICROARCH.md, cycle constants), not by HBM or the matrix pipe -- see DESIGN.md.
#include "conv_common.h"
#include <type_traits>

typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));

struct DgradSmallParams {
    const float* in;   // [N, Cin, H, W] (frame stride in_ns)
    long in_ns;
    const unsigned char* wpk;  // rfn_pack_conv_weight_bf16x3 buffer of the [Cout][Cin][3][3] logical weight
    float* out1;
    float* out2;
    long out1_ns, out2_ns;
    int Cin, Cout, cout_split, acc1, acc2, CoutP;
    int N, H;
};

#define DS_MFMA(acc, a, b) acc = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a, b, acc, 0, 0, 0)

template <int W, int MT, int TPX>
__global__ __launch_bounds__(256) void dgrad_small_kernel(const DgradSmallParams p) {
    constexpr int RPT = 32 / W;           // rows per 32-pixel tile
    constexpr int R = TPX * RPT;          // rows per wave
    constexpr int BR = 4 * R;             // rows per block
    constexpr int IW = W + 2, IH = BR + 2, IPOS = IW * IH;
    constexpr int IMG_BYTES = 2 * 2 * IPOS * 16;         // [plane][group][pos] x 16 B
    constexpr int WFR = 9 * 2 * MT;                        // 1-KB weight fragments per chunk: [tap][plane][mt]
    constexpr int WGT_BYTES = WFR * 1024;
    constexpr int QPR = W / 4, NITEMS = 2 * IH * QPR;      // staging items: (group, image row, quad of 4 pixels)
    constexpr int ITEMS = (NITEMS + 255) / 256;            // per thread (the last round may cover only the first waves)
    constexpr int NFI = R + 2 - (RPT - 1);                 // distinct input fragment rows per wave: i = -1 .. R - RPT + 1
    constexpr int NSTEP = 3 * NFI;                         // (dx, input fragment row) steps of a chunk
    static_assert(ITEMS * 8 <= NSTEP && ITEMS * 4 <= NSTEP, "staging pieces must fit the compute steps");
    extern __shared__ __attribute__((aligned(16))) unsigned char lds[];
    // lds: image[2] | weights[2]
    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int l31 = lane & 31, kk = lane >> 5;
    const int HW = p.H * W;
    const int bands = p.H / BR;
    const int n = blockIdx.x / bands, rb = (blockIdx.x - n * bands) * BR;
    const int nchunks = p.Cin >> 4;
    // bounds-checked buffer over the whole input (descriptor from kernel arguments only: wave-uniform by construction);
    // masked lanes point past the end and read zeros.  The host guarantees N * in_ns * 4 < 0xFFFFFF00.
    const auto rs_in = __builtin_amdgcn_make_buffer_rsrc(const_cast<float*>(p.in), 0,
                                                         (unsigned)((long)p.N * p.in_ns * 4), 0x00020000);
    const unsigned frame_off = (unsigned)((long)n * p.in_ns * 4);

    // ---- staging items of this thread.  8 channels x 4 pixels arrive as 8 dwordx4 loads and leave as 4 (hi, lo) pairs of
    // 16-byte LDS units; the two halo columns are zeroed once below and never written again.
    unsigned s_off[ITEMS];  // byte offset of (channel 8g, row, quad) inside the frame's chunk 0 (0xFFFFFF00: masked)
    int s_dst[ITEMS];       // 16-B unit index inside a plane of the first pixel: g * IPOS + iy * IW + 1 + 4 q
    constexpr int PADU = (2 * IMG_BYTES + 2 * WGT_BYTES) / 16;  // 8 scratch units behind the buffers: dead items write there
    int s_lo[ITEMS], s_bufu[ITEMS];  // lo-plane unit index; units to add for image buffer 1 (0 for dead items)
#pragma unroll
    for (int it = 0; it < ITEMS; ++it) {
        const int item = tid + it * 256;
        const bool live = item < NITEMS;
        const int g = live ? item / (IH * QPR) : 0, rem = live ? item - g * (IH * QPR) : 0;
        const int iy = rem / QPR, q = rem - iy * QPR;
        const int gy = rb - 1 + iy;
        const bool ok = live && gy >= 0 && gy < p.H;
        s_off[it] = ok ? frame_off + (unsigned)((g * 8 * HW + gy * W + 4 * q) * 4) : 0xFFFFFF00u;
        s_dst[it] = live ? g * IPOS + iy * IW + 1 + 4 * q : PADU;
        s_lo[it] = live ? s_dst[it] + 2 * IPOS : PADU + 4;
        s_bufu[it] = live ? IMG_BYTES / 16 : 0;
    }
    typedef unsigned u32x4_ __attribute__((ext_vector_type(4)));
    // load j (channel) of item `it` of chunk c
    auto img_load1 = [&](f32x4 (&raw)[ITEMS][8], const int c, const int it, const int j) {
        const unsigned off = s_off[it] == 0xFFFFFF00u ? 0xFFFFFF00u : s_off[it] + (unsigned)((c * 16 + j) * HW * 4);
        const u32x4_ v = __builtin_amdgcn_raw_buffer_load_b128(rs_in, off, 0, 0);
        raw[it][j] = __builtin_bit_cast(f32x4, v);
    };
    // pixel e of item `it`: split into bf16 hi / lo, one 16-byte unit each
    auto img_store1 = [&](const f32x4 (&raw)[ITEMS][8], const int buf, const int it, const int e) {
        bf16x8* img = reinterpret_cast<bf16x8*>(lds) + buf * s_bufu[it] + e;
        bf16x8 hi, lo;
#pragma unroll
        for (int j = 0; j < 8; ++j) {
            const float v = raw[it][j][e];
            const __bf16 h = (__bf16)v;
            hi[j] = h;
            lo[j] = (__bf16)(v - (float)h);
        }
        img[s_dst[it]] = hi;
        img[s_lo[it]] = lo;
    };
    // weights of chunk c -> LDS buffer `buf`: fragment f = (tap*2 + plane)*MT + mt, lane = (row, kk) reads the pack unit
    // (((c*9 + tap)*2 + plane)*2 + kk)*CoutP + mt*32 + row
    auto wgt_dma = [&](const int c, const int buf) {
        unsigned char* dst0 = lds + 2 * IMG_BYTES + buf * WGT_BYTES;
        for (int f = wave; f < WFR; f += 4) {
            const int mt = f % MT, tp = f / MT;  // tp = tap*2 + plane
            const unsigned char* src = p.wpk + ((((long)c * 9 * 2 + tp) * 2 + kk) * p.CoutP + mt * 32 + l31) * 16;
            __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)src,
                                             (__attribute__((address_space(3))) void*)(dst0 + f * 1024), 16, 0, 0);
        }
    };
    {   // zero both image buffers once (halo columns stay zero)
        f32x4* z4 = reinterpret_cast<f32x4*>(lds);
        for (int i = tid; i < 2 * IMG_BYTES / 16; i += 256) z4[i] = f32x4{0.f, 0.f, 0.f, 0.f};
        __syncthreads();
    }

    f32x16 acc[TPX][MT];
#pragma unroll
    for (int t = 0; t < TPX; ++t)
#pragma unroll
        for (int m = 0; m < MT; ++m)
#pragma unroll
            for (int r = 0; r < 16; ++r) acc[t][m][r] = 0.f;

    // this lane's position of (input fragment row i = -1, dx = -1) inside the staged band: pixel l31 of a tile
    const int prow = l31 / W, pcol = l31 - prow * W;
    const int pos0 = (wave * R + prow) * IW + pcol;  // + (i + 1) * IW + (dx + 1)

    // ---- three-stage pipeline over the 16-channel chunks: while the MFMAs of chunk c run, the registers loaded during
    // chunk c-1 (chunk c+1's image) are split and written to the other LDS buffer, and chunk c+2 is being loaded.
    f32x4 rawA[ITEMS][8], rawB[ITEMS][8];
    wgt_dma(0, 0);
#pragma unroll
    for (int it = 0; it < ITEMS; ++it)
#pragma unroll
        for (int j = 0; j < 8; ++j) img_load1(rawA, 0, it, j);
#pragma unroll
    for (int it = 0; it < ITEMS; ++it)
#pragma unroll
        for (int j = 0; j < 8; ++j) img_load1(rawB, 1, it, j);
#pragma unroll
    for (int it = 0; it < ITEMS; ++it)
#pragma unroll
        for (int e = 0; e < 4; ++e) img_store1(rawA, 0, it, e);
    asm volatile("s_waitcnt vmcnt(0) lgkmcnt(0)" ::: "memory");
    __syncthreads();

    // one chunk: rawC holds chunk c+1 (to be stored into the other buffer), rawN receives chunk c+2
    auto chunk = [&](auto st_, auto ld_, const int c, f32x4 (&rawC)[ITEMS][8], f32x4 (&rawN)[ITEMS][8]) {
        constexpr bool st = decltype(st_)::value, ld = decltype(ld_)::value;  // is there a chunk c+1 / c+2
        const int buf = c & 1;
        if (st) wgt_dma(c + 1, buf ^ 1);
        const bf16x8* img = reinterpret_cast<const bf16x8*>(lds + buf * IMG_BYTES) + kk * IPOS + pos0;
        const bf16x8* wl = reinterpret_cast<const bf16x8*>(lds + 2 * IMG_BYTES + buf * WGT_BYTES) + lane;
#pragma unroll
        for (int dx = 0; dx < 3; ++dx) {
            bf16x8 A[3][MT][2];  // [dy][mt][plane]
#pragma unroll
            for (int dy = 0; dy < 3; ++dy)
#pragma unroll
                for (int m = 0; m < MT; ++m)
#pragma unroll
                    for (int pl = 0; pl < 2; ++pl) A[dy][m][pl] = wl[((((dy * 3 + dx) * 2 + pl) * MT) + m) * 64];
            bf16x8 B[2][2];
            B[0][0] = img[dx];
            B[0][1] = img[2 * IPOS + dx];
#pragma unroll
            for (int fi = 0; fi < NFI; ++fi) {  // input fragment first row i = fi - 1 (relative to the wave's rows)
                const int cur = fi & 1, step = dx * NFI + fi;
                if (fi + 1 < NFI) {
                    B[cur ^ 1][0] = img[(fi + 1) * IW + dx];
                    B[cur ^ 1][1] = img[2 * IPOS + (fi + 1) * IW + dx];
                }
                __builtin_amdgcn_sched_barrier(0);
                // the (up to) 3 x MT accumulators this fragment feeds take turns: no back-to-back dependent MFMAs
#pragma unroll
                for (int combo = 0; combo < 3; ++combo) {
#pragma unroll
                    for (int dy = 0; dy < 3; ++dy) {
                        const int ro = (fi - 1) - (dy - 1);  // output tile's first row
                        if (ro >= 0 && ro <= (TPX - 1) * RPT && ro % RPT == 0) {
#pragma unroll
                            for (int m = 0; m < MT; ++m) {
                                if (combo == 0) DS_MFMA(acc[ro / RPT][m], A[dy][m][1], B[cur][0]);
                                if (combo == 1) DS_MFMA(acc[ro / RPT][m], A[dy][m][0], B[cur][1]);
                                if (combo == 2) DS_MFMA(acc[ro / RPT][m], A[dy][m][0], B[cur][0]);
                            }
                        }
                    }
                }
                // staging pieces in the shadow of those MFMAs: one pixel of chunk c+1 (split + LDS write) and one channel
                // load of chunk c+2 (into the other register set)
                // (spread evenly over the NSTEP steps: piece k of NP goes to step floor(k * NSTEP / NP))
                {
                    constexpr int NPS = ITEMS * 4, NPL = ITEMS * 8;
                    const int ks = (step * NPS + NSTEP - 1) / NSTEP;  // first store piece with slot >= step
                    if (st && ks < NPS && (ks * NSTEP) / NPS == step) img_store1(rawC, buf ^ 1, ks / 4, ks % 4);
                    const int kl = (step * NPL + NSTEP - 1) / NSTEP;
                    if (ld && kl < NPL && (kl * NSTEP) / NPL == step) img_load1(rawN, c + 2, kl / 8, kl % 8);
                }
                __builtin_amdgcn_sched_barrier(0);
            }
        }
        // my share of chunk c+1's weights (DMA) has landed when at most the image loads issued after it are in flight
        if (ld) asm volatile("s_waitcnt vmcnt(%0)" ::"n"(ITEMS * 8) : "memory");
        else asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
        __builtin_amdgcn_s_barrier();
    };
    {   // (nchunks is even: the host requires Cin % 32 == 0)
        using T_ = std::integral_constant<bool, true>;
        using F_ = std::integral_constant<bool, false>;
        int c = 0;
        for (; c + 3 < nchunks; c += 2) {
            chunk(T_{}, T_{}, c, rawB, rawA);
            chunk(T_{}, T_{}, c + 1, rawA, rawB);
        }
        chunk(T_{}, F_{}, c, rawB, rawA);
        chunk(F_{}, F_{}, c + 1, rawA, rawB);
    }

    // ---- epilogue: D layout col = lane & 31 (pixel), row = (r & 3) + 8 (r >> 2) + 4 kk (channel).  Accumulating outputs
    // are read for a whole tile first (16 loads in flight), then added and stored.
#pragma unroll
    for (int t = 0; t < TPX; ++t) {
        const int y = rb + wave * R + t * RPT + prow;
        const long pix = (long)y * W + pcol;
#pragma unroll
        for (int m = 0; m < MT; ++m) {
            float* dst[16];
            float old[16];
#pragma unroll
            for (int r = 0; r < 16; ++r) {
                const int co = m * 32 + (r & 3) + 8 * (r >> 2) + 4 * kk;
                const bool first = co < p.cout_split;
                dst[r] = co >= p.Cout ? nullptr
                         : first ? p.out1 + n * p.out1_ns + (long)co * HW + pix
                                 : p.out2 + n * p.out2_ns + (long)(co - p.cout_split) * HW + pix;
                old[r] = (dst[r] && (first ? p.acc1 : p.acc2)) ? *dst[r] : 0.f;
            }
#pragma unroll
            for (int r = 0; r < 16; ++r)
                if (dst[r]) *dst[r] = acc[t][m][r] + old[r];
        }
    }
}

template <int W, int MT, int TPX>
static int launch_dgrad_small(const DgradSmallParams& p, hipStream_t st) {
    constexpr int RPT = 32 / W, R = TPX * RPT, BR = 4 * R, IPOS = (W + 2) * (BR + 2);
    constexpr size_t lds = 2 * (size_t)(2 * 2 * IPOS * 16) + 2 * (size_t)(9 * 2 * MT * 1024) + 128;
    static_assert(lds <= 160 * 1024, "LDS budget");
    auto* k = dgrad_small_kernel<W, MT, TPX>;
    if (lds > 65536) (void)hipFuncSetAttribute((const void*)k, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
    hipLaunchKernelGGL(k, dim3(p.N * (p.H / BR)), dim3(256), lds, st, p);
    return 0;
}

extern "C" int rfn_dgrad_small_supported(int N, int Cin, int Cout, int H, int W) {
    if (N <= 0 || Cin <= 0 || (Cin & 31) || Cout <= 0 || Cout > 64 || H != W) return 0;
    if ((long)N * Cin * H * W * 4 >= 0xFFFFFF00L) return 0;  // 32-bit buffer offsets
    if (!(W == 32 || W == 16)) return 0;
    return 1;
}

// out[n, co, y, x] = sum_ci sum_tap in[n, ci, y+dy, x+dx] * w'[co][ci][tap] with w' the logical weight the pack buffer was
// built from (mode 1 of the pack kernel = data gradient of a forward weight); output channels [0, cout_split) go to out1,
// the rest to out2; acc1 / acc2: add into the existing values.
extern "C" int rfn_conv3x3_smallcout_bf16x3(const float* in, long in_ns, int Cin, const float* wpk, float* out1,
                                            long out1_ns, float* out2, long out2_ns, int Cout, int cout_split, int acc1,
                                            int acc2, int N, int H, int W, rfn_stream_t stream) {
    RFN_CHECK_ARG(in && wpk && out1 && N >= 0, -1);
    RFN_CHECK_ARG(rfn_dgrad_small_supported(N > 0 ? N : 1, Cin, Cout, H, W), -2);
    RFN_CHECK_ARG(cout_split > 0 && cout_split <= Cout && (cout_split == Cout || out2), -3);
    RFN_CHECK_ARG(((uintptr_t)wpk & 15) == 0 && ((uintptr_t)in & 15) == 0 && (in_ns & 3) == 0, -4);
    RFN_CHECK_ARG((long)N * in_ns * 4 < 0xFFFFFF00L && in_ns >= (long)Cin * H * W, -5);
    if (N == 0) return 0;
    DgradSmallParams p;
    p.in = in; p.in_ns = in_ns; p.wpk = reinterpret_cast<const unsigned char*>(wpk);
    p.out1 = out1; p.out2 = out2; p.out1_ns = out1_ns; p.out2_ns = out2_ns;
    p.Cin = Cin; p.Cout = Cout; p.cout_split = cout_split; p.acc1 = acc1; p.acc2 = acc2;
    p.CoutP = ((Cout + 255) / 256) * 256;
    p.N = N; p.H = H;
    hipStream_t st = (hipStream_t)stream;
    const int MT = Cout <= 32 ? 1 : 2;
    if (W == 32) {
        if (MT == 1) launch_dgrad_small<32, 1, 4>(p, st);
        else launch_dgrad_small<32, 2, 4>(p, st);
    } else {
        if (MT == 1) launch_dgrad_small<16, 1, 2>(p, st);
        else launch_dgrad_small<16, 2, 2>(p, st);
    }
    RFN_LAUNCH_CHECK();
    return 0;
}
