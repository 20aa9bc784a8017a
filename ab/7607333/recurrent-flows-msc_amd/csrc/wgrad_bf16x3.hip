// Weight gradients as one split-precision GEMM:  gw[m][n] += Σ_{frames, pixels} A[f][m][p] * B[f][n][p].
//
// Every weight gradient of the path is brought to this 1x1 form by expanding the SMALL operand in HBM:
//   * 1x1 conv:                 A = grad [M = Cout],       B = input [N = Cin]
//   * 3x3 conv, Cin <= Cout:    A = grad,                  B = im2col3x3(input) [N = 9*Cin]   (rfn_im2col3x3_f32)
//   * 3x3 conv, Cout <  Cin:    A = tap_scatter(grad) [M = 9*Cout] (rfn_tap_scatter_f32),     B = input
// so the shifted-window reads of a 3x3 weight gradient never reach this kernel and the 8 consecutive k (pixels) an
// MFMA operand lane needs are 8 consecutive fp32 of one channel plane in NCHW: staged with 16-byte loads, split into
// bf16 hi/lo (see conv_bf16x3.hip) and stored as [row][k-group] 16-byte units with an odd row stride, so both fragment
// reads are conflict-free ds_read_b128.  K (all pixels of all frames) is split over gridDim.x; a workgroup sweeps its
// stages with register prefetch and emits float atomics once.
#include "conv_common.h"

typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));

struct GemmWgradParams {
    const float* a;
    const float* b;
    long a_ns, b_ns;
    int M, N;
    float* gw;  // [M][N]
    int F, HW;  // frames, pixels per frame (HW % 4 == 0)
    long total; // F*HW
    int n_stages;
    // implicit 3x3 mode (IMPL = 1): operand B row n = ci*9 + tap is the input plane ci shifted by the tap, read straight
    // from the convolution's (two-source) input -- b = in1, b2 = in2 -- instead of from an im2col buffer in HBM
    const float* b2;
    long b2_ns;
    int C1, C2, H, W;
    // grouped launch (G > 0): G independent gradients of the SAME shape in one launch -- the K steps of a flow level at
    // the deep levels / small batches, where a single gradient is a latency-class problem.  blockIdx.z = grp * mtiles + mt.
    int G, mtiles;
    const float* ga[16];
    const float* gb[16];
    const float* gb2[16];
    float* ggw[16];
};

// operands of group `grp` of a grouped launch (or the single gradient's).  The group tables are read with
// COMPILE-TIME indices: a run-time index into the by-value parameter struct keeps the whole struct in scratch memory
// (632 bytes per lane, every field read through it -- what every kernel of this file did until round 3).
struct WgOperands { const float* a; const float* b; const float* b2; float* gw; };
__device__ __forceinline__ WgOperands wg_operands(const GemmWgradParams& p, const int grp) {
    WgOperands o = {p.a, p.b, p.b2, p.gw};
    if (p.G > 0) {
#pragma unroll
        for (int i = 0; i < 16; ++i)
            if (i == grp) {
                o.a = p.ga[i];
                o.b = p.gb[i];
                o.b2 = p.gb2[i];
                o.gw = p.ggw[i];
            }
    }
    return o;
}

// WM x WN waves (4 or 8) of TM x TN 32x32 tiles each.  The 8-wave 256-row configurations read both operands of the
// big level-0 / level-1 gradients exactly once (a 128 x 128 tiling of a 256 x 256 gradient reads each of them twice,
// and those launches sit on the HBM roof).
template <int WM, int WN, int TM, int TN, int KP, int IMPL = 0>
__global__ __launch_bounds__(64 * WM * WN) void gemm_wgrad_b3_kernel(const GemmWgradParams p) {
    constexpr int NT = 64 * WM * WN;
    constexpr int BM = 32 * TM * WM, BN = 32 * TN * WN;
    constexpr int NU = KP / 8;        // 16-byte units (8 pixels) per row and stage
    constexpr int RS = NU + 1;        // odd-ish row stride in units -> conflict-free b128 fragment reads
    constexpr int AU = (BM * NU + NT - 1) / NT;  // units staged per thread (A)
    constexpr int BU = (BN * NU + NT - 1) / NT;
    constexpr bool AX = (BM * NU) % NT == 0, BX = (BN * NU) % NT == 0;  // exact: no tail guard
    static_assert(WM * WN == 4 || WM * WN == 8, "tile/stage shape");
    extern __shared__ __attribute__((aligned(16))) unsigned char lds_raw[];
    bf16x8* Ah = reinterpret_cast<bf16x8*>(lds_raw);
    bf16x8* Al = Ah + BM * RS;
    bf16x8* Bh = Al + BM * RS;
    bf16x8* Bl = Bh + BN * RS;
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int wm = wave / WN, wn = wave % WN;
    const int l31 = lane & 31, kk = lane >> 5;
    const int grp = p.G > 0 ? (int)blockIdx.z / p.mtiles : 0;
    const int m0 = (p.G > 0 ? (int)blockIdx.z - grp * p.mtiles : (int)blockIdx.z) * BM, n0 = blockIdx.y * BN;
    const WgOperands ops_ = wg_operands(p, grp);
    const float* const pa_ = ops_.a;
    const float* const pb_ = ops_.b;
    const float* const pb2_ = ops_.b2;
    float* const pgw_ = ops_.gw;

    f32x16 acc[TM][TN];
#pragma unroll
    for (int i = 0; i < TM; ++i)
#pragma unroll
        for (int j = 0; j < TN; ++j)
#pragma unroll
            for (int r = 0; r < 16; ++r) acc[i][j][r] = 0.f;

    // staging: unit e = tid + NT*u  ->  row = e / NU, k-group = e % NU   (consecutive threads = consecutive pixels)
    float4 ast[AU][2], bst[BU][2];
    // All units of a thread share the k-group (NT % NU == 0), so the (frame, pixel) split of a stage's two 4-pixel
    // groups is computed once per stage and thread -- the only divisions of the loop (32-bit: total < 2^31 is checked
    // on the host).  A group never straddles a frame because HW % 4 == 0.
    static_assert(NT % NU == 0, "k-group must be a per-thread constant");
    const int kg = tid % NU, r0 = tid / NU;   // unit u of this thread: row r0 + u*(NT/NU), k-group kg
    const unsigned uHW = (unsigned)p.HW;
    // The loads of the NEXT stage are issued ahead of the MFMAs of the current one and are not touched until the next
    // commit: unconditional (clamped rows / pixels, no select on a loaded value, no control flow) -- otherwise the
    // compiler waits for HBM right behind the loads and nothing overlaps.  Masks are applied in commit().
    bool okq[2] = {false, false};
    // implicit mode: per unit (fixed row n = tap*Cin + ci) the plane pointer, frame stride and tap offsets; per stage the
    // 8 pixels of a unit lie in one image row (W % 8 == 0), shifted by dx they need one element beyond either end
    const float* uplane[IMPL ? BU : 1];
    long uns[IMPL ? BU : 1];
    int udy[IMPL ? BU : 1], udx[IMPL ? BU : 1];
    float bedge[IMPL ? BU : 1];
    unsigned rowmask = 0, edgemask = 0;
    if (IMPL) {
#pragma unroll
        for (int u = 0; u < BU; ++u) {
            int row = n0 + r0 + u * (NT / NU);
            if (row >= p.N) row = 0;
            // row = ci*9 + tap: the output [Cout][9*Cin] IS the torch weight layout [Cout][Cin][3][3]
            const int ci = row / 9, tap = row - ci * 9;
            udy[u] = tap / 3 - 1;
            udx[u] = tap % 3 - 1;
            const bool first = ci < p.C1;
            uplane[u] = first ? pb_ + (long)ci * p.HW : pb2_ + (long)(ci - p.C1) * p.HW;
            uns[u] = first ? p.b_ns : p.b2_ns;
        }
    }
    auto prefetch = [&](int stage) {
        const unsigned q0 = (unsigned)stage * KP + 8u * kg;
        long offa[2], offb[2];
#pragma unroll
        for (int h = 0; h < 2; ++h) {
            const unsigned q = q0 + 4u * h;
            okq[h] = stage < p.n_stages && q < (unsigned)p.total;
            const unsigned qq = okq[h] ? q : 0u;
            const unsigned f = qq / uHW, px = qq - f * uHW;
            offa[h] = (long)f * p.a_ns + px;
            offb[h] = (long)f * p.b_ns + px;
        }
#pragma unroll
        for (int u = 0; u < AU; ++u) {
            const int row = m0 + r0 + u * (NT / NU);
            const float* base = pa_ + (long)(row < p.M ? row : 0) * p.HW;
#pragma unroll
            for (int h = 0; h < 2; ++h) ast[u][h] = *reinterpret_cast<const float4*>(base + offa[h]);
        }
        if (IMPL) {
            const unsigned qq = okq[0] ? q0 : 0u;
            const unsigned f = qq / uHW, pix = qq - f * uHW;
            const int y = (int)(pix / (unsigned)p.W), x0 = (int)(pix - (unsigned)y * (unsigned)p.W);
            rowmask = 0;
            edgemask = 0;
#pragma unroll
            for (int u = 0; u < BU; ++u) {
                const int yy = y + udy[u];
                const bool rok = yy >= 0 && yy < p.H;
                const float* src = uplane[u] + (long)f * uns[u] + (rok ? yy : y) * p.W + x0;
                bst[u][0] = *reinterpret_cast<const float4*>(src);
                bst[u][1] = *reinterpret_cast<const float4*>(src + 4);
                const bool eok = udx[u] < 0 ? x0 > 0 : x0 + 8 < p.W;  // the element beyond the end, inside the row?
                bedge[u] = src[eok ? (udx[u] < 0 ? -1 : 8) : 0];
                rowmask |= (rok ? 1u : 0u) << u;
                edgemask |= (eok ? 1u : 0u) << u;
            }
        } else {
#pragma unroll
            for (int u = 0; u < BU; ++u) {
                const int row = n0 + r0 + u * (NT / NU);
                const float* base = pb_ + (long)(row < p.N ? row : 0) * p.HW;
#pragma unroll
                for (int h = 0; h < 2; ++h) bst[u][h] = *reinterpret_cast<const float4*>(base + offb[h]);
            }
        }
    };
    auto split_store_shift = [&](const float4 (&src)[2], float edge, int dx, bool okr, bool eok, bf16x8* hi_p, bf16x8* lo_p) {
        const float v[8] = {src[0].x, src[0].y, src[0].z, src[0].w, src[1].x, src[1].y, src[1].z, src[1].w};
        bf16x8 hi, lo;
#pragma unroll
        for (int c = 0; c < 8; ++c) {
            const float left = c == 0 ? (eok ? edge : 0.f) : v[c > 0 ? c - 1 : 0];
            const float right = c == 7 ? (eok ? edge : 0.f) : v[c < 7 ? c + 1 : 7];
            float x = dx < 0 ? left : (dx > 0 ? right : v[c]);
            x = (okr && okq[0]) ? x : 0.f;
            const __bf16 h = (__bf16)x;
            hi[c] = h;
            lo[c] = (__bf16)(x - (float)h);
        }
        *hi_p = hi;
        *lo_p = lo;
    };
    auto split_store = [&](const float4 (&src)[2], bool okr, bf16x8* hi_p, bf16x8* lo_p) {
        const float v[8] = {src[0].x, src[0].y, src[0].z, src[0].w, src[1].x, src[1].y, src[1].z, src[1].w};
        bf16x8 hi, lo;
#pragma unroll
        for (int c = 0; c < 8; ++c) {
            const float x = (okr && okq[c >> 2]) ? v[c] : 0.f;
            const __bf16 h = (__bf16)x;
            hi[c] = h;
            lo[c] = (__bf16)(x - (float)h);
        }
        *hi_p = hi;
        *lo_p = lo;
    };
    auto commit = [&]() {
#pragma unroll
        for (int u = 0; u < AU; ++u) {
            const int o = (r0 + u * (NT / NU)) * RS + kg;
            if (AX || r0 + u * (NT / NU) < BM) split_store(ast[u], m0 + r0 + u * (NT / NU) < p.M, Ah + o, Al + o);
        }
#pragma unroll
        for (int u = 0; u < BU; ++u) {
            const int o = (r0 + u * (NT / NU)) * RS + kg;
            if (BX || r0 + u * (NT / NU) < BN) {
                const bool okr = n0 + r0 + u * (NT / NU) < p.N;
                if (IMPL)
                    split_store_shift(bst[u], bedge[u], udx[u], okr && ((rowmask >> u) & 1u), (edgemask >> u) & 1u, Bh + o,
                                      Bl + o);
                else
                    split_store(bst[u], okr, Bh + o, Bl + o);
            }
        }
    };

    const int arow = (wm * TM * 32 + l31) * RS + kk;  // + i*32*RS + 2*s
    const int brow = (wn * TN * 32 + l31) * RS + kk;
    int stage = blockIdx.x;
    prefetch(stage);
    for (; stage < p.n_stages; stage += gridDim.x) {
        __syncthreads();
        commit();
        __syncthreads();
        prefetch(stage + gridDim.x);        // always issued; past the end it re-reads pixel 0 and is never committed
        __builtin_amdgcn_sched_barrier(0);  // keep the loads ahead of the MFMAs (the scheduler sinks them)
        // 8-wave tiles with 64-pixel stages: one k-step of fragments live at a time (register budget 256)
#pragma unroll WM * WN == 8 && KP > 32 ? 1 : KP / 16
        for (int s = 0; s < KP / 16; ++s) {
            bf16x8 ah[TM], al[TM], bh[TN], bl[TN];
#pragma unroll
            for (int i = 0; i < TM; ++i) {
                ah[i] = Ah[arow + i * 32 * RS + 2 * s];
                al[i] = Al[arow + i * 32 * RS + 2 * s];
            }
#pragma unroll
            for (int j = 0; j < TN; ++j) {
                bh[j] = Bh[brow + j * 32 * RS + 2 * s];
                bl[j] = Bl[brow + j * 32 * RS + 2 * s];
            }
#pragma unroll
            for (int i = 0; i < TM; ++i)
#pragma unroll
                for (int j = 0; j < TN; ++j) {
                    acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(al[i], bh[j], acc[i][j], 0, 0, 0);
                    acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(ah[i], bl[j], acc[i][j], 0, 0, 0);
                    acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(ah[i], bh[j], acc[i][j], 0, 0, 0);
                }
        }
    }
    // D[i = m][j = n]: col = lane&31 = n (32 consecutive floats of a gw row = one 128-byte atomic segment)
#pragma unroll
    for (int i = 0; i < TM; ++i)
#pragma unroll
        for (int j = 0; j < TN; ++j) {
            const int n = n0 + (wn * TN + j) * 32 + l31;
#pragma unroll
            for (int r = 0; r < 16; ++r) {
                const int m = m0 + (wm * TM + i) * 32 + (r & 3) + 8 * (r >> 2) + 4 * kk;
                if (m < p.M && n < p.N) atomicAdd(&pgw_[(long)m * p.N + n], acc[i][j][r]);
            }
        }
}

// ------------------------------------------------------------------------------------------------ LDS-DMA variant
// The big gradients of the shallow flow levels (256 x 256 and 9C x 256 over 10^5 .. 10^6 pixels) read two 637 MB
// operands per launch and sit on the HBM roof -- if the loads are kept in flight.  The register-staged kernel above has
// ONE stage in flight per workgroup, and only while its MFMAs run: load -> convert -> LDS -> MFMA serialise into
// ~23 k cycles per 64-pixel stage against ~14 k of pure data movement.  Here the raw fp32 rows go HBM -> LDS by LDS-DMA
// (global_load_lds_dwordx4, no registers) into a ring of NST stages of KP pixels, two to three stages ahead of the
// MFMAs, and the fp32 -> bf16 (hi, lo) split moves to the fragment reads (each wave converts the fragments it consumes:
// 3x the conversions of a cooperative commit, issued in the shadow of the MFMAs).
//   * one DMA wave-instruction = 1 KB = 64 / UPR rows x UPR units of 16 B (UPR = KP / 4): coalesced row segments; LDS
//     image of a stage: [row][UPR units];
//   * the unit order inside a row is XOR-swizzled with (row / (16 / UPR)) & (UPR - 1) -- the LDS destination of a DMA
//     is linear in the lane, but WHICH global unit a lane fetches is free -- so the fragment reads (lane = row, two
//     ds_read_b128 per 8-pixel fragment) are conflict-free in the hardware's 16-lane groups (MI355X guide, LDS table);
//   * rows beyond M / N are clamped to the last valid row (their products are never written);
//   * ONE barrier per stage and one continuous software pipeline over all stages: the barrier at the top of stage t
//     certifies that everybody's pieces of stage t + 1 have landed (its first fragments are read during stage t) and
//     that everybody has left stage t - 1, whose slot the pieces of stage t + NST - 1 then fill -- issued one at a time
//     between the MFMAs.  Per unit (k-step, A tile): raw fragment of unit u + 2 read from LDS, TN x 3 MFMAs of unit u,
//     fragment of unit u + 1 split; the B fragments of the next k-step are read during its unit 0 and split, one tile
//     per unit, from unit 1 on.
// K split, epilogue atomics and operand roles are those of gemm_wgrad_b3_kernel (non-implicit, non-grouped).
template <int N_>
__device__ __forceinline__ void wg_wait_vm() { asm volatile("s_waitcnt vmcnt(%0)" ::"n"(N_) : "memory"); }

__device__ __forceinline__ void wg_split8(const f32x4 lo4, const f32x4 hi4, bf16x8& hi, bf16x8& lo) {
    const float v[8] = {lo4[0], lo4[1], lo4[2], lo4[3], hi4[0], hi4[1], hi4[2], hi4[3]};
#pragma unroll
    for (int c = 0; c < 8; ++c) {
        const __bf16 h = (__bf16)v[c];
        hi[c] = h;
        lo[c] = (__bf16)(v[c] - (float)h);
    }
}

template <int WM, int WN, int TM, int TN, int KP, int NST>
__global__ __launch_bounds__(64 * WM * WN) void gemm_wgrad_dma_kernel(const GemmWgradParams p) {
    constexpr int NW = WM * WN;
    constexpr int BM = 32 * TM * WM, BN = 32 * TN * WN;
    constexpr int UPR = KP / 4, RPP = 64 / UPR, SWD = 16 / UPR;          // units per row, rows per piece, swizzle divisor
    constexpr int RB = KP * 4;                                            // bytes per row and stage
    constexpr int ROWS = BM + BN, PIECES = ROWS / RPP, PPW = PIECES / NW;  // 1-KB DMA pieces per stage / per wave
    constexpr int STAGE = ROWS * RB;                                      // bytes per ring slot
    constexpr int NK = KP / 16, NU_ = NK * TM;                            // k-steps / units per stage
    static_assert(KP == 16 || KP == 32, "stage width");
    static_assert(PIECES % NW == 0 && BM % 16 == 0 && NST >= 2, "pieces shared evenly; swizzle period divides BM");
    // NST >= 3: the software pipeline runs across stage boundaries (the first fragments of stage t + 1 are read during
    // stage t).  NST == 2 (the 256 x 256 tile: two 64-KB slots of 32 pixels; 16-pixel stages would fit four slots but
    // their 64-byte row segments stream at half the rate): stage t + 1 is in flight while stage t is consumed, the
    // pipeline drains and refills at every stage boundary.
    constexpr bool LOOK = NST >= 3;
    static_assert(TM % 2 == 0 && TM >= TN + 1, "pipeline parities");
    extern __shared__ __attribute__((aligned(16))) unsigned char lds_raw[];
    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int wm = wave / WN, wn = wave % WN;
    const int l31 = lane & 31, kk = lane >> 5;
    // grouped launch (G > 0): blockIdx.z = group * mtiles + row tile; the groups share shapes and strides
    const int grp = p.G > 0 ? (int)blockIdx.z / p.mtiles : 0;
    const int m0 = (p.G > 0 ? (int)blockIdx.z - grp * p.mtiles : (int)blockIdx.z) * BM, n0 = blockIdx.y * BN;
    const WgOperands ops_ = wg_operands(p, grp);
    const float* const pa_ = ops_.a;
    const float* const pb_ = ops_.b;
    float* const pgw_ = ops_.gw;

    f32x16 acc[TM][TN];
#pragma unroll
    for (int i = 0; i < TM; ++i)
#pragma unroll
        for (int j = 0; j < TN; ++j)
#pragma unroll
            for (int r = 0; r < 16; ++r) acc[i][j][r] = 0.f;

    // this lane's share of a stage: PPW pieces; piece q = wave + j NW covers stacked rows RPP q .. (A rows, then B rows)
    const float* prow[PPW];   // row base + this lane's (swizzled) unit
    bool pisa[PPW];
#pragma unroll
    for (int j = 0; j < PPW; ++j) {
        const int r = RPP * (wave + j * NW) + lane / UPR;               // stacked row
        const int u = (lane & (UPR - 1)) ^ ((r / SWD) & (UPR - 1));     // global unit that lands in LDS unit lane % UPR
        pisa[j] = r < BM;
        if (pisa[j]) {
            int row = m0 + r;
            row = row < p.M ? row : p.M - 1;
            prow[j] = pa_ + (long)row * p.HW + 4 * u;
        } else {
            int row = n0 + r - BM;
            row = row < p.N ? row : p.N - 1;
            prow[j] = pb_ + (long)row * p.HW + 4 * u;
        }
    }
    const unsigned uHW = (unsigned)p.HW;
    const int S = gridDim.x;
    const int nmine = (p.n_stages - (int)blockIdx.x + S - 1) / S;   // my stages: blockIdx.x, + S, ...
    // frame / pixel offsets of my stage t (clamped to my last one: pieces past the end re-read it and are never used)
    long offa = 0, offb = 0;
    auto stage_offsets = [&](int t) {
        t = t < nmine ? t : nmine - 1;
        const unsigned q0 = (unsigned)(blockIdx.x + t * S) * KP;
        const unsigned f = q0 / uHW, px = q0 - f * uHW;          // a stage never straddles a frame (HW % KP == 0)
        offa = (long)f * p.a_ns + px;
        offb = (long)f * p.b_ns + px;
    };
    auto piece = [&](const int j, const int slot) {
        __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)(prow[j] + (pisa[j] ? offa : offb)),
                                         (__attribute__((address_space(3))) void*)(lds_raw + slot * STAGE + (wave + j * NW) * 1024),
                                         16, 0, 0);
    };

    // fragment byte offsets inside a slot: row * RB + ((2 (2 s + kk) + h) ^ swz(row)) * 16
    int aoff[TM], boff[TN];
#pragma unroll
    for (int i = 0; i < TM; ++i) {
        const int r = (wm * TM + i) * 32 + l31;
        aoff[i] = r * RB + (((2 * kk) ^ ((r / SWD) & (UPR - 1))) << 4);
    }
#pragma unroll
    for (int j = 0; j < TN; ++j) {
        const int r = BM + (wn * TN + j) * 32 + l31;
        boff[j] = r * RB + (((2 * kk) ^ ((r / SWD) & (UPR - 1))) << 4);
    }
    auto raw = [&](const unsigned char* sb, const int off, const int s_, f32x4 (&r)[2]) {
        const int o = off ^ (s_ << 6);
        r[0] = *reinterpret_cast<const f32x4*>(sb + o);
        r[1] = *reinterpret_cast<const f32x4*>(sb + (o ^ 16));
    };

    // prologue: stages 0 .. NST-2 in flight; (LOOK) stage 0 landed, its first fragments read and split
#pragma unroll
    for (int t = 0; t < NST - 1; ++t) {
        stage_offsets(t);
#pragma unroll
        for (int j = 0; j < PPW; ++j) piece(j, t);
    }
    f32x4 ra[2][2], rb[TN][2];
    bf16x8 ah[2], al[2], bh[2][TN], bl[2][TN];
    auto first_fragments = [&](const unsigned char* sb0) {
#pragma unroll
        for (int j = 0; j < TN; ++j) raw(sb0, boff[j], 0, rb[j]);
        raw(sb0, aoff[0], 0, ra[0]);
        raw(sb0, aoff[1], 0, ra[1]);
#pragma unroll
        for (int j = 0; j < TN; ++j) wg_split8(rb[j][0], rb[j][1], bh[0][j], bl[0][j]);
        wg_split8(ra[0][0], ra[0][1], ah[0], al[0]);
    };
    if (LOOK) {
        wg_wait_vm<(NST - 2) * PPW>();
        __builtin_amdgcn_s_barrier();
        first_fragments(lds_raw);
    }

    int slot = 0;
    for (int it = 0; it < nmine; ++it) {
        // top of stage `it`.  LOOK: my pieces of stage it + 1 have landed once only those of the stages issued after it
        // are outstanding; the barrier makes that everybody's pieces and frees the slot of stage it - 1.  Two slots: the
        // same for stage `it` itself (nothing younger is outstanding).
        wg_wait_vm<LOOK ? (NST - 3) * PPW : 0>();
        __builtin_amdgcn_s_barrier();
        const int nslot = slot + 1 == NST ? 0 : slot + 1;
        const int fslot = slot == 0 ? NST - 1 : slot - 1;          // slot of stage it - 1 = slot of stage it + NST - 1
        const unsigned char* sb = lds_raw + slot * STAGE;
        const unsigned char* sn = lds_raw + nslot * STAGE;
        stage_offsets(it + NST - 1);
        if (!LOOK) {
            // two slots: the next stage's pieces have exactly this stage to land -- issued before anything else
#pragma unroll
            for (int j = 0; j < PPW; ++j) piece(j, fslot);
            first_fragments(sb);
        }
#pragma unroll
        for (int u = 0; u < NU_; ++u) {
            const int s_ = u / TM, i = u % TM, cur = u & 1, kb = s_ & 1;
            // raw A fragment of unit u + 2 (this stage or the next one)
            if (u + 2 < NU_) raw(sb, aoff[(u + 2) % TM], (u + 2) / TM, ra[cur]);
            else if (LOOK) raw(sn, aoff[(u + 2) % TM], 0, ra[cur]);
            // raw B fragments of the next k-step
            if (i == 0) {
#pragma unroll
                for (int j = 0; j < TN; ++j) {
                    if (s_ + 1 < NK) raw(sb, boff[j], s_ + 1, rb[j]);
                    else if (LOOK) raw(sn, boff[j], 0, rb[j]);
                }
            }
            // the DMA pieces of the stage that will fill the freed slot (NST - 2 stages to land): one per MFMA gap of
            // the first unit(s)
            if (LOOK && u == 0) {
#pragma unroll
                for (int j = 0; j < PPW; ++j) piece(j, fslot);
            }
#pragma unroll
            for (int j = 0; j < TN; ++j) {
                acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(al[cur], bh[kb][j], acc[i][j], 0, 0, 0);
                acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(ah[cur], bl[kb][j], acc[i][j], 0, 0, 0);
                acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(ah[cur], bh[kb][j], acc[i][j], 0, 0, 0);
            }
            if (LOOK || u + 1 < NU_) wg_split8(ra[cur ^ 1][0], ra[cur ^ 1][1], ah[cur ^ 1], al[cur ^ 1]);
            if (i >= 1 && i <= TN && (LOOK || s_ + 1 < NK))
                wg_split8(rb[i - 1][0], rb[i - 1][1], bh[kb ^ 1][i - 1], bl[kb ^ 1][i - 1]);
        }
        if (LOOK && (NK & 1)) {   // an odd number of k-steps per stage: the next stage starts on the other B buffer
#pragma unroll
            for (int j = 0; j < TN; ++j) {
                bh[0][j] = bh[1][j];
                bl[0][j] = bl[1][j];
            }
        }
        // the schedule of the stage, spelled out for the compiler (left alone it issues the conversions in long VALU
        // runs and the MFMAs back to back, i.e. one after the other): after every MFMA up to six VALU, one LDS read
        // and (first units) one DMA piece in its shadow
        if (!LOOK) {
            __builtin_amdgcn_sched_group_barrier(0x020, PPW, 0);
            __builtin_amdgcn_sched_group_barrier(0x100, 2 * TN + 4, 0);
            __builtin_amdgcn_sched_group_barrier(0x002, 30 * (TN + 1), 0);
        }
#pragma unroll
        for (int g = 0; g < NU_ * TN * 3; ++g) {
            __builtin_amdgcn_sched_group_barrier(0x008, 1, 0);
            __builtin_amdgcn_sched_group_barrier(0x002, 6, 0);
            __builtin_amdgcn_sched_group_barrier(0x100, 1, 0);
            if (LOOK && g < PPW) __builtin_amdgcn_sched_group_barrier(0x020, 1, 0);
        }
        slot = nslot;
    }
    wg_wait_vm<0>();   // (pieces past the end)
#pragma unroll
    for (int i = 0; i < TM; ++i)
#pragma unroll
        for (int j = 0; j < TN; ++j) {
            const int n = n0 + (wn * TN + j) * 32 + l31;
#pragma unroll
            for (int r = 0; r < 16; ++r) {
                const int m = m0 + (wm * TM + i) * 32 + (r & 3) + 8 * (r >> 2) + 4 * kk;
                if (m < p.M && n < p.N) atomicAdd(&pgw_[(long)m * p.N + n], acc[i][j][r]);
            }
        }
}

template <int WM, int WN, int TM, int TN, int KP, int NST>
static void launch_gemm_wgrad_dma(GemmWgradParams& p, hipStream_t s) {
    constexpr int BM = 32 * TM * WM, BN = 32 * TN * WN;
    p.n_stages = (int)(p.total / KP);
    const size_t lds = (size_t)NST * (BM + BN) * KP * 4;
    const int tiles = ceil_div(p.M, BM) * ceil_div(p.N, BN);
    const int G = p.G > 0 ? p.G : 1;
    int S = 256 / (tiles * G);   // one 8-wave workgroup per CU (see launch_gemm_wgrad)
    if (S > p.n_stages / 8) S = p.n_stages / 8;
    if (S < 1) S = 1;
    auto kern = gemm_wgrad_dma_kernel<WM, WN, TM, TN, KP, NST>;
    (void)hipFuncSetAttribute((const void*)kern, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
    p.mtiles = ceil_div(p.M, BM);
    dim3 grid(S, ceil_div(p.N, BN), p.mtiles * G);
    hipLaunchKernelGGL(kern, grid, dim3(64 * WM * WN), lds, s, p);
}

// Implicit 3x3 form of the ring kernel (conv1's weight gradient at the shallow levels: gw[256][ci*9 + tap], A = the
// 256-channel gradient image, B = the 18 / 36 input planes shifted by the tap).  The BIG operand A goes through the DMA
// ring (two slots of 256 rows x 32 pixels, split at the fragment reads) exactly as above; the SMALL operand keeps the
// cooperative register staging of gemm_wgrad_b3_kernel<..., IMPL = 1> -- its 9 Cin rows are built from L2-resident
// planes with the shift and the (hi, lo) split, into a double buffer of bf16 planes: the loads of stage t + 1 are issued
// at the top of stage t, committed at its end (which also certifies that the DMA pieces issued before them have landed:
// vector-memory operations complete in order), and the barrier at the top of stage t + 1 publishes both.
template <int WM, int WN, int TM, int TN>
__global__ __launch_bounds__(64 * WM * WN) void gemm_wgrad_dma_impl_kernel(const GemmWgradParams p) {
    constexpr int NW = WM * WN, NT = 64 * NW, KP = 32, NK = KP / 16;
    constexpr int BM = 32 * TM * WM, BN = 32 * TN * WN;
    constexpr int PIECES = BM / 8, PPW = PIECES / NW;   // 1-KB DMA pieces (8 rows x 128 B) of the A rows per stage / wave
    constexpr int ASLOT = BM * 128;
    constexpr int NU = KP / 8, RS = NU + 1;             // B planes: 16-byte units per row, row stride (conflict-free)
    constexpr int BU = (BN * NU + NT - 1) / NT;         // B units staged per thread
    constexpr int BPLANE = BN * RS;                     // units per plane
    static_assert(PIECES % NW == 0 && NT % NU == 0 && TM % 2 == 0, "shares / parities");
    extern __shared__ __attribute__((aligned(16))) unsigned char lds_raw[];
    bf16x8* const Bbase = reinterpret_cast<bf16x8*>(lds_raw + 2 * ASLOT);   // [buffer 2][plane 2][BN][RS]
    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int wm = wave / WN, wn = wave % WN;
    const int l31 = lane & 31, kk = lane >> 5;
    // grouped launch (G > 0): blockIdx.z = group * mtiles + row tile; the groups share shapes and strides
    const int grp = p.G > 0 ? (int)blockIdx.z / p.mtiles : 0;
    const int m0 = (p.G > 0 ? (int)blockIdx.z - grp * p.mtiles : (int)blockIdx.z) * BM, n0 = blockIdx.y * BN;
    const WgOperands ops_ = wg_operands(p, grp);
    const float* const pa_ = ops_.a;
    const float* const pb_ = ops_.b;
    const float* const pb2_ = ops_.b2;
    float* const pgw_ = ops_.gw;

    f32x16 acc[TM][TN];
#pragma unroll
    for (int i = 0; i < TM; ++i)
#pragma unroll
        for (int j = 0; j < TN; ++j)
#pragma unroll
            for (int r = 0; r < 16; ++r) acc[i][j][r] = 0.f;

    // ---- A: DMA pieces of this lane (rows 8 q .. 8 q + 7 of the tile, swizzled units: see gemm_wgrad_dma_kernel)
    const float* prow[PPW];
#pragma unroll
    for (int j = 0; j < PPW; ++j) {
        const int r = 8 * (wave + j * NW) + (lane >> 3);
        const int u = (lane & 7) ^ ((r >> 1) & 7);
        int row = m0 + r;
        row = row < p.M ? row : p.M - 1;
        prow[j] = pa_ + (long)row * p.HW + 4 * u;
    }
    int aoff[TM];
#pragma unroll
    for (int i = 0; i < TM; ++i) {
        const int r = (wm * TM + i) * 32 + l31;
        aoff[i] = r * 128 + (((2 * kk) ^ ((r >> 1) & 7)) << 4);
    }
    // ---- B: this thread's units (fixed row n = ci * 9 + tap, k-group kg): plane, frame stride, tap
    const int kg = tid % NU, r0 = tid / NU;
    const float* uplane[BU];
    long uns[BU];
    int udy[BU], udx[BU];
#pragma unroll
    for (int u = 0; u < BU; ++u) {
        int row = n0 + r0 + u * (NT / NU);
        if (row >= p.N) row = 0;
        const int ci = row / 9, tap = row - ci * 9;
        udy[u] = tap / 3 - 1;
        udx[u] = tap % 3 - 1;
        const bool first = ci < p.C1;
        uplane[u] = first ? pb_ + (long)ci * p.HW : pb2_ + (long)(ci - p.C1) * p.HW;
        uns[u] = first ? p.b_ns : p.b2_ns;
    }
    const unsigned uHW = (unsigned)p.HW;
    const int S = gridDim.x;
    const int nmine = (p.n_stages - (int)blockIdx.x + S - 1) / S;
    float4 bst[BU][2];
    float bedge[BU];
    unsigned rowmask = 0, edgemask = 0;
    // loads of my stage t: DMA pieces of A into `slot`, B units into registers (clamped to my last stage)
    auto fetch = [&](int t, const int slot) {
        t = t < nmine ? t : nmine - 1;
        const unsigned q0 = (unsigned)(blockIdx.x + t * S) * KP;
        const unsigned f = q0 / uHW, px = q0 - f * uHW;
        const long offa = (long)f * p.a_ns + px;
#pragma unroll
        for (int j = 0; j < PPW; ++j)
            __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)(prow[j] + offa),
                                             (__attribute__((address_space(3))) void*)(lds_raw + slot * ASLOT + (wave + j * NW) * 1024),
                                             16, 0, 0);
        const unsigned pix = px + 8u * kg;
        const int y = (int)(pix / (unsigned)p.W), x0 = (int)(pix - (unsigned)y * (unsigned)p.W);
        rowmask = 0;
        edgemask = 0;
#pragma unroll
        for (int u = 0; u < BU; ++u) {
            const int yy = y + udy[u];
            const bool rok = yy >= 0 && yy < p.H;
            const float* src = uplane[u] + (long)f * uns[u] + (rok ? yy : y) * p.W + x0;
            bst[u][0] = *reinterpret_cast<const float4*>(src);
            bst[u][1] = *reinterpret_cast<const float4*>(src + 4);
            const bool eok = udx[u] < 0 ? x0 > 0 : x0 + 8 < p.W;
            bedge[u] = src[eok ? (udx[u] < 0 ? -1 : 8) : 0];
            rowmask |= (rok ? 1u : 0u) << u;
            edgemask |= (eok ? 1u : 0u) << u;
        }
    };
    auto commit = [&](const int buf) {
        bf16x8* Bh = Bbase + buf * 2 * BPLANE;
        bf16x8* Bl = Bh + BPLANE;
#pragma unroll
        for (int u = 0; u < BU; ++u) {
            const int row = r0 + u * (NT / NU);
            if (row < BN) {
                const bool okr = n0 + row < p.N && ((rowmask >> u) & 1u);
                const bool eok = (edgemask >> u) & 1u;
                const float v[8] = {bst[u][0].x, bst[u][0].y, bst[u][0].z, bst[u][0].w,
                                    bst[u][1].x, bst[u][1].y, bst[u][1].z, bst[u][1].w};
                bf16x8 hi, lo;
#pragma unroll
                for (int c = 0; c < 8; ++c) {
                    const float left = c == 0 ? (eok ? bedge[u] : 0.f) : v[c > 0 ? c - 1 : 0];
                    const float right = c == 7 ? (eok ? bedge[u] : 0.f) : v[c < 7 ? c + 1 : 7];
                    float x = udx[u] < 0 ? left : (udx[u] > 0 ? right : v[c]);
                    x = okr ? x : 0.f;
                    const __bf16 h = (__bf16)x;
                    hi[c] = h;
                    lo[c] = (__bf16)(x - (float)h);
                }
                Bh[row * RS + kg] = hi;
                Bl[row * RS + kg] = lo;
            }
        }
    };
    auto rawA = [&](const unsigned char* sb, const int i, const int s_, f32x4 (&r)[2]) {
        const int o = aoff[i] ^ (s_ << 6);
        r[0] = *reinterpret_cast<const f32x4*>(sb + o);
        r[1] = *reinterpret_cast<const f32x4*>(sb + (o ^ 16));
    };
    const int brow = (wn * TN * 32 + l31) * RS + kk;

    // prologue: stage 0 loaded and committed
    fetch(0, 0);
    commit(0);
    int slot = 0;
    for (int it = 0; it < nmine; ++it) {
        wg_wait_vm<0>();                    // (my pieces of this stage: older than the B loads just committed)
        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
        __builtin_amdgcn_s_barrier();       // everybody's pieces and planes of stage `it`; the other slot / buffer is free
        fetch(it + 1, slot ^ 1);
        const unsigned char* sb = lds_raw + slot * ASLOT;
        const bf16x8* Bh = Bbase + slot * 2 * BPLANE;
        const bf16x8* Bl = Bh + BPLANE;
        f32x4 ra[2][TM][2];
        bf16x8 ah[2][TM], al[2][TM];
#pragma unroll
        for (int i = 0; i < TM; ++i) rawA(sb, i, 0, ra[0][i]);
#pragma unroll
        for (int i = 0; i < TM; ++i) wg_split8(ra[0][i][0], ra[0][i][1], ah[0][i], al[0][i]);
#pragma unroll
        for (int s_ = 0; s_ < NK; ++s_) {
            const int cur = s_ & 1;
            if (s_ + 1 < NK) {
#pragma unroll
                for (int i = 0; i < TM; ++i) rawA(sb, i, s_ + 1, ra[cur ^ 1][i]);
            }
            bf16x8 bh[TN], bl[TN];
#pragma unroll
            for (int j = 0; j < TN; ++j) {
                bh[j] = Bh[brow + j * 32 * RS + 2 * s_];
                bl[j] = Bl[brow + j * 32 * RS + 2 * s_];
            }
#pragma unroll
            for (int i = 0; i < TM; ++i)
#pragma unroll
                for (int j = 0; j < TN; ++j) {
                    acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(al[cur][i], bh[j], acc[i][j], 0, 0, 0);
                    acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(ah[cur][i], bl[j], acc[i][j], 0, 0, 0);
                    acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(ah[cur][i], bh[j], acc[i][j], 0, 0, 0);
                }
            if (s_ + 1 < NK) {
#pragma unroll
                for (int i = 0; i < TM; ++i) wg_split8(ra[cur ^ 1][i][0], ra[cur ^ 1][i][1], ah[cur ^ 1][i], al[cur ^ 1][i]);
            }
        }
        commit(slot ^ 1);   // B planes of stage it + 1 (nobody reads that buffer before the next barrier)
        slot ^= 1;
    }
    wg_wait_vm<0>();
#pragma unroll
    for (int i = 0; i < TM; ++i)
#pragma unroll
        for (int j = 0; j < TN; ++j) {
            const int n = n0 + (wn * TN + j) * 32 + l31;
#pragma unroll
            for (int r = 0; r < 16; ++r) {
                const int m = m0 + (wm * TM + i) * 32 + (r & 3) + 8 * (r >> 2) + 4 * kk;
                if (m < p.M && n < p.N) atomicAdd(&pgw_[(long)m * p.N + n], acc[i][j][r]);
            }
        }
}

template <int WM, int WN, int TM, int TN>
static void launch_gemm_wgrad_dma_impl(GemmWgradParams& p, hipStream_t s) {
    constexpr int BM = 32 * TM * WM, BN = 32 * TN * WN;
    p.n_stages = (int)(p.total / 32);
    const size_t lds = (size_t)2 * BM * 128 + (size_t)2 * 2 * BN * 5 * 16;
    const int tiles = ceil_div(p.M, BM) * ceil_div(p.N, BN);
    const int G = p.G > 0 ? p.G : 1;
    int S = 256 / (tiles * G);
    if (S > p.n_stages / 8) S = p.n_stages / 8;
    if (S < 1) S = 1;
    auto kern = gemm_wgrad_dma_impl_kernel<WM, WN, TM, TN>;
    (void)hipFuncSetAttribute((const void*)kern, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
    p.mtiles = ceil_div(p.M, BM);
    dim3 grid(S, ceil_div(p.N, BN), p.mtiles * G);
    hipLaunchKernelGGL(kern, grid, dim3(64 * WM * WN), lds, s, p);
}

// shapes the LDS-DMA kernel takes: a plain (1x1-form) gradient -- or a group of them -- over whole 32-pixel stages
static bool wgrad_dma_ok(const GemmWgradParams& p) {
    static const int off = getenv("RFN_WGRAD_DMA") ? atoi(getenv("RFN_WGRAD_DMA")) == 0 : 0;
    // (grouped: the K gradients of a level together are the problem size)
    return !off && p.HW % 32 == 0 && p.total * (p.G > 0 ? p.G : 1) >= 100000 && p.total >= 2048 && p.a_ns % 4 == 0 &&
           p.b_ns % 4 == 0 && p.N > 128;
}

template <int WM, int WN, int TM, int TN, int KP, int IMPL = 0>
static void launch_gemm_wgrad(GemmWgradParams& p, hipStream_t s) {
    constexpr int BM = 32 * TM * WM, BN = 32 * TN * WN;
    p.n_stages = (int)((p.total + KP - 1) / KP);
    size_t lds = (size_t)2 * (BM + BN) * (KP / 8 + 1) * 16;
    int tiles = ceil_div(p.M, BM) * ceil_div(p.N, BN);
    // K split: every workgroup ends with BM x BN float atomics into the SAME gw tile, and the chip retires only ~1.3 TB/s
    // of atomic bytes -- 1024 workgroups x 64 KB is 50 us of atomics on a problem whose GEMM takes 10.  So: as many
    // workgroups as the CUs can hold at once (8-wave tiles: one per CU, 4-wave: two), and at least 4 stages each.
    static const int sdiv = getenv("RFN_WGRAD_SPLIT") ? atoi(getenv("RFN_WGRAD_SPLIT")) : 0;
    const int G = p.G > 0 ? p.G : 1;
    int S = (sdiv > 0 ? sdiv : (WM * WN == 8 ? 256 : 512)) / (tiles * G);
    if (S > p.n_stages / 4) S = p.n_stages / 4;
    if (S < 1) S = 1;
    auto kern = gemm_wgrad_b3_kernel<WM, WN, TM, TN, KP, IMPL>;
    if (lds > 65536) (void)hipFuncSetAttribute((const void*)kern, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
    p.mtiles = ceil_div(p.M, BM);
    dim3 grid(S, ceil_div(p.N, BN), p.mtiles * G);
    hipLaunchKernelGGL(kern, grid, dim3(64 * WM * WN), lds, s, p);
}

static void select_gemm_wgrad(GemmWgradParams& p, hipStream_t s) {
    const int M = p.M, Nc = p.N;
    static const int variant = getenv("RFN_WGRAD_VARIANT") ? atoi(getenv("RFN_WGRAD_VARIANT")) : 0;
    if (variant == 0 && wgrad_dma_ok(p)) {
        if (M >= 192 && Nc % 256 == 0) return launch_gemm_wgrad_dma<2, 4, 4, 2, 32, 2>(p, s);   // 256 x 256, two slots of 64 KB
        if (M <= 64 && Nc % 256 == 0) return launch_gemm_wgrad_dma<1, 8, 2, 1, 32, 3>(p, s);   // 64 x 256, ring of 3 x 40 KB
    }
    const bool big = variant != 1 && M > 128 && Nc > 128 && p.total >= 100000;
    if (big && ceil_div(Nc, 192) * 192 < ceil_div(Nc, 256) * 256) {
        if (variant == 2)
            launch_gemm_wgrad<4, 2, 2, 3, 32>(p, s);
        else
            launch_gemm_wgrad<4, 2, 2, 3, 64>(p, s);   // 256 x 192, 8 waves, 64-pixel stages
    } else if (big && variant == 2)
        launch_gemm_wgrad<2, 4, 4, 2, 32>(p, s);
    else if (big)
        launch_gemm_wgrad<2, 4, 4, 2, 64>(p, s);   // 256 x 256, 8 waves, 64-pixel stages
    else if (M <= 64)
        launch_gemm_wgrad<1, 4, 2, 2, 32>(p, s);   // 64 x 256
    else if (Nc <= 64)
        launch_gemm_wgrad<4, 1, 2, 2, 32>(p, s);   // 256 x 64
    else
        launch_gemm_wgrad<2, 2, 2, 2, 64>(p, s);   // 128 x 128
}

extern "C" int rfn_gemm_wgrad_bf16x3(const float* a, long a_ns, int M, const float* b, long b_ns, int Nc, float* gw,
                                     int F, int HW, rfn_stream_t stream) {
    RFN_CHECK_ARG(a && b && gw && M > 0 && Nc > 0 && F >= 0 && HW > 0, -1);
    RFN_CHECK_ARG(HW % 4 == 0 && a_ns % 4 == 0 && b_ns % 4 == 0, -2);
    RFN_CHECK_ARG((((uintptr_t)a | (uintptr_t)b) & 15) == 0, -3);
    RFN_CHECK_ARG((long)F * HW < (1L << 31) - 4096, -4);
    if (F == 0) return 0;
    GemmWgradParams p;
    memset(&p, 0, sizeof(p));
    p.a = a; p.b = b; p.a_ns = a_ns; p.b_ns = b_ns; p.M = M; p.N = Nc; p.gw = gw; p.F = F; p.HW = HW;
    p.total = (long)F * HW;
    select_gemm_wgrad(p, (hipStream_t)stream);
    RFN_LAUNCH_CHECK();
    return 0;
}

// G (<= 16) weight gradients of one shape in ONE launch: gw[g][M][Nc] += sum a[g] b[g]^T (same strides, F, HW for all).
extern "C" int rfn_gemm_wgrad_grouped_bf16x3(const float* const* a, long a_ns, int M, const float* const* b, long b_ns,
                                             int Nc, float* const* gw, int G, int F, int HW, rfn_stream_t stream) {
    RFN_CHECK_ARG(a && b && gw && G >= 1 && G <= 16 && M > 0 && Nc > 0 && F >= 0 && HW > 0, -1);
    RFN_CHECK_ARG(HW % 4 == 0 && a_ns % 4 == 0 && b_ns % 4 == 0, -2);
    RFN_CHECK_ARG((long)F * HW < (1L << 31) - 4096, -4);
    if (F == 0) return 0;
    GemmWgradParams p;
    memset(&p, 0, sizeof(p));
    for (int g = 0; g < G; ++g) {
        RFN_CHECK_ARG(a[g] && b[g] && gw[g] && (((uintptr_t)a[g] | (uintptr_t)b[g]) & 15) == 0, -3);
        p.ga[g] = a[g]; p.gb[g] = b[g]; p.gb2[g] = b[g]; p.ggw[g] = gw[g];
    }
    p.G = G; p.a = a[0]; p.b = b[0]; p.gw = gw[0];
    p.a_ns = a_ns; p.b_ns = b_ns; p.M = M; p.N = Nc; p.F = F; p.HW = HW;
    p.total = (long)F * HW;
    select_gemm_wgrad(p, (hipStream_t)stream);
    RFN_LAUNCH_CHECK();
    return 0;
}

// 3x3 weight gradient without the im2col buffer: gw[co][ci][tap] = sum_{frames,pixels} g[co][px] * in[ci][px + tap]
// (pad 1), the shifted planes built while staging (rows of one image row, W % 8 == 0).  The output is the torch weight
// layout [Cout][Cin][3][3] (the GEMM on rfn_im2col3x3_f32's buffer gives [Cout][tap][Cin] instead).
static void select_wgrad_implicit(GemmWgradParams& p, hipStream_t s) {
    static const int dma_off = getenv("RFN_WGRAD_DMA") ? atoi(getenv("RFN_WGRAD_DMA")) == 0 : 0;
    // (grouped: the K gradients of a level together are the problem size)
    if (p.M > 128 && p.total * (p.G > 0 ? p.G : 1) >= 100000 && p.total >= 2048 && !dma_off && p.HW % 32 == 0 && p.a_ns % 4 == 0)
        launch_gemm_wgrad_dma_impl<4, 2, 2, 3>(p, s);   // 256 x 192, 8 waves, A through the DMA ring
    else if (p.M > 128 && p.total >= 100000)
        launch_gemm_wgrad<4, 2, 2, 3, 64, 1>(p, s);   // 256 x 192, 8 waves
    else if (p.M <= 32 && p.G == 0)
        // few output channels (the 16- / 32-channel blocks of the extractor / upscaler on 64x64 and 32x32 maps): one 32-row
        // tile instead of 128 rows of which 16 are real (the clamped rows were 8x the loads and MFMAs of the gradient)
        launch_gemm_wgrad<1, 4, 1, 2, 32, 1>(p, s);   // 32 x 256
    else
        launch_gemm_wgrad<2, 2, 2, 2, 64, 1>(p, s);   // 128 x 128
}

extern "C" int rfn_conv3x3_wgrad_implicit_bf16x3(const float* g, long g_ns, int Cout, const float* in1, long in1_ns, int C1,
                                                 const float* in2, long in2_ns, int C2, float* gw, int F, int H, int W,
                                                 rfn_stream_t stream) {
    RFN_CHECK_ARG(g && in1 && gw && Cout > 0 && C1 > 0 && C2 >= 0 && (C2 == 0 || in2) && F >= 0 && H > 0 && W > 0, -1);
    RFN_CHECK_ARG(W % 8 == 0 && g_ns % 4 == 0 && in1_ns % 4 == 0 && (C2 == 0 || in2_ns % 4 == 0), -2);
    RFN_CHECK_ARG((((uintptr_t)g | (uintptr_t)in1 | (uintptr_t)(C2 ? in2 : in1)) & 15) == 0, -3);
    RFN_CHECK_ARG((long)F * H * W < (1L << 31) - 4096, -4);
    if (F == 0) return 0;
    GemmWgradParams p;
    memset(&p, 0, sizeof(p));
    p.a = g; p.a_ns = g_ns; p.M = Cout; p.b = in1; p.b_ns = in1_ns; p.b2 = C2 ? in2 : in1; p.b2_ns = C2 ? in2_ns : in1_ns;
    p.C1 = C1; p.C2 = C2; p.H = H; p.W = W; p.N = 9 * (C1 + C2); p.gw = gw; p.F = F; p.HW = H * W;
    p.total = (long)F * H * W;
    select_wgrad_implicit(p, (hipStream_t)stream);
    RFN_LAUNCH_CHECK();
    return 0;
}

// grouped form: G gradients of one shape (per group: g, in1, in2 (ignored when C2 == 0), gw)
extern "C" int rfn_conv3x3_wgrad_implicit_grouped_bf16x3(const float* const* g, long g_ns, int Cout,
                                                         const float* const* in1, long in1_ns, int C1,
                                                         const float* const* in2, long in2_ns, int C2, float* const* gw,
                                                         int G, int F, int H, int W, rfn_stream_t stream) {
    RFN_CHECK_ARG(g && in1 && gw && G >= 1 && G <= 16 && Cout > 0 && C1 > 0 && C2 >= 0 && (C2 == 0 || in2) && F >= 0, -1);
    RFN_CHECK_ARG(H > 0 && W > 0 && W % 8 == 0 && g_ns % 4 == 0 && in1_ns % 4 == 0 && (C2 == 0 || in2_ns % 4 == 0), -2);
    RFN_CHECK_ARG((long)F * H * W < (1L << 31) - 4096, -4);
    if (F == 0) return 0;
    GemmWgradParams p;
    memset(&p, 0, sizeof(p));
    for (int i = 0; i < G; ++i) {
        RFN_CHECK_ARG(g[i] && in1[i] && gw[i] && (C2 == 0 || in2[i]), -3);
        RFN_CHECK_ARG((((uintptr_t)g[i] | (uintptr_t)in1[i] | (uintptr_t)(C2 ? in2[i] : in1[i])) & 15) == 0, -3);
        p.ga[i] = g[i]; p.gb[i] = in1[i]; p.gb2[i] = C2 ? in2[i] : in1[i]; p.ggw[i] = gw[i];
    }
    p.G = G; p.a = g[0]; p.b = in1[0]; p.b2 = p.gb2[0]; p.gw = gw[0];
    p.a_ns = g_ns; p.M = Cout; p.b_ns = in1_ns; p.b2_ns = C2 ? in2_ns : in1_ns;
    p.C1 = C1; p.C2 = C2; p.H = H; p.W = W; p.N = 9 * (C1 + C2); p.F = F; p.HW = H * W;
    p.total = (long)F * H * W;
    select_wgrad_implicit(p, (hipStream_t)stream);
    RFN_LAUNCH_CHECK();
    return 0;
}

// ------------------------------------------------------------------------------------------------ im2col (3x3, pad 1)
// X9[n][tap*Cin + ci][y][x] = in[n][ci][y+dy-1][x+dx-1] (0 outside), two-source input like the convolutions.
__global__ void im2col3x3_kernel(const float* __restrict__ in1, long in1_ns, int C1, const float* __restrict__ in2,
                                 long in2_ns, int C2, float* __restrict__ out, int N, int H, int W) {
    const int Cin = C1 + C2;
    const long HW = (long)H * W, total = (long)N * 9 * Cin * HW;
    for (long idx = (long)blockIdx.x * blockDim.x + threadIdx.x; idx < total; idx += (long)gridDim.x * blockDim.x) {
        const int x = (int)(idx % W);
        long r = idx / W;
        const int y = (int)(r % H);
        r /= H;
        const int tc = (int)(r % (9 * Cin));
        const long n = r / (9 * Cin);
        const int t = tc / Cin, ci = tc - t * Cin;
        const int yy = y + t / 3 - 1, xx = x + t % 3 - 1;
        float v = 0.f;
        if (yy >= 0 && yy < H && xx >= 0 && xx < W)
            v = ci < C1 ? in1[n * in1_ns + (long)ci * HW + (long)yy * W + xx]
                        : in2[n * in2_ns + (long)(ci - C1) * HW + (long)yy * W + xx];
        out[idx] = v;
    }
}
// W % 4 == 0: one thread = 4 consecutive pixels of one (frame, tap*Cin+ci, y) row -> one 16-byte store, the index
// divisions amortised over 4 elements (the element-wise kernel above spends its time in them).
__global__ void im2col3x3_v4_kernel(const float* __restrict__ in1, long in1_ns, int C1, const float* __restrict__ in2,
                                    long in2_ns, int C2, float* __restrict__ out, int N, int H, int W) {
    const int Cin = C1 + C2, HW4 = H * W / 4, W4 = W / 4;
    const long HW = (long)H * W, total = (long)N * 9 * Cin * HW4;
    for (long idx = (long)blockIdx.x * blockDim.x + threadIdx.x; idx < total; idx += (long)gridDim.x * blockDim.x) {
        const int q = (int)(idx % HW4);
        const long r = idx / HW4;
        const int tc = (int)(r % (9 * Cin));
        const long n = r / (9 * Cin);
        const int t = tc / Cin, ci = tc - t * Cin;
        const int y = q / W4, x0 = (q - y * W4) * 4;
        const int yy = y + t / 3 - 1, dx = t % 3 - 1;
        float4 v = {0.f, 0.f, 0.f, 0.f};
        if (yy >= 0 && yy < H) {
            const float* src = (ci < C1 ? in1 + n * in1_ns + (long)ci * HW : in2 + n * in2_ns + (long)(ci - C1) * HW) +
                               (long)yy * W;
            if (dx == 0) {
                v = *reinterpret_cast<const float4*>(src + x0);
            } else {
                const float4 c = *reinterpret_cast<const float4*>(src + x0);
                if (dx < 0) {
                    v.x = x0 > 0 ? src[x0 - 1] : 0.f;
                    v.y = c.x; v.z = c.y; v.w = c.z;
                } else {
                    v.x = c.y; v.y = c.z; v.z = c.w;
                    v.w = x0 + 4 < W ? src[x0 + 4] : 0.f;
                }
            }
        }
        *reinterpret_cast<float4*>(out + idx * 4) = v;
    }
}
extern "C" int rfn_im2col3x3_f32(const float* in1, long in1_ns, int C1, const float* in2, long in2_ns, int C2,
                                 float* out, int N, int H, int W, rfn_stream_t stream) {
    RFN_CHECK_ARG(in1 && out && C1 > 0 && C2 >= 0 && (C2 == 0 || in2) && N >= 0 && H > 0 && W > 0, -1);
    if (N == 0) return 0;
    hipStream_t s = (hipStream_t)stream;
    const long HWl = (long)H * W;
    if (W % 4 == 0 && in1_ns % 4 == 0 && (C2 == 0 || in2_ns % 4 == 0) && HWl % 4 == 0 &&
        (((uintptr_t)in1 | (uintptr_t)out | (uintptr_t)(C2 ? in2 : in1)) & 15) == 0) {
        long tot4 = (long)N * 9 * (C1 + C2) * (HWl / 4);
        int grid4 = (int)((tot4 + 255) / 256 < 16384 ? (tot4 + 255) / 256 : 16384);
        hipLaunchKernelGGL(im2col3x3_v4_kernel, dim3(grid4), dim3(256), 0, s, in1, in1_ns, C1, in2, in2_ns, C2, out, N, H, W);
        RFN_LAUNCH_CHECK();
        return 0;
    }
    long tot = (long)N * 9 * (C1 + C2) * H * W;
    int grid = (int)((tot + 255) / 256 < 8192 ? (tot + 255) / 256 : 8192);
    hipLaunchKernelGGL(im2col3x3_kernel, dim3(grid), dim3(256), 0, (hipStream_t)stream, in1, in1_ns, C1, in2, in2_ns, C2,
                       out, N, H, W);
    RFN_LAUNCH_CHECK();
    return 0;
}
