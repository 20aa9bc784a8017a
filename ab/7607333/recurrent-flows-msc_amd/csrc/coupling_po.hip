// Fused coupling network for the shallow flow levels ("pixel-owning" layout), fp32-grade split arithmetic on f16 MFMA.
//
// Reference op sequence (Flow/glow_modules.py:232-238, 119-121, 139-142):
//     h1 = act(ActNorm(conv3x3(cat(z1, cond))))      Cin = C/2 + Cc  ->  Hd = 256
//     h2 = act(ActNorm(conv1x1(h1)))                  256 -> 256
//     o  = (conv3x3(h2) + b3) * exp(3 logs3)          256 -> C
// The unfused kernels (conv_bf16x3.hip) own OUTPUT-CHANNEL slabs per wave and pass the 256-channel hidden tensors
// through HBM (level 0: 637 MB each, written once and read once or twice per layer).  Here a wave owns 32 PIXELS and
// carries them through the whole chain:
//   * an MFMA 32x32 accumulator tile D[channel][pixel] has the pixel on the lane and 16 channels in registers, which is
//     exactly the B-operand layout of the next 32x32x16 MFMA whose K runs over those channels (MI355X guide: "an
//     accumulator tile as the next MFMA's operand"; the k order inside a step is permuted, so the 1x1 weights are
//     packed in that order).  h1 / h2 never leave the register file on their way to the next layer;
//   * conv2 runs K-MAJOR: as soon as four h1 tiles (128 channels) exist they are converted and multiplied into all
//     eight conv2 accumulators (8 independent MFMA chains), so h1 needs no storage at all and h2 = the 8 accumulators;
//   * the weights are the shared operand: every wave of every workgroup streams the same pre-split, fragment-ordered
//     weight groups from L2 into LDS with LDS-DMA (global_load_lds_dwordx4), two groups ahead in a ring of three
//     slots behind counted vmcnt waits;
//     A fragments are single conflict-free ds_read_b128;
//   * the first 3x3 convolution reads its im2col B fragments from a haloed, pre-split image of the workgroup's 128
//     pixels (z1 | cond, a few KB) staged in LDS once per round -- loaded and converted one round ahead;
//   * the last 3x3 convolution runs tap-expanded: P[tap*C + co][pixel] = Σ_c w3[co][c][tap] h2[c][pixel] is one more
//     1x1 product on the register-resident h2 (9C <= 96 rows), and the cross-pixel part o[co] = Σ_tap P[tap,co][px+tap]
//     is left to the gather kernel (shell.hip) -- 9C floats per pixel instead of the 256-channel h2.
// h1 and h2 are still WRITTEN once (the backward pass needs them); they are never read back in the forward pass.
//
// Arithmetic ("f16x3s").  bits/dim parity (north_star: 1e-4) is decided by the forward pass, and two bf16 pieces per
// operand ("bf16x3", 16 significant bits) are measurably not enough when the flow is ill-conditioned
// (tools/precision_study.py, canonical model at T=10: 1e-4..5e-4 against 1e-5 for fp32).  This kernel splits every
// operand into two FP16 pieces after an exact power-of-two scaling,
//     x * 2^e = hi + lo,   hi = fp16(x 2^e),  lo = fp16(x 2^e - hi)        (22 significant bits),
// and forms a*b as  hi*hi + hi*lo + lo*hi  with three v_mfma_f32_32x32x16_f16 accumulating in fp32; the scale is undone
// in the epilogue.  Same MFMA count as bf16x3, error at the fp32 level (study: 1.0e-5 vs 1.1e-5 for a 3-piece / 6-MFMA
// bf16 split).  fp16 has 5 exponent bits, so the scales are dynamic and chosen where the data is:
//     weights        one scale per convolution, max|w| -> [2^14, 2^15)            (pack kernel)
//     conv1 input    one scale per round (block maximum of the staged image)
//     h1 -> conv2    per PIXEL (lane): running maximum over the h1 tiles produced so far; when a later tile raises it,
//                    the eight conv2 accumulators of that pixel are rescaled by the (exact) power of two
//     h2 -> conv3    per pixel, maximum over all 256 channels (all of h2 is in registers before conv3 starts)
#include "conv_common.h"

typedef _Float16 f16x8 __attribute__((ext_vector_type(8)));

#define PO_HD 256          // hidden width (8 output tiles of 32)
#define PO_TAP_PAIRS 5     // 9 taps in pairs (the lane half selects the tap of a pair); the 10th tap is zero weight
#define PO_WAVES 4
#define PO_ROUND_PX (32 * PO_WAVES)

// ---- geometry of the forward weight stream (host + device).  Unit: fragment = 64 lanes x 16 B = 1 KB; two planes
// (hi, lo) per logical fragment.  Fragment 0 is a header: floats {1/s_w1, 1/s_w2, 1/s_w3}.  Then, consumed in order:
//   for quad u = 0, 1  (conv1 output tiles 4u .. 4u+3):
//       NG conv1 groups of 5 k-steps:   ((k*4 + t)*2 + plane)                                    40 fragments
//       4 conv2 groups of 2 k-steps (h1 k-steps 8u + 2g + {0,1}, all 8 output tiles):
//                                        ((k*8 + a2)*2 + plane)                                  32 fragments
//   4 conv3 groups of 4 k-steps:         ((k*NP + jt)*2 + plane)                                 8 NP fragments
struct POGeom {
    int NG, NP, NS1;
    int quad_frags;    // fragments per quad
    int c3;            // first fragment of the conv3 part
    int total_frags;
};
// bwd: the stream of the backward kernel (data-gradient chain): no conv3 part
__host__ __device__ static inline POGeom po_geom(int Cin, int C, bool bwd = false) {
    POGeom g;
    g.NG = (Cin + 7) / 8;
    g.NP = bwd ? 0 : (9 * C + 31) / 32;
    g.NS1 = PO_TAP_PAIRS * g.NG;
    g.quad_frags = g.NG * 40 + 4 * 32;
    g.c3 = 1 + 2 * g.quad_frags;
    g.total_frags = g.c3 + 4 * 8 * g.NP;
    return g;
}

// Grouping of the stream for the LDS ring (the stream itself is linear in the k-step, so the grouping is the kernel's
// choice): conv1 groups of GS1 k-steps (8 GS1 fragments; NG1 = 5 NG / GS1 groups per quad), conv2 groups of 2 k-steps
// (32 fragments), conv3 groups of GS3 k-steps (2 NP GS3 fragments).  The canonical shallow levels (NG <= 5) take
// GS1 = 5, GS3 = 4 (slots of 40 KB); wider inputs (NG = 9: level 2 of the canonical flow, the BAIR flow's level 0) need
// the LDS for the input image and take GS1 = 3, GS3 = 2 (slots of 32 KB).
__host__ __device__ constexpr int po_gs1(int NG) { return NG > 5 ? 3 : 5; }
__host__ __device__ constexpr int po_gs3(int NP) { return NP > 3 ? 2 : 4; }
// group `idx` (0 .. 2 NG1 + 8 + 16 / GS3 - 1) of a round in consumption order: first fragment and fragment count
__host__ __device__ constexpr int po_group_base(int NG, int NP, int idx) {
    const int QF = NG * 40 + 128, GS1 = po_gs1(NG), NG1 = 5 * NG / GS1, F1 = 8 * GS1;
    return idx < NG1 ? 1 + idx * F1
         : idx < NG1 + 4 ? 1 + NG * 40 + (idx - NG1) * 32
         : idx < 2 * NG1 + 4 ? 1 + QF + (idx - NG1 - 4) * F1
         : idx < 2 * NG1 + 8 ? 1 + QF + NG * 40 + (idx - 2 * NG1 - 4) * 32
         : 1 + 2 * QF + (idx - 2 * NG1 - 8) * 2 * NP * po_gs3(NP);
}
// positions of the haloed input image of a round on W x W maps (see the kernel)
__host__ __device__ constexpr int po_image_positions(int W) {
    return W * W >= PO_ROUND_PX ? (PO_ROUND_PX / W + 2) * (W + 2) : (PO_ROUND_PX / (W * W)) * (W + 2) * (W + 2);
}
__host__ __device__ constexpr int po_slot_frags(int NG, int NP) {   // fragments (KB) per ring slot
    const int f1 = 8 * po_gs1(NG), f3 = 2 * NP * po_gs3(NP);
    return (f1 > 32 ? f1 : 32) > f3 ? (f1 > 32 ? f1 : 32) : f3;
}
__host__ __device__ constexpr int po_group_size(int NG, int NP, int idx) {
    const int GS1 = po_gs1(NG), NG1 = 5 * NG / GS1, F1 = 8 * GS1;
    return idx < NG1 ? F1 : idx < NG1 + 4 ? 32 : idx < 2 * NG1 + 4 ? F1 : idx < 2 * NG1 + 8 ? 32 : 2 * NP * po_gs3(NP);
}

// ------------------------------------------------------------------------------------------------ weight stream
// One fragment: lane (r = lane & 31, kk = lane >> 5) holds A[row r][8 k-values] of one plane of the scaled weights.
//   conv1 tile a (a < 8), k-step s = tp*NG + g:   row = 32a + r, tap = 2tp + kk, element j <-> input channel 8g + j
//       (zero beyond Cin / tap 9)                                                   value w1[row][ci][tap]
//   conv2 tile a2, k-step s < 16:  row = 32a2 + r, element j <-> h1 channel 16s + 8(j>>2) + 4kk + (j&3)
//       (the register order of an accumulator tile)                                value w2[row][c1]
//   conv3 row tile jt, k-step s < 16:  row R = 32jt + r = tap*C + co (zero beyond 9C), element j <-> h2 channel
//       16s + 8(j>>2) + 4kk + (j&3)                                                 value w3[co][c2][tap]
struct POPackDesc {   // mirrors rfn_po_pack_desc in include/rfn_hip.h
    const float* w1;  // [256][Cin][3][3]
    const float* w2;  // [256][256][1][1]
    const float* w3;  // [C][256][3][3]
    float* dst;       // total_frags * 1 KB
    int Cin, C;
};

// exact power-of-two scale that puts m = max|x| into [2^14, 2^15) (fp16: largest finite 65504), and its inverse
__host__ __device__ static inline void po_scale_for_max(float m, float* sc, float* inv) {
    unsigned bits;
    memcpy(&bits, &m, 4);
    int E = (int)((bits >> 23) & 0xff);
    if (E < 40) E = 40;          // zero / denormal-small maxima: any moderate scale does
    if (E > 250) E = 250;        // inf / nan input: results are garbage either way
    const unsigned sb = (unsigned)(268 - E) << 23, ib = (unsigned)(E - 14) << 23;
    memcpy(sc, &sb, 4);
    memcpy(inv, &ib, 4);
}

__device__ __forceinline__ void po_split_f16(const float (&v)[8], const float sc, f16x8& hi, f16x8& lo) {
#pragma unroll
    for (int j = 0; j < 8; ++j) {
        const float x = v[j] * sc;
        const _Float16 h = (_Float16)x;
        hi[j] = h;
        lo[j] = (_Float16)(x - (float)h);
    }
}

// grid (3, n): block (c, d) = max |w| of convolution c of descriptor d -> header {1/s_w1, 1/s_w2, 1/s_w3}
// BWD (stream of the backward kernel, see coupling_po_bwd below): "conv1" = w3 transposed and mirrored, "conv2" = w2
// transposed; header {1/s_w3, 1/s_w2, 1}
template <bool BWD>
__global__ __launch_bounds__(256) void po_pack_scale_kernel(const POPackDesc* __restrict__ descs) {
    __shared__ float sm[4];
    const POPackDesc d = descs[blockIdx.y];
    const int c = blockIdx.x;
    const float* w = BWD ? (c == 0 ? d.w3 : d.w2) : (c == 0 ? d.w1 : (c == 1 ? d.w2 : d.w3));
    long n = c == 0 ? (long)PO_HD * d.Cin * 9 : (c == 1 ? (long)PO_HD * PO_HD : (long)d.C * PO_HD * 9);
    if (BWD) n = c == 0 ? (long)d.C * PO_HD * 9 : (c == 1 ? (long)PO_HD * PO_HD : 0);
    float m = 0.f;
    for (long i = threadIdx.x; i < n; i += 256) m = fmaxf(m, fabsf(w[i]));
#pragma unroll
    for (int off = 32; off > 0; off >>= 1) m = fmaxf(m, __shfl_xor(m, off, 64));
    if ((threadIdx.x & 63) == 0) sm[threadIdx.x >> 6] = m;
    __syncthreads();
    if (threadIdx.x == 0) {
        m = fmaxf(fmaxf(sm[0], sm[1]), fmaxf(sm[2], sm[3]));
        float sc, inv;
        po_scale_for_max(m, &sc, &inv);
        d.dst[c] = inv;
    }
}

template <bool BWD>
__global__ __launch_bounds__(256) void po_pack_fwd_kernel(const POPackDesc* __restrict__ descs) {
    const POPackDesc d = descs[blockIdx.y];
    // BWD: the first product is the data gradient of conv3 written as a 3x3 convolution of the C-channel gradient
    // image: w1'[c2][co][tap] = w3[co][c2][8 - tap] ("Cin" = C); the second is w2'[c1][c2] = w2[c2][c1]; no third
    const int Cin = BWD ? d.C : d.Cin;
    const POGeom g = po_geom(Cin, d.C, BWD);
    f16x8* dst = reinterpret_cast<f16x8*>(d.dst);
    float scw[3];
#pragma unroll
    for (int c = 0; c < 3; ++c) scw[c] = 1.0f / d.dst[c];   // written by po_pack_scale_kernel (exact powers of two)
    // one thread per (item, lane); item = one k-step of one row tile of one of the three products (2 planes)
    const int n1 = 8 * g.NS1, n2 = 8 * 16, n3 = g.NP * 16;
    const long n_items = (long)(n1 + n2 + n3) * 64;
    for (long it = (long)blockIdx.x * blockDim.x + threadIdx.x; it < n_items; it += (long)gridDim.x * blockDim.x) {
        const int lane = (int)(it & 63);
        int k = (int)(it >> 6);
        const int r = lane & 31, kk = lane >> 5;
        float v[8];
#pragma unroll
        for (int j = 0; j < 8; ++j) v[j] = 0.f;
        long frag0;
        float sc;
        if (k < n1) {
            const int a = k / g.NS1, s = k % g.NS1;
            frag0 = 1 + (long)(a >> 2) * g.quad_frags + (s / 5) * 40 + ((s % 5) * 4 + (a & 3)) * 2;
            sc = scw[0];
            const int tp = s / g.NG, gg = s % g.NG, tap = 2 * tp + kk;
            if (tap < 9) {
#pragma unroll
                for (int j = 0; j < 8; ++j) {
                    const int ci = 8 * gg + j;
                    if (ci < Cin)
                        v[j] = BWD ? d.w3[((long)ci * PO_HD + (32 * a + r)) * 9 + (8 - tap)]
                                   : d.w1[((long)(32 * a + r) * Cin + ci) * 9 + tap];
                }
            }
        } else if (k < n1 + n2) {
            k -= n1;
            const int a2 = k >> 4, s = k & 15;
            frag0 = 1 + (long)(s >> 3) * g.quad_frags + g.NG * 40 + ((s & 7) >> 1) * 32 + ((s & 1) * 8 + a2) * 2;
            sc = scw[1];
#pragma unroll
            for (int j = 0; j < 8; ++j) {
                const int c1 = 16 * s + 8 * (j >> 2) + 4 * kk + (j & 3);
                v[j] = BWD ? d.w2[(long)c1 * PO_HD + (32 * a2 + r)] : d.w2[(long)(32 * a2 + r) * PO_HD + c1];
            }
        } else {
            k -= n1 + n2;
            const int jt = k >> 4, s = k & 15;
            frag0 = g.c3 + (long)(s >> 2) * (8 * g.NP) + ((s & 3) * g.NP + jt) * 2;
            sc = scw[2];
            const int R = 32 * jt + r;
            if (R < 9 * d.C) {
                const int tap = R / d.C, co = R % d.C;
#pragma unroll
                for (int j = 0; j < 8; ++j) {
                    const int c2 = 16 * s + 8 * (j >> 2) + 4 * kk + (j & 3);
                    v[j] = d.w3[((long)co * PO_HD + c2) * 9 + tap];
                }
            }
        }
        f16x8 hi, lo;
        po_split_f16(v, sc, hi, lo);
        dst[frag0 * 64 + lane] = hi;
        dst[(frag0 + 1) * 64 + lane] = lo;
    }
}

extern "C" long rfn_coupling_po_packed_bytes(int Cin, int C) { return (long)po_geom(Cin, C).total_frags * 1024; }
extern "C" long rfn_coupling_po_bwd_packed_bytes(int C) { return (long)po_geom(C, C, true).total_frags * 1024; }

extern "C" int rfn_coupling_po_pack(const void* descs_device, int n, rfn_stream_t stream) {
    RFN_CHECK_ARG(descs_device && n >= 0, -1);
    if (n == 0) return 0;
    hipLaunchKernelGGL(po_pack_scale_kernel<false>, dim3(3, n), dim3(256), 0, (hipStream_t)stream,
                       reinterpret_cast<const POPackDesc*>(descs_device));
    hipLaunchKernelGGL(po_pack_fwd_kernel<false>, dim3(16, n), dim3(256), 0, (hipStream_t)stream,
                       reinterpret_cast<const POPackDesc*>(descs_device));
    RFN_LAUNCH_CHECK();
    return 0;
}

/* the backward kernel's stream of n coupling nets (descs: w2, w3, dst = rfn_coupling_po_bwd_packed_bytes(C) bytes, C;
 * w1 / Cin are not read) */
extern "C" int rfn_coupling_po_pack_bwd(const void* descs_device, int n, rfn_stream_t stream) {
    RFN_CHECK_ARG(descs_device && n >= 0, -1);
    if (n == 0) return 0;
    hipLaunchKernelGGL(po_pack_scale_kernel<true>, dim3(3, n), dim3(256), 0, (hipStream_t)stream,
                       reinterpret_cast<const POPackDesc*>(descs_device));
    hipLaunchKernelGGL(po_pack_fwd_kernel<true>, dim3(16, n), dim3(256), 0, (hipStream_t)stream,
                       reinterpret_cast<const POPackDesc*>(descs_device));
    RFN_LAUNCH_CHECK();
    return 0;
}

// ------------------------------------------------------------------------------------------------ forward kernel
struct POFwdParams {
    const float* z;     long z_ns;      // [N, >=Ch, H, W]: conv1 reads channels [0, Ch)
    const float* cond;  long cond_ns;   // [N, Cc, H, W]
    const unsigned char* wpk;           // forward weight stream (po_pack_fwd_kernel)
    const float* n1b; const float* n1l; const float* n2b; const float* n2l;   // ActNorm (bias, logs) of the hidden layers
    float* h1; long h1_ns; float* h2; long h2_ns;   // [N, 256, H, W] saved activations
    float* P;  long P_ns;                           // [N, 9C, H, W] tap-expanded conv3 output (no bias / scale)
    int Ch, Cc, C, N, H, W, logW, act;
    int rpf_shift;   // log2(rounds per frame), a round = 128 consecutive pixels of one frame
    int n_rounds;    // N * H * W / 128
    int IW, IPOS;    // haloed image of a round: (128/W + 2) rows x (W + 2) columns
    // activation masks, one uint4 per (round, thread): bit 16 (a & 1) + r of word a >> 1 is set when register r of the
    // thread's tile a (channels 32 a + 4 kk + 8 (r >> 2) + (r & 3) of its pixel) is NOT in the activation's linear
    // region (value <= 0).  Forward: written (m1 for h1, m2 for h2; nullptr = not wanted).  Backward: read (m1 masks the
    // first stage = h2's mask, m2 the second = h1's).
    unsigned* m1; unsigned* m2;
    float* part;     // backward: [gridDim.x][2][256] per-workgroup sums over pixels of the two stages' outputs
};

#define PO_MFMA(acc, a, b) acc = __builtin_amdgcn_mfma_f32_32x32x16_f16(a, b, acc, 0, 0, 0)

template <int N>
__device__ __forceinline__ void po_wait_vm() { asm volatile("s_waitcnt vmcnt(%0)" ::"n"(N) : "memory"); }

// one 32-channel x 32-pixel accumulator tile: v = act((acc * u + bias) * exp(logs)) in place, fp32 store of the 16
// values of this lane (channels 4kk + 8q + i of the tile at this lane's pixel), running max of |v|.
// pb -> bias of the tile's first channel of this lane half (exp(logs) 256 floats further);
// rsrc / voff: buffer descriptor of the output tensor and this lane's byte offset of (frame, first channel, pixel).
// mask: bit SH + r is set when value r is outside the activation's linear region (the backward kernel's act'(y))
template <int ACT, int SH>
__device__ __forceinline__ void po_epilogue(f32x16& acc, const float u, const float* pb, const __amdgpu_buffer_rsrc_t rsrc,
                                            const unsigned voff, const unsigned ch_bytes, float& vmax, unsigned& mask) {
#pragma unroll
    for (int q = 0; q < 4; ++q) {
        const f32x4 b4 = *reinterpret_cast<const f32x4*>(pb + 8 * q);
        const f32x4 e4 = *reinterpret_cast<const f32x4*>(pb + 256 + 8 * q);
#pragma unroll
        for (int i = 0; i < 4; ++i) {
            float t = fmaf(acc[4 * q + i], u, b4[i]) * e4[i];
            if (ACT == 1) t = fmaxf(t, 0.f);
            if (ACT == 2) t = fmaxf(t, 0.2f * t);
            if (ACT != 0) mask |= t > 0.f ? 0u : (1u << (SH + 4 * q + i));
            acc[4 * q + i] = t;
            vmax = fmaxf(vmax, fabsf(t));
            __builtin_amdgcn_raw_buffer_store_b32(__float_as_uint(t), rsrc, voff, (8 * q + i) * ch_bytes, 0);
        }
    }
}

// Backward counterpart: acc holds the gradient wrt the layer's activated output h = act((a + b) exp(l)); the tile
// becomes ga = acc * u * act'(.) * exp(l) = the gradient wrt the convolution output a (what the weight gradient and the
// next data gradient consume) and is stored.  Its sum over pixels (the ActNorm bias gradient; the gradient of the logs
// follows from the weight gradient, see po_bwd_finish_kernel) is gathered without leaving the registers: a two-level
// butterfly over the lanes of a quad turns each group of four registers into ONE register whose lane j of the quad
// holds the quad's sum of register j (9 VALU per 4 values), which is added to the running sums psum[4] of this tile
// (kept across all rounds of the workgroup; reduced over the quads once, at the end of the kernel).
// (A first version added the quad sums to LDS with ds_add_f32: an LDS float atomic costs the CU about 32 cycles per
// wave instruction however few lanes are active -- 1024 of them per round doubled the round time.)
template <int ACT>
__device__ __forceinline__ void po_epilogue_bwd(f32x16& acc, const float u, const float* pe,
                                                const __amdgpu_buffer_rsrc_t rsrc, const unsigned voff,
                                                const unsigned ch_bytes, float& vmax, const unsigned mbits,
                                                float (&psum)[4], const bool bit0, const bool bit1) {
#pragma unroll
    for (int q = 0; q < 4; ++q) {
        const f32x4 e4 = *reinterpret_cast<const f32x4*>(pe + 8 * q);
        float t4[4];
#pragma unroll
        for (int i = 0; i < 4; ++i) {
            float t = acc[4 * q + i] * u * e4[i];
            const bool off = (mbits >> (4 * q + i)) & 1u;
            if (ACT == 1) t = off ? 0.f : t;
            if (ACT == 2) t = off ? 0.2f * t : t;
            acc[4 * q + i] = t;
            t4[i] = t;
            vmax = fmaxf(vmax, fabsf(t));
            __builtin_amdgcn_raw_buffer_store_b32(__float_as_uint(t), rsrc, voff, (8 * q + i) * ch_bytes, 0);
        }
        // lane pair (l, l^1): even lanes gather register 0 (2), odd lanes register 1 (3)
        float s01 = (bit0 ? t4[1] : t4[0]) +
                    __int_as_float(__builtin_amdgcn_update_dpp(0, __float_as_int(bit0 ? t4[0] : t4[1]), 0xB1, 0xF, 0xF, true));
        float s23 = (bit0 ? t4[3] : t4[2]) +
                    __int_as_float(__builtin_amdgcn_update_dpp(0, __float_as_int(bit0 ? t4[2] : t4[3]), 0xB1, 0xF, 0xF, true));
        // lane pairs (l, l^2): lanes 0,1 of the quad keep registers 0,1, lanes 2,3 registers 2,3
        psum[q] += (bit1 ? s23 : s01) +
                   __int_as_float(__builtin_amdgcn_update_dpp(0, __float_as_int(bit1 ? s01 : s23), 0x4E, 0xF, 0xF, true));
    }
}

// biased exponent of the power-of-two scale for a maximum m (see po_scale_for_max): scale = 2^(141 - E), clamped
__device__ __forceinline__ int po_exp_of(const float m) {
    int E = (int)((__float_as_uint(m) >> 23) & 0xff);
    E = E < 40 ? 40 : E;
    return E > 250 ? 250 : E;
}

// BWD = the data-gradient chain of the same network on the same machinery (backward of glow_modules.py:232-238 from
// the gradient `go` at conv3's output; NP = 0):
//     gh2 = conv3^T go          a 3x3 convolution of the C-channel image go with w3 transposed + mirrored  ("conv1")
//     ga2 = gh2 act'(h2) exp(l2)    stored: the weight gradient of conv2 and of conv3's ... consume it    (epilogue)
//     gh1 = w2^T ga2            K-major on the register-resident ga2                                       ("conv2")
//     ga1 = gh1 act'(h1) exp(l1)    stored: weight gradient and data gradient of conv1 consume it          (epilogue)
// act'(.) comes from the 1-bit masks the forward kernel wrote in this kernel's own (round, thread, register) order, so
// the 637 MB activations are not read here at all; per-channel sums of ga2 / ga1 (ActNorm bias gradients) leave as
// per-workgroup partial rows.  Parameter roles in BWD: z = go (Ch = C, Cc = 0), n1l = l2, n2l = l1, h1 = ga2, h2 = ga1,
// m1 = mask of h2, m2 = mask of h1.
template <int NG, int NP, int LOGW, int ACT, bool BWD>
__global__ __launch_bounds__(64 * PO_WAVES) void coupling_po_fwd_kernel(const POFwdParams p) {
    // square maps of side W = 2^LOGW.  A round is 128 consecutive pixels: 128 / W image rows of one frame (H W >= 128) or
    // FPR = 128 / (H W) whole frames (8x8 maps: two).  Its haloed image: per frame part (RROWS + 2) rows of IW columns.
    constexpr int W = 1 << LOGW, HW = W * W, IW = W + 2;
    constexpr int FPR = HW >= PO_ROUND_PX ? 1 : PO_ROUND_PX / HW;
    constexpr int RROWS = HW >= PO_ROUND_PX ? PO_ROUND_PX / W : W;
    constexpr int FPOS = (RROWS + 2) * IW, IPOS = FPR * FPOS;
    constexpr int PO_ITEMS = (NG * IPOS + 64 * PO_WAVES - 1) / (64 * PO_WAVES);   // staging items per thread
    constexpr int RPF_SHIFT = HW >= PO_ROUND_PX ? 2 * LOGW - 7 : 0;   // log2(rounds per frame)
    constexpr int GS1 = po_gs1(NG), NG1 = 5 * NG / GS1;   // conv1: k-steps per ring group, groups per quad
    constexpr int GS3 = po_gs3(NP), NG3 = 16 / GS3;       // conv3: k-steps per ring group, groups
    constexpr bool RT = NG > 5;   // conv1 as run-time loops over channel groups inside the (unrolled) tap pairs
    // The next round's image is loaded one quad ahead into registers (8 PO_ITEMS of them) -- where the register file has
    // room.  The wide instantiations are at the 512-register limit without them (scratch spills otherwise) and load the
    // image when its LDS buffer is free, right before converting it: the tensors were written by the launch before this
    // one and come from L2 / the Infinity Cache.
    constexpr bool LATE = RT || (BWD && NG > 1);
    constexpr bool NOHOIST = LATE || NG == 5;   // staging decomposition recomputed every round (registers, see stage_load)
    static_assert((5 * NG) % GS1 == 0 && (!RT || NG % GS1 == 0), "conv1 grouping");
    constexpr int G3 = 2 * NP * GS3;           // fragments per conv3 group
    constexpr int Y1 = 2 * GS1, Y3 = G3 / 4;   // LDS-DMA instructions per wave of a conv1 / conv3 group (conv2: 8)
    static_assert(G3 % 4 == 0, "a group's fragments are shared evenly by the four waves");
    constexpr int SLOTF = po_slot_frags(NG, NP);
    constexpr int SLOT = SLOTF * 1024;
    extern __shared__ __attribute__((aligned(16))) unsigned char lds[];
    constexpr int NGRP = 2 * NG1 + 8 + (BWD ? 0 : NG3);                    // weight groups per round
    float* par = reinterpret_cast<float*>(lds + 3 * SLOT);                 // [4][256]: b1, exp(l1), b2, exp(l2)
    float* red = par + 1024;                                               // [8] block reductions
    f16x8* img = reinterpret_cast<f16x8*>(lds + 3 * SLOT + 4096 + 64);      // [plane 2][NG][IPOS]
    float* gsum = reinterpret_cast<float*>(lds + 3 * SLOT + 4096 + 64 + 2 * NG * IPOS * 16);   // BWD: [4 waves][2][256]
    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int l31 = lane & 31, kk = lane >> 5;
    const int Cin = p.Ch + p.Cc;
    constexpr unsigned ch_bytes = (unsigned)HW * 4u;

    for (int c = tid; c < 256; c += 64 * PO_WAVES) {
        par[c] = BWD ? 0.f : p.n1b[c];
        par[256 + c] = expf(p.n1l[c]);
        par[512 + c] = BWD ? 0.f : p.n2b[c];
        par[768 + c] = expf(p.n2l[c]);
    }
    // BWD: running per-channel sums of the two stages' outputs, packed: psN[a][q] lane j of a quad <-> channel
    // 32 a + 4 kk + 8 q + j, summed over the quad's pixels of every round of this workgroup
    float ps1[BWD ? 8 : 1][4], ps2[BWD ? 8 : 1][4];
#pragma unroll
    for (int a = 0; a < (BWD ? 8 : 1); ++a)
#pragma unroll
        for (int q = 0; q < 4; ++q) ps1[a][q] = ps2[a][q] = 0.f;
    const bool lbit0 = lane & 1, lbit1 = lane & 2;
    const float* hdr = reinterpret_cast<const float*>(p.wpk);
    const float inv_w1 = hdr[0], inv_w2 = hdr[1], inv_w3 = hdr[2];
    __syncthreads();

    int sl = 2;   // LDS slot of the group being consumed (0 -> 1 -> 2 -> 0 at every group boundary)
    // LDS-DMA of `nfr` fragments starting at stream fragment `base` into slot `slot`: wave w moves w, w+4, ...
    auto dma = [&](const int base, const int nfr, const int slot) {
        const unsigned char* src = p.wpk + (long)(base + wave) * 1024 + lane * 16;
        unsigned char* dst = lds + slot * SLOT + wave * 1024;
        for (int i = wave; i < nfr; i += PO_WAVES, src += PO_WAVES * 1024, dst += PO_WAVES * 1024)
            __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)src,
                                             (__attribute__((address_space(3))) void*)dst, 16, 0, 0);
    };
    // Boundary into group IDX of the round.  The ring holds the group being consumed, the next one (landing) and the one
    // after it, whose DMA is issued here into the slot everybody has just left.  Counted wait: YOUNGER = a lower bound of
    // the vector-memory operations this wave issued after the DMA of group IDX (the DMA of group IDX+1 = a quarter of its
    // fragments, plus activation stores; capped by the 6-bit counter) -- then my share of group IDX has landed; the
    // barrier makes that everybody's share.
#define PO_BOUNDARY(IDX, YOUNGER)                                                                          \
    do {                                                                                                   \
        po_wait_vm<(YOUNGER) < 63 ? (YOUNGER) : 63>();                                                     \
        __builtin_amdgcn_s_barrier();                                                                      \
        const int free_slot = sl;                                                                          \
        sl = sl == 2 ? 0 : sl + 1;                                                                         \
        const int n2_ = (IDX) + 2;                                                                         \
        if (n2_ < NGRP) dma(po_group_base(NG, NP, n2_), po_group_size(NG, NP, n2_), free_slot);            \
        else if (more) dma(po_group_base(NG, NP, n2_ - NGRP), po_group_size(NG, NP, n2_ - NGRP), free_slot); \
    } while (0)

    const auto rs_h1 = __builtin_amdgcn_make_buffer_rsrc(p.h1, 0, 0xFFFFFFFFu, 0x00020000);
    const auto rs_h2 = __builtin_amdgcn_make_buffer_rsrc(p.h2, 0, 0xFFFFFFFFu, 0x00020000);

    // ---- staging of the haloed, scaled, pre-split input image of a round, in two halves: (1) loads into registers,
    // (2) block maximum -> scale, conversion, LDS.  Item = (8-channel group, image position).
    float raw[PO_ITEMS][8];
    auto stage_load = [&](const int rnd) {
        const int n0_ = FPR > 1 ? rnd * FPR : rnd >> RPF_SHIFT;
        const int y0_ = FPR > 1 ? 0 : ((rnd & ((1 << RPF_SHIFT) - 1)) * PO_ROUND_PX) >> LOGW;
        // (wide instantiations: the items' decomposition is recomputed every round -- hoisted out of the round loop it is
        // 70+ registers the kernel does not have, i.e. scratch)
        int tid_ = tid;
        if (NOHOIST) asm volatile("" : "+v"(tid_));
#pragma unroll
        for (int it = 0; it < PO_ITEMS; ++it) {
            const int item = tid_ + it * 64 * PO_WAVES;
            const bool live = item < NG * IPOS;
            const int g = live ? item / IPOS : 0, pos = live ? item - g * IPOS : 0;
            const int fr = FPR > 1 ? pos / FPOS : 0, rem = pos - fr * FPOS;
            const int iy = rem / IW, ix = rem - iy * IW;
            const int gy = y0_ - 1 + iy, gx = ix - 1;
            const bool ok = live && gy >= 0 && gy < W && gx >= 0 && gx < W;
            const int off = ok ? gy * W + gx : 0;
            const float* zb = p.z + (long)(n0_ + fr) * p.z_ns;
            const float* cb = p.cond + (long)(n0_ + fr) * p.cond_ns;
#pragma unroll
            for (int j = 0; j < 8; ++j) {
                const int ci = 8 * g + j;
                const int cc = ci < Cin ? ci : 0;
                const float* src = cc < p.Ch ? zb + (long)cc * HW : cb + (long)(cc - p.Ch) * HW;
                raw[it][j] = src[off];   // (masked in stage_finish: no use of the value here, the load stays in flight)
            }
        }
    };
    float u1_next = 0.f;   // 1 / (image scale * weight scale) of the image staged last
    auto stage_finish = [&](const int rnd) {
        const int y0_ = FPR > 1 ? 0 : ((rnd & ((1 << RPF_SHIFT) - 1)) * PO_ROUND_PX) >> LOGW;
        float vm = 0.f;
        int tid_ = tid;
        if (NOHOIST) asm volatile("" : "+v"(tid_));
#pragma unroll
        for (int it = 0; it < PO_ITEMS; ++it) {
            const int item = tid_ + it * 64 * PO_WAVES;
            const bool live = item < NG * IPOS;
            const int g = live ? item / IPOS : 0, pos = live ? item - g * IPOS : 0;
            const int rem = FPR > 1 ? pos % FPOS : pos;
            const int iy = rem / IW, ix = rem - iy * IW;
            const int gy = y0_ - 1 + iy, gx = ix - 1;
            const bool ok = live && gy >= 0 && gy < W && gx >= 0 && gx < W;
#pragma unroll
            for (int j = 0; j < 8; ++j) {
                raw[it][j] = (ok && 8 * g + j < Cin) ? raw[it][j] : 0.f;
                vm = fmaxf(vm, fabsf(raw[it][j]));
            }
        }
#pragma unroll
        for (int off = 32; off > 0; off >>= 1) vm = fmaxf(vm, __shfl_xor(vm, off, 64));
        if (lane == 0) red[wave] = vm;
        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
        __builtin_amdgcn_s_barrier();
        vm = fmaxf(fmaxf(red[0], red[1]), fmaxf(red[2], red[3]));
        const int Eimg = po_exp_of(vm);
        const float sc_img = __uint_as_float((unsigned)(268 - Eimg) << 23);
        u1_next = __uint_as_float((unsigned)(Eimg - 14) << 23) * inv_w1;   // undoes image and weight scale
#pragma unroll
        for (int it = 0; it < PO_ITEMS; ++it) {
            const int item = tid_ + it * 64 * PO_WAVES;
            if (item < NG * IPOS) {
                const int g = item / IPOS, pos = item - g * IPOS;
                f16x8 hi, lo;
                po_split_f16(raw[it], sc_img, hi, lo);
                img[(0 * NG + g) * IPOS + pos] = hi;
                img[(1 * NG + g) * IPOS + pos] = lo;
            }
        }
        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");   // image (and red[] reads) done before the next barrier
    };

    int round = blockIdx.x;
    if (round < p.n_rounds) {
        const bool more = true;
        (void)more;
        dma(po_group_base(NG, NP, 0), po_group_size(NG, NP, 0), 0);
        dma(po_group_base(NG, NP, 1), po_group_size(NG, NP, 1), 1);
        stage_load(round);
        stage_finish(round);
    }
    for (; round < p.n_rounds; round += gridDim.x) {
        const bool more = round + (int)gridDim.x < p.n_rounds;
        // this lane's frame, its pixel there, and the first image row of the round in that frame
        const int rp = 32 * wave + l31;                                      // pixel of the round
        const int fr = FPR > 1 ? rp >> (2 * LOGW) : 0;                       // frame of the round
        const int n = FPR > 1 ? round * FPR + fr : round >> RPF_SHIFT;
        const int pix = FPR > 1 ? rp & (HW - 1) : (round & ((1 << RPF_SHIFT) - 1)) * PO_ROUND_PX + rp;
        const int y0 = FPR > 1 ? 0 : ((round & ((1 << RPF_SHIFT) - 1)) * PO_ROUND_PX) >> LOGW;
        const float u1 = u1_next;

        // its five tap-pair positions inside the image (lane half = tap of the pair)
        const int iy = (pix >> LOGW) - y0 + 1, ix = (pix & (W - 1)) + 1;
        const f16x8* bp[PO_TAP_PAIRS];
#pragma unroll
        for (int tp = 0; tp < PO_TAP_PAIRS; ++tp) {
            const int tA = 2 * tp, tB = 2 * tp + 1 < 9 ? 2 * tp + 1 : 4;   // the padding tap reads the centre (zero weight)
            const int dy = kk ? (tB / 3 - 1) : (tA / 3 - 1);
            const int dx = kk ? (tB % 3 - 1) : (tA % 3 - 1);
            bp[tp] = img + fr * FPOS + (iy + dy) * IW + ix + dx;
        }
        // byte offset of (frame n, channel 4kk, this pixel) in h1 / h2 (both [N,256,H,W] with frame strides h*_ns)
        const unsigned vo1 = (unsigned)(((long)n * p.h1_ns + (long)(4 * kk) * HW + pix) * 4);
        const unsigned vo2 = (unsigned)(((long)n * p.h2_ns + (long)(4 * kk) * HW + pix) * 4);
        const float* par1 = par + 4 * kk;
        const float* par2 = par + 512 + 4 * kk;

        // activation masks of this thread's (round, registers): written forward, read backward
        unsigned mk1[4] = {0u, 0u, 0u, 0u}, mk2[4] = {0u, 0u, 0u, 0u};   // (forward: dead, see the epilogues)
        if (BWD && ACT != 0) {
            const uint4 a_ = reinterpret_cast<const uint4*>(p.m1)[(long)round * (64 * PO_WAVES) + tid];
            const uint4 b_ = reinterpret_cast<const uint4*>(p.m2)[(long)round * (64 * PO_WAVES) + tid];
            mk1[0] = a_.x; mk1[1] = a_.y; mk1[2] = a_.z; mk1[3] = a_.w;
            mk2[0] = b_.x; mk2[1] = b_.y; mk2[2] = b_.z; mk2[3] = b_.w;
        }

        f32x16 acc2[8];   // conv2 accumulators = h2 (all 256 channels of this lane's pixel half)
#pragma unroll
        for (int a2 = 0; a2 < 8; ++a2)
#pragma unroll
            for (int r = 0; r < 16; ++r) acc2[a2][r] = 0.f;
        int Erun = 40;    // biased exponent of this pixel's running h1 maximum (scale of what is in acc2)

#pragma unroll
        for (int u = 0; u < 2; ++u) {
            if (u == 1 && more && !LATE) stage_load(round + (int)gridDim.x);   // next round's image: in flight under this quad
            // ================= conv1, output tiles 4u .. 4u+3: NG groups of 5 k-steps, four MFMA chains
            f32x16 Q[4];
#pragma unroll
            for (int t = 0; t < 4; ++t)
#pragma unroll
                for (int r = 0; r < 16; ++r) Q[t][r] = 0.f;
            // one k-step: three MFMAs per output tile (lo*hi, hi*lo, hi*hi), the next k-step's fragments read in between
#define PO_C1_STEP(LAST, ANEXT, BNEXT)                                                                     \
            do {                                                                                           \
                __builtin_amdgcn_sched_barrier(0);                                                         \
                _Pragma("unroll") for (int t = 0; t < 4; ++t) PO_MFMA(Q[t], A[cur][t][1], B[cur][0]);      \
                __builtin_amdgcn_sched_barrier(0);                                                         \
                if (!(LAST)) {                                                                             \
                    _Pragma("unroll") for (int t = 0; t < 4; ++t) {                                        \
                        A[nxt][t][0] = fr[((ANEXT) * 4 + t) * 2 * 64];                                     \
                        A[nxt][t][1] = fr[(((ANEXT) * 4 + t) * 2 + 1) * 64];                               \
                    }                                                                                      \
                    _Pragma("unroll") for (int pl = 0; pl < 2; ++pl) B[nxt][pl] = BNEXT;                   \
                }                                                                                          \
                __builtin_amdgcn_sched_barrier(0);                                                         \
                _Pragma("unroll") for (int t = 0; t < 4; ++t) PO_MFMA(Q[t], A[cur][t][0], B[cur][1]);      \
                _Pragma("unroll") for (int t = 0; t < 4; ++t) PO_MFMA(Q[t], A[cur][t][0], B[cur][0]);      \
                __builtin_amdgcn_sched_barrier(0);                                                         \
            } while (0)
            if constexpr (!RT) {
#pragma unroll
            for (int gi = 0; gi < NG1; ++gi) {
                // younger than this group's DMA: the DMA of the next group (Y1 / 8 instructions per wave)
                // (+ the 8 PO_ITEMS image loads of the next round while they are in flight: quad 1, first two groups)
                if (u == 1 && gi < 2 && more && !LATE) {
                    if (gi + 1 < NG1) PO_BOUNDARY(u * (NG1 + 4) + gi, Y1 + 8 * PO_ITEMS);
                    else PO_BOUNDARY(u * (NG1 + 4) + gi, 8 + 8 * PO_ITEMS);
                } else {
                    if (gi + 1 < NG1) PO_BOUNDARY(u * (NG1 + 4) + gi, Y1);
                    else PO_BOUNDARY(u * (NG1 + 4) + gi, 8);
                }
                const f16x8* fr = reinterpret_cast<const f16x8*>(lds + sl * SLOT) + lane;
                f16x8 A[2][4][2], B[2][2];
#pragma unroll
                for (int t = 0; t < 4; ++t) {
                    A[0][t][0] = fr[(t * 2) * 64];
                    A[0][t][1] = fr[(t * 2 + 1) * 64];
                }
#pragma unroll
                for (int pl = 0; pl < 2; ++pl) B[0][pl] = bp[(GS1 * gi) / NG][(pl * NG + (GS1 * gi) % NG) * IPOS];
#pragma unroll
                for (int k = 0; k < GS1; ++k) {
                    const int s = GS1 * gi + k, cur = k & 1, nxt = cur ^ 1;
                    PO_C1_STEP(k + 1 == GS1, k + 1, bp[(s + 1) / NG][(pl * NG + (s + 1) % NG) * IPOS]);
                }
            }
            } else {
            // wide inputs: k-step s = tp * NG + g with the tap pairs unrolled and the channel groups of a tap pair in a
            // run-time loop over ring groups of GS1 (a fully unrolled round is 10 NG k-steps with compile-time
            // addresses: at NG = 9 hipcc spills hundreds of SGPRs and its AGPR pass fails)
#pragma unroll
            for (int tp = 0; tp < PO_TAP_PAIRS; ++tp) {
#pragma unroll 1
                for (int gg = 0; gg < NG / GS1; ++gg) {
                    const int gq = tp * (NG / GS1) + gg;          // ring group of the quad
                    const int gidx = u * (NG1 + 4) + gq;          // ring group of the round
                    // younger: the next group's DMA (at least Y1 instructions) -- and, in quad 1, the image loads of the
                    // next round, issued before its first group and still in flight during its first two
                    if (u == 1 && tp == 0 && gg < 2 && more && !LATE) PO_BOUNDARY(gidx, Y1 + 8 * PO_ITEMS);
                    else PO_BOUNDARY(gidx, Y1);
                    const f16x8* fr = reinterpret_cast<const f16x8*>(lds + sl * SLOT) + lane;
                    const f16x8* bq = bp[tp] + (GS1 * gg) * IPOS;   // channel group GS1 gg of this tap pair, plane hi
                    f16x8 A[2][4][2], B[2][2];
#pragma unroll
                    for (int t = 0; t < 4; ++t) {
                        A[0][t][0] = fr[(t * 2) * 64];
                        A[0][t][1] = fr[(t * 2 + 1) * 64];
                    }
#pragma unroll
                    for (int pl = 0; pl < 2; ++pl) B[0][pl] = bq[(pl * NG) * IPOS];
#pragma unroll
                    for (int k = 0; k < GS1; ++k) {
                        const int cur = k & 1, nxt = cur ^ 1;
                        PO_C1_STEP(k + 1 == GS1, k + 1, bq[(pl * NG + k + 1) * IPOS]);
                    }
                }
            }
            }
#undef PO_C1_STEP
            // ---- epilogue of the four tiles: ActNorm + act, store h1, per-pixel maximum
            float vmax = 0.f;
            unsigned mq[2] = {0u, 0u};   // forward: the quad's two mask words (words 2u, 2u+1 of the thread's uint4), stored at once
#pragma unroll
            for (int t = 0; t < 4; ++t) {
                const int a = 4 * u + t;   // tile: mask word a >> 1, bits 16 (a & 1) ..
                if (BWD)
                    po_epilogue_bwd<ACT>(Q[t], u1, par1 + 256 + 32 * a, rs_h1, vo1 + (unsigned)(32 * a) * ch_bytes, ch_bytes,
                                         vmax, mk1[a >> 1] >> (16 * (a & 1)), ps1[BWD ? a : 0], lbit0, lbit1);
                else if (t & 1)
                    po_epilogue<ACT, 16>(Q[t], u1, par1 + 32 * a, rs_h1, vo1 + (unsigned)(32 * a) * ch_bytes, ch_bytes,
                                         vmax, mq[t >> 1]);
                else
                    po_epilogue<ACT, 0>(Q[t], u1, par1 + 32 * a, rs_h1, vo1 + (unsigned)(32 * a) * ch_bytes, ch_bytes,
                                        vmax, mq[t >> 1]);
            }
            if (!BWD && ACT != 0 && p.m1)
                reinterpret_cast<uint2*>(p.m1)[((long)round * (64 * PO_WAVES) + tid) * 2 + u] = make_uint2(mq[0], mq[1]);
            vmax = fmaxf(vmax, __shfl_xor(vmax, 32, 64));   // the other lane half holds the pixel's other channels
            const int Enew = max(Erun, po_exp_of(vmax));
            if (u > 0 && __any(Enew != Erun)) {
                // a larger h1 value appeared for some pixel: bring its conv2 partial sums to the new (coarser) scale
                const int dE = Erun - Enew;   // <= 0
                const float f = __uint_as_float((unsigned)(dE < -126 ? 0 : 127 + dE) << 23);
#pragma unroll
                for (int a2 = 0; a2 < 8; ++a2)
#pragma unroll
                    for (int r = 0; r < 16; ++r) acc2[a2][r] *= f;
            }
            Erun = Enew;
            const float sc_h = __uint_as_float((unsigned)(268 - Erun) << 23);
            f16x8 hq[8][2];   // the quad's 8 k-steps as B fragments (hi, lo)
#pragma unroll
            for (int t = 0; t < 4; ++t)
#pragma unroll
                for (int sp = 0; sp < 2; ++sp) {
                    float w8[8];
#pragma unroll
                    for (int j = 0; j < 8; ++j) w8[j] = Q[t][8 * sp + j];
                    po_split_f16(w8, sc_h, hq[2 * t + sp][0], hq[2 * t + sp][1]);
                }

            // ================= conv2, K-major: h1 k-steps 8u .. 8u+7 into all eight accumulators; 4 groups of 2 k-steps,
            // each k-step in two halves of four output tiles (four MFMA chains, fragment reads one half ahead)
#pragma unroll
            for (int g2 = 0; g2 < 4; ++g2) {
                // younger: next group's DMA (8; after the last one 10 = conv1 or 2 NP = conv3) and, for the first two groups,
                // the 64 h1 stores of the quad's epilogue
                if (g2 < 2) PO_BOUNDARY(u * (NG1 + 4) + NG1 + g2, 8 + 64);
                else if (g2 == 2) PO_BOUNDARY(u * (NG1 + 4) + NG1 + g2, 8);
                else if (u == 0) PO_BOUNDARY(u * (NG1 + 4) + NG1 + g2, Y1);
                else if (!BWD) PO_BOUNDARY(u * (NG1 + 4) + NG1 + g2, Y3);
                else if (more) PO_BOUNDARY(u * (NG1 + 4) + NG1 + g2, Y1);   // (next: the next round's first group)
                else PO_BOUNDARY(u * (NG1 + 4) + NG1 + g2, 0);
                const f16x8* fr = reinterpret_cast<const f16x8*>(lds + sl * SLOT) + lane;
                f16x8 A[2][4][2];
#pragma unroll
                for (int t = 0; t < 4; ++t) {
                    A[0][t][0] = fr[(t * 2) * 64];
                    A[0][t][1] = fr[(t * 2 + 1) * 64];
                }
#pragma unroll
                for (int hstep = 0; hstep < 4; ++hstep) {   // (k-step of the group, half)
                    const int kq = 2 * g2 + (hstep >> 1), half = hstep & 1, cur = hstep & 1, nxt = cur ^ 1;
                    __builtin_amdgcn_sched_barrier(0);
#pragma unroll
                    for (int t = 0; t < 4; ++t) PO_MFMA(acc2[4 * half + t], A[cur][t][1], hq[kq][0]);
                    __builtin_amdgcn_sched_barrier(0);
                    if (hstep + 1 < 4) {
#pragma unroll
                        for (int t = 0; t < 4; ++t) {
                            A[nxt][t][0] = fr[(((hstep + 1) * 4 + t) * 2) * 64];
                            A[nxt][t][1] = fr[(((hstep + 1) * 4 + t) * 2 + 1) * 64];
                        }
                    }
                    __builtin_amdgcn_sched_barrier(0);
#pragma unroll
                    for (int t = 0; t < 4; ++t) PO_MFMA(acc2[4 * half + t], A[cur][t][0], hq[kq][1]);
#pragma unroll
                    for (int t = 0; t < 4; ++t) PO_MFMA(acc2[4 * half + t], A[cur][t][0], hq[kq][0]);
                    __builtin_amdgcn_sched_barrier(0);
                }
            }
        }

        // the image buffer is free (every wave passed four barriers since its last conv1 read): next round's image
        if (more) {
            if (LATE) stage_load(round + (int)gridDim.x);
            stage_finish(round + (int)gridDim.x);
        }

        // ================= h2 = act(ActNorm(conv2)): store, per-pixel maximum over all 256 channels, conversion
        const float u2 = __uint_as_float((unsigned)(Erun - 14) << 23) * inv_w2;
        float vmax2 = 0.f;
#pragma unroll
        for (int a2 = 0; a2 < 8; ++a2) {
            if (BWD) {
                po_epilogue_bwd<ACT>(acc2[a2], u2, par2 + 256 + 32 * a2, rs_h2, vo2 + (unsigned)(32 * a2) * ch_bytes,
                                     ch_bytes, vmax2, mk2[a2 >> 1] >> (16 * (a2 & 1)), ps2[BWD ? a2 : 0], lbit0, lbit1);
            } else {
                unsigned mw = 0u;   // forward: one mask word per pair of tiles, stored as soon as it is complete
                if (a2 & 1) continue;
                po_epilogue<ACT, 0>(acc2[a2], u2, par2 + 32 * a2, rs_h2, vo2 + (unsigned)(32 * a2) * ch_bytes, ch_bytes,
                                    vmax2, mw);
                po_epilogue<ACT, 16>(acc2[a2 + 1], u2, par2 + 32 * (a2 + 1), rs_h2, vo2 + (unsigned)(32 * (a2 + 1)) * ch_bytes,
                                     ch_bytes, vmax2, mw);
                if (ACT != 0 && p.m2) p.m2[((long)round * (64 * PO_WAVES) + tid) * 4 + (a2 >> 1)] = mw;
            }
        }
        if constexpr (!BWD) {
        vmax2 = fmaxf(vmax2, __shfl_xor(vmax2, 32, 64));
        const int E3 = po_exp_of(vmax2);
        const float sc3 = __uint_as_float((unsigned)(268 - E3) << 23);
        const float u3 = __uint_as_float((unsigned)(E3 - 14) << 23) * inv_w3;
        // h2 as B fragments: converted at once (NP <= 3) or, where NP accumulators + fragments leave no room for all of
        // them beside h2 itself, one k-step at a time from the accumulators as conv3 proceeds
        constexpr bool LAZY3 = NP > 3;
        f16x8 h2f[LAZY3 ? 1 : 16][2];
        if constexpr (!LAZY3) {
#pragma unroll
        for (int a2 = 0; a2 < 8; ++a2)
#pragma unroll
            for (int sp = 0; sp < 2; ++sp) {
                float w8[8];
#pragma unroll
                for (int j = 0; j < 8; ++j) w8[j] = acc2[a2][8 * sp + j];
                po_split_f16(w8, sc3, h2f[2 * a2 + sp][0], h2f[2 * a2 + sp][1]);
            }
        }

        // ================= tap-expanded conv3: NG3 groups of GS3 k-steps, NP chains
        f32x16 Pacc[NP];
#pragma unroll
        for (int jt = 0; jt < NP; ++jt)
#pragma unroll
            for (int r = 0; r < 16; ++r) Pacc[jt][r] = 0.f;
#pragma unroll
        for (int g3 = 0; g3 < NG3; ++g3) {
            // younger: next group's DMA (Y3; after the last one the next round's first group, if any) and, for the first
            // two groups, the 128 h2 stores
            if (g3 < 2) PO_BOUNDARY(2 * NG1 + 8 + g3, Y3 + 128);
            else if (g3 + 1 < NG3) PO_BOUNDARY(2 * NG1 + 8 + g3, Y3);
            else if (more) PO_BOUNDARY(2 * NG1 + 8 + g3, Y1);
            else PO_BOUNDARY(2 * NG1 + 8 + g3, 0);
            const f16x8* fr = reinterpret_cast<const f16x8*>(lds + sl * SLOT) + lane;
            f16x8 A[2][NP][2];
#pragma unroll
            for (int jt = 0; jt < NP; ++jt) {
                A[0][jt][0] = fr[(jt * 2) * 64];
                A[0][jt][1] = fr[(jt * 2 + 1) * 64];
            }
#pragma unroll
            for (int k = 0; k < GS3; ++k) {
                const int s = GS3 * g3 + k, cur = k & 1, nxt = cur ^ 1;
                const int sf = LAZY3 ? 0 : s;
                if constexpr (LAZY3) {
                    float w8[8];
#pragma unroll
                    for (int j = 0; j < 8; ++j) w8[j] = acc2[s >> 1][8 * (s & 1) + j];
                    po_split_f16(w8, sc3, h2f[0][0], h2f[0][1]);
                }
                __builtin_amdgcn_sched_barrier(0);
#pragma unroll
                for (int jt = 0; jt < NP; ++jt) PO_MFMA(Pacc[jt], A[cur][jt][1], h2f[sf][0]);
                __builtin_amdgcn_sched_barrier(0);
                if (k + 1 < GS3) {
#pragma unroll
                    for (int jt = 0; jt < NP; ++jt) {
                        A[nxt][jt][0] = fr[(((k + 1) * NP + jt) * 2) * 64];
                        A[nxt][jt][1] = fr[(((k + 1) * NP + jt) * 2 + 1) * 64];
                    }
                }
                __builtin_amdgcn_sched_barrier(0);
#pragma unroll
                for (int jt = 0; jt < NP; ++jt) PO_MFMA(Pacc[jt], A[cur][jt][0], h2f[sf][1]);
#pragma unroll
                for (int jt = 0; jt < NP; ++jt) PO_MFMA(Pacc[jt], A[cur][jt][0], h2f[sf][0]);
                __builtin_amdgcn_sched_barrier(0);
            }
        }
        // ---- tap-expanded conv3 output
        float* Po = p.P + (long)n * p.P_ns + pix;
#pragma unroll
        for (int jt = 0; jt < NP; ++jt)
#pragma unroll
            for (int r = 0; r < 16; ++r) {
                const int R = 32 * jt + (r & 3) + 8 * (r >> 2) + 4 * kk;
                if (R < 9 * p.C) Po[(long)R * HW] = Pacc[jt][r] * u3;
            }
        }   // !BWD
    }
#undef PO_BOUNDARY
    if (BWD) {
        // running sums: over the 8 quads of each 32-lane half (two rotations inside the 16-lane rows, then the other
        // row), each wave into its own LDS row, then the four rows together -> the workgroup's row of partial sums
#pragma unroll
        for (int st = 0; st < 2; ++st)
#pragma unroll
            for (int a = 0; a < 8; ++a)
#pragma unroll
                for (int q = 0; q < 4; ++q) {
                    float v = st == 0 ? ps1[a][q] : ps2[a][q];
                    v += __int_as_float(__builtin_amdgcn_update_dpp(0, __float_as_int(v), 0x124, 0xF, 0xF, true));  // row_ror:4
                    v += __int_as_float(__builtin_amdgcn_update_dpp(0, __float_as_int(v), 0x128, 0xF, 0xF, true));  // row_ror:8
                    v += __shfl_xor(v, 16, 64);
                    if (l31 < 4) gsum[(wave * 2 + st) * 256 + 32 * a + 4 * kk + 8 * q + l31] = v;
                }
        __syncthreads();
        for (int c = tid; c < 512; c += 64 * PO_WAVES)
            p.part[(long)blockIdx.x * 512 + c] = (gsum[c] + gsum[512 + c]) + (gsum[1024 + c] + gsum[1536 + c]);
    }
}

template <int NG, int NP, int LOGW, int ACT, bool BWD>
static int launch_po_fwd_t(const POFwdParams& p, hipStream_t s) {
    constexpr int SLOTF = po_slot_frags(NG, NP);
    constexpr int W = 1 << LOGW, IPOS = po_image_positions(W);
    const size_t ldsz = (size_t)3 * SLOTF * 1024 + 4096 + 64 + (size_t)2 * NG * IPOS * 16 + (BWD ? PO_WAVES * 512 * 4 : 0);
    if (ldsz > 160 * 1024 || IPOS != p.IPOS) {
        rfn_set_error("coupling_po: %zu bytes of LDS / %d image positions (expected %d)", ldsz, p.IPOS, IPOS);
        return -5;
    }
    auto kern = coupling_po_fwd_kernel<NG, NP, LOGW, ACT, BWD>;
    (void)hipFuncSetAttribute((const void*)kern, hipFuncAttributeMaxDynamicSharedMemorySize, (int)ldsz);
    const int grid = p.n_rounds < 256 ? p.n_rounds : 256;
    hipLaunchKernelGGL(kern, dim3(grid), dim3(64 * PO_WAVES), ldsz, s, p);
    return 0;
}

template <int NG, int NP, int LOGW, bool BWD = false>
static int launch_po_fwd(const POFwdParams& p, hipStream_t s) {
    if (p.act == 1) return launch_po_fwd_t<NG, NP, LOGW, 1, BWD>(p, s);
    if (p.act == 2) return launch_po_fwd_t<NG, NP, LOGW, 2, BWD>(p, s);
    return launch_po_fwd_t<NG, NP, LOGW, 0, BWD>(p, s);
}

// instantiations of the forward kernel: (channel groups of conv1's input, row tiles of the tap-expanded conv3, map side)
//   (3, 2, 32) / (5, 3, 16)   levels 0 / 1 of the canonical flow (Cin 18 / 36, C 4 / 8)
//   (9, 5, 8)                 level 2 of the canonical flow (Cin 72, C 16): two frames per round
//   (9, 4, 32)                level 0 of the BAIR-shaped flow (C = 12, Cin 65..72: 6 + 64 `with_skip` condition channels)
static bool po_fwd_instance(int NG, int NP, int W) {
    return (NG == 3 && NP == 2 && W == 32) || (NG == 5 && NP == 3 && W == 16) || (NG == 9 && NP == 5 && W == 8) ||
           (NG == 9 && NP == 4 && W == 32);
}
// shapes the fused kernel takes (the host asks before choosing this path)
extern "C" int rfn_coupling_po_supported(int N, int C, int Cc, int Hd, int H, int W) {
    if (N <= 0 || C <= 0 || C % 2 || Cc < 0 || H <= 0 || W <= 0) return 0;
    const POGeom g = po_geom(C / 2 + Cc, C);
    const bool pow2 = (H & (H - 1)) == 0 && (W & (W - 1)) == 0;
    const bool inst = H == W && po_fwd_instance(g.NG, g.NP, W);
    // output tensors are addressed with 32-bit byte offsets (buffer stores)
    const bool small = (long)N * PO_HD * H * W * 4 < (1L << 32);
    // a round is 128 consecutive pixels of the (frame, pixel) sequence: whole rounds only
    return Hd == PO_HD && pow2 && W <= PO_ROUND_PX && ((long)N * H * W) % PO_ROUND_PX == 0 && inst && small;
}

/* floats per activation-mask tensor (as fp32 elements: 4 per (round, thread)) and per partial-sum buffer of the
 * backward kernel for N frames of H x W pixels */
extern "C" long rfn_coupling_po_mask_floats(int N, int H, int W) { return (long)N * H * W / PO_ROUND_PX * (64 * PO_WAVES) * 4; }
extern "C" long rfn_coupling_po_bwd_part_floats(int N, int H, int W) {
    const long r = (long)N * H * W / PO_ROUND_PX;
    return (r < 256 ? r : 256) * 512;
}

static void po_fill_geometry(POFwdParams& p, int N, int C, int H, int W, int act) {
    p.C = C; p.N = N; p.H = H; p.W = W; p.logW = ilog2(W); p.act = act;
    p.rpf_shift = H * W >= PO_ROUND_PX ? ilog2(H * W / PO_ROUND_PX) : 0;
    p.n_rounds = (int)((long)N * H * W / PO_ROUND_PX);
    p.IW = W + 2;
    p.IPOS = po_image_positions(W);
}

/* ---- a5 (fused)  AffineCoupling.net forward  (Flow/glow_modules.py:232-238 with :119-121, :139-142): see the header
 * of this file.  z: output of ActNorm+InvConv (channels [0, C/2) are read), cond: the condition tensor.
 * Outputs: h1, h2 [N,256,H,W] (saved for the backward pass) and P [N, 9C, H, W] with
 * P[tap*C + co] = Σ_c w3[co][c][tap] h2[c] -- rfn_tap_gather_f32 turns P into the Conv2dZeros output.
 * m1 / m2 (optional, rfn_coupling_po_mask_floats(N, H, W) floats each): 1-bit activation masks of h1 / h2 in the
 * backward kernel's own order (rfn_coupling_po_bwd). */
extern "C" int rfn_coupling_po_fwd(const float* z, long z_ns, const float* cond, long cond_ns, const void* wpk,
                                   const float* n1b, const float* n1l, const float* n2b, const float* n2l, float* h1,
                                   long h1_ns, float* h2, long h2_ns, float* P, long P_ns, float* m1, float* m2, int N,
                                   int C, int Cc, int H, int W, int act, rfn_stream_t stream) {
    RFN_CHECK_ARG(z && wpk && n1b && n1l && n2b && n2l && h1 && h2 && P && (Cc == 0 || cond), -1);
    RFN_CHECK_ARG(rfn_coupling_po_supported(N, C, Cc, PO_HD, H, W), -2);
    RFN_CHECK_ARG(((uintptr_t)wpk & 15) == 0 && ((uintptr_t)m1 & 15) == 0 && ((uintptr_t)m2 & 15) == 0, -3);
    RFN_CHECK_ARG(h1_ns * 4L * N < (1L << 32) && h2_ns * 4L * N < (1L << 32), -4);
    POFwdParams p;
    memset(&p, 0, sizeof(p));
    p.z = z; p.z_ns = z_ns; p.cond = cond ? cond : z; p.cond_ns = cond_ns;
    p.wpk = reinterpret_cast<const unsigned char*>(wpk);
    p.n1b = n1b; p.n1l = n1l; p.n2b = n2b; p.n2l = n2l;
    p.h1 = h1; p.h1_ns = h1_ns; p.h2 = h2; p.h2_ns = h2_ns; p.P = P; p.P_ns = P_ns;
    p.m1 = reinterpret_cast<unsigned*>(m1); p.m2 = reinterpret_cast<unsigned*>(m2);
    p.Ch = C / 2; p.Cc = Cc;
    po_fill_geometry(p, N, C, H, W, act);
    const POGeom g = po_geom(p.Ch + Cc, C);
    int rc = -4;
    if (g.NG == 3 && g.NP == 2 && W == 32) rc = launch_po_fwd<3, 2, 5>(p, (hipStream_t)stream);
    if (g.NG == 5 && g.NP == 3 && W == 16) rc = launch_po_fwd<5, 3, 4>(p, (hipStream_t)stream);
    if (g.NG == 9 && g.NP == 5 && W == 8) rc = launch_po_fwd<9, 5, 3>(p, (hipStream_t)stream);
    if (g.NG == 9 && g.NP == 4 && W == 32) rc = launch_po_fwd<9, 4, 5>(p, (hipStream_t)stream);
    if (rc) return rc;
    RFN_LAUNCH_CHECK();
    return 0;
}

/* ---- a5 (fused, backward)  data-gradient chain of AffineCoupling.net (backward of Flow/glow_modules.py:232-238):
 * from go = gradient at conv3's output [N, C, H, W] (dense frames, stride go_ns) to
 *     ga2 = (conv3^T go) act'(h2) exp(n2l)   [N, 256, H, W]   gradient at conv2's output
 *     ga1 = (w2^T ga2)   act'(h1) exp(n1l)   [N, 256, H, W]   gradient at conv1's output
 * in ONE kernel (same machinery and arithmetic as the forward kernel: f16x3s, a wave owns 32 pixels, ga2 stays in
 * registers on its way into the second product).  act'(.) is read from the forward kernel's 1-bit masks m_h1 / m_h2.
 * wpk: rfn_coupling_po_pack_bwd stream.  part: rfn_coupling_po_bwd_part_floats(N, H, W) floats, WRITTEN: per-workgroup
 * sums over pixels of ga2 ([.][0][256]) and ga1 ([.][1][256]); rfn_coupling_po_bwd_finish turns them (and the weight
 * gradients) into the ActNorm gradients. */
// instantiations: (channel groups of the gradient image, map side): (1, 32) / (1, 16) levels 0 / 1 of the canonical flow
// (C 4 / 8), (2, 8) its level 2 (C 16), (2, 32) level 0 of the BAIR-shaped flow (C 12) -- the levels whose forward pass
// runs on the fused kernel (the backward kernel needs its activation masks)
static bool po_bwd_instance(int NG, int W) {
    return (NG == 1 && (W == 32 || W == 16)) || (NG == 2 && (W == 8 || W == 32));
}
extern "C" int rfn_coupling_po_bwd_supported(int N, int C, int H, int W) {
    if (N <= 0 || C <= 0 || H != W) return 0;
    const bool small = (long)N * PO_HD * H * W * 4 < (1L << 32);
    return po_bwd_instance((C + 7) / 8, W) && ((long)N * H * W) % PO_ROUND_PX == 0 && small;
}
extern "C" int rfn_coupling_po_bwd(const float* go, long go_ns, const void* wpk, const float* n1l, const float* n2l,
                                   const float* m_h1, const float* m_h2, float* ga2, long ga2_ns, float* ga1,
                                   long ga1_ns, float* part, int N, int C, int H, int W, int act, rfn_stream_t stream) {
    RFN_CHECK_ARG(go && wpk && n1l && n2l && ga2 && ga1 && part && (act == 0 || (m_h1 && m_h2)), -1);
    RFN_CHECK_ARG(rfn_coupling_po_bwd_supported(N, C, H, W), -2);
    RFN_CHECK_ARG(((uintptr_t)wpk & 15) == 0 && ((uintptr_t)m_h1 & 15) == 0 && ((uintptr_t)m_h2 & 15) == 0, -3);
    RFN_CHECK_ARG(ga1_ns * 4L * N < (1L << 32) && ga2_ns * 4L * N < (1L << 32), -4);
    POFwdParams p;
    memset(&p, 0, sizeof(p));
    p.z = go; p.z_ns = go_ns; p.cond = go; p.cond_ns = 0;
    p.wpk = reinterpret_cast<const unsigned char*>(wpk);
    p.n1b = n2l; p.n1l = n2l; p.n2b = n1l; p.n2l = n1l;     // stage 1 scales by exp(l2), stage 2 by exp(l1); no biases
    p.h1 = ga2; p.h1_ns = ga2_ns; p.h2 = ga1; p.h2_ns = ga1_ns;
    p.m1 = reinterpret_cast<unsigned*>(const_cast<float*>(m_h2));
    p.m2 = reinterpret_cast<unsigned*>(const_cast<float*>(m_h1));
    p.part = part;
    p.Ch = C; p.Cc = 0;
    po_fill_geometry(p, N, C, H, W, act);
    int rc = -4;
    const int NGb = (C + 7) / 8;
    if (NGb == 1 && W == 32) rc = launch_po_fwd<1, 0, 5, true>(p, (hipStream_t)stream);
    if (NGb == 1 && W == 16) rc = launch_po_fwd<1, 0, 4, true>(p, (hipStream_t)stream);
    if (NGb == 2 && W == 8) rc = launch_po_fwd<2, 0, 3, true>(p, (hipStream_t)stream);
    if (NGb == 2 && W == 32) rc = launch_po_fwd<2, 0, 5, true>(p, (hipStream_t)stream);
    if (rc) return rc;
    RFN_LAUNCH_CHECK();
    return 0;
}

/* ActNorm gradients of the two hidden layers of up to 16 coupling nets (the K steps of a level) in one launch:
 *     gnb[c] = sum over workgroups of part[.][layer][c]                      (= sum over pixels of ga)
 *     gnl[c] = sum_k w[c][k] gw[c][k] + nb[c] gnb[c]
 * The second line is sum over pixels of gy*y (the gradient of the ActNorm logs, glow_modules.py:38-45 backward) rewritten
 * with y = (a + nb) exp(nl), ga = gy exp(nl) and sum_pix ga[c] a[c] = sum_k w[c][k] gw[c][k] (a = w x, gw = the weight
 * gradient): the kernel that made ga never needs the activations themselves. */
#define PO_FIN_MAX 16
struct POFinishParams {
    const float* part[PO_FIN_MAX];
    const float* w1[PO_FIN_MAX]; const float* gw1[PO_FIN_MAX]; const float* n1b[PO_FIN_MAX];
    const float* w2[PO_FIN_MAX]; const float* gw2[PO_FIN_MAX]; const float* n2b[PO_FIN_MAX];
    float* out[PO_FIN_MAX];   // [4][256]: gn1b, gn1l, gn2b, gn2l
    int nblk, K1;             // rows of `part`; elements per output channel of w1 (Cin * 9)
};
// grid (8 groups of 32 channels, 2 layers, n nets), 256 threads
__global__ __launch_bounds__(256) void po_bwd_finish_kernel(const POFinishParams p) {
    __shared__ float red[8][32];
    const int g = blockIdx.z, layer = blockIdx.y, c0 = 32 * blockIdx.x, tid = threadIdx.x;
    // part rows hold [stage 0 = ga2 sums (layer 2), stage 1 = ga1 sums (layer 1)]: 8 row-slices x 32 channels
    {
        const int bl = tid >> 5, c = tid & 31;
        const float* part = p.part[g] + (layer == 0 ? 256 : 0) + c0 + c;
        float sgb = 0.f;
        for (int b = bl; b < p.nblk; b += 8) sgb += part[(long)b * 512];
        red[bl][c] = sgb;
    }
    __syncthreads();
    const int wave = tid >> 6, lane = tid & 63;
    const int K = layer == 0 ? p.K1 : PO_HD;
    const float* wb = layer == 0 ? p.w1[g] : p.w2[g];
    const float* gwb = layer == 0 ? p.gw1[g] : p.gw2[g];
    const float* nbp = layer == 0 ? p.n1b[g] : p.n2b[g];
    float* out = p.out[g] + layer * 512;
    for (int ci = wave; ci < 32; ci += 4) {     // a wave per channel: coalesced rows of w and gw
        const int c = c0 + ci;
        float dot = 0.f;
        for (int k = lane; k < K; k += 64) dot = fmaf(wb[(long)c * K + k], gwb[(long)c * K + k], dot);
        dot = wave_sum_dpp(dot);
        if (lane == 0) {
            const float gb = ((red[0][ci] + red[1][ci]) + (red[2][ci] + red[3][ci])) +
                             ((red[4][ci] + red[5][ci]) + (red[6][ci] + red[7][ci]));
            out[c] = gb;
            out[256 + c] = fmaf(nbp[c], gb, dot);
        }
    }
}
extern "C" int rfn_coupling_po_bwd_finish(const float* const* part, const float* const* w1, const float* const* gw1,
                                          const float* const* n1b, const float* const* w2, const float* const* gw2,
                                          const float* const* n2b, float* const* out, int n, int nblk, int K1,
                                          rfn_stream_t stream) {
    RFN_CHECK_ARG(part && w1 && gw1 && n1b && w2 && gw2 && n2b && out && n >= 1 && n <= PO_FIN_MAX && nblk >= 1 && K1 >= 1, -1);
    POFinishParams p;
    memset(&p, 0, sizeof(p));
    for (int i = 0; i < n; ++i) {
        RFN_CHECK_ARG(part[i] && w1[i] && gw1[i] && n1b[i] && w2[i] && gw2[i] && n2b[i] && out[i], -2);
        p.part[i] = (const float*)part[i];
        p.w1[i] = (const float*)w1[i]; p.gw1[i] = (const float*)gw1[i]; p.n1b[i] = (const float*)n1b[i];
        p.w2[i] = (const float*)w2[i]; p.gw2[i] = (const float*)gw2[i]; p.n2b[i] = (const float*)n2b[i];
        p.out[i] = (float*)out[i];
    }
    p.nblk = nblk; p.K1 = K1;
    hipLaunchKernelGGL(po_bwd_finish_kernel, dim3(8, 2, n), dim3(256), 0, (hipStream_t)stream, p);
    RFN_LAUNCH_CHECK();
    return 0;
}
