/* rfn_hip.h — C ABI of librfn_hip.so: hand-written gfx950 (MI355X / CDNA4) kernels for the RFN hot path.
 *
 * The reference (cdglissov/recurrent-flows-msc) has no FFI: its boundary for this path is the Python class API of
 * `Flow/glow_modules.py`, `Flow/glow.py` and `Utils/modules.py` (ConvLSTM).  Each entry point below replaces the
 * torch-op sequence of the cited reference lines; the Python host side (recurrent-flows-msc_amd/{Flow,Utils,RFN})
 * keeps the reference's class names, signatures and state_dict keys and calls these through ctypes
 * (recurrent-flows-msc_amd/rfn_hip/lib.py).  See INTEGRATION.md for the binding a maintainer would add.
 *
 * Conventions
 *   - every tensor is fp32, NCHW, device memory, 4-byte aligned; "ns" arguments are the frame (dim-0) stride in
 *     ELEMENTS, so a channel-slice view of a bigger tensor can be passed without a copy; the channel stride is H*W;
 *   - N is the number of frames in the launch (the host time-batches B*(T-1) frames), HW = H*W;
 *   - all functions enqueue on `stream` (a hipStream_t passed as void*) and never synchronise; the only allocation the
 *     library makes is ONE grow-only scratch buffer per device for the split-K convolutions (few-pixel shapes: the K
 *     slices write partial outputs there and a second kernel adds them in a fixed order -- no float atomics, results are
 *     bit-reproducible).  It grows on first use of a larger shape (hipMalloc; refused with an error while the stream is
 *     being captured into a hipGraph: run the shape eagerly once first) and serves every stream of the device, so
 *     split-K convolutions on DIFFERENT streams of one device must not overlap in time;
 *   - return value: 0 on success, otherwise a hipError_t / negative argument-check code; rfn_last_error() gives text;
 *   - global state: the (thread-local) last-error string, and five developer knobs that the kernel selectors read ONCE
 *     from the environment at their first call and then keep for the life of the process: RFN_CONV_WS (0: no
 *     weight-stationary kernels), RFN_CONV_VARIANT, RFN_WGRAD_VARIANT, RFN_WGRAD_SPLIT, RFN_WGRAD_BPX128 (tile / split-K
 *     experiments; unset = production choice).  They pick between kernels that compute the same result; nothing else is
 *     cached between calls, and the library is safe to call from several host threads on distinct streams.
 */
#ifndef RFN_HIP_H
#define RFN_HIP_H

#ifdef __cplusplus
extern "C" {
#endif

typedef void* rfn_stream_t;

int rfn_abi_version(void);
const char* rfn_last_error(void);

/* ---- a1  Squeeze2d.forward  (Flow/glow_modules.py:298-310): out[b,4c+2i+j,h,w] = in[b,c,2h+i,2w+j]; undo = inverse.
 * C,H,W are the INPUT dims.  x_ns / y_ns: frame strides of input / output. Bit-exact copy. */
int rfn_squeeze2d_f32(const float* x, long x_ns, float* y, long y_ns, int N, int C, int H, int W, int undo,
                      rfn_stream_t stream);

/* ---- a3  ActNorm.initialize  (glow_modules.py:22-31): per-channel mean and UNBIASED variance over (N,H,W).
 * mean[C], var[C] are overwritten. */
int rfn_channel_stats_f32(const float* x, long x_ns, float* mean, float* var_unbiased, int N, int C, int HW,
                          rfn_stream_t stream);

/* ---- a3+a4  ActNorm.forward + InvConv.forward fused  (glow_modules.py:38-45, 209-216):
 *   y = (x + bias[c]) * exp(logs[c]);  z[n,:,p] = Wm · y[n,:,p]      (Wm is the C×C matrix built on the host from
 *   P,L,U,log_s, glow_modules.py:188-205).  The log-det terms (Σlogs + Σlog_s)·HW are parameter-only and are added
 *   by the host. */
int rfn_actnorm_invconv_fwd_f32(const float* x, long x_ns, const float* bias, const float* logs, const float* Wm,
                                float* z, long z_ns, int N, int C, int HW, rfn_stream_t stream);
/* backward of the above: given gz, recomputes y; gx = exp(logs)·Wmᵀgz; gW += Σ gz yᵀ; gbias += Σ gy·exp(logs);
 * glogs += Σ gy·y.  gW[C*C], gbias[C], glogs[C] are ACCUMULATED into (caller zeroes them). */
int rfn_actnorm_invconv_bwd_f32(const float* x, long x_ns, const float* bias, const float* logs, const float* Wm,
                                const float* gz, long gz_ns, float* gx, long gx_ns, float* gW, float* gbias,
                                float* glogs, int N, int C, int HW, rfn_stream_t stream);
/* ---- a4  InvConv.get_weight for the K steps of a flow level  (glow_modules.py:178-207, forward direction):
 *   W[k] = P[k] (lower[k] o tril(-1) + I)(upper[k] o triu(+1) + diag(sign_s[k] * exp(log_s[k]))),  W is [K][C][C];
 *   *logdet  = H*W * sum_k sum(log_s[k])   (written; the K steps are added in order by one workgroup: no atomics).
 * The five parameter arguments are HOST arrays of K device pointers (one per step: no stacking copies); K <=
 * RFN_INVCONV_MAX_STEPS, C <= RFN_INVCONV_MAX_CHANNELS (three C x C matrices in LDS).  Backward: from gW [K][C][C] and gc (gradient of the scalar, may be NULL) to
 * g_lower, g_upper [K][C][C] (zero outside their triangles) and g_log_s [K][C]. */
#define RFN_INVCONV_MAX_STEPS 32
#define RFN_INVCONV_MAX_CHANNELS 96
int rfn_invconv_weights_fwd_f32(const float* const* p, const float* const* lower, const float* const* upper,
                                const float* const* log_s, const float* const* sign_s, float* W, float* logdet, int K,
                                int C, int HW, rfn_stream_t stream);
int rfn_invconv_weights_bwd_f32(const float* const* p, const float* const* lower, const float* const* upper,
                                const float* const* log_s, const float* const* sign_s, const float* gW, const float* gc,
                                float* g_lower, float* g_upper, float* g_log_s, int K, int C, int HW, rfn_stream_t stream);
/* reverse direction (glow_modules.py:47-52, 217-221):  x = (Winv · z) * exp(-logs) - bias. */
int rfn_invconv_actnorm_rev_f32(const float* z, long z_ns, const float* bias, const float* logs, const float* Winv,
                                float* x, long x_ns, int N, int C, int HW, rfn_stream_t stream);

/* ---- a5.1/a5.2  Conv2dNorm / Conv2dZeros / ConvLSTM conv  (glow_modules.py:106-147, Utils/modules.py:335-340,368):
 * implicit-GEMM convolution on fp32 MFMA (v_mfma_f32_32x32x2_f32), stride 1, "same" padding, ks ∈ {1,3}.
 * Input channels [0,C1) are read from in1, [C1,C1+C2) from in2 (the torch.cat of glow_modules.py:273,355 and
 * Utils/modules.py:367 is never materialised); in2 may be NULL when C2 == 0.
 * wpk: weights packed by rfn_pack_conv_weight_f32.
 * Output channels [0,cout_split) go to out1, [cout_split,Cout) to out2 (out2 may be NULL when cout_split == Cout);
 * acc1/acc2 != 0 accumulates (out += result) instead of overwriting.
 * Epilogue (ep_mode), applied per output channel c before the store:
 *   0: y = a                                   (plain; used for dgrad)
 *   1: y = act((a + p0[c]) * exp(p1[c]))       (Conv2dNorm = conv + ActNorm, then ActFun; act 0 none,1 relu,2 leaky .2)
 *   2: y = (a + p0[c]) * exp(3*p1[c])          (Conv2dZeros)
 *   3: y = a + p0[c]                           (conv with bias; ConvLSTM)
 */
int rfn_conv2d_fwd_f32(const float* in1, long in1_ns, int C1, const float* in2, long in2_ns, int C2,
                       const float* wpk, float* out1, long out1_ns, float* out2, long out2_ns, int Cout,
                       int cout_split, int acc1, int acc2, int N, int H, int W, int ks, int ep_mode,
                       const float* p0, const float* p1, int act, rfn_stream_t stream);

/* ---- split-precision variant of the same convolution ("bf16x3"): every fp32 operand x is split on the fly into two
 * bf16 numbers hi + lo and a*b is formed as a_hi*b_hi + a_hi*b_lo + a_lo*b_hi by three v_mfma_f32_32x32x16_bf16 with
 * fp32 accumulation (relative error of a product <= 2^-16; measured nll error ~1e-5 relative, budget 1e-4).  Inputs,
 * outputs, epilogues and argument meaning are identical to rfn_conv2d_fwd_f32; wpk must come from
 * rfn_pack_conv_weight_bf16x3 (which also performs the hi/lo split of the weights). */
int rfn_conv2d_fwd_bf16x3(const float* in1, long in1_ns, int C1, const float* in2, long in2_ns, int C2,
                          const float* wpk, float* out1, long out1_ns, float* out2, long out2_ns, int Cout,
                          int cout_split, int acc1, int acc2, int N, int H, int W, int ks, int ep_mode,
                          const float* p0, const float* p1, int act, rfn_stream_t stream);
long rfn_packed_weight_size_bf16x3(int Cout, int Cin, int ks); /* in floats (4-byte units) */
int rfn_pack_conv_weight_bf16x3(const float* w, float* wpk, int Cout, int Cin, int ks, int transpose_flip,
                                rfn_stream_t stream);
/* "bf16x6": three bf16 pieces per operand (24 significant bits), six MFMAs per product -- fp32-grade results at twice
 * the cost of bf16x3 and a third of the fp32-MFMA kernel's; the forward convolutions of the flow levels the fused kernel
 * does not take ('mixed' arithmetic).  Same arguments as the bf16x3 functions; generic tile kernel only. */
long rfn_packed_weight_size_bf16x6(int Cout, int Cin, int ks);
int rfn_pack_conv_weight_bf16x6(const float* w, float* wpk, int Cout, int Cin, int ks, int transpose_flip,
                                rfn_stream_t stream);
int rfn_conv2d_fwd_bf16x6(const float* in1, long in1_ns, int C1, const float* in2, long in2_ns, int C2,
                          const float* wpk, float* out1, long out1_ns, float* out2, long out2_ns, int Cout,
                          int cout_split, int acc1, int acc2, int N, int H, int W, int ks, int ep_mode,
                          const float* p0, const float* p1, int act, rfn_stream_t stream);

/* 3x3 / pad 1 convolution of an image with 1 .. 4 channels into 16 or 32 feature maps as plain fp32 FMAs (exact), one
 * thread per pixel: the first convolution of the frame extractor (Utils/modules.py:81, VGG_downscaler's 3x3 / stride 1 /
 * bias-free convolution applied to the input frames, called from RFN/RFN_new.py:127;
 * w is the torch weight [Cout][Cin][3][3], no epilogue), and its weight gradient for one input channel and 16 outputs
 * (gw [16][1][3][3], accumulated with float atomics: the caller zeroes it). */
int rfn_conv3x3_fewcin_supported(int Cin, int Cout);
int rfn_conv3x3_fewcin_fwd_f32(const float* in, long in_ns, int Cin, const float* w, float* out, long out_ns, int Cout,
                               int N, int H, int W, rfn_stream_t stream);
int rfn_conv3x3_c1_wgrad16_f32(const float* in, long in_ns, const float* g, long g_ns, float* gw, int N, int H, int W,
                               rfn_stream_t stream);

/* Data-gradient convolution fused with the backward of the PRODUCER conv's Conv2dNorm epilogue (ActNorm + ActFun,
 * glow_modules.py:139-142 + Utils/modules.py:8-19): with y = act((u+b)*exp(logs)) saved from the forward pass,
 *   g  = conv(gin, wpk)                    (wpk packed with transpose_flip = 1 / mode 1)
 *   out = g * act'(y) * exp(logs[c])       (= grad wrt u, what the producer's weight- and data-gradient consume)
 *   part[0][c] += Σ out  (= grad b),  part[1][c] += Σ g*y  (= grad logs)   over all frames and pixels
 * part is [2][Cout], accumulated with float atomics (one per workgroup / wave and channel): the caller zeroes it.
 * Cout % 64 == 0.  (rfn_conv2d_dgrad_act_rows_bf16x3: number of partial sums per channel the launch adds; informative.) */
int rfn_conv2d_dgrad_act_rows_bf16x3(int N, int H, int W, int ks, int Cout, int Cin);
int rfn_conv2d_dgrad_act_bf16x3(const float* gin, long gin_ns, int Cin, const float* wpk, const float* y, long y_ns,
                                const float* logs, int act, float* out, long out_ns, float* part, int Cout, int N,
                                int H, int W, int ks, rfn_stream_t stream);

/* Pack MANY weights in one launch (a whole flow: ~300 descriptors per training step instead of ~370 launches).
 * descs_device: device array of n rfn_pack_desc; mode 0 forward, 1 data-gradient (transposed, taps mirrored),
 * 2 tap-expanded 1x1 form of a 3x3 conv with tiny Cout (w'[tap*Cout+co][ci] = w[co][ci][tap]); mode + 4: three planes
 * (bf16x6) instead of two.  Each wpk must hold rfn_packed_weight_size_bf16x3 (or _bf16x6) floats of the LOGICAL conv
 * (mode 2: Cout' = 9*Cout, ks' = 1). */
typedef struct {
    const float* w;
    float* wpk;
    int Cout, Cin, ks, mode;
} rfn_pack_desc;
int rfn_pack_conv_weights_batched_bf16x3(const void* descs_device, int n, rfn_stream_t stream);
/* the same for descriptors in HOST memory (packs queued by the host between two launches): 64 per launch */
int rfn_pack_conv_weights_hostdescs_bf16x3(const void* descs_host, int n, rfn_stream_t stream);

/* ---- a5 fused  AffineCoupling.net forward for the shallow levels (Flow/glow_modules.py:232-238 with :119-121 and
 * :139-142), csrc/coupling_po.hip: conv3x3 -> ActNorm -> act -> conv1x1 -> ActNorm -> act -> tap-expanded conv3x3 in
 * ONE kernel; the 256-channel hidden activations pass from layer to layer in registers (they are still written once:
 * the backward pass needs them).  Arithmetic "f16x3s": every operand is scaled by a power of two (per tensor for
 * weights, per block for the staged input, per pixel for the hidden activations) and split into two fp16 pieces
 * (22 significant bits); a product is three v_mfma_f32_32x32x16_f16 with fp32 accumulation, the scale undone in fp32.
 *   rfn_coupling_po_supported  1 when (N, C, Cc, Hd, H, W) is a shape the kernel takes (else use the unfused kernels):
 *       Hd = 256, square power-of-two maps, N*H*W a multiple of 128 (a workgroup round is 128 consecutive pixels of the
 *       (frame, pixel) sequence: rows of one frame, or two whole 8x8 frames) and one of the instantiated channel-group
 *       counts (NG = ceil((C/2 + Cc) / 8), NP = ceil(9C / 32), W): (3, 2, 32) and (5, 3, 16) = levels 0 / 1 of the
 *       canonical flow, (9, 5, 8) = its level 2, (9, 4, 32) = a C = 12 flow with 59..66 condition channels;
 *   rfn_coupling_po_packed_bytes / rfn_coupling_po_pack  the fragment-ordered weight stream of one coupling net
 *       (descs_device: device array of n rfn_po_pack_desc; one launch packs every net of a flow);
 *   rfn_coupling_po_fwd  z: output of ActNorm+InvConv [N, >=C/2, H, W] (channels [0, C/2) are read), cond [N, Cc, H, W];
 *       n1b/n1l, n2b/n2l: ActNorm bias / logs [256] of the hidden layers; act 0 none, 1 relu, 2 leaky .2.
 *       Outputs h1, h2 [N, 256, H, W] and P [N, 9C, H, W], P[tap*C + co] = sum_c w3[co][c][tap] h2[c]
 *       (rfn_tap_gather_f32 turns P into the Conv2dZeros output); m1 / m2 (nullable, rfn_coupling_po_mask_floats(N, H, W)
 *       floats each, 16-byte aligned): 1-bit masks "h <= 0" of h1 / h2 in the backward kernel's (round, thread,
 *       register) order -- act'(.) for rfn_coupling_po_bwd, 1/32 of the activations' bytes.
 * Backward of the same three convolutions' DATA path (backward of glow_modules.py:232-238), one kernel:
 *   rfn_coupling_po_bwd_supported / rfn_coupling_po_bwd_packed_bytes / rfn_coupling_po_pack_bwd  as above for the
 *       backward stream (w3 transposed + mirrored, w2 transposed; descs: w2, w3, dst, C are read); gradient images of
 *       C <= 8 channels on 32x32 / 16x16 maps and of 9..16 channels on 8x8 / 32x32 maps;
 *   rfn_coupling_po_bwd  go [N, C, H, W] = gradient at conv3's output ->
 *       ga2 = (conv3^T go) act'(h2) exp(n2l), ga1 = (w2^T ga2) act'(h1) exp(n1l)  [N, 256, H, W] each (the gradients at
 *       the outputs of conv2 / conv1: operands of the weight gradients and of conv1's data gradient), and
 *       part (rfn_coupling_po_bwd_part_floats(N, H, W) floats, written): per-workgroup sums over pixels of ga2 | ga1;
 *   rfn_coupling_po_bwd_finish  for n <= 16 nets (host arrays of n device pointers): ActNorm gradients
 *       out[i] = [gn1b | gn1l | gn2b | gn2l] (4 x 256) with gnb = sum of the part rows (nblk of them) and
 *       gnl[c] = sum_k w[c][k] gw[c][k] + nb[c] gnb[c]  (K1 = Cin * 9 elements per row of w1 / gw1, 256 of w2 / gw2). */
typedef struct {
    const float* w1;  /* [256][C/2 + Cc][3][3] */
    const float* w2;  /* [256][256][1][1] */
    const float* w3;  /* [C][256][3][3] */
    float* dst;       /* rfn_coupling_po_packed_bytes(Cin, C) bytes, 16-byte aligned */
    int Cin, C;
} rfn_po_pack_desc;
int rfn_coupling_po_supported(int N, int C, int Cc, int Hd, int H, int W);
long rfn_coupling_po_packed_bytes(int Cin, int C);
int rfn_coupling_po_pack(const void* descs_device, int n, rfn_stream_t stream);
int rfn_coupling_po_fwd(const float* z, long z_ns, const float* cond, long cond_ns, const void* wpk, const float* n1b,
                        const float* n1l, const float* n2b, const float* n2l, float* h1, long h1_ns, float* h2,
                        long h2_ns, float* P, long P_ns, float* m1, float* m2, int N, int C, int Cc, int H, int W, int act,
                        rfn_stream_t stream);
long rfn_coupling_po_mask_floats(int N, int H, int W);
int rfn_coupling_po_bwd_supported(int N, int C, int H, int W);
long rfn_coupling_po_bwd_packed_bytes(int C);
int rfn_coupling_po_pack_bwd(const void* descs_device, int n, rfn_stream_t stream);
long rfn_coupling_po_bwd_part_floats(int N, int H, int W);
int rfn_coupling_po_bwd(const float* go, long go_ns, const void* wpk, const float* n1l, const float* n2l,
                        const float* m_h1, const float* m_h2, float* ga2, long ga2_ns, float* ga1, long ga1_ns,
                        float* part, int N, int C, int H, int W, int act, rfn_stream_t stream);
int rfn_coupling_po_bwd_finish(const float* const* part, const float* const* w1, const float* const* gw1,
                               const float* const* n1b, const float* const* w2, const float* const* gw2,
                               const float* const* n2b, float* const* out, int n, int nblk, int K1,
                               rfn_stream_t stream);

/* ---- a5/a6 fused shell tail of GlowStep.forward (Flow/glow_modules.py:119-121 + :276-285): with P (tap-expanded
 * Conv2dZeros output, [N,9C,H,W]) the 3x3 shift-and-add, bias and exp(3 logs) scale are applied here and the result is
 * also written to o_out [N,C,H,W]; without P, o_in holds that result.  z [N,C,H,W] (frame stride z_ns): channels
 * [C/2, C) <- (z2 + o[0::2]) * exp(clamp(o[1::2])); logdet[n] = sum of the clamped log-scales is WRITTEN. */
int rfn_gather_affine_f32(const float* P, const float* o_in, long o_ns, const float* b3, const float* l3, float* o_out,
                          float* z, long z_ns, const float* scale, const float* scale_shift, float* logdet,
                          int clamp_type, int N, int C, int H, int W, rfn_stream_t stream);
/* backward of the affine coupling and of the Conv2dZeros epilogue in one launch: from gout (grad of the step's output)
 * and glogdet [N] (may be NULL) to gz (whole tensor: first half copied from gout, second half the coupling gradient) and
 * gpre = grad at the convolution output (what the weight / data gradient of conv3 consume).  gscale, gscale_shift [C/2]
 * (realnvp clamp only), gb3, gl3 [C] are ACCUMULATED into (caller zeroes them). */
int rfn_affine_zeros_bwd_f32(const float* zout, long zout_ns, const float* o, long o_ns, const float* gout, long gout_ns,
                             const float* glogdet, const float* scale, const float* scale_shift, const float* l3,
                             float* gz, long gz_ns, float* gpre, long gpre_ns, float* gscale, float* gscale_shift,
                             float* gb3, float* gl3, int clamp_type, int N, int C, int HW, rfn_stream_t stream);

/* ---- a6 shell BETWEEN two consecutive Glow steps of a level, one launch each way (Flow/glow.py:31-36 unrolled over
 * the K steps of a level).  Forward: the coupling tail of step k (as rfn_gather_affine_f32) then, when Wm != NULL, the
 * ActNorm + InvConv head of step k+1: znext = Wm ((z' + bias) * exp(logs)).  With P == o_in == NULL only the head runs
 * (first step of a level; z is then read only).
 * Log-det WITHOUT float atomics (a forward pass is bit-reproducible): `logdet` is this launch's buffer of per-workgroup
 * partial sums (rfn_glow_shell_fwd_ld_floats(N, C, H, W) floats, WRITTEN); rfn_logdet_reduce_f32 adds the partials of
 * n_launch consecutive such buffers per frame in a fixed order into logdet [N] (accumulate = 1: added to it). */
int rfn_glow_shell_fwd_f32(float* z, long z_ns, const float* P, const float* o_in, long o_ns, const float* b3,
                           const float* l3, float* o_out, const float* scale, const float* scale_shift, float* logdet,
                           int clamp_type, const float* bias, const float* logs, const float* Wm, float* znext,
                           long znext_ns, int ld_const, int N, int C, int H, int W, rfn_stream_t stream);
long rfn_glow_shell_fwd_ld_floats(int N, int C, int H, int W);
int rfn_logdet_reduce_f32(const float* part, int n_launch, float* logdet, int accumulate, int N, int C, int H, int W,
                          rfn_stream_t stream);
/* (ld_const = 1: the head also adds its ActNorm's parameter-only log-det term H*W * sum_c logs[c] to logdet[n],
 * glow_modules.py:47-52; the backward kernels below then add H*W * sum_n glogdet[n] to glogs.) */
/* Backward: rfn_actnorm_invconv_bwd_f32 of step k+1 (x = its input = step k's output, gz = gradient wrt its
 * post-InvConv tensor; gW, gbias, glogs accumulated) whose result feeds rfn_affine_zeros_bwd_f32 of step k from
 * registers (o .. gl3 are step k's, same meaning as there). */
int rfn_glow_shell_bwd_f32(const float* x, long x_ns, const float* bias, const float* logs, const float* Wm,
                           const float* gz, long gz_ns, float* gW, float* gbias, float* glogs, const float* o, long o_ns,
                           const float* glogdet, const float* scale, const float* scale_shift, const float* l3,
                           float* gz_prev, long gz_prev_ns, float* gpre, long gpre_ns, float* gscale,
                           float* gscale_shift, float* gb3, float* gl3, int clamp_type, int ld_const, int N, int C,
                           int HW, rfn_stream_t stream);
int rfn_actnorm_invconv_bwd_ld_f32(const float* x, long x_ns, const float* bias, const float* logs, const float* Wm,
                                   const float* gz, long gz_ns, float* gx, long gx_ns, float* gW, float* gbias,
                                   float* glogs, const float* glogdet, int N, int C, int HW, rfn_stream_t stream);

/* ---- 3x3 convolution (stride 1, pad 1) with at most 64 output channels and Cin % 16 == 0 input channels on 32x32 or
 * 16x16 maps, bf16x3 arithmetic, one input tensor: the data gradient of the first coupling-net convolution at the two
 * finest flow levels (256 -> C/2 + Cc channels; backward of Flow/glow_modules.py:232-238).  wpk: the
 * rfn_pack_conv_weight_bf16x3 buffer of the logical weight (transpose_flip = 1 of the forward weight for a data
 * gradient).  Output channels [0, cout_split) go to out1, the rest to out2; acc1 / acc2: add into what is there. */
int rfn_dgrad_small_supported(int N, int Cin, int Cout, int H, int W);
int rfn_conv3x3_smallcout_bf16x3(const float* in, long in_ns, int Cin, const float* wpk, float* out1, long out1_ns,
                                 float* out2, long out2_ns, int Cout, int cout_split, int acc1, int acc2, int N, int H,
                                 int W, rfn_stream_t stream);

/* Split-precision weight-gradient GEMM: gw[M][Nc] += Σ_{f,p} a[f][m][p] * b[f][n][p]  (a: [F,M,HW] frame stride a_ns,
 * b: [F,Nc,HW] frame stride b_ns; HW % 4 == 0, 16-byte aligned bases).  gw is accumulated with float atomics (caller
 * zeroes it).  1x1 weight gradients use it directly (a = output grad, b = conv input); 3x3 ones first expand the
 * smaller operand: b = rfn_im2col3x3_f32(input) [9*Cin rows, tap-major] or a = rfn_tap_scatter_f32(grad) [9*Cout rows].
 * Kernel choice (same arithmetic, same result up to summation order): F*HW >= 100000 pixels, HW % 32 == 0 and
 * Nc % 256 == 0 with M >= 192 or M <= 64 (grouped form: G * F*HW >= 100000) -> the LDS-DMA ring kernel (raw fp32 rows HBM -> LDS by global_load_lds, split
 * at the fragment reads; RFN_WGRAD_DMA=0 disables it); otherwise the register-staged tilings (RFN_WGRAD_VARIANT,
 * RFN_WGRAD_SPLIT: tile / K-split experiments). */
int rfn_gemm_wgrad_bf16x3(const float* a, long a_ns, int M, const float* b, long b_ns, int Nc, float* gw, int F, int HW,
                          rfn_stream_t stream);
/* G (<= 16) gradients of ONE shape in one launch (the K steps of a flow level, where a single gradient is a
 * latency-class problem): a, b, gw are host arrays of G device pointers; strides and sizes are shared. */
int rfn_gemm_wgrad_grouped_bf16x3(const float* const* a, long a_ns, int M, const float* const* b, long b_ns, int Nc,
                                  float* const* gw, int G, int F, int HW, rfn_stream_t stream);
/* out[n][tap*Cin+ci][y][x] = in[n][ci][y+dy-1][x+dx-1] (0 outside); two-source input; out dense [N,9*Cin,H,W]. */
/* 3x3 (pad 1) weight gradient without the im2col buffer (W % 8 == 0): gw[Cout][9*(C1+C2)] += sum over frames and
 * pixels of g[co][px] * in[ci][px + tap], column index ci*9 + tap: gw is the torch weight layout [Cout][Cin][3][3].
 * gw must be zeroed by the caller (split over pixel stages, atomic combine). */
int rfn_conv3x3_wgrad_implicit_bf16x3(const float* g, long g_ns, int Cout, const float* in1, long in1_ns, int C1,
                                      const float* in2, long in2_ns, int C2, float* gw, int F, int H, int W,
                                      rfn_stream_t stream);
int rfn_conv3x3_wgrad_implicit_grouped_bf16x3(const float* const* g, long g_ns, int Cout, const float* const* in1,
                                              long in1_ns, int C1, const float* const* in2, long in2_ns, int C2,
                                              float* const* gw, int G, int F, int H, int W, rfn_stream_t stream);
int rfn_im2col3x3_f32(const float* in1, long in1_ns, int C1, const float* in2, long in2_ns, int C2, float* out, int N,
                      int H, int W, rfn_stream_t stream);

/* number of floats of a packed weight buffer for (Cout, Cin, ks) */
long rfn_packed_weight_size(int Cout, int Cin, int ks);
/* Pack torch-layout weights w[Cout][Cin][ks][ks] for rfn_conv2d_fwd_f32.
 * transpose_flip = 0: forward conv.  transpose_flip = 1: the data-gradient conv (roles of Cin/Cout swapped, taps
 * mirrored), i.e. the packed buffer then describes a conv with Cin'=Cout inputs and Cout'=Cin outputs. */
int rfn_pack_conv_weight_f32(const float* w, float* wpk, int Cout, int Cin, int ks, int transpose_flip,
                             rfn_stream_t stream);

/* weight gradient, tap-major: gwt[ks*ks][Cout][Cin] += Σ_{n,y,x} g[n,co,y,x] · in[n,ci,y+dy,x+dx]  (two-source
 * input as above).  gwt is ACCUMULATED into with float atomics (caller zeroes it); tap-major so that one MFMA
 * accumulator register is a contiguous 128-byte atomic segment. */
int rfn_conv2d_wgrad_f32(const float* in1, long in1_ns, int C1, const float* in2, long in2_ns, int C2,
                         const float* g, long g_ns, int Cout, float* gwt, int N, int H, int W, int ks,
                         rfn_stream_t stream);
/* gw[Cout][Cin][ks][ks] (torch layout) = (accumulate ? gw : 0) + transpose of gwt[ks*ks][Cout][Cin]. */
int rfn_wgrad_finish_f32(const float* gwt, float* gw, int Cout, int Cin, int ks, int accumulate, rfn_stream_t stream);

/* Tap-expanded form of a 3x3 convolution with very few output channels (Conv2dZeros at the shallow flow levels,
 * glow_modules.py:106-121 with Cout = 4 / 8): the conv runs as a 1x1 rfn_conv2d_fwd_f32 to 9*C channels
 * P[n][tap*C+co] (dense [N,9C,H,W]); rfn_tap_gather_f32 then forms
 *   o[n][co][y][x] = (Σ_tap P[n][tap*C+co][y+dy-1][x+dx-1] + bias[co]) * exp(3*logs[co])   (bias/logs both NULL: plain sum).
 * rfn_tap_scatter_f32 is the adjoint data movement used for the weight gradient:
 *   Gs[n][tap*C+co][y][x] = g[n][co][y-dy+1][x-dx+1] (0 outside); a 1x1 rfn_conv2d_wgrad_f32 of Gs gives gW[co][ci][tap].
 * All tensors dense NCHW. */
int rfn_tap_gather_f32(const float* P, const float* bias, const float* logs, float* o, int N, int C, int H, int W,
                       rfn_stream_t stream);
int rfn_tap_scatter_f32(const float* g, float* Gs, int N, int C, int H, int W, rfn_stream_t stream);

/* backward through  y = act((u + b[c]) * exp(l[c]))  (ep_mode 1) or  y = (u + b[c]) * exp(3 l[c])  (ep_mode 2),
 * given y (saved forward output) and gy:   gu (may alias gy) ;  gb[c] += ...;  gl[c] += ...  (accumulated). */
int rfn_conv_epilogue_bwd_f32(const float* y, long y_ns, const float* gy, long gy_ns, float* gu, long gu_ns,
                              const float* logs, float* gb, float* gl, int N, int C, int HW, int ep_mode, int act,
                              rfn_stream_t stream);

/* ---- a5  AffineCoupling.forward, everything after the coupling net  (glow_modules.py:276-291):
 * o = NN output [N,C,HW] with shift = o[:,0::2], s = o[:,1::2];  ls = clamp(s) (clamp_type 0 realnvp:
 * scale[c]*tanh(s)+scale_shift[c]; 1 glow: log sigmoid(s+2); 2 softclamp: 2.5*0.636*atan(s/2.5); 3 none);
 * forward (reverse=0): z2 <- (z2 + shift) * exp(ls), logdet[n] += Σ ls;  reverse=1: z2 <- z2*exp(-ls) - shift,
 * logdet[n] -= Σ ls.   z2 = channels [C/2, C) of z, updated IN PLACE.  logdet may be NULL. */
int rfn_affine_coupling_f32(float* z, long z_ns, const float* o, long o_ns, const float* scale,
                            const float* scale_shift, float* logdet, int clamp_type, int reverse, int N, int C, int HW,
                            rfn_stream_t stream);
/* backward of the forward direction.  zout = output of the forward (z1 | z2'), gout = grad wrt it, glogdet[N] = grad
 * wrt logdet.  Writes gz2 into channels [C/2,C) of gz (channels [0,C/2) of gz are NOT touched), go [N,C,HW] (grad wrt
 * the NN output) and accumulates gscale[C/2], gscale_shift[C/2]. */
int rfn_affine_coupling_bwd_f32(const float* zout, long zout_ns, const float* o, long o_ns, const float* gout,
                                long gout_ns, const float* glogdet, const float* scale, const float* scale_shift,
                                float* gz, long gz_ns, float* go, long go_ns, float* gscale, float* gscale_shift,
                                int clamp_type, int N, int C, int HW, rfn_stream_t stream);

/* ---- a7/a8  Gaussian log-likelihood reductions  (glow_modules.py:358-365, glow.py:135-140):
 * logp[n] += Σ_{c,p} log N(z[n,c,p]; mean, std).
 *   layout 0 ("cross", Split2d): mean = o[:,2c], raw = o[:,2c+1];  layout 1 ("split", base prior): mean = o[:,c],
 *   raw = o[:,Cz+c].   std_mode 0: softplus(raw)+1e-8 ; 1: exp(raw).   o has 2*Cz channels, z has Cz. */
int rfn_gauss_logp_f32(const float* z, long z_ns, const float* o, long o_ns, float* logp, int layout, int std_mode,
                       int N, int Cz, int HW, rfn_stream_t stream);
/* backward: gz (written) and go (written) from glogp[N]. */
int rfn_gauss_logp_bwd_f32(const float* z, long z_ns, const float* o, long o_ns, const float* glogp, float* gz,
                           long gz_ns, float* go, long go_ns, int layout, int std_mode, int N, int Cz, int HW,
                           rfn_stream_t stream);
/* reverse / sampling (glow_modules.py:366-369, glow.py:153-154): z = mean + std*temperature*eps. */
int rfn_gauss_sample_f32(const float* o, long o_ns, const float* eps, float* z, long z_ns, float temperature, int layout,
                         int std_mode, int N, int Cz, int HW, rfn_stream_t stream);

/* ---- a10  SRNN latent step of RFN.loss (RFN/RFN_new.py:167-184,206-207 with SimpleParamNet's chunk + softplus,
 * Utils/modules.py:240-244): enc, pri = [B, 2*Z*HW] outputs of the encoder / prior parameter convs (loc half | raw scale
 * half); ps = softplus(pri_raw), es = softplus(enc_raw), pm = pri_loc, em = enc_loc (+ pm when res_q);
 *   zt = pm + ps*eps_p,  zxt = em + es*eps_q,  kl = KL(N(em,es)||N(pm,ps)) element-wise,  em/es also returned.
 * ZHW = Z*H*W.  The backward takes the gradients of the five outputs (any may be NULL) and writes g_enc, g_pri; g_zt and
 * g_zxt are read with a row stride in floats (>= ZHW), so a channel slice of a wider gradient needs no copy. */
int rfn_latent_step_fwd_f32(const float* enc, const float* pri, const float* eps_p, const float* eps_q, float* zt,
                            float* zxt, float* kl, float* em, float* es, int B, int ZHW, int res_q, rfn_stream_t stream);
int rfn_latent_step_bwd_f32(const float* enc, const float* pri, const float* eps_p, const float* eps_q,
                            const float* g_zt, long g_zt_ns, const float* g_zxt, long g_zxt_ns, const float* g_kl,
                            const float* g_em, const float* g_es, float* g_enc, float* g_pri, int B, int ZHW, int res_q,
                            rfn_stream_t stream);

/* ---- a10  the per-timestep parameter nets (SimpleParamNet, Utils/modules.py:216-244, called once per frame by
 * RFN.loss, RFN/RFN_new.py:167-179) on small maps: a 3x3 / pad 1 convolution on an H x W <= 16 pixel map is the dense
 * product out[b][(co,po)] = bias[co] + sum x[b][(ci,pi)] * w[co][ci][tap(po,pi)] over NCHW-contiguous samples.
 * rfn_smallmap_pack_bf16x3 writes w [Cout][Cin][3][3] in MFMA fragment order, split into bf16 (hi, lo), for the forward
 * product (transpose 0: K = Cin*HW, N = Cout*HW) or the data gradient (transpose 1: K = Cout*HW, N = Cin*HW);
 * rfn_smallmap_packed_size gives the buffer size in bytes.  rfn_smallmap_dense_bf16x3: out[B][N] = a'[B][K] * packed
 * (+ bias[n / HW] + add[B][N], each optional, then leaky_relu(slope_out) when act_out), where a' = a, or a * (y > 0 ? 1 : slope_in) when y is given
 * (backward of an in-place leaky_relu whose result is y); a_out (optional) receives a'.  K % 8 == 0. */
long rfn_smallmap_packed_size(int Cout, int Cin, int H, int W, int transpose);
int rfn_smallmap_pack_bf16x3(const float* w, int Cout, int Cin, int H, int W, int transpose, float* packed,
                             rfn_stream_t stream);
/* The same for n matrices in ceil(n / 64) launches: host array of descriptors (the packs of one training step -- latent
 * nets, ConvLSTM, the 2x2 flow level -- are queued by the host and flushed before their first consumer). */
typedef struct {
    const float* w;   /* [Cout][Cin][3][3] */
    float* packed;    /* rfn_smallmap_packed_size(Cout, Cin, H, W, transpose) bytes, 16-byte aligned */
    int Cout, Cin, H, W, transpose, pad_;
} rfn_smallmap_pack_desc;
int rfn_smallmap_pack_batched_bf16x3(const void* descs_host, int n, rfn_stream_t stream);
int rfn_smallmap_dense_bf16x3(const float* a, const float* y, float slope_in, const float* packed, const float* bias,
                              const float* add, int act_out, float slope_out, float* out, float* a_out, int B, int K, int N,
                              int HW, rfn_stream_t stream);

/* Two rfn_smallmap_dense_bf16x3 products (suffix 0 / 1) in one launch: the encoder and the prior layer of a
 * timestep (RFN_new.py:167-179) have no data dependence on each other.  Same B and HW; K, N per product. */
int rfn_smallmap_dense_pair_bf16x3(const float* a0, const float* y0, float slope_in0, const float* packed0,
                                   const float* bias0, const float* add0, int act_out0, float slope_out0, float* out0,
                                   float* a_out0, int K0, int N0, const float* a1, const float* y1, float slope_in1,
                                   const float* packed1, const float* bias1, const float* add1, int act_out1,
                                   float slope_out1, float* out1, float* a_out1, int K1, int N1, int B, int HW,
                                   rfn_stream_t stream);

/* The same dense product behind the interface of rfn_conv2d_fwd_bf16x3 (ks = 3 implied): N frames of an H x W <= 16
 * map, two-source input, ep_mode 0-3, output channels split at cout_split; acc1 is a bit mask (1: add into out1, 2: add
 * into out2).  `packed` from
 * rfn_smallmap_pack_bf16x3 of the [Cout][C1+C2][3][3] weight (transpose 0), or transpose 1 of the FORWARD weight for a
 * data gradient.  (C1*H*W) % 8 == 0 and ((C1+C2)*H*W) % 8 == 0.  Used for the coupling convolutions of the two deepest
 * flow levels, where a launch is a few thousand pixels against megabytes of weights. */
int rfn_smallmap_conv_bf16x3(const float* in1, long in1_ns, int C1, const float* in2, long in2_ns, int C2,
                             const float* packed, float* out1, long out1_ns, float* out2, long out2_ns, int Cout,
                             int cout_split, int acc1, int N, int H, int W, int ep_mode, const float* p0,
                             const float* p1, int act, rfn_stream_t stream);

/* ---- callers of the path (VGG extractor / upscaler, Utils/modules.py:43-213): BatchNorm2d (training mode) + the
 * activation that follows it on a step-major time-batched tensor x [S*B, C, HW] with the statistics of EACH step's B
 * samples, as the reference's per-timestep calls compute them (RFN_new.py:126-128,191-194).  mean / var (biased) are
 * [S*C]; act: 0 none, 1 relu, 2 leaky_relu(slope), 3 tanh; gamma / beta may both be NULL.  Backward: g' = g*act'(u) with u = xhat*gamma + beta recomputed from x (the output is not read),
 * sg = sum g', sgx = sum g'*xhat per (step, channel), gx = gamma*rstd*(g' - sg/n - xhat*sgx/n). */
/* One BatchNorm layer of the time-batched extractor / upscaler in two launches each way.  Forward: partial sums + apply;
 * the apply kernel derives the moments from the partial sums, writes mean / var [S*C] (kept for the backward) and applies
 * the S running-statistics updates of the reference's step-wise calls in closed form: run <- decay*run + sum_s
 * coef[s]*stat[s] (run_mean / run_var [C] both or neither, coef for the mean and coef_u for the unbiased variance are [S]
 * device arrays, decay = (1-momentum)^S); num_batches_tracked (int64, optional) += S.
 * Backward: per-step partial sums + apply; ggamma / gbeta [C] (both or neither) are written.  acc / sums = scratch of
 * rfn_stepbn_scratch_floats(S, B, C) floats (no zeroing needed: no atomics, results are deterministic). */
long rfn_stepbn_scratch_floats(int S, int B, int C);
int rfn_stepbn_fwd_f32(const float* x, const float* gamma, const float* beta, float* y, float* mean, float* var, float* acc,
                       float* run_mean, float* run_var, const float* coef, const float* coef_u, float decay,
                       long long* num_batches_tracked, int S, int B, int C, int HW, float eps, int act, float slope,
                       rfn_stream_t stream);
int rfn_stepbn_bwd_f32(const float* x, const float* gamma, const float* beta, const float* g, const float* mean,
                       const float* var, float* sums, float* gx, float* ggamma, float* gbeta, int S, int B, int C, int HW,
                       float eps, int act, float slope, int stage, int world, rfn_stream_t stream);
/* Synchronised BatchNorm over `world` data-parallel ranks (same mathematics as one process on the
 * global batch): forward = rfn_stepbn_fwd_f32 for the local moments, the caller combines the ranks' moments, then
 * rfn_stepbn_apply_f32 normalises with the GIVEN statistics; backward = rfn_stepbn_bwd_f32 with stage 1 (partial sums
 * only), the caller adds `sums` over the ranks, then stage 2 (apply; gx uses the global means of g' and g'*xhat, the
 * parameter gradients are the sums divided by `world`: the reducer's rank average then gives the global-batch gradient).
 * stage 0 / world 1 = the single-process behaviour. */
int rfn_stepbn_apply_f32(const float* x, const float* gamma, const float* beta, float* y, const float* mean,
                         const float* var, int S, int B, int C, int HW, float eps, int act, float slope,
                         rfn_stream_t stream);

/* ---- a9  ConvLSTMLayer.forward gate update  (Utils/modules.py:370-377): cc = conv output [N,4*Hc,HW] in gate order
 * i,f,o,g;  i=σ(cc_i+Wci∘c) f=σ(cc_f+Wcf∘c) g=tanh(cc_g) c'=f∘c+i∘g o=σ(cc_o+Wco∘c') h'=o∘tanh(c').
 * Wci/Wcf/Wco [Hc*HW] may be NULL (== 0, which is what the reference trains with).  gates [N,4*Hc,HW] receives the
 * post-nonlinearity i,f,o,g for the backward pass (may be NULL). */
int rfn_convlstm_gates_fwd_f32(const float* cc, const float* c_prev, long c_ns, const float* Wci, const float* Wcf,
                               const float* Wco, float* h_out, long h_ns, float* c_out, long co_ns, float* gates,
                               int N, int Hc, int HW, rfn_stream_t stream);
/* backward: from gh, gc_next (either may be NULL), saved gates, c_prev, c_out -> gcc [N,4Hc,HW], gc_prev.
 * The peephole terms enter the state gradients; gradients w.r.t. Wci/Wcf/Wco themselves are not produced (the
 * reference never trains them on a GPU: Utils/modules.py:385-393 creates them as non-leaf tensors). */
int rfn_convlstm_gates_bwd_f32(const float* gates, const float* c_prev, long c_ns, const float* c_out, long co_ns,
                               const float* gh, long gh_ns, const float* gc_next, long gcn_ns, const float* Wci,
                               const float* Wcf, const float* Wco, float* gcc, float* gc_prev, long gcp_ns, int N,
                               int Hc, int HW, rfn_stream_t stream);

/* ---- the optimizer of the training step  (RFN/trainer.py:96: torch.optim.Adam with its defaults; the arithmetic is
 * torch's: m <- m + (1-b1)(g - m), v <- b2 v + (1-b2) g^2, p <- p - lr/(1-b1^s) * m / (sqrt(v)/sqrt(1-b2^s) + eps), with
 * g += weight_decay*p first when weight_decay != 0) for ALL parameter tensors in one launch.  `table` and `chunks` are
 * DEVICE arrays built by the host: one rfn_adam_entry per tensor (s = t - step_offset is that tensor's step count, >= 1)
 * and one (tensor index, chunk index) int pair per workgroup, a chunk being rfn_adam_chunk_elems() consecutive elements.
 * p, m, v are updated in place; g is read only.  The hyper-parameters arrive as doubles (Python floats): 1 - beta and the
 * bias corrections are formed in double and rounded once, the per-element arithmetic is fp32. */
typedef struct rfn_adam_entry {
    float* p;
    const float* g;
    float* m;
    float* v;
    long n;
    int step_offset;
    int reserved;
} rfn_adam_entry;
int rfn_adam_chunk_elems(void);
int rfn_adam_step_f32(const rfn_adam_entry* table, const int* chunks, int n_chunks, double lr, double beta1, double beta2,
                      double eps, double weight_decay, int t, rfn_stream_t stream);

#ifdef __cplusplus
}
#endif
#endif /* RFN_HIP_H */
