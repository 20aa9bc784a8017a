// HBM-bound "shell" kernels of the Glow step and the ConvLSTM gate update (gfx950).
// Everything here is a streaming pass: coalesced NCHW loads along the pixel dimension (consecutive lanes =
// consecutive pixels), per-frame / per-channel reductions done with wave shuffles + one LDS hop, never a GEMM.
#include "common.h"
#include "../../include/rfn_hip.h"
#include <stdarg.h>

static thread_local char g_err[512] = "";
void rfn_set_error(const char* fmt, ...) {
    va_list ap;
    va_start(ap, fmt);
    vsnprintf(g_err, sizeof(g_err), fmt, ap);
    va_end(ap);
}
extern "C" const char* rfn_last_error(void) { return g_err; }
extern "C" int rfn_abi_version(void) { return 1; }

#include <map>
#include <mutex>
#include <vector>
float* rfn_workspace(hipStream_t s, size_t floats) {
    // ONE buffer per device, whatever the stream: a hipGraph is captured on a fresh stream of its own, which must find
    // the buffer its eager warm-up runs (on other streams) have grown.  Consequence, stated in include/rfn_hip.h: split-K
    // convolutions issued on DIFFERENT streams of one device must not overlap in time.
    static std::mutex mu;
    static std::map<int, std::pair<float*, size_t>> tab;
    std::lock_guard<std::mutex> lock(mu);
    int dev = 0;
    (void)hipGetDevice(&dev);
    auto& e = tab[dev];
    if (e.second >= floats) return e.first;
    hipStreamCaptureStatus st = hipStreamCaptureStatusNone;
    if (hipStreamIsCapturing(s, &st) == hipSuccess && st != hipStreamCaptureStatusNone) {
        rfn_set_error("split-K workspace must grow to %zu floats during a hipGraph capture: run this shape eagerly first", floats);
        return nullptr;
    }
    // An outgrown buffer is NOT freed: a captured hipGraph has its address baked into kernel arguments and may be replayed
    // after an eager call of a larger shape made the buffer grow (generation between training steps).  Growth is geometric,
    // so the retired buffers add up to less than the live one; all of them go with the process.
    static std::vector<float*> retired;
    if (e.first) retired.push_back(e.first);
    e.first = nullptr;
    e.second = 0;
    const size_t want = 2 * floats + 1024;
    float* ptr = nullptr;
    if (hipMalloc(&ptr, want * sizeof(float)) != hipSuccess) {
        rfn_set_error("split-K workspace: hipMalloc of %zu bytes failed", want * sizeof(float));
        return nullptr;
    }
    e.first = ptr;
    e.second = want;
    return ptr;
}

// ------------------------------------------------------------------------------------------------ squeeze2d
// forward: each thread reads one float2 (input row 2h+i, cols 2w,2w+1) and writes the two output planes j=0,1.
__global__ void squeeze2d_fwd_kernel(const float* __restrict__ x, long x_ns, float* __restrict__ y, long y_ns, int N,
                                     int C, int H, int W) {
    const int W2 = W >> 1, H2 = H >> 1;
    const long total = (long)N * C * H * W2;
    for (long idx = (long)blockIdx.x * blockDim.x + threadIdx.x; idx < total; idx += (long)gridDim.x * blockDim.x) {
        int w = (int)(idx % W2);
        long r = idx / W2;
        int hin = (int)(r % H);
        r /= H;
        int c = (int)(r % C);
        int n = (int)(r / C);
        const float* src = x + n * x_ns + ((long)c * H + hin) * W + 2 * w;
        float v0 = src[0], v1 = src[1];
        int i = hin & 1, h = hin >> 1;
        float* dst = y + n * y_ns + ((long)(4 * c + 2 * i) * H2 + h) * W2 + w;
        dst[0] = v0;
        dst[(long)H2 * W2] = v1;
    }
}
// undo: C,H,W are the INPUT dims (C multiple of 4); output is [C/4, 2H, 2W].
__global__ void squeeze2d_undo_kernel(const float* __restrict__ x, long x_ns, float* __restrict__ y, long y_ns, int N,
                                      int C, int H, int W) {
    const int Co = C >> 2, Ho = H * 2, Wo = W * 2;
    const long total = (long)N * Co * Ho * W;
    for (long idx = (long)blockIdx.x * blockDim.x + threadIdx.x; idx < total; idx += (long)gridDim.x * blockDim.x) {
        int w = (int)(idx % W);
        long r = idx / W;
        int hout = (int)(r % Ho);
        r /= Ho;
        int c = (int)(r % Co);
        int n = (int)(r / Co);
        int i = hout & 1, h = hout >> 1;
        const float* src = x + n * x_ns + ((long)(4 * c + 2 * i) * H + h) * W + w;
        float v0 = src[0], v1 = src[(long)H * W];
        float* dst = y + n * y_ns + ((long)c * Ho + hout) * Wo + 2 * w;
        dst[0] = v0;
        dst[1] = v1;
    }
}

extern "C" int rfn_squeeze2d_f32(const float* x, long x_ns, float* y, long y_ns, int N, int C, int H, int W, int undo,
                                 rfn_stream_t stream) {
    RFN_CHECK_ARG(x && y && N >= 0 && C > 0 && H > 0 && W > 0, -1);
    if (N == 0) return 0;
    hipStream_t s = (hipStream_t)stream;
    if (!undo) {
        RFN_CHECK_ARG((H % 2 == 0) && (W % 2 == 0), -2);
        long total = (long)N * C * H * (W / 2);
        int grid = (int)((total + 255) / 256 < 4096 ? (total + 255) / 256 : 4096);
        hipLaunchKernelGGL(squeeze2d_fwd_kernel, dim3(grid), dim3(256), 0, s, x, x_ns, y, y_ns, N, C, H, W);
    } else {
        RFN_CHECK_ARG(C % 4 == 0, -2);
        long total = (long)N * (C / 4) * (H * 2) * W;
        int grid = (int)((total + 255) / 256 < 4096 ? (total + 255) / 256 : 4096);
        hipLaunchKernelGGL(squeeze2d_undo_kernel, dim3(grid), dim3(256), 0, s, x, x_ns, y, y_ns, N, C, H, W);
    }
    RFN_LAUNCH_CHECK();
    return 0;
}

// ------------------------------------------------------------------------------------------------ channel stats
// One block per channel: pass 1 mean, pass 2 Σ(x-mean)² / (n-1)   (ActNorm data dependent init; one-time cost).
__global__ __launch_bounds__(256) void channel_stats_kernel(const float* __restrict__ x, long x_ns, float* mean,
                                                            float* var, int N, int C, int HW) {
    __shared__ float sm[4];
    const int c = blockIdx.x;
    const long cnt = (long)N * HW;
    float s = 0.f;
    for (long i = threadIdx.x; i < cnt; i += 256) {
        int n = (int)(i / HW), p = (int)(i % HW);
        s += x[n * x_ns + (long)c * HW + p];
    }
    float m = block_sum_256(s, sm) / (float)cnt;
    float q = 0.f;
    for (long i = threadIdx.x; i < cnt; i += 256) {
        int n = (int)(i / HW), p = (int)(i % HW);
        float d = x[n * x_ns + (long)c * HW + p] - m;
        q += d * d;
    }
    q = block_sum_256(q, sm);
    if (threadIdx.x == 0) {
        mean[c] = m;
        var[c] = q / (float)(cnt - 1);
    }
}
extern "C" int rfn_channel_stats_f32(const float* x, long x_ns, float* mean, float* var_unbiased, int N, int C, int HW,
                                     rfn_stream_t stream) {
    RFN_CHECK_ARG(x && mean && var_unbiased && N > 0 && C > 0 && HW > 0, -1);
    hipLaunchKernelGGL(channel_stats_kernel, dim3(C), dim3(256), 0, (hipStream_t)stream, x, x_ns, mean, var_unbiased, N,
                       C, HW);
    RFN_LAUNCH_CHECK();
    return 0;
}

__device__ __forceinline__ float clamp_ls(float s, int clamp_type, float sc, float sh) {
    switch (clamp_type) {
        case 0: return sc * tanhf(s) + sh;
        case 1: return -log1pf(expf(-(s + 2.0f)));  // log sigmoid(s+2)
        case 2: return 2.5f * 0.636f * atanf(s / 2.5f);
        default: return s;
    }
}
// d ls / d s
__device__ __forceinline__ float clamp_ls_grad(float s, int clamp_type, float sc) {
    switch (clamp_type) {
        case 0: {
            float t = tanhf(s);
            return sc * (1.0f - t * t);
        }
        case 1: return 1.0f / (1.0f + expf(s + 2.0f));  // 1 - sigmoid(s+2)
        case 2: {
            float r = s / 2.5f;
            return 0.636f / (1.0f + r * r);
        }
        default: return 1.0f;
    }
}


// Backward of the affine coupling + Conv2dZeros epilogue of the PREVIOUS Glow step, appended to the ActNorm/InvConv
// backward of this step (whose gx IS the previous step's output gradient and whose x IS its output): see
// rfn_glow_shell_bwd_f32.
struct ShellBwdTail {
    const float* o;        // previous step's coupling-net output [N, C, HW]
    long o_ns;
    const float* glogdet;  // [N] or NULL
    const float* scale;    // realnvp clamp parameters [C/2] (clamp_type 0)
    const float* scale_shift;
    const float* l3;       // Conv2dZeros logs [C]
    float* gz;             // out: gradient wrt the previous step's post-InvConv tensor [N, C, HW] (whole tensor)
    long gz_ns;
    float* gpre;           // out: gradient at the previous step's conv3 output
    long gpre_ns;
    float* gscale;         // accumulated [C/2] (clamp_type 0)
    float* gshift;
    float* gb3;            // accumulated [C]
    float* gl3;
    int clamp_type;
    int ld_const;  // this step's ActNorm also contributes HW * sum_c logs[c] to every frame's log-det (added by the forward
                   // shell kernel): its gradient HW * glogdet[n] is added to glogs here (glogdet must be given)
};

// ------------------------------------------------------------------------------------------------ actnorm + invconv
// A block owns PB consecutive "global pixels" q = n*HW + p.  y = (x+b)*exp(l) is staged in LDS as [C][PB]
// (lane-consecutive pixels -> conflict free); every thread then forms the C outputs of its pixel with W read through
// the scalar cache (uniform indices).
template <int REV>
__global__ __launch_bounds__(256) void actnorm_invconv_kernel(const float* __restrict__ x, long x_ns,
                                                              const float* __restrict__ bias,
                                                              const float* __restrict__ logs,
                                                              const float* __restrict__ Wm, float* __restrict__ z,
                                                              long z_ns, int N, int C, int HW, int PB) {
    extern __shared__ float lds[];  // [C][PB]
    // thread = (pixel px of the block, channel group ig): all 256 threads stage and compute whatever PB is -- at the deep
    // levels (C = 32, 64; a few thousand pixels) PB shrinks to 32 so that the launch still has tens of blocks and a
    // thread forms C/8 outputs instead of all C.
    const int px = threadIdx.x & (PB - 1), ig = threadIdx.x / PB, NG = 256 / PB;
    const long q = (long)blockIdx.x * PB + px;
    const bool valid = q < (long)N * HW;
    int n = 0, p = 0;
    if (valid) {
        n = (int)(q / HW);
        p = (int)(q % HW);
    }
    const float* src = x + n * x_ns + p;
    for (int c = ig; c < C; c += NG) {
        float v = valid ? src[(long)c * HW] : 0.f;
        if (!REV) v = (v + bias[c]) * expf(logs[c]);
        lds[c * PB + px] = v;
    }
    __syncthreads();
    if (valid) {
        float* dst = z + n * z_ns + p;
        for (int i = ig; i < C; i += NG) {
            float a = 0.f;
            const float* wr = Wm + (long)i * C;
#pragma unroll 8
            for (int j = 0; j < C; ++j) a = fmaf(wr[j], lds[j * PB + px], a);
            if (REV) a = a * expf(-logs[i]) - bias[i];
            dst[(long)i * HW] = a;
        }
    }
}

static int shell_pb(int C) {
    int PB = 256;
    while ((long)C * PB * 4 > 49152 && PB > 64) PB >>= 1;
    return PB;
}

static int launch_actnorm_invconv(int rev, const float* x, long x_ns, const float* bias, const float* logs,
                                  const float* Wm, float* z, long z_ns, int N, int C, int HW, hipStream_t s) {
    int PB = shell_pb(C);
    long tot = (long)N * HW;
    while (PB > 32 && tot / PB < 256) PB >>= 1;  // few pixels: more, smaller blocks with the channels split over threads
    size_t lds = (size_t)C * PB * 4;
    if (lds > 160 * 1024) {
        rfn_set_error("actnorm_invconv: C=%d too large for the LDS-staged kernel", C);
        return -3;
    }
    int grid = (int)((tot + PB - 1) / PB);
    if (rev) {
        if (lds > 65536)
            (void)hipFuncSetAttribute((const void*)actnorm_invconv_kernel<1>, hipFuncAttributeMaxDynamicSharedMemorySize,
                                (int)lds);
        hipLaunchKernelGGL(actnorm_invconv_kernel<1>, dim3(grid), dim3(256), lds, s, x, x_ns, bias, logs, Wm, z, z_ns, N,
                           C, HW, PB);
    } else {
        if (lds > 65536)
            (void)hipFuncSetAttribute((const void*)actnorm_invconv_kernel<0>, hipFuncAttributeMaxDynamicSharedMemorySize,
                                (int)lds);
        hipLaunchKernelGGL(actnorm_invconv_kernel<0>, dim3(grid), dim3(256), lds, s, x, x_ns, bias, logs, Wm, z, z_ns, N,
                           C, HW, PB);
    }
    return 0;
}

extern "C" int rfn_actnorm_invconv_fwd_f32(const float* x, long x_ns, const float* bias, const float* logs,
                                           const float* Wm, float* z, long z_ns, int N, int C, int HW,
                                           rfn_stream_t stream) {
    RFN_CHECK_ARG(x && bias && logs && Wm && z && N >= 0 && C > 0 && HW > 0, -1);
    if (N == 0) return 0;
    int rc = launch_actnorm_invconv(0, x, x_ns, bias, logs, Wm, z, z_ns, N, C, HW, (hipStream_t)stream);
    if (rc) return rc;
    RFN_LAUNCH_CHECK();
    return 0;
}
extern "C" int rfn_invconv_actnorm_rev_f32(const float* zin, long z_ns, const float* bias, const float* logs,
                                           const float* Winv, float* x, long x_ns, int N, int C, int HW,
                                           rfn_stream_t stream) {
    RFN_CHECK_ARG(zin && bias && logs && Winv && x && N >= 0 && C > 0 && HW > 0, -1);
    if (N == 0) return 0;
    int rc = launch_actnorm_invconv(1, zin, z_ns, bias, logs, Winv, x, x_ns, N, C, HW, (hipStream_t)stream);
    if (rc) return rc;
    RFN_LAUNCH_CHECK();
    return 0;
}

// backward.  LDS: Y[C][PBS] and G[C][PBS] with PBS = PB+1 (odd stride: the gW phase reads rows with a stride), plus
// block-level accumulators Wacc[C*C], Bacc[C], Lacc[C].  A block sweeps many pixel tiles (grid-stride) and issues its
// global float atomics ONCE at the end: a few hundred blocks x (C*C + 2C) atomics instead of one set per 256 pixels —
// same-address atomics serialise at the memory side (MI355X_MICROARCH.md "Global float atomics": 14x slower).
// TAIL: the gx part runs over channel PAIRS (j, j + C/2) so that the thread that forms the previous step's output
// gradient at both halves of a pair applies that step's coupling / Conv2dZeros-epilogue backward on the spot (gx itself
// is not written).
template <bool TAIL>
__global__ __launch_bounds__(256) void actnorm_invconv_bwd_kernel(
    const float* __restrict__ x, long x_ns, const float* __restrict__ bias, const float* __restrict__ logs,
    const float* __restrict__ Wm, const float* __restrict__ gz, long gz_ns, float* __restrict__ gx, long gx_ns,
    float* __restrict__ gW, float* __restrict__ gbias, float* __restrict__ glogs, int N, int C, int HW, int PB,
    int ntiles, const ShellBwdTail tl) {
    extern __shared__ float lds[];
    const int PBS = PB + 1;
    float* Y = lds;                   // [C][PBS]
    float* G = lds + C * PBS;         // [C][PBS]
    float* Wacc = lds + 2 * C * PBS;  // [C*C]
    float* Bacc = Wacc + C * C;       // [C]
    float* Lacc = Bacc + C;           // [C]
    float* Tacc = Lacc + C;           // TAIL: [gb3 C][gl3 C][gscale C/2][gshift C/2]
    const int t = threadIdx.x;
    const int E = C * C;
    for (int e = t; e < E + 2 * C + (TAIL ? 3 * C : 0); e += 256) Wacc[e] = 0.f;
    // all 256 threads stage: thread = (pixel px, channel group cg); PB is a power of two <= 256
    const int px = t & (PB - 1), cg = t / PB, ncg = 256 / PB;
    const long total = (long)N * HW;
    for (int tile = blockIdx.x; tile < ntiles; tile += gridDim.x) {
        const long q = (long)tile * PB + px;
        const bool valid = q < total;
        const long qq = valid ? q : 0;  // masked pixels read pixel 0 (always mapped) and are zeroed after the load
        const int n = (int)(qq / HW), p = (int)(qq % HW);
        __syncthreads();  // previous tile fully consumed (also orders the accumulator zeroing)
        const float* xs = x + n * x_ns + p;
        const float* gs = gz + n * gz_ns + p;
#pragma unroll 8
        for (int c = cg; c < C; c += ncg) {
            const float xv = xs[(long)c * HW], gv = gs[(long)c * HW];
            Y[c * PBS + px] = valid ? (xv + bias[c]) * expf(logs[c]) : 0.f;
            G[c * PBS + px] = valid ? gv : 0.f;
        }
        __syncthreads();
        // gW[i][j] += Σ_p gz_i(p) y_j(p): small C -> 256/(C*C) threads per entry split the pixels (LDS atomic
        // combine); large C -> every thread owns entries e, e+256, ... (plain LDS read-modify-write, single owner)
        if (E <= 256) {
            // (every y-block staged the same tile: the pixels of the tile are split over them)
            const int e = t % E, grp = t / E, G_ = 256 / E;
            if (grp < G_) {
                const float* gi = G + (e / C) * PBS;
                const float* yj = Y + (e % C) * PBS;
                // (four independent partial sums: the LDS reads of four pixels are in flight instead of one)
                float a0 = 0.f, a1 = 0.f, a2 = 0.f, a3 = 0.f;
                const int st = G_ * gridDim.y;
                int pp = grp + G_ * blockIdx.y;
                for (; pp + 3 * st < PB; pp += 4 * st) {
                    a0 = fmaf(gi[pp], yj[pp], a0);
                    a1 = fmaf(gi[pp + st], yj[pp + st], a1);
                    a2 = fmaf(gi[pp + 2 * st], yj[pp + 2 * st], a2);
                    a3 = fmaf(gi[pp + 3 * st], yj[pp + 3 * st], a3);
                }
                for (; pp < PB; pp += st) a0 = fmaf(gi[pp], yj[pp], a0);
                atomicAdd(&Wacc[e], (a0 + a1) + (a2 + a3));
            }
        } else {
            // entries are split over blockIdx.y (deep levels have few pixel tiles: 38 at the 2x2 level)
            for (int e = blockIdx.y * 256 + t; e < E; e += 256 * gridDim.y) {
                const float* gi = G + (e / C) * PBS;
                const float* yj = Y + (e % C) * PBS;
                float a0 = 0.f, a1 = 0.f, a2 = 0.f, a3 = 0.f;   // (PB is a multiple of 4)
                for (int pp = 0; pp < PB; pp += 4) {
                    a0 = fmaf(gi[pp], yj[pp], a0);
                    a1 = fmaf(gi[pp + 1], yj[pp + 1], a1);
                    a2 = fmaf(gi[pp + 2], yj[pp + 2], a2);
                    a3 = fmaf(gi[pp + 3], yj[pp + 3], a3);
                }
                Wacc[e] += (a0 + a1) + (a2 + a3);
            }
        }
        // gy_j = Σ_i W[i][j] gz_i ; gx_j = gy_j * exp(logs_j) ; gbias_j += Σ gx_j ; glogs_j += Σ gy_j y_j.
        // Thread (px, cg) takes output channels j = cg, cg+ncg, ...: j is wave-uniform (PB >= 64), so W comes through the
        // scalar cache and wave_sum adds over 64 pixels; masked pixels hold zeros.
        // The output channels are split over blockIdx.y as well (every y-block staged the same tile): at the deep levels a
        // launch has only a handful of pixel tiles and one block per tile ran C/ncg x C serial steps per thread.
        // gradient of the parameter-only log-det term HW * sum_c logs[c] (one contribution per frame: its pixel 0)
        const float ldc = (tl.ld_const && valid && p == 0) ? (float)HW * tl.glogdet[n] : 0.f;
        if (!TAIL) {
            for (int j = cg + ncg * blockIdx.y; j < C; j += ncg * gridDim.y) {
                float a = 0.f;
#pragma unroll 8
                for (int i = 0; i < C; ++i) a = fmaf(Wm[(long)i * C + j], G[i * PBS + px], a);
                const float gxv = a * expf(logs[j]);
                if (valid) gx[n * gx_ns + (long)j * HW + p] = gxv;
                const float s1 = wave_sum_dpp(gxv);
                const float s2 = wave_sum_dpp(a * Y[j * PBS + px] + ldc);
                if ((t & 63) == 0) {
                    atomicAdd(&Bacc[j], s1);
                    atomicAdd(&Lacc[j], s2);
                }
            }
        } else {
            const int Ch = C >> 1;
            // the coupling backward's global operands (o[2j], o[2j+1], x[j+Ch] of this pixel) are loaded one channel
            // pair AHEAD of the matrix-vector product that needs them: used right after their load they cost one
            // global-memory latency per pair (4 pairs per thread at C = 16: 8 of the launch's 30 us)
            const int j0 = cg + ncg * blockIdx.y, jst = ncg * gridDim.y;
            const float gld = (valid && tl.glogdet) ? tl.glogdet[n] : 0.f;
            const float* op = tl.o + n * tl.o_ns + p;
            float o0n = 0.f, svn = 0.f, zon = 0.f;
            if (valid && j0 < Ch) {
                o0n = op[(long)(2 * j0) * HW];
                svn = op[(long)(2 * j0 + 1) * HW];
                zon = xs[(long)(j0 + Ch) * HW];
            }
            for (int j = j0; j < Ch; j += jst) {
                const float o0 = o0n, sv = svn, zo = zon;
                if (valid && j + jst < Ch) {
                    o0n = op[(long)(2 * (j + jst)) * HW];
                    svn = op[(long)(2 * (j + jst) + 1) * HW];
                    zon = xs[(long)(j + jst + Ch) * HW];
                }
                float a1 = 0.f, a2 = 0.f;
#pragma unroll 8
                for (int i = 0; i < C; ++i) {
                    const float gv = G[i * PBS + px];
                    a1 = fmaf(Wm[(long)i * C + j], gv, a1);
                    a2 = fmaf(Wm[(long)i * C + j + Ch], gv, a2);
                }
                const float g1 = a1 * expf(logs[j]), g2 = a2 * expf(logs[j + Ch]);  // previous step's gout at (j, j+Ch)
                // previous step: z2' = (z2 + o[2j]) * exp(ls(o[2j+1])), its output z2' is this step's x at channel j+Ch
                float sc = 0.f, sh = 0.f;
                if (tl.clamp_type == 0) {
                    sc = tl.scale[j];
                    sh = tl.scale_shift[j];
                }
                const float ls = clamp_ls(sv, tl.clamp_type, sc, sh);
                const float gls = valid ? g2 * zo + gld : 0.f;
                const float gzv = g2 * expf(ls);
                const float go1 = gls * clamp_ls_grad(sv, tl.clamp_type, sc);
                const float u0 = gzv * expf(3.f * tl.l3[2 * j]), u1 = go1 * expf(3.f * tl.l3[2 * j + 1]);
                if (valid) {
                    tl.gz[n * tl.gz_ns + (long)j * HW + p] = g1;
                    tl.gz[n * tl.gz_ns + (long)(j + Ch) * HW + p] = gzv;
                    tl.gpre[n * tl.gpre_ns + (long)(2 * j) * HW + p] = u0;
                    tl.gpre[n * tl.gpre_ns + (long)(2 * j + 1) * HW + p] = u1;
                }
                const float s1a = wave_sum_dpp(g1), s1b = wave_sum_dpp(g2);
                const float s2a = wave_sum_dpp(a1 * Y[j * PBS + px] + ldc), s2b = wave_sum_dpp(a2 * Y[(j + Ch) * PBS + px] + ldc);
                const float tb0 = wave_sum_dpp(u0), tb1 = wave_sum_dpp(u1);
                const float tl0 = wave_sum_dpp(gzv * o0), tl1 = wave_sum_dpp(go1 * sv);
                float tsc = 0.f, tsh = 0.f;
                if (tl.clamp_type == 0) {
                    tsc = wave_sum_dpp(gls * tanhf(sv));
                    tsh = wave_sum_dpp(gls);
                }
                if ((t & 63) == 0) {
                    atomicAdd(&Bacc[j], s1a);
                    atomicAdd(&Bacc[j + Ch], s1b);
                    atomicAdd(&Lacc[j], s2a);
                    atomicAdd(&Lacc[j + Ch], s2b);
                    atomicAdd(&Tacc[2 * j], tb0);
                    atomicAdd(&Tacc[2 * j + 1], tb1);
                    atomicAdd(&Tacc[C + 2 * j], 3.f * tl0);
                    atomicAdd(&Tacc[C + 2 * j + 1], 3.f * tl1);
                    if (tl.clamp_type == 0) {
                        atomicAdd(&Tacc[2 * C + j], tsc);
                        atomicAdd(&Tacc[2 * C + Ch + j], tsh);
                    }
                }
            }
        }
    }
    __syncthreads();
    if (E <= 256) {
        for (int e = t; e < E; e += 256) atomicAdd(&gW[e], Wacc[e]);
    } else {
        for (int e = blockIdx.y * 256 + t; e < E; e += 256 * gridDim.y) atomicAdd(&gW[e], Wacc[e]);  // owned entries
    }
    for (int c = t; c < C; c += 256) {  // channels this y-block did not compute hold zeros
        if (Bacc[c] != 0.f) atomicAdd(&gbias[c], Bacc[c]);
        if (Lacc[c] != 0.f) atomicAdd(&glogs[c], Lacc[c]);
        if (TAIL) {
            if (Tacc[c] != 0.f) atomicAdd(&tl.gb3[c], Tacc[c]);
            if (Tacc[C + c] != 0.f) atomicAdd(&tl.gl3[c], Tacc[C + c]);
            if (tl.clamp_type == 0) {
                const int Ch = C >> 1;
                float* dst = c < Ch ? tl.gscale + c : tl.gshift + (c - Ch);
                if (Tacc[2 * C + c] != 0.f) atomicAdd(dst, Tacc[2 * C + c]);
            }
        }
    }
}

// Small channel counts (the two finest flow levels, C = 4 and 8, where N*HW is largest): one thread per pixel, all
// C values in registers, every parameter-gradient partial in registers across a grid-stride sweep; the block's
// C*C + 2C sums meet through wave shuffles + LDS and reach global memory as one atomic each.
template <int C, bool TAIL>
__global__ __launch_bounds__(256) void actnorm_invconv_bwd_small_kernel(
    const float* __restrict__ x, long x_ns, const float* __restrict__ bias, const float* __restrict__ logs,
    const float* __restrict__ Wm, const float* __restrict__ gz, long gz_ns, float* __restrict__ gx, long gx_ns,
    float* __restrict__ gW, float* __restrict__ gbias, float* __restrict__ glogs, int N, int HW,
    const ShellBwdTail tl) {
    constexpr int Ch = C / 2;
    __shared__ float red[C * C + 2 * C + 3 * C];  // TAIL: + [gb3 C][gl3 C][gscale Ch][gshift Ch]
    for (int e = threadIdx.x; e < C * C + 2 * C + 3 * C; e += 256) red[e] = 0.f;
    __syncthreads();
    float w[C][C], b[C], es[C];
#pragma unroll
    for (int i = 0; i < C; ++i) {
        b[i] = bias[i];
        es[i] = expf(logs[i]);
#pragma unroll
        for (int j = 0; j < C; ++j) w[i][j] = Wm[i * C + j];
    }
    float aW[C][C], ab[C], al[C];
    float tb[C], tlg[C], tsc[Ch], tsh[Ch];  // TAIL accumulators
    float sc[Ch], sh[Ch], e3[C];
#pragma unroll
    for (int i = 0; i < C; ++i) {
        ab[i] = 0.f;
        al[i] = 0.f;
        tb[i] = 0.f;
        tlg[i] = 0.f;
        e3[i] = TAIL ? expf(3.f * tl.l3[i]) : 1.f;
#pragma unroll
        for (int j = 0; j < C; ++j) aW[i][j] = 0.f;
    }
#pragma unroll
    for (int j = 0; j < Ch; ++j) {
        tsc[j] = 0.f;
        tsh[j] = 0.f;
        sc[j] = (TAIL && tl.clamp_type == 0) ? tl.scale[j] : 0.f;
        sh[j] = (TAIL && tl.clamp_type == 0) ? tl.scale_shift[j] : 0.f;
    }
    const long total = (long)N * HW;
    for (long q = (long)blockIdx.x * 256 + threadIdx.x; q < total; q += (long)gridDim.x * 256) {
        const long n = q / HW;
        const int p = (int)(q - n * HW);
        float xr[C], y[C], g[C], gxr[C], ov[C];
        float gld = 0.f;
        // every load of the iteration is issued up front (one memory round trip per pixel instead of two)
#pragma unroll
        for (int c = 0; c < C; ++c) {
            xr[c] = x[n * x_ns + (long)c * HW + p];
            g[c] = gz[n * gz_ns + (long)c * HW + p];
            ov[c] = TAIL ? tl.o[n * tl.o_ns + (long)c * HW + p] : 0.f;
        }
        if (TAIL && tl.glogdet) gld = tl.glogdet[n];
#pragma unroll
        for (int c = 0; c < C; ++c) y[c] = (xr[c] + b[c]) * es[c];
#pragma unroll
        for (int j = 0; j < C; ++j) {
            float gy = 0.f;
#pragma unroll
            for (int i = 0; i < C; ++i) {
                gy = fmaf(w[i][j], g[i], gy);
                aW[i][j] = fmaf(g[i], y[j], aW[i][j]);
            }
            const float gxv = gy * es[j];
            gxr[j] = gxv;
            if (!TAIL) gx[n * gx_ns + (long)j * HW + p] = gxv;
            ab[j] += gxv;
            al[j] = fmaf(gy, y[j], al[j]);
        }
        if (tl.ld_const && p == 0) {
            const float ldc = (float)HW * (TAIL ? gld : tl.glogdet[n]);
#pragma unroll
            for (int c = 0; c < C; ++c) al[c] += ldc;
        }
        if (TAIL) {
#pragma unroll
            for (int j = 0; j < Ch; ++j) {
                const float o0 = ov[2 * j];
                const float sv = ov[2 * j + 1];
                const float ls = clamp_ls(sv, tl.clamp_type, sc[j], sh[j]);
                const float gls = gxr[j + Ch] * xr[j + Ch] + gld;
                const float gzv = gxr[j + Ch] * expf(ls);
                const float go1 = gls * clamp_ls_grad(sv, tl.clamp_type, sc[j]);
                const float u0 = gzv * e3[2 * j], u1 = go1 * e3[2 * j + 1];
                tl.gz[n * tl.gz_ns + (long)j * HW + p] = gxr[j];
                tl.gz[n * tl.gz_ns + (long)(j + Ch) * HW + p] = gzv;
                tl.gpre[n * tl.gpre_ns + (long)(2 * j) * HW + p] = u0;
                tl.gpre[n * tl.gpre_ns + (long)(2 * j + 1) * HW + p] = u1;
                tb[2 * j] += u0;
                tb[2 * j + 1] += u1;
                tlg[2 * j] = fmaf(gzv, o0, tlg[2 * j]);
                tlg[2 * j + 1] = fmaf(go1, sv, tlg[2 * j + 1]);
                if (tl.clamp_type == 0) {
                    tsc[j] = fmaf(gls, tanhf(sv), tsc[j]);
                    tsh[j] += gls;
                }
            }
        }
    }
    const bool lead = (threadIdx.x & 63) == 0;
#pragma unroll
    for (int i = 0; i < C; ++i) {
#pragma unroll
        for (int j = 0; j < C; ++j) {
            const float v = wave_sum_dpp(aW[i][j]);
            if (lead) atomicAdd(&red[i * C + j], v);
        }
        const float vb = wave_sum_dpp(ab[i]), vl = wave_sum_dpp(al[i]);
        if (lead) {
            atomicAdd(&red[C * C + i], vb);
            atomicAdd(&red[C * C + C + i], vl);
        }
        if (TAIL) {
            const float v0 = wave_sum_dpp(tb[i]), v1 = wave_sum_dpp(tlg[i]);
            if (lead) {
                atomicAdd(&red[C * C + 2 * C + i], v0);
                atomicAdd(&red[C * C + 3 * C + i], 3.f * v1);
            }
        }
    }
    if (TAIL && tl.clamp_type == 0) {
#pragma unroll
        for (int j = 0; j < Ch; ++j) {
            const float v0 = wave_sum_dpp(tsc[j]), v1 = wave_sum_dpp(tsh[j]);
            if (lead) {
                atomicAdd(&red[C * C + 4 * C + j], v0);
                atomicAdd(&red[C * C + 4 * C + Ch + j], v1);
            }
        }
    }
    __syncthreads();
    for (int e = threadIdx.x; e < C * C; e += 256) atomicAdd(&gW[e], red[e]);
    if (threadIdx.x < C) {
        atomicAdd(&gbias[threadIdx.x], red[C * C + threadIdx.x]);
        atomicAdd(&glogs[threadIdx.x], red[C * C + C + threadIdx.x]);
        if (TAIL) {
            atomicAdd(&tl.gb3[threadIdx.x], red[C * C + 2 * C + threadIdx.x]);
            atomicAdd(&tl.gl3[threadIdx.x], red[C * C + 3 * C + threadIdx.x]);
            if (tl.clamp_type == 0) {
                float* dst = threadIdx.x < Ch ? tl.gscale + threadIdx.x : tl.gshift + (threadIdx.x - Ch);
                atomicAdd(dst, red[C * C + 4 * C + threadIdx.x]);
            }
        }
    }
}

template <bool TAIL>
static int launch_actnorm_invconv_bwd(const float* x, long x_ns, const float* bias, const float* logs, const float* Wm,
                                      const float* gz, long gz_ns, float* gx, long gx_ns, float* gW, float* gbias,
                                      float* glogs, int N, int C, int HW, const ShellBwdTail& tl, hipStream_t st) {
    if (C == 4 || C == 8) {
        long tot = (long)N * HW;
        // few, fat blocks: every block ends with C*C+2C same-address atomics, which serialise at the memory side
        // (512 / 1024 blocks measured slower: 256-deep same-address atomic chains per accumulator are the budget)
        int grid = (int)((tot + 255) / 256 < 256 ? (tot + 255) / 256 : 256);
        if (C == 4)
            hipLaunchKernelGGL((actnorm_invconv_bwd_small_kernel<4, TAIL>), dim3(grid), dim3(256), 0, st, x, x_ns, bias,
                               logs, Wm, gz, gz_ns, gx, gx_ns, gW, gbias, glogs, N, HW, tl);
        else
            hipLaunchKernelGGL((actnorm_invconv_bwd_small_kernel<8, TAIL>), dim3(grid), dim3(256), 0, st, x, x_ns, bias,
                               logs, Wm, gz, gz_ns, gx, gx_ns, gW, gbias, glogs, N, HW, tl);
        return 0;
    }
    int PB = 256;
    const size_t extra = ((size_t)C * C + 2 * C + (TAIL ? 3 * C : 0)) * 4;
    while ((size_t)2 * C * (PB + 1) * 4 + extra > 65536 && PB > 64) PB >>= 1;
    size_t lds = (size_t)2 * C * (PB + 1) * 4 + extra;
    if (lds > 160 * 1024) {
        rfn_set_error("actnorm_invconv_bwd: C=%d too large", C);
        return -3;
    }
    if (lds > 65536)
        (void)hipFuncSetAttribute((const void*)actnorm_invconv_bwd_kernel<TAIL>,
                                  hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
    long tot = (long)N * HW;
    int ntiles = (int)((tot + PB - 1) / PB);
    int grid = ntiles < 512 ? ntiles : 512;
    // blockIdx.y splits the gW entries (C*C > 256) and the output channels of the gx part: aim at <= 2 channels per thread
    int ny = (C * C > 256) ? (C * C + 511) / 512 : 1;
    const int jpt = C * PB / 256;  // channels per thread without a split
    if (ny < jpt / 2) ny = jpt / 2;
    if (ny > 8) ny = 8;
    while (ny > 1 && grid * ny > 512) ny >>= 1;  // one wave of workgroups: the y-blocks re-stage the same tile
    hipLaunchKernelGGL(actnorm_invconv_bwd_kernel<TAIL>, dim3(grid, ny), dim3(256), lds, st, x, x_ns, bias, logs, Wm, gz,
                       gz_ns, gx, gx_ns, gW, gbias, glogs, N, C, HW, PB, ntiles, tl);
    return 0;
}

extern "C" int rfn_actnorm_invconv_bwd_f32(const float* x, long x_ns, const float* bias, const float* logs,
                                           const float* Wm, const float* gz, long gz_ns, float* gx, long gx_ns,
                                           float* gW, float* gbias, float* glogs, int N, int C, int HW,
                                           rfn_stream_t stream) {
    RFN_CHECK_ARG(x && bias && logs && Wm && gz && gx && gW && gbias && glogs && N >= 0 && C > 0 && HW > 0, -1);
    if (N == 0) return 0;
    ShellBwdTail tl = {};
    int rc = launch_actnorm_invconv_bwd<false>(x, x_ns, bias, logs, Wm, gz, gz_ns, gx, gx_ns, gW, gbias, glogs, N, C, HW,
                                               tl, (hipStream_t)stream);
    if (rc) return rc;
    RFN_LAUNCH_CHECK();
    return 0;
}

// rfn_actnorm_invconv_bwd_f32 for a step whose ActNorm also put HW * sum_c logs[c] into every frame's log-det (the level
// node's forward shell kernel adds that term): glogs additionally receives HW * sum_n glogdet[n].
extern "C" int rfn_actnorm_invconv_bwd_ld_f32(const float* x, long x_ns, const float* bias, const float* logs,
                                              const float* Wm, const float* gz, long gz_ns, float* gx, long gx_ns,
                                              float* gW, float* gbias, float* glogs, const float* glogdet, int N, int C,
                                              int HW, rfn_stream_t stream) {
    RFN_CHECK_ARG(x && bias && logs && Wm && gz && gx && gW && gbias && glogs && N >= 0 && C > 0 && HW > 0, -1);
    if (N == 0) return 0;
    ShellBwdTail tl = {};
    tl.glogdet = glogdet;
    tl.ld_const = glogdet ? 1 : 0;
    int rc = launch_actnorm_invconv_bwd<false>(x, x_ns, bias, logs, Wm, gz, gz_ns, gx, gx_ns, gW, gbias, glogs, N, C, HW,
                                               tl, (hipStream_t)stream);
    if (rc) return rc;
    RFN_LAUNCH_CHECK();
    return 0;
}

// Backward shell between two consecutive Glow steps in ONE launch: ActNorm + InvConv backward of step k+1 (x = its
// input = step k's output, gz = gradient wrt its post-InvConv tensor) followed by the affine-coupling and Conv2dZeros
// epilogue backward of step k, fed from registers (the gradient wrt step k's output never goes to HBM).
extern "C" int rfn_glow_shell_bwd_f32(const float* x, long x_ns, const float* bias, const float* logs, const float* Wm,
                                      const float* gz, long gz_ns, float* gW, float* gbias, float* glogs,
                                      const float* o, long o_ns, const float* glogdet, const float* scale,
                                      const float* scale_shift, const float* l3, float* gz_prev, long gz_prev_ns,
                                      float* gpre, long gpre_ns, float* gscale, float* gscale_shift, float* gb3,
                                      float* gl3, int clamp_type, int ld_const, int N, int C, int HW,
                                      rfn_stream_t stream) {
    RFN_CHECK_ARG(x && bias && logs && Wm && gz && gW && gbias && glogs && N >= 0 && C > 0 && (C % 2 == 0) && HW > 0, -1);
    RFN_CHECK_ARG(o && l3 && gz_prev && gpre && gb3 && gl3, -2);
    RFN_CHECK_ARG(clamp_type != 0 || (scale && scale_shift && gscale && gscale_shift), -3);
    if (N == 0) return 0;
    ShellBwdTail tl = {o, o_ns, glogdet, scale, scale_shift, l3, gz_prev, gz_prev_ns, gpre, gpre_ns, gscale,
                       gscale_shift, gb3, gl3, clamp_type, (ld_const && glogdet) ? 1 : 0};
    int rc = launch_actnorm_invconv_bwd<true>(x, x_ns, bias, logs, Wm, gz, gz_ns, nullptr, 0, gW, gbias, glogs, N, C, HW,
                                              tl, (hipStream_t)stream);
    if (rc) return rc;
    RFN_LAUNCH_CHECK();
    return 0;
}

// ------------------------------------------------------------------------------------------------ affine coupling
// one block per frame; elements e in [0, C/2*HW): channel j = e / HW
__global__ __launch_bounds__(256) void affine_coupling_kernel(float* __restrict__ z, long z_ns,
                                                              const float* __restrict__ o, long o_ns,
                                                              const float* __restrict__ scale,
                                                              const float* __restrict__ scale_shift,
                                                              float* __restrict__ logdet, int clamp_type, int reverse,
                                                              int C, int HW) {
    __shared__ float sm[4];
    const int n = blockIdx.x, Ch = C >> 1;
    float* z2 = z + n * z_ns + (long)Ch * HW;
    const float* on = o + n * o_ns;
    float acc = 0.f;
    for (int e = threadIdx.x; e < Ch * HW; e += 256) {
        int j = e / HW, p = e - j * HW;
        float shift = on[(long)(2 * j) * HW + p];
        float s = on[(long)(2 * j + 1) * HW + p];
        float sc = 0.f, sh = 0.f;
        if (clamp_type == 0) {
            sc = scale[j];
            sh = scale_shift[j];
        }
        float ls = clamp_ls(s, clamp_type, sc, sh);
        float v = z2[e];
        if (!reverse)
            v = (v + shift) * expf(ls);
        else
            v = v * expf(-ls) - shift;
        z2[e] = v;
        acc += ls;
    }
    if (logdet) {
        float tot = block_sum_256(acc, sm);
        if (threadIdx.x == 0) logdet[n] += reverse ? -tot : tot;
    }
}
extern "C" int rfn_affine_coupling_f32(float* z, long z_ns, const float* o, long o_ns, const float* scale,
                                       const float* scale_shift, float* logdet, int clamp_type, int reverse, int N,
                                       int C, int HW, rfn_stream_t stream) {
    RFN_CHECK_ARG(z && o && N >= 0 && C > 0 && (C % 2 == 0) && HW > 0, -1);
    RFN_CHECK_ARG(clamp_type != 0 || (scale && scale_shift), -2);
    if (N == 0) return 0;
    hipLaunchKernelGGL(affine_coupling_kernel, dim3(N), dim3(256), 0, (hipStream_t)stream, z, z_ns, o, o_ns, scale,
                       scale_shift, logdet, clamp_type, reverse, C, HW);
    RFN_LAUNCH_CHECK();
    return 0;
}

// backward: grid (X, C/2).  Block (bx, j) sweeps channel j over a strided share of the N*HW (frame, pixel) pairs —
// every thread busy at every level (C/2 = 2 at the finest level), per-channel parameter sums in registers, one atomic
// pair per block.
__global__ __launch_bounds__(256) void affine_coupling_bwd_kernel(
    const float* __restrict__ zout, long zout_ns, const float* __restrict__ o, long o_ns,
    const float* __restrict__ gout, long gout_ns, const float* __restrict__ glogdet, const float* __restrict__ scale,
    const float* __restrict__ scale_shift, float* __restrict__ gz, long gz_ns, float* __restrict__ go, long go_ns,
    float* __restrict__ gscale, float* __restrict__ gscale_shift, int clamp_type, int N, int C, int HW) {
    __shared__ float sm[4];
    const int Ch = C >> 1, j = blockIdx.y;
    float sc = 0.f, sh = 0.f;
    if (clamp_type == 0) {
        sc = scale[j];
        sh = scale_shift[j];
    }
    float a_sc = 0.f, a_sh = 0.f;
    const long total = (long)N * HW;
    for (long q = (long)blockIdx.x * 256 + threadIdx.x; q < total; q += (long)gridDim.x * 256) {
        const long n = q / HW;
        const int p = (int)(q - n * HW);
        const float s = o[n * o_ns + (long)(2 * j + 1) * HW + p];
        const float ls = clamp_ls(s, clamp_type, sc, sh);
        const float e = expf(ls);
        const float g = gout[n * gout_ns + (long)(Ch + j) * HW + p];
        const float gls = g * zout[n * zout_ns + (long)(Ch + j) * HW + p] + (glogdet ? glogdet[n] : 0.f);
        const float gzv = g * e;
        gz[n * gz_ns + (long)(Ch + j) * HW + p] = gzv;
        go[n * go_ns + (long)(2 * j) * HW + p] = gzv;  // d/dshift
        go[n * go_ns + (long)(2 * j + 1) * HW + p] = gls * clamp_ls_grad(s, clamp_type, sc);
        if (clamp_type == 0) {
            a_sc += gls * tanhf(s);
            a_sh += gls;
        }
    }
    if (clamp_type == 0) {
        const float t_sc = block_sum_256(a_sc, sm);
        const float t_sh = block_sum_256(a_sh, sm);
        if (threadIdx.x == 0) {
            atomicAdd(&gscale[j], t_sc);
            atomicAdd(&gscale_shift[j], t_sh);
        }
    }
}
extern "C" int rfn_affine_coupling_bwd_f32(const float* zout, long zout_ns, const float* o, long o_ns,
                                           const float* gout, long gout_ns, const float* glogdet, const float* scale,
                                           const float* scale_shift, float* gz, long gz_ns, float* go, long go_ns,
                                           float* gscale, float* gscale_shift, int clamp_type, int N, int C, int HW,
                                           rfn_stream_t stream) {
    RFN_CHECK_ARG(zout && o && gout && gz && go && N >= 0 && C > 0 && (C % 2 == 0) && HW > 0, -1);
    RFN_CHECK_ARG(clamp_type != 0 || (scale && scale_shift && gscale && gscale_shift), -2);
    if (N == 0) return 0;
    {
        const int Ch = C / 2;
        long tot = (long)N * HW;
        int gx_ = (int)((tot + 255) / 256);
        int cap = 512 / Ch < 1 ? 1 : 512 / Ch;
        if (gx_ > cap) gx_ = cap;
        hipLaunchKernelGGL(affine_coupling_bwd_kernel, dim3(gx_, Ch), dim3(256), 0, (hipStream_t)stream, zout, zout_ns, o,
                           o_ns, gout, gout_ns, glogdet, scale, scale_shift, gz, gz_ns, go, go_ns, gscale, gscale_shift,
                           clamp_type, N, C, HW);
    }
    RFN_LAUNCH_CHECK();
    return 0;
}

// ---- fused forward shell tail of a Glow step: (tap-expanded) Conv2dZeros output -> affine coupling -> per-frame log-det.
// One block per frame.  With P: o[c] = (sum_tap P[tap*C + c][y+dy-1][x+dx-1] + b3[c]) * exp(3 l3[c]) is formed here (and
// written to o_out for the backward pass); without P, o is read from o_in.  z2 <- (z2 + o[2j]) * exp(ls), ls = clamp(o[2j+1]);
// logdet[n] = sum ls is WRITTEN (no zero-filled accumulator, no separate gather / affine launches).
__global__ __launch_bounds__(256) void gather_affine_kernel(const float* __restrict__ P, const float* __restrict__ o_in,
                                                            long o_ns, const float* __restrict__ b3,
                                                            const float* __restrict__ l3, float* __restrict__ o_out,
                                                            float* __restrict__ z, long z_ns,
                                                            const float* __restrict__ scale,
                                                            const float* __restrict__ scale_shift,
                                                            float* __restrict__ logdet, int clamp_type, int C, int H,
                                                            int W) {
    __shared__ float sm[4];
    const int n = blockIdx.x, Ch = C >> 1, HW = H * W;
    float* z2 = z + n * z_ns + (long)Ch * HW;
    const float* Pn = P ? P + (long)n * 9 * C * HW : nullptr;
    const float* on = o_in ? o_in + n * o_ns : nullptr;
    float* oo = o_out ? o_out + (long)n * C * HW : nullptr;
    float acc = 0.f;
    for (int e = threadIdx.x; e < Ch * HW; e += 256) {
        const int j = e / HW, p = e - j * HW;
        float shift, s;
        if (Pn) {
            const int y = p / W, x = p - y * W;
            float a0 = 0.f, a1 = 0.f;
#pragma unroll
            for (int t = 0; t < 9; ++t) {
                const int yy = y + t / 3 - 1, xx = x + t % 3 - 1;
                if (yy >= 0 && yy < H && xx >= 0 && xx < W) {
                    const float* src = Pn + ((long)t * C + 2 * j) * HW + (long)yy * W + xx;
                    a0 += src[0];
                    a1 += src[HW];
                }
            }
            shift = (a0 + b3[2 * j]) * expf(3.f * l3[2 * j]);
            s = (a1 + b3[2 * j + 1]) * expf(3.f * l3[2 * j + 1]);
            oo[(long)(2 * j) * HW + p] = shift;
            oo[(long)(2 * j + 1) * HW + p] = s;
        } else {
            shift = on[(long)(2 * j) * HW + p];
            s = on[(long)(2 * j + 1) * HW + p];
        }
        float sc = 0.f, sh = 0.f;
        if (clamp_type == 0) {
            sc = scale[j];
            sh = scale_shift[j];
        }
        const float ls = clamp_ls(s, clamp_type, sc, sh);
        z2[e] = (z2[e] + shift) * expf(ls);
        acc += ls;
    }
    const float tot = block_sum_256(acc, sm);
    if (threadIdx.x == 0) logdet[n] = tot;
}
extern "C" int rfn_gather_affine_f32(const float* P, const float* o_in, long o_ns, const float* b3, const float* l3,
                                     float* o_out, float* z, long z_ns, const float* scale, const float* scale_shift,
                                     float* logdet, int clamp_type, int N, int C, int H, int W, rfn_stream_t stream) {
    RFN_CHECK_ARG(z && logdet && N >= 0 && C > 0 && (C % 2 == 0) && H > 0 && W > 0, -1);
    RFN_CHECK_ARG((P && b3 && l3 && o_out && !o_in) || (!P && o_in), -2);
    RFN_CHECK_ARG(clamp_type != 0 || (scale && scale_shift), -3);
    if (N == 0) return 0;
    hipLaunchKernelGGL(gather_affine_kernel, dim3(N), dim3(256), 0, (hipStream_t)stream, P, o_in, o_ns, b3, l3, o_out, z,
                       z_ns, scale, scale_shift, logdet, clamp_type, C, H, W);
    RFN_LAUNCH_CHECK();
    return 0;
}

// ---- forward shell between two consecutive Glow steps in ONE launch: the coupling tail of step k (tap gather / bias /
// exp(3 logs) of its Conv2dZeros output, affine coupling in place, per-frame log-det ACCUMULATED) and the ActNorm +
// InvConv head of step k+1 on the freshly coupled values (never re-read from HBM).  Either half may be absent: head only
// (first step of a level), tail only (last step).  Block = PB consecutive global pixels, thread = (pixel, channel group).
struct ShellFwdParams {
    float* z;  // tail: step k's post-InvConv tensor, coupled in place (-> its output); head only: the head's input
    long z_ns;
    const float* P;     // tail, tap-expanded conv3 output [N, 9C, H, W] (then b3, l3, o_out are used) ...
    const float* o_in;  // ... or the finished coupling-net output [N, C, H, W]
    long o_ns;
    const float* b3;
    const float* l3;
    float* o_out;
    const float* scale;
    const float* scale_shift;
    float* logdet;  // this launch's log-det partials [n_blocks][LDS_] (LDS_ = ld_slots), WRITTEN: slot s of block b =
                    // the block's sum for frame (b * PB) / HW + s; rfn_logdet_reduce_f32 adds them in a fixed order
    int clamp_type, tail;
    const float* bias;  // head: next step's ActNorm parameters and C x C matrix; znext = W ((v + bias) * exp(logs))
    const float* logs;
    const float* Wm;
    float* znext;
    long znext_ns;
    int head;
    int N, C, H, W, PB;
    int ld_const;  // head: also add the step's parameter-only log-det term HW * sum_c logs[c] to logdet[n]
    int ld_slots;  // frames a block can touch: (PB - 1) / HW + 2
};

struct __attribute__((aligned(16))) f32x4_s { float x, y, z, w; };
__global__ __launch_bounds__(256) void glow_shell_fwd_kernel(const ShellFwdParams q_) {
    extern __shared__ float lds[];  // head: [C][PB]
    __shared__ float red[256];      // log-det shares of the block's threads
    const ShellFwdParams& a = q_;
    const int PB = a.PB, C = a.C, Ch = C >> 1, HW = a.H * a.W;
    const int px = threadIdx.x & (PB - 1), ig = threadIdx.x / PB, NG = 256 / PB;
    const long q = (long)blockIdx.x * PB + px;
    const bool valid = q < (long)a.N * HW;
    int n = 0, p = 0;
    if (valid) {
        n = (int)(q / HW);
        p = (int)(q % HW);
    }
    float* src = a.z + n * a.z_ns + p;
    const int y = p / a.W, x = p - y * a.W;
    float lsacc = 0.f;
    for (int c = ig; c < C; c += NG) {
        float v = valid ? src[(long)c * HW] : 0.f;
        if (a.tail && c >= Ch && valid) {
            const int j = c - Ch;
            float shift, s;
            if (a.P) {
                const float* Pn = a.P + (long)n * 9 * C * HW;
                float a0 = 0.f, a1 = 0.f;
#pragma unroll
                for (int t = 0; t < 9; ++t) {
                    const int yy = y + t / 3 - 1, xx = x + t % 3 - 1;
                    if (yy >= 0 && yy < a.H && xx >= 0 && xx < a.W) {
                        const float* sp = Pn + ((long)t * C + 2 * j) * HW + (long)yy * a.W + xx;
                        a0 += sp[0];
                        a1 += sp[HW];
                    }
                }
                shift = (a0 + a.b3[2 * j]) * expf(3.f * a.l3[2 * j]);
                s = (a1 + a.b3[2 * j + 1]) * expf(3.f * a.l3[2 * j + 1]);
                float* oo = a.o_out + (long)n * C * HW + p;
                oo[(long)(2 * j) * HW] = shift;
                oo[(long)(2 * j + 1) * HW] = s;
            } else {
                shift = a.o_in[n * a.o_ns + (long)(2 * j) * HW + p];
                s = a.o_in[n * a.o_ns + (long)(2 * j + 1) * HW + p];
            }
            float sc = 0.f, sh = 0.f;
            if (a.clamp_type == 0) {
                sc = a.scale[j];
                sh = a.scale_shift[j];
            }
            const float ls = clamp_ls(s, a.clamp_type, sc, sh);
            v = (v + shift) * expf(ls);
            src[(long)c * HW] = v;
            lsacc += ls;
        }
        if (a.head) lds[c * PB + px] = (v + a.bias[c]) * expf(a.logs[c]);
    }
    if (a.logdet) {
        // log-det contribution of this block per frame, WITHOUT atomics (a forward pass must not depend on the order in
        // which workgroups finish): every thread's share meets in LDS and is added per frame in a fixed order
        float contrib = lsacc;
        if (a.ld_const && a.head && valid && p == 0 && ig == 0) {  // once per frame: HW * sum_c logs[c]
            float cs = 0.f;
            for (int c = 0; c < C; ++c) cs += a.logs[c];
            contrib += cs * (float)HW;
        }
        const int n0 = (int)(((long)blockIdx.x * PB) / HW);
        float* out = a.logdet + (long)blockIdx.x * a.ld_slots;
        if ((long)n0 * HW <= (long)blockIdx.x * PB && ((long)blockIdx.x * PB + PB) <= (long)(n0 + 1) * HW) {
            // the whole block lies inside frame n0 (the shallow levels): DPP butterfly per wave, four waves in order
            const float tot = wave_sum_dpp(contrib);
            if ((threadIdx.x & 63) == 0) red[threadIdx.x >> 6] = tot;
            __syncthreads();
            if (threadIdx.x == 0) out[0] = (red[0] + red[1]) + (red[2] + red[3]);
            if (threadIdx.x > 0 && threadIdx.x < a.ld_slots) out[threadIdx.x] = 0.f;
        } else if ((HW & (HW - 1)) == 0 && HW < PB && HW >= 4) {
            // power-of-two maps smaller than the block (the deep levels): the block starts on a frame boundary.  Groups of
            // G = min(HW, 64) consecutive threads lie in one frame: DPP sums inside the group, then thread f adds the
            // groups of frame f in increasing order
            const int G = HW < 64 ? HW : 64;
            float v = valid ? contrib : 0.f;
            if (G == 64) {
                v = wave_sum_dpp(v);
            } else {
#define RFN_DPP_ADD_(ctrl) v += __int_as_float(__builtin_amdgcn_update_dpp(0, __float_as_int(v), ctrl, 0xF, 0xF, true))
                RFN_DPP_ADD_(0xB1);                 // quad_perm [1,0,3,2]
                RFN_DPP_ADD_(0x4E);                 // quad_perm [2,3,0,1]
                if (G >= 8) RFN_DPP_ADD_(0x141);    // row_half_mirror
                if (G >= 16) RFN_DPP_ADD_(0x140);   // row_mirror
#undef RFN_DPP_ADD_
                if (G >= 32) v += __shfl_xor(v, 16, 64);
            }
            if ((threadIdx.x & (G - 1)) == 0) red[threadIdx.x / G] = v;
            __syncthreads();
            const int F = PB / HW, ngrp = 256 / G, gpf = HW / G;   // groups per (frame, channel group)
            if (threadIdx.x < F) {
                float tot = 0.f;
                for (int k = 0; k < ngrp; ++k)
                    if (((k * G) & (PB - 1)) / HW == (int)threadIdx.x) tot += red[k];
                (void)gpf;
                out[threadIdx.x] = tot;
            } else if (threadIdx.x < a.ld_slots) {
                out[threadIdx.x] = 0.f;
            }
        } else {
            red[threadIdx.x] = valid ? contrib : 0.f;
            __syncthreads();
            if (threadIdx.x < a.ld_slots) {   // slot s: frame n0 + s; its threads are the pixels px with that frame
                const int fs = n0 + threadIdx.x;
                float tot = 0.f;
                for (int t = 0; t < 256; ++t) {
                    const long qq = (long)blockIdx.x * PB + (t & (PB - 1));
                    if ((int)(qq / HW) == fs) tot += red[t];
                }
                out[threadIdx.x] = tot;
            }
        }
    }
    if (!a.head) return;
    if (C >= 16 && C % NG == 0 && (C / NG) % 2 == 0) {
        // deep levels: the C x C matrix goes through LDS, transposed (Wt[j][i]); a thread forms OPT = C / NG CONSECUTIVE
        // outputs, four (or two) at a time: per input channel one read of y and one broadcast read of the weights.  (Read
        // from global memory per (i, j) -- with PB = 32 a wave spans two channel groups, so not even through the scalar
        // cache -- the 64 x 64 product of the 2x2 level was 12 of the launch's 18 us.)
        float* Wt = lds + C * PB;
        for (int e = threadIdx.x; e < C * C; e += 256) Wt[e] = a.Wm[(e % C) * C + e / C];   // Wt[j][i] = W[i][j]
        __syncthreads();
        if (valid) {
            float* dst = a.znext + n * a.znext_ns + p;
            const int OPT = C / NG;
            if (OPT % 4 == 0 && C % 4 == 0) {
                for (int k0 = 0; k0 < OPT; k0 += 4) {
                    const int i0 = ig * OPT + k0;
                    float acc[4] = {0.f, 0.f, 0.f, 0.f};
#pragma unroll 4
                    for (int j = 0; j < C; ++j) {
                        const float yv = lds[j * PB + px];
                        const f32x4_s w4 = *reinterpret_cast<const f32x4_s*>(Wt + j * C + i0);
                        acc[0] = fmaf(w4.x, yv, acc[0]);
                        acc[1] = fmaf(w4.y, yv, acc[1]);
                        acc[2] = fmaf(w4.z, yv, acc[2]);
                        acc[3] = fmaf(w4.w, yv, acc[3]);
                    }
#pragma unroll
                    for (int k = 0; k < 4; ++k) dst[(long)(i0 + k) * HW] = acc[k];
                }
            } else {
                for (int k0 = 0; k0 < OPT; k0 += 2) {
                    const int i0 = ig * OPT + k0;
                    float acc0 = 0.f, acc1 = 0.f;
#pragma unroll 4
                    for (int j = 0; j < C; ++j) {
                        const float yv = lds[j * PB + px];
                        acc0 = fmaf(Wt[j * C + i0], yv, acc0);
                        acc1 = fmaf(Wt[j * C + i0 + 1], yv, acc1);
                    }
                    dst[(long)i0 * HW] = acc0;
                    dst[(long)(i0 + 1) * HW] = acc1;
                }
            }
        }
        return;
    }
    __syncthreads();
    if (valid) {
        float* dst = a.znext + n * a.znext_ns + p;
        for (int i = ig; i < C; i += NG) {
            float acc = 0.f;
            const float* wr = a.Wm + (long)i * C;
#pragma unroll 8
            for (int j = 0; j < C; ++j) acc = fmaf(wr[j], lds[j * PB + px], acc);
            dst[(long)i * HW] = acc;
        }
    }
}

// pixels per block of glow_shell_fwd_kernel for N frames of HW pixels and C channels (also fixes the layout of its log-det
// partials: rfn_glow_shell_fwd_ld_floats)
static int shell_fwd_pb(int N, int C, int HW) {
    int PB = shell_pb(C);
    const long tot = (long)N * HW;
    while (PB > 32 && tot / PB < 256) PB >>= 1;
    return PB;
}
/* floats of ONE launch's log-det partials ([blocks][slots]) */
extern "C" long rfn_glow_shell_fwd_ld_floats(int N, int C, int H, int W) {
    if (N <= 0 || C <= 0 || H <= 0 || W <= 0) return 0;
    const int HW = H * W, PB = shell_fwd_pb(N, C, HW);
    return (((long)N * HW + PB - 1) / PB) * ((PB - 1) / HW + 2);
}
// logdet[n] (+)= sum over launches k (ascending) and over the blocks b that touch frame n (ascending) of part[k][b][n - n0(b)]
__global__ __launch_bounds__(256) void logdet_reduce_kernel(const float* __restrict__ part, long per_launch, int n_launch,
                                                            int N, int HW, int PB, int slots, float* __restrict__ logdet,
                                                            int accumulate) {
    const int n = blockIdx.x * 256 + threadIdx.x;
    if (n >= N) return;
    const long b_lo = ((long)n * HW) / PB, b_hi = ((long)(n + 1) * HW - 1) / PB;
    float acc = accumulate ? logdet[n] : 0.f;
    for (int k = 0; k < n_launch; ++k) {
        const float* pk = part + (long)k * per_launch;
        for (long b = b_lo; b <= b_hi; ++b) acc += pk[b * slots + (n - (int)((b * PB) / HW))];
    }
    logdet[n] = acc;
}
/* second pass of the level's log-det: part = n_launch consecutive partial buffers of rfn_glow_shell_fwd_f32 (each
 * rfn_glow_shell_fwd_ld_floats(N, C, H, W) floats) -> logdet [N], written (accumulate = 0) or added to */
extern "C" int rfn_logdet_reduce_f32(const float* part, int n_launch, float* logdet, int accumulate, int N, int C, int H,
                                     int W, rfn_stream_t stream) {
    RFN_CHECK_ARG(part && logdet && n_launch >= 1 && N >= 0 && C > 0 && H > 0 && W > 0, -1);
    if (N == 0) return 0;
    const int HW = H * W, PB = shell_fwd_pb(N, C, HW);
    hipLaunchKernelGGL(logdet_reduce_kernel, dim3((N + 255) / 256), dim3(256), 0, (hipStream_t)stream, part,
                       rfn_glow_shell_fwd_ld_floats(N, C, H, W), n_launch, N, HW, PB, (PB - 1) / HW + 2, logdet, accumulate);
    RFN_LAUNCH_CHECK();
    return 0;
}

extern "C" int rfn_glow_shell_fwd_f32(float* z, long z_ns, const float* P, const float* o_in, long o_ns,
                                      const float* b3, const float* l3, float* o_out, const float* scale,
                                      const float* scale_shift, float* logdet, int clamp_type, const float* bias,
                                      const float* logs, const float* Wm, float* znext, long znext_ns, int ld_const,
                                      int N, int C, int H, int W, rfn_stream_t stream) {
    RFN_CHECK_ARG(z && N >= 0 && C > 0 && (C % 2 == 0) && H > 0 && W > 0, -1);
    const int tail = (P || o_in) ? 1 : 0, head = Wm ? 1 : 0;
    RFN_CHECK_ARG(tail || head, -2);
    RFN_CHECK_ARG(!tail || (logdet && ((P && b3 && l3 && o_out && !o_in) || (!P && o_in))), -3);
    RFN_CHECK_ARG(!tail || clamp_type != 0 || (scale && scale_shift), -4);
    RFN_CHECK_ARG(!head || (bias && logs && znext), -5);
    RFN_CHECK_ARG(!ld_const || (head && logdet), -7);
    if (N == 0) return 0;
    const int HW = H * W;
    const int PB = shell_fwd_pb(N, C, HW);
    const long tot = (long)N * HW;
    size_t lds = head ? (size_t)C * PB * 4 + (C >= 16 ? (size_t)C * C * 4 : 0) : 0;
    if (lds > 160 * 1024) {
        rfn_set_error("glow_shell_fwd: C=%d too large for the LDS-staged kernel", C);
        return -6;
    }
    if (lds > 65536)
        (void)hipFuncSetAttribute((const void*)glow_shell_fwd_kernel, hipFuncAttributeMaxDynamicSharedMemorySize,
                                  (int)lds);
    ShellFwdParams a = {z, z_ns, P, o_in, o_ns, b3, l3, o_out, scale, scale_shift, logdet, clamp_type, tail,
                        bias, logs, Wm, znext, znext_ns, head, N, C, H, W, PB, ld_const, (PB - 1) / HW + 2};
    hipLaunchKernelGGL(glow_shell_fwd_kernel, dim3((unsigned)((tot + PB - 1) / PB)), dim3(256), lds, (hipStream_t)stream,
                       a);
    RFN_LAUNCH_CHECK();
    return 0;
}

// ---- fused backward shell head of a Glow step: affine coupling backward + Conv2dZeros epilogue backward.
// grid (X, C/2): block (bx, j) sweeps channel pair j over a strided share of the N*HW (frame, pixel) pairs.
//   gz[:, :C/2]      = gout[:, :C/2]                       (z1 passes through; the conv1 data gradient is added later)
//   gz[:, C/2 + j]   = gout * exp(ls)
//   go[2j]  = gz2 (d/dshift),  go[2j+1] = (gout * zout + glogdet[n]) * dls/ds
//   gpre[c] = go[c] * exp(3 l3[c])                         (gradient at the conv output, what wgrad3 / dgrad3 consume)
//   gscale, gshift (realnvp clamp), gb3[c] += sum gpre[c], gl3[c] += 3 sum go[c] * o[c]
__global__ __launch_bounds__(256) void affine_zeros_bwd_kernel(
    const float* __restrict__ zout, long zout_ns, const float* __restrict__ o, long o_ns,
    const float* __restrict__ gout, long gout_ns, const float* __restrict__ glogdet, const float* __restrict__ scale,
    const float* __restrict__ scale_shift, const float* __restrict__ l3, float* __restrict__ gz, long gz_ns,
    float* __restrict__ gpre, long gpre_ns, float* __restrict__ gscale, float* __restrict__ gscale_shift,
    float* __restrict__ gb3, float* __restrict__ gl3, int clamp_type, int N, int C, int HW) {
    __shared__ float sm[4];
    const int Ch = C >> 1, j = blockIdx.y;
    float sc = 0.f, sh = 0.f;
    if (clamp_type == 0) {
        sc = scale[j];
        sh = scale_shift[j];
    }
    const float e0 = expf(3.f * l3[2 * j]), e1 = expf(3.f * l3[2 * j + 1]);
    float a_sc = 0.f, a_sh = 0.f, a_b0 = 0.f, a_b1 = 0.f, a_l0 = 0.f, a_l1 = 0.f;
    const long total = (long)N * HW;
    for (long q = (long)blockIdx.x * 256 + threadIdx.x; q < total; q += (long)gridDim.x * 256) {
        const long n = q / HW;
        const int p = (int)(q - n * HW);
        const float o0 = o[n * o_ns + (long)(2 * j) * HW + p];
        const float s = o[n * o_ns + (long)(2 * j + 1) * HW + p];
        const float ls = clamp_ls(s, clamp_type, sc, sh);
        const float e = expf(ls);
        const float g = gout[n * gout_ns + (long)(Ch + j) * HW + p];
        const float gls = g * zout[n * zout_ns + (long)(Ch + j) * HW + p] + (glogdet ? glogdet[n] : 0.f);
        const float gzv = g * e;
        gz[n * gz_ns + (long)j * HW + p] = gout[n * gout_ns + (long)j * HW + p];
        gz[n * gz_ns + (long)(Ch + j) * HW + p] = gzv;
        const float go1 = gls * clamp_ls_grad(s, clamp_type, sc);
        const float u0 = gzv * e0, u1 = go1 * e1;
        gpre[n * gpre_ns + (long)(2 * j) * HW + p] = u0;
        gpre[n * gpre_ns + (long)(2 * j + 1) * HW + p] = u1;
        a_b0 += u0;
        a_b1 += u1;
        a_l0 += gzv * o0;
        a_l1 += go1 * s;
        if (clamp_type == 0) {
            a_sc += gls * tanhf(s);
            a_sh += gls;
        }
    }
    const float t_b0 = block_sum_256(a_b0, sm), t_b1 = block_sum_256(a_b1, sm);
    const float t_l0 = block_sum_256(a_l0, sm), t_l1 = block_sum_256(a_l1, sm);
    float t_sc = 0.f, t_sh = 0.f;
    if (clamp_type == 0) {
        t_sc = block_sum_256(a_sc, sm);
        t_sh = block_sum_256(a_sh, sm);
    }
    if (threadIdx.x == 0) {
        atomicAdd(&gb3[2 * j], t_b0);
        atomicAdd(&gb3[2 * j + 1], t_b1);
        atomicAdd(&gl3[2 * j], 3.f * t_l0);
        atomicAdd(&gl3[2 * j + 1], 3.f * t_l1);
        if (clamp_type == 0) {
            atomicAdd(&gscale[j], t_sc);
            atomicAdd(&gscale_shift[j], t_sh);
        }
    }
}
extern "C" int rfn_affine_zeros_bwd_f32(const float* zout, long zout_ns, const float* o, long o_ns, const float* gout,
                                        long gout_ns, const float* glogdet, const float* scale, const float* scale_shift,
                                        const float* l3, float* gz, long gz_ns, float* gpre, long gpre_ns, float* gscale,
                                        float* gscale_shift, float* gb3, float* gl3, int clamp_type, int N, int C,
                                        int HW, rfn_stream_t stream) {
    RFN_CHECK_ARG(zout && o && gout && l3 && gz && gpre && gb3 && gl3 && N >= 0 && C > 0 && (C % 2 == 0) && HW > 0, -1);
    RFN_CHECK_ARG(clamp_type != 0 || (scale && scale_shift && gscale && gscale_shift), -2);
    if (N == 0) return 0;
    const int Ch = C / 2;
    long tot = (long)N * HW;
    int gx_ = (int)((tot + 255) / 256);
    int cap = 512 / Ch < 1 ? 1 : 512 / Ch;
    if (gx_ > cap) gx_ = cap;
    hipLaunchKernelGGL(affine_zeros_bwd_kernel, dim3(gx_, Ch), dim3(256), 0, (hipStream_t)stream, zout, zout_ns, o, o_ns,
                       gout, gout_ns, glogdet, scale, scale_shift, l3, gz, gz_ns, gpre, gpre_ns, gscale, gscale_shift,
                       gb3, gl3, clamp_type, N, C, HW);
    RFN_LAUNCH_CHECK();
    return 0;
}

// ------------------------------------------------------------------------------------------------ gaussian log-prob
#define RFN_HALF_LOG_2PI 0.91893853320467274178f
__device__ __forceinline__ float softplusf_(float x) { return x > 20.f ? x : log1pf(expf(x)); }  // torch threshold=20

__device__ __forceinline__ void gauss_params(const float* on, int layout, int c, int Cz, int HW, int p, float& mean,
                                             float& raw) {
    if (layout == 0) {
        mean = on[(long)(2 * c) * HW + p];
        raw = on[(long)(2 * c + 1) * HW + p];
    } else {
        mean = on[(long)c * HW + p];
        raw = on[(long)(Cz + c) * HW + p];
    }
}

__global__ __launch_bounds__(256) void gauss_logp_kernel(const float* __restrict__ z, long z_ns,
                                                         const float* __restrict__ o, long o_ns,
                                                         float* __restrict__ logp, int layout, int std_mode, int Cz,
                                                         int HW) {
    __shared__ float sm[4];
    const int n = blockIdx.x;
    const float* zn = z + n * z_ns;
    const float* on = o + n * o_ns;
    float acc = 0.f;
    for (int e = threadIdx.x; e < Cz * HW; e += 256) {
        int c = e / HW, p = e - c * HW;
        float mean, raw;
        gauss_params(on, layout, c, Cz, HW, p, mean, raw);
        float logstd, std;
        if (std_mode == 0) {
            std = softplusf_(raw) + 1e-8f;
            logstd = logf(std);
        } else {
            std = expf(raw);
            logstd = raw;
        }
        float d = zn[e] - mean;
        acc += -(d * d) / (2.f * std * std) - logstd - RFN_HALF_LOG_2PI;
    }
    float tot = block_sum_256(acc, sm);
    if (threadIdx.x == 0) logp[n] += tot;
}
extern "C" int rfn_gauss_logp_f32(const float* z, long z_ns, const float* o, long o_ns, float* logp, int layout,
                                  int std_mode, int N, int Cz, int HW, rfn_stream_t stream) {
    RFN_CHECK_ARG(z && o && logp && N >= 0 && Cz > 0 && HW > 0, -1);
    if (N == 0) return 0;
    hipLaunchKernelGGL(gauss_logp_kernel, dim3(N), dim3(256), 0, (hipStream_t)stream, z, z_ns, o, o_ns, logp, layout,
                       std_mode, Cz, HW);
    RFN_LAUNCH_CHECK();
    return 0;
}

__global__ __launch_bounds__(256) void gauss_logp_bwd_kernel(const float* __restrict__ z, long z_ns,
                                                             const float* __restrict__ o, long o_ns,
                                                             const float* __restrict__ glogp, float* __restrict__ gz,
                                                             long gz_ns, float* __restrict__ go, long go_ns, int layout,
                                                             int std_mode, int Cz, int HW) {
    const int n = blockIdx.x;
    const float* zn = z + n * z_ns;
    const float* on = o + n * o_ns;
    float* gzn = gz + n * gz_ns;
    float* gon = go + n * go_ns;
    const float g = glogp[n];
    for (int e = threadIdx.x; e < Cz * HW; e += 256) {
        int c = e / HW, p = e - c * HW;
        float mean, raw;
        gauss_params(on, layout, c, Cz, HW, p, mean, raw);
        float std, dstd_draw;
        if (std_mode == 0) {
            std = softplusf_(raw) + 1e-8f;
            dstd_draw = raw > 20.f ? 1.f : 1.f / (1.f + expf(-raw));
        } else {
            std = expf(raw);
            dstd_draw = std;
        }
        float d = zn[e] - mean;
        float inv = 1.f / (std * std);
        float gzv = -d * inv * g;                        // d logp / d z
        float gstd = (d * d * inv / std - 1.f / std) * g;  // d logp / d std
        gzn[e] = gzv;
        if (layout == 0) {
            gon[(long)(2 * c) * HW + p] = -gzv;
            gon[(long)(2 * c + 1) * HW + p] = gstd * dstd_draw;
        } else {
            gon[(long)c * HW + p] = -gzv;
            gon[(long)(Cz + c) * HW + p] = gstd * dstd_draw;
        }
    }
}
extern "C" int rfn_gauss_logp_bwd_f32(const float* z, long z_ns, const float* o, long o_ns, const float* glogp,
                                      float* gz, long gz_ns, float* go, long go_ns, int layout, int std_mode, int N,
                                      int Cz, int HW, rfn_stream_t stream) {
    RFN_CHECK_ARG(z && o && glogp && gz && go && N >= 0 && Cz > 0 && HW > 0, -1);
    if (N == 0) return 0;
    hipLaunchKernelGGL(gauss_logp_bwd_kernel, dim3(N), dim3(256), 0, (hipStream_t)stream, z, z_ns, o, o_ns, glogp, gz,
                       gz_ns, go, go_ns, layout, std_mode, Cz, HW);
    RFN_LAUNCH_CHECK();
    return 0;
}

__global__ __launch_bounds__(256) void gauss_sample_kernel(const float* __restrict__ o, long o_ns,
                                                           const float* __restrict__ eps, float* __restrict__ z,
                                                           long z_ns, float temperature, int layout, int std_mode,
                                                           int Cz, int HW) {
    const int n = blockIdx.x;
    const float* on = o + n * o_ns;
    for (int e = threadIdx.x; e < Cz * HW; e += 256) {
        int c = e / HW, p = e - c * HW;
        float mean, raw;
        gauss_params(on, layout, c, Cz, HW, p, mean, raw);
        float std = std_mode == 0 ? softplusf_(raw) + 1e-8f : expf(raw);
        z[n * z_ns + e] = mean + std * temperature * eps[(long)n * Cz * HW + e];
    }
}
extern "C" int rfn_gauss_sample_f32(const float* o, long o_ns, const float* eps, float* z, long z_ns, float temperature,
                                    int layout, int std_mode, int N, int Cz, int HW, rfn_stream_t stream) {
    RFN_CHECK_ARG(o && eps && z && N >= 0 && Cz > 0 && HW > 0, -1);
    if (N == 0) return 0;
    hipLaunchKernelGGL(gauss_sample_kernel, dim3(N), dim3(256), 0, (hipStream_t)stream, o, o_ns, eps, z, z_ns,
                       temperature, layout, std_mode, Cz, HW);
    RFN_LAUNCH_CHECK();
    return 0;
}

// ------------------------------------------------------------------------------------------------ conv epilogue bwd
// grid (C, S): block (c, s) sweeps frames n = s, s+S, ... of channel c.
__global__ __launch_bounds__(256) void conv_epilogue_bwd_kernel(const float* __restrict__ y, long y_ns,
                                                                const float* __restrict__ gy, long gy_ns,
                                                                float* __restrict__ gu, long gu_ns,
                                                                const float* __restrict__ logs, float* __restrict__ gb,
                                                                float* __restrict__ gl, int N, int C, int HW,
                                                                int ep_mode, int act) {
    __shared__ float sm[4];
    const int c = blockIdx.x;
    float e = 1.f;
    if (ep_mode == 1) e = expf(logs[c]);
    if (ep_mode == 2) e = expf(3.f * logs[c]);
    float a_b = 0.f, a_l = 0.f;
    // (frame, pixel) flattened: at the deep levels a frame has 4 or 16 pixels and a per-frame loop idles the block
    const long total = (long)N * HW;
    for (long i = (long)blockIdx.y * 256 + threadIdx.x; i < total; i += (long)gridDim.y * 256) {
        const long n = i / HW;
        const int p = (int)(i - n * HW);
        const long off = (long)c * HW + p;
        const float g = gy[n * gy_ns + off];
        const float yv = y ? y[n * y_ns + off] : 0.f;
        float slope = 1.f;
        if (ep_mode == 1) {
            if (act == 1) slope = yv > 0.f ? 1.f : 0.f;
            if (act == 2) slope = yv > 0.f ? 1.f : 0.2f;
        }
        const float u = g * slope * e;
        gu[n * gu_ns + off] = u;
        a_b += u;
        a_l += g * yv;
    }
    float tb = block_sum_256(a_b, sm);
    float tl = block_sum_256(a_l, sm);
    if (threadIdx.x == 0) {
        if (gb) atomicAdd(&gb[c], tb);
        if (gl && ep_mode != 3) atomicAdd(&gl[c], ep_mode == 2 ? 3.f * tl : tl);
    }
}
extern "C" int rfn_conv_epilogue_bwd_f32(const float* y, long y_ns, const float* gy, long gy_ns, float* gu, long gu_ns,
                                         const float* logs, float* gb, float* gl, int N, int C, int HW, int ep_mode,
                                         int act, rfn_stream_t stream) {
    RFN_CHECK_ARG(gy && gu && N >= 0 && C > 0 && HW > 0, -1);
    RFN_CHECK_ARG(ep_mode == 3 || (y && logs), -2);
    if (N == 0) return 0;
    int S = 2048 / C;
    if (S < 1) S = 1;
    const long per = ((long)N * HW + 1023) / 1024;  // at least ~4 elements per thread
    if (S > per) S = (int)per;
    if (S < 1) S = 1;
    hipLaunchKernelGGL(conv_epilogue_bwd_kernel, dim3(C, S), dim3(256), 0, (hipStream_t)stream, y, y_ns, gy, gy_ns, gu,
                       gu_ns, logs, gb, gl, N, C, HW, ep_mode, act);
    RFN_LAUNCH_CHECK();
    return 0;
}

// ------------------------------------------------------------------------------------------------ tap gather / scatter
// A 3x3 convolution with very few output channels (Conv2dZeros at the shallow flow levels: 256 -> 4 / 8) wastes a
// 32-row MFMA tile.  It is computed instead as a 1x1 convolution to 9*C "tap-expanded" channels
//   P[n][tap*C + co][p] = Σ_ci W[co][ci][tap] x[n][ci][p]
// followed by this shift-and-add:  o[n][co][y][x] = (Σ_tap P[n][tap*C+co][y+dy-1][x+dx-1] + b[co]) * exp(3 l[co]).
// tap_scatter is its adjoint data movement for the weight gradient:
//   Gs[n][tap*C + co][y'][x'] = g[n][co][y'-dy+1][x'-dx+1]   (0 outside), so gW[co][ci][tap] = Σ Gs[tap*C+co] · x[ci].
__global__ void tap_gather_kernel(const float* __restrict__ P, const float* __restrict__ b, const float* __restrict__ l,
                                  float* __restrict__ o, int N, int C, int H, int W) {
    const long HW = (long)H * W, total = (long)N * C * HW;
    for (long idx = (long)blockIdx.x * blockDim.x + threadIdx.x; idx < total; idx += (long)gridDim.x * blockDim.x) {
        const int x = (int)(idx % W);
        long r = idx / W;
        const int y = (int)(r % H);
        r /= H;
        const int co = (int)(r % C);
        const long n = r / C;
        const float* Pn = P + n * 9 * C * HW;
        float a = 0.f;
#pragma unroll
        for (int t = 0; t < 9; ++t) {
            const int yy = y + t / 3 - 1, xx = x + t % 3 - 1;
            if (yy >= 0 && yy < H && xx >= 0 && xx < W) a += Pn[((long)t * C + co) * HW + (long)yy * W + xx];
        }
        o[idx] = b ? (a + b[co]) * expf(3.f * l[co]) : a;
    }
}
__global__ void tap_scatter_kernel(const float* __restrict__ g, float* __restrict__ Gs, int N, int C, int H, int W) {
    const long HW = (long)H * W, total = (long)N * 9 * C * HW;
    for (long idx = (long)blockIdx.x * blockDim.x + threadIdx.x; idx < total; idx += (long)gridDim.x * blockDim.x) {
        const int x = (int)(idx % W);
        long r = idx / W;
        const int y = (int)(r % H);
        r /= H;
        const int tc = (int)(r % (9 * C));
        const long n = r / (9 * C);
        const int t = tc / C, co = tc - t * C;
        const int yy = y - (t / 3 - 1), xx = x - (t % 3 - 1);
        float v = 0.f;
        if (yy >= 0 && yy < H && xx >= 0 && xx < W) v = g[(n * C + co) * HW + (long)yy * W + xx];
        Gs[idx] = v;
    }
}
// W % 4 == 0: one thread = 4 consecutive pixels of one (frame, tap*C+co, y) row, one 16-byte store, 32-bit index math
__global__ void tap_scatter_v4_kernel(const float* __restrict__ g, float* __restrict__ Gs, int N, int C, int H, int W) {
    const int HW4 = H * W / 4, W4 = W / 4;
    const long HW = (long)H * W, total = (long)N * 9 * C * HW4;
    for (long idx = (long)blockIdx.x * blockDim.x + threadIdx.x; idx < total; idx += (long)gridDim.x * blockDim.x) {
        const int q = (int)(idx % HW4);
        const long r = idx / HW4;
        const int tc = (int)(r % (9 * C));
        const long n = r / (9 * C);
        const int t = tc / C, co = tc - t * C;
        const int y = q / W4, x0 = (q - y * W4) * 4;
        const int yy = y - (t / 3 - 1), dx = -(t % 3 - 1);  // source = (yy, x + dx)
        float4 v = {0.f, 0.f, 0.f, 0.f};
        if (yy >= 0 && yy < H) {
            const float* src = g + (n * C + co) * HW + (long)yy * W;
            const float4 c = *reinterpret_cast<const float4*>(src + x0);
            if (dx == 0) {
                v = c;
            } else if (dx < 0) {
                v.x = x0 > 0 ? src[x0 - 1] : 0.f;
                v.y = c.x; v.z = c.y; v.w = c.z;
            } else {
                v.x = c.y; v.y = c.z; v.z = c.w;
                v.w = x0 + 4 < W ? src[x0 + 4] : 0.f;
            }
        }
        *reinterpret_cast<float4*>(Gs + idx * 4) = v;
    }
}
extern "C" int rfn_tap_gather_f32(const float* P, const float* bias, const float* logs, float* o, int N, int C, int H,
                                  int W, rfn_stream_t stream) {
    RFN_CHECK_ARG(P && o && N >= 0 && C > 0 && H > 0 && W > 0 && ((bias && logs) || (!bias && !logs)), -1);
    if (N == 0) return 0;
    long tot = (long)N * C * H * W;
    int grid = (int)((tot + 255) / 256 < 4096 ? (tot + 255) / 256 : 4096);
    hipLaunchKernelGGL(tap_gather_kernel, dim3(grid), dim3(256), 0, (hipStream_t)stream, P, bias, logs, o, N, C, H, W);
    RFN_LAUNCH_CHECK();
    return 0;
}
extern "C" int rfn_tap_scatter_f32(const float* g, float* Gs, int N, int C, int H, int W, rfn_stream_t stream) {
    RFN_CHECK_ARG(g && Gs && N >= 0 && C > 0 && H > 0 && W > 0, -1);
    if (N == 0) return 0;
    if (W % 4 == 0 && (((uintptr_t)g | (uintptr_t)Gs) & 15) == 0) {
        long tot4 = (long)N * 9 * C * (H * W / 4);
        int grid4 = (int)((tot4 + 255) / 256 < 16384 ? (tot4 + 255) / 256 : 16384);
        hipLaunchKernelGGL(tap_scatter_v4_kernel, dim3(grid4), dim3(256), 0, (hipStream_t)stream, g, Gs, N, C, H, W);
        RFN_LAUNCH_CHECK();
        return 0;
    }
    long tot = (long)N * 9 * C * H * W;
    int grid = (int)((tot + 255) / 256 < 8192 ? (tot + 255) / 256 : 8192);
    hipLaunchKernelGGL(tap_scatter_kernel, dim3(grid), dim3(256), 0, (hipStream_t)stream, g, Gs, N, C, H, W);
    RFN_LAUNCH_CHECK();
    return 0;
}


// ------------------------------------------------------------------------------------------------ per-step BatchNorm
// The reference runs its VGG extractor / upscaler once per timestep (RFN_new.py:126-128,191-194), so every BatchNorm
// sees the B samples of ONE step.  Time-batched here: x is [S*B, C, HW] step-major and the statistics are per (step,
// channel) over (B, HW).  Four kernels replace the permute-copy / batch_norm / permute-copy / activation chain (and its
// backward): statistics (two-pass, second pass from L2), apply + activation, backward reduction, backward apply.
// act: 0 none, 1 relu, 2 leaky_relu(slope), 3 tanh.
__device__ __forceinline__ float stepbn_act(float u, int act, float slope) {
    if (act == 1) return u > 0.f ? u : 0.f;
    if (act == 2) return u > 0.f ? u : slope * u;
    if (act == 3) return tanhf(u);
    return u;
}
__device__ __forceinline__ float stepbn_dact(float u, int act, float slope) {  // from the activation's INPUT u = xhat*g + b
    if (act == 1) return u > 0.f ? 1.f : 0.f;                                  // (recomputed: saves a pass over y)
    if (act == 2) return u > 0.f ? 1.f : slope;
    if (act == 3) {
        const float t = tanhf(u);
        return 1.f - t * t;
    }
    return 1.f;
}
// block (sc, j) = frames j, j+gridDim.y, ... of one (step, channel): shifted sums s1 = sum(x-K), s2 = sum((x-K)^2) with
// K = the pair's first element (keeps the one-pass variance well conditioned).  Every block stores ITS partial pair at
// acc[j][sc][2] (no atomics, nothing to zero, deterministic); the consumers add the gridDim.y <= STEPBN_MAX_SPLIT partials.
constexpr int STEPBN_MAX_SPLIT = 16;
static int stepbn_split(int S, int B, int C) {  // workgroups per (step, channel): enough blocks for the chip, <= B
    int ny = 2048 / (S * C);
    if (ny > STEPBN_MAX_SPLIT) ny = STEPBN_MAX_SPLIT;
    if (ny > B) ny = B;
    return ny < 1 ? 1 : ny;
}
__global__ __launch_bounds__(256) void stepbn_stats_kernel(const float* __restrict__ x, float* __restrict__ acc, int B,
                                                           int C, int HW) {
    __shared__ float sm[4];
    const int s = blockIdx.x / C, c = blockIdx.x - s * C;
    const float* base = x + ((long)s * B * C + c) * HW;
    const long fs = (long)C * HW;
    const float K = base[0];
    float a = 0.f, v = 0.f;
    if ((HW & 3) == 0 && (((uintptr_t)x) & 15) == 0) {  // 16-byte loads (frames and channel planes stay aligned)
        // (frame, pixel quad) pairs of this block flattened over the threads: small maps keep all 256 lanes busy
        const int HW4 = HW >> 2, nb = (B - (int)blockIdx.y + (int)gridDim.y - 1) / (int)gridDim.y;
        for (int idx = threadIdx.x; idx < nb * HW4; idx += 256) {
            const int bi = idx / HW4, p = idx - bi * HW4;
            const float4 q = reinterpret_cast<const float4*>(base + (blockIdx.y + (long)bi * gridDim.y) * fs)[p];
            const float d0 = q.x - K, d1 = q.y - K, d2 = q.z - K, d3 = q.w - K;
            a += (d0 + d1) + (d2 + d3);
            v = fmaf(d0, d0, fmaf(d1, d1, fmaf(d2, d2, fmaf(d3, d3, v))));
        }
    } else {
        for (int b = blockIdx.y; b < B; b += gridDim.y) {
            const float* row = base + b * fs;
            for (int p = threadIdx.x; p < HW; p += 256) {
                const float d = row[p] - K;
                a += d;
                v = fmaf(d, d, v);
            }
        }
    }
    const float ta = block_sum_256(a, sm);
    const float tv = block_sum_256(v, sm);
    if (threadIdx.x == 0) {
        float* dst = acc + ((long)blockIdx.y * gridDim.x + blockIdx.x) * 2;
        dst[0] = ta;
        dst[1] = tv;
    }
}
// elementwise kernels: VEC = 4 consecutive pixels per thread when HW % 4 == 0 (16-byte accesses); 32-bit index math
// (the host checks total < 2^31).  The statistics come straight from the stats kernel's partial sums (no finalize launch);
// ONE extra workgroup (the first: dispatched first, so no streaming block is delayed by it) writes mean / var for the
// backward and applies the S running-statistics updates of the step-wise calls in closed form.
struct StepBnFused {
    const float* acc;      // [ny][S*C][2] partial shifted sums of stepbn_stats_kernel
    float* mean_out;
    float* var_out;
    float* run_mean;       // optional [C]: r <- decay r + sum_s coef[s] mean[s]   (and var with coef_u)
    float* run_var;
    const float* coef;
    const float* coef_u;
    float decay;
    long long* nbt;        // optional: += S
    int S, ny;
};
__device__ __forceinline__ void stepbn_moments(const float* __restrict__ x, const float* __restrict__ acc, int sc, int s,
                                               int c, int B, int C, int HW, int ny, int SC, float& m, float& v) {
    if (!acc) return;   // statistics given by the caller (m, v preloaded): synchronised BatchNorm
    const float K = x[((long)s * B * C + c) * HW];
    const float n = (float)B * HW;
    float s1 = 0.f, s2 = 0.f;
    for (int y = 0; y < ny; ++y) {
        s1 += acc[((long)y * SC + sc) * 2];
        s2 += acc[((long)y * SC + sc) * 2 + 1];
    }
    const float m1 = s1 / n;
    m = K + m1;
    const float vv = s2 / n - m1 * m1;
    v = vv > 0.f ? vv : 0.f;
}
template <int VEC>
__global__ void stepbn_apply_kernel(const float* __restrict__ x, const float* __restrict__ gamma,
                                    const float* __restrict__ beta, float* __restrict__ y, unsigned total, int B, int C,
                                    int HW, float eps, int act, float slope, StepBnFused f) {
    const int SC = f.S * C;
    const unsigned nblk = gridDim.x - 1, bid = blockIdx.x - 1;
    if (blockIdx.x == 0) {
        if (!f.acc) return;   // given statistics: nothing to finalise
        for (int sc = threadIdx.x; sc < SC; sc += blockDim.x) {
            float m, v;
            stepbn_moments(x, f.acc, sc, sc / C, sc % C, B, C, HW, f.ny, SC, m, v);
            f.mean_out[sc] = m;
            f.var_out[sc] = v;
        }
        if (f.run_mean) {
            for (int c = threadIdx.x; c < C; c += blockDim.x) {  // loads only inside the loop: they pipeline
                float em = 0.f, ev = 0.f;
#pragma unroll 4
                for (int s = 0; s < f.S; ++s) {
                    float m, v;
                    stepbn_moments(x, f.acc, s * C + c, s, c, B, C, HW, f.ny, SC, m, v);
                    em = fmaf(f.coef[s], m, em);
                    ev = fmaf(f.coef_u[s], v, ev);
                }
                f.run_mean[c] = fmaf(f.decay, f.run_mean[c], em);
                f.run_var[c] = fmaf(f.decay, f.run_var[c], ev);
            }
        }
        if (f.nbt && threadIdx.x == 0) *f.nbt += f.S;
        return;
    }
    const unsigned nvec = total / VEC;
    for (unsigned iv = bid * blockDim.x + threadIdx.x; iv < nvec; iv += nblk * blockDim.x) {
        const unsigned idx = iv * VEC;
        const unsigned r = idx / (unsigned)HW;  // frame * C + c
        const int c = (int)(r % (unsigned)C);
        const int s = (int)(r / (unsigned)C / (unsigned)B);
        float m = f.acc ? 0.f : f.mean_out[s * C + c], vr = f.acc ? 0.f : f.var_out[s * C + c];
        stepbn_moments(x, f.acc, s * C + c, s, c, B, C, HW, f.ny, SC, m, vr);
        const float rstd = rsqrtf(vr + eps);
        const float ga = gamma ? gamma[c] : 1.f, be = gamma ? beta[c] : 0.f;
        float v[VEC];
        if (VEC == 4) {
            const float4 t = *reinterpret_cast<const float4*>(x + idx);
            v[0] = t.x; v[1 % VEC] = t.y; v[2 % VEC] = t.z; v[3 % VEC] = t.w;
        } else {
            v[0] = x[idx];
        }
#pragma unroll
        for (int j = 0; j < VEC; ++j) v[j] = stepbn_act((v[j] - m) * rstd * ga + be, act, slope);
        if (VEC == 4)
            *reinterpret_cast<float4*>(y + idx) = float4{v[0], v[1 % VEC], v[2 % VEC], v[3 % VEC]};
        else
            y[idx] = v[0];
    }
}
// block (sc, j): partial sums of g' and g' * xhat over frames j, j+gridDim.y, ...  with g' = g * act'(y), stored at
// sg[j][sc] / sgx[j][sc] (no atomics, nothing to zero)
__global__ __launch_bounds__(256) void stepbn_bwd_reduce_kernel(const float* __restrict__ x,
                                                                const float* __restrict__ gamma,
                                                                const float* __restrict__ beta,
                                                                const float* __restrict__ g,
                                                                const float* __restrict__ mean,
                                                                const float* __restrict__ var, float* __restrict__ sg,
                                                                float* __restrict__ sgx, int B, int C, int HW, float eps,
                                                                int act, float slope) {
    __shared__ float sm[4];
    const int s = blockIdx.x / C, c = blockIdx.x - s * C;
    const long off = ((long)s * B * C + c) * HW, fs = (long)C * HW;
    const float m = mean[blockIdx.x], rstd = rsqrtf(var[blockIdx.x] + eps);
    const float ga = gamma ? gamma[c] : 1.f, be = gamma ? beta[c] : 0.f;
    float a = 0.f, ax = 0.f;
    if ((HW & 3) == 0 && ((((uintptr_t)x) | ((uintptr_t)g)) & 15) == 0) {  // 16-byte loads
        const int HW4 = HW >> 2, nb = (B - (int)blockIdx.y + (int)gridDim.y - 1) / (int)gridDim.y;
        for (int idx = threadIdx.x; idx < nb * HW4; idx += 256) {
            const int bi = idx / HW4, p = idx - bi * HW4;
            const long e0 = off + (blockIdx.y + (long)bi * gridDim.y) * fs;
            const float4 xq = reinterpret_cast<const float4*>(x + e0)[p], gq = reinterpret_cast<const float4*>(g + e0)[p];
            const float xv[4] = {xq.x, xq.y, xq.z, xq.w}, gv[4] = {gq.x, gq.y, gq.z, gq.w};
#pragma unroll
            for (int i = 0; i < 4; ++i) {
                const float xh = (xv[i] - m) * rstd;
                const float gp = gv[i] * stepbn_dact(xh * ga + be, act, slope);
                a += gp;
                ax = fmaf(gp, xh, ax);
            }
        }
    } else {
        for (int b = blockIdx.y; b < B; b += gridDim.y) {
            const long e0 = off + b * fs;
            for (int p = threadIdx.x; p < HW; p += 256) {
                const float xh = (x[e0 + p] - m) * rstd;
                const float gp = g[e0 + p] * stepbn_dact(xh * ga + be, act, slope);
                a += gp;
                ax = fmaf(gp, xh, ax);
            }
        }
    }
    const float ta = block_sum_256(a, sm);
    const float tx = block_sum_256(ax, sm);
    if (threadIdx.x == 0) {
        sg[(long)blockIdx.y * gridDim.x + blockIdx.x] = ta;
        sgx[(long)blockIdx.y * gridDim.x + blockIdx.x] = tx;
    }
}
template <int VEC>
__global__ void stepbn_bwd_apply_kernel(const float* __restrict__ x, const float* __restrict__ beta,
                                        const float* __restrict__ g, const float* __restrict__ mean,
                                        const float* __restrict__ var, const float* __restrict__ gamma,
                                        const float* __restrict__ sg, const float* __restrict__ sgx,
                                        float* __restrict__ gx, unsigned total, int B, int C, int HW, float eps, int act,
                                        float slope, float* __restrict__ ggamma, float* __restrict__ gbeta, int S, int ny,
                                        int world) {
    // (one extra workgroup, the first, for the parameter gradients = the per-step sums added over the steps)
    const int SC = S * C;
    const unsigned nblk = ggamma ? gridDim.x - 1 : gridDim.x, bid = ggamma ? blockIdx.x - 1 : blockIdx.x;
    if (ggamma && blockIdx.x == 0) {
        for (int c = threadIdx.x; c < C; c += blockDim.x) {
            float a = 0.f, b = 0.f;
            for (int i = 0; i < ny * S; ++i) {  // [ny][S][C]: all partials of channel c
                a += sgx[(long)i * C + c];
                b += sg[(long)i * C + c];
            }
            // (world > 1: the partial sums were added over the ranks for gx; a parameter gradient is this rank's
            // share of a rank average, so the sum over ranks is divided by their number)
            ggamma[c] = a / (float)world;
            gbeta[c] = b / (float)world;
        }
        return;
    }
    const float inv_n = 1.f / ((float)(B * HW) * (float)world);
    const unsigned nvec = total / VEC;
    for (unsigned iv = bid * blockDim.x + threadIdx.x; iv < nvec; iv += nblk * blockDim.x) {
        const unsigned idx = iv * VEC;
        const unsigned r = idx / (unsigned)HW;
        const int c = (int)(r % (unsigned)C);
        const int s = (int)(r / (unsigned)C / (unsigned)B);
        const int sc = s * C + c;
        const float rstd = rsqrtf(var[sc] + eps), m = mean[sc];
        const float w = gamma ? gamma[c] : 1.f, be = gamma ? beta[c] : 0.f;
        float k1 = 0.f, k2 = 0.f;
        for (int yy = 0; yy < ny; ++yy) {
            k1 += sg[(long)yy * SC + sc];
            k2 += sgx[(long)yy * SC + sc];
        }
        k1 *= inv_n;
        k2 *= inv_n;
        float xv[VEC], gv[VEC];
        if (VEC == 4) {
            const float4 t = *reinterpret_cast<const float4*>(x + idx);
            const float4 u = *reinterpret_cast<const float4*>(g + idx);
            xv[0] = t.x; xv[1 % VEC] = t.y; xv[2 % VEC] = t.z; xv[3 % VEC] = t.w;
            gv[0] = u.x; gv[1 % VEC] = u.y; gv[2 % VEC] = u.z; gv[3 % VEC] = u.w;
        } else {
            xv[0] = x[idx];
            gv[0] = g[idx];
        }
#pragma unroll
        for (int j = 0; j < VEC; ++j) {
            const float xh = (xv[j] - m) * rstd;
            const float gp = gv[j] * stepbn_dact(xh * w + be, act, slope);
            xv[j] = w * rstd * (gp - k1 - xh * k2);
        }
        if (VEC == 4)
            *reinterpret_cast<float4*>(gx + idx) = float4{xv[0], xv[1 % VEC], xv[2 % VEC], xv[3 % VEC]};
        else
            gx[idx] = xv[0];
    }
}
// floats of scratch rfn_stepbn_fwd_f32 (acc) / rfn_stepbn_bwd_f32 (sums) need for these sizes
extern "C" long rfn_stepbn_scratch_floats(int S, int B, int C) {
    if (S <= 0 || B <= 0 || C <= 0) return 0;
    return 2L * S * C * stepbn_split(S, B, C);
}
// stats + normalise + activate + running statistics in TWO launches (partial sums, apply): what per_step_batchnorm_act
// needs of one BatchNorm layer.  mean / var [S*C] are outputs (kept for the backward); run_mean / run_var [C] (both or
// neither) receive r <- decay r + sum_s coef[s] stat[s] with coef / coef_u [S] on the device; nbt (optional) += S.
extern "C" int rfn_stepbn_fwd_f32(const float* x, const float* gamma, const float* beta, float* y, float* mean, float* var,
                                  float* acc, float* run_mean, float* run_var, const float* coef, const float* coef_u,
                                  float decay, long long* nbt, int S, int B, int C, int HW, float eps, int act, float slope,
                                  rfn_stream_t stream) {
    RFN_CHECK_ARG(x && y && mean && var && acc && S > 0 && B > 0 && C > 0 && HW > 0, -1);
    RFN_CHECK_ARG(((gamma && beta) || (!gamma && !beta)) && ((run_mean && run_var && coef && coef_u) || (!run_mean && !run_var)),
                  -2);
    const long total = (long)S * B * C * HW;
    RFN_CHECK_ARG(total < (1L << 31), -3);
    hipStream_t st = (hipStream_t)stream;
    const int ny = stepbn_split(S, B, C);
    hipLaunchKernelGGL(stepbn_stats_kernel, dim3(S * C, ny), dim3(256), 0, st, x, acc, B, C, HW);
    StepBnFused f;
    f.acc = acc; f.mean_out = mean; f.var_out = var; f.run_mean = run_mean; f.run_var = run_var; f.coef = coef;
    f.coef_u = coef_u; f.decay = decay; f.nbt = nbt; f.S = S; f.ny = ny;
    const bool v4 = HW % 4 == 0 && (((uintptr_t)x | (uintptr_t)y) & 15) == 0;
    const long nthr = v4 ? total / 4 : total;
    const int grid = (int)((nthr + 255) / 256 < 16384 ? (nthr + 255) / 256 : 16384);
    if (v4)
        hipLaunchKernelGGL(stepbn_apply_kernel<4>, dim3(grid + 1), dim3(256), 0, st, x, gamma, beta, y, (unsigned)total, B, C,
                           HW, eps, act, slope, f);
    else
        hipLaunchKernelGGL(stepbn_apply_kernel<1>, dim3(grid + 1), dim3(256), 0, st, x, gamma, beta, y, (unsigned)total, B, C,
                           HW, eps, act, slope, f);
    RFN_LAUNCH_CHECK();
    return 0;
}
/* normalise + activate with GIVEN per-step statistics mean / var [S*C] (synchronised BatchNorm across data-parallel
 * ranks: the caller combined the ranks' moments); one launch */
extern "C" int rfn_stepbn_apply_f32(const float* x, const float* gamma, const float* beta, float* y, const float* mean,
                                    const float* var, int S, int B, int C, int HW, float eps, int act, float slope,
                                    rfn_stream_t stream) {
    RFN_CHECK_ARG(x && y && mean && var && S > 0 && B > 0 && C > 0 && HW > 0, -1);
    RFN_CHECK_ARG((gamma && beta) || (!gamma && !beta), -2);
    const long total = (long)S * B * C * HW;
    RFN_CHECK_ARG(total < (1L << 31), -3);
    StepBnFused f;
    memset(&f, 0, sizeof(f));
    f.mean_out = const_cast<float*>(mean); f.var_out = const_cast<float*>(var); f.S = S;
    const bool v4 = HW % 4 == 0 && (((uintptr_t)x | (uintptr_t)y) & 15) == 0;
    const long nthr = v4 ? total / 4 : total;
    const int grid = (int)((nthr + 255) / 256 < 16384 ? (nthr + 255) / 256 : 16384);
    if (v4)
        hipLaunchKernelGGL(stepbn_apply_kernel<4>, dim3(grid + 1), dim3(256), 0, (hipStream_t)stream, x, gamma, beta, y,
                           (unsigned)total, B, C, HW, eps, act, slope, f);
    else
        hipLaunchKernelGGL(stepbn_apply_kernel<1>, dim3(grid + 1), dim3(256), 0, (hipStream_t)stream, x, gamma, beta, y,
                           (unsigned)total, B, C, HW, eps, act, slope, f);
    RFN_LAUNCH_CHECK();
    return 0;
}
// the whole backward of one layer in TWO launches (per-step partial sums, apply): sums = scratch
// [rfn_stepbn_scratch_floats] (sg | sgx); ggamma / gbeta [C] (both or neither) = the parameter gradients, written by the
// apply kernel
extern "C" int rfn_stepbn_bwd_f32(const float* x, const float* gamma, const float* beta, const float* g, const float* mean,
                                  const float* var, float* sums, float* gx, float* ggamma, float* gbeta, int S, int B,
                                  int C, int HW, float eps, int act, float slope, int stage, int world,
                                  rfn_stream_t stream) {
    // stage 0: both launches; 1: the partial sums only; 2: the apply only (the caller has added `sums` over `world`
    // data-parallel ranks in between: synchronised BatchNorm; mean / var are the global statistics)
    RFN_CHECK_ARG(x && g && mean && var && sums && gx && S > 0 && B > 0 && C > 0 && HW > 0, -1);
    RFN_CHECK_ARG(stage >= 0 && stage <= 2 && world >= 1, -4);
    RFN_CHECK_ARG(((gamma && beta) || (!gamma && !beta)) && ((ggamma && gbeta) || (!ggamma && !gbeta)), -2);
    const long total = (long)S * B * C * HW;
    RFN_CHECK_ARG(total < (1L << 31), -3);
    hipStream_t st = (hipStream_t)stream;
    const int ny = stepbn_split(S, B, C);
    float* sg = sums;
    float* sgx = sums + (long)ny * S * C;
    if (stage != 2)
        hipLaunchKernelGGL(stepbn_bwd_reduce_kernel, dim3(S * C, ny), dim3(256), 0, st, x, gamma, beta, g, mean, var, sg, sgx,
                           B, C, HW, eps, act, slope);
    if (stage == 1) {
        RFN_LAUNCH_CHECK();
        return 0;
    }
    const bool v4 = HW % 4 == 0 && (((uintptr_t)x | (uintptr_t)g | (uintptr_t)gx) & 15) == 0;
    const long nthr = v4 ? total / 4 : total;
    const int grid = (int)((nthr + 255) / 256 < 16384 ? (nthr + 255) / 256 : 16384);
    const int extra = ggamma ? 1 : 0;
    if (v4)
        hipLaunchKernelGGL(stepbn_bwd_apply_kernel<4>, dim3(grid + extra), dim3(256), 0, st, x, beta, g, mean, var, gamma,
                           sg, sgx, gx, (unsigned)total, B, C, HW, eps, act, slope, ggamma, gbeta, S, ny, world);
    else
        hipLaunchKernelGGL(stepbn_bwd_apply_kernel<1>, dim3(grid + extra), dim3(256), 0, st, x, beta, g, mean, var, gamma,
                           sg, sgx, gx, (unsigned)total, B, C, HW, eps, act, slope, ggamma, gbeta, S, ny, world);
    RFN_LAUNCH_CHECK();
    return 0;
}

// ------------------------------------------------------------------------------------------------ InvConv.get_weight
// Flow/glow_modules.py:178-207 for the K steps of a flow level in one launch each way (the reference -- and a torch
// restatement -- spends ~17 launches per level forward and ~20 backward on this C x C algebra):
//   Lm = lower o tril(-1) + I,   Um = upper o triu(+1) + diag(sign_s * exp(log_s)),   W = P Lm Um,
//   dlogdet = HW * sum(log_s)  (summed over the K steps, in step order, into ONE scalar that is WRITTEN: no atomics).
// Backward, given gW and the gradient gc of the scalar:  gT = P^T gW;  g_lower = (gT Um^T) o tril(-1);
//   g_upper = (Lm^T gT) o triu(+1);  g_log_s = diag(Lm^T gT) * sign_s * exp(log_s) + gc * HW.
// One workgroup per step, the three matrices in LDS (C <= RFN_INVCONV_MAX_CHANNELS).  Parameters arrive as pointer arrays in the kernel
// arguments (no stacking copies): K <= RFN_INVCONV_MAX_STEPS.
struct InvConvWeightsParams {
    const float* p[RFN_INVCONV_MAX_STEPS];
    const float* lower[RFN_INVCONV_MAX_STEPS];
    const float* upper[RFN_INVCONV_MAX_STEPS];
    const float* log_s[RFN_INVCONV_MAX_STEPS];
    const float* sign_s[RFN_INVCONV_MAX_STEPS];
    float* W;            // [K][C][C]
    float* logdet;       // scalar, written
    const float* gW;     // backward: [K][C][C]
    const float* gc;     // backward: gradient of the scalar (may be null)
    float* g_lower;      // [K][C][C]
    float* g_upper;      // [K][C][C]
    float* g_log_s;      // [K][C]
    int C;
    float hw;
};
// LDS matrices have row stride C + 1: column walks (transposed operands) are bank-conflict free
__device__ __forceinline__ void invconv_load_LU(const InvConvWeightsParams& q, int k, float* Lm, float* Um) {
    const int C = q.C, CP = C + 1;
    for (int i = threadIdx.x; i < C * C; i += blockDim.x) {
        const int r = i / C, c = i - r * C;
        Lm[r * CP + c] = r > c ? q.lower[k][i] : (r == c ? 1.f : 0.f);
        Um[r * CP + c] = r < c ? q.upper[k][i] : (r == c ? q.sign_s[k][r] * expf(q.log_s[k][r]) : 0.f);
    }
}
__device__ __forceinline__ void invconv_load(const float* __restrict__ src, float* dst, int C) {
    for (int i = threadIdx.x; i < C * C; i += blockDim.x) dst[(i / C) * (C + 1) + i % C] = src[i];
}
// out(r, c) = sum_j A(r, j) * B(j, c) for the C x C matrices in LDS (row stride C + 1), each thread a 4 x 4 block of the
// result (8 LDS reads per 16 multiply-adds); TA / TB: the operand is read transposed.  C % 4 == 0 or the scalar tail runs.
template <bool TA, bool TB, typename Store>
__device__ __forceinline__ void invconv_mm(const float* __restrict__ A, const float* __restrict__ B, int C, Store store) {
    const int CP = C + 1, nb = C >> 2;
    for (int blk = threadIdx.x; blk < nb * nb; blk += blockDim.x) {
        const int r0 = (blk / nb) * 4, c0 = (blk % nb) * 4;
        float acc[4][4] = {};
        for (int j = 0; j < C; ++j) {
            float av[4], bv[4];
#pragma unroll
            for (int i = 0; i < 4; ++i) {
                av[i] = TA ? A[j * CP + r0 + i] : A[(r0 + i) * CP + j];
                bv[i] = TB ? B[(c0 + i) * CP + j] : B[j * CP + c0 + i];
            }
#pragma unroll
            for (int i = 0; i < 4; ++i)
#pragma unroll
                for (int k = 0; k < 4; ++k) acc[i][k] = fmaf(av[i], bv[k], acc[i][k]);
        }
#pragma unroll
        for (int i = 0; i < 4; ++i)
#pragma unroll
            for (int k = 0; k < 4; ++k) store(r0 + i, c0 + k, acc[i][k]);
    }
    const int Cm = nb * 4;  // ragged edge (C % 4 != 0): the last rows and columns one element at a time
    for (int i = threadIdx.x; i < C * C; i += blockDim.x) {
        const int r = i / C, c = i - r * C;
        if (r < Cm && c < Cm) continue;
        float a = 0.f;
        for (int j = 0; j < C; ++j)
            a = fmaf(TA ? A[j * CP + r] : A[r * CP + j], TB ? B[c * CP + j] : B[j * CP + c], a);
        store(r, c, a);
    }
}
__global__ __launch_bounds__(256) void invconv_weights_fwd_kernel(const InvConvWeightsParams q) {
    extern __shared__ float sm_iw[];
    const int C = q.C, CP = C + 1, k = blockIdx.x;
    float* B0 = sm_iw;            // Lm, then P
    float* B1 = B0 + C * CP;      // Um
    float* B2 = B1 + C * CP;      // T = Lm Um
    invconv_load_LU(q, k, B0, B1);
    __syncthreads();
    invconv_mm<false, false>(B0, B1, C, [&](int r, int c, float v) { B2[r * CP + c] = v; });
    __syncthreads();
    invconv_load(q.p[k], B0, C);
    __syncthreads();
    float* W = q.W + (long)k * C * C;
    invconv_mm<false, false>(B0, B2, C, [&](int r, int c, float v) { W[r * C + c] = v; });  // W = P T
    if (k == 0 && threadIdx.x < 64) {  // one wave of ONE block: HW * sum over the steps (in order) of sum(log_s): no atomics
        float tot = 0.f;
        for (int kk = 0; kk < (int)gridDim.x; ++kk) {
            float v = 0.f;
            for (int c = threadIdx.x; c < C; c += 64) v += q.log_s[kk][c];
            tot += wave_sum(v) * q.hw;
        }
        if (threadIdx.x == 0) *q.logdet = tot;
    }
}
__global__ __launch_bounds__(256) void invconv_weights_bwd_kernel(const InvConvWeightsParams q) {
    extern __shared__ float sm_iw[];
    const int C = q.C, CP = C + 1, k = blockIdx.x;
    float* B0 = sm_iw;            // P, then Lm
    float* B1 = B0 + C * CP;      // gW, then Um
    float* B2 = B1 + C * CP;      // gT = P^T gW
    invconv_load(q.p[k], B0, C);
    invconv_load(q.gW + (long)k * C * C, B1, C);
    __syncthreads();
    invconv_mm<true, false>(B0, B1, C, [&](int r, int c, float v) { B2[r * CP + c] = v; });
    __syncthreads();
    invconv_load_LU(q, k, B0, B1);
    __syncthreads();
    const float* Um = B1;
    const float gc = q.gc ? q.gc[0] * q.hw : 0.f;
    float* gl = q.g_lower + (long)k * C * C;
    float* gu = q.g_upper + (long)k * C * C;
    float* gs = q.g_log_s + (long)k * C;
    // g_lower = (gT Um^T) o tril(-1);  g_upper = (Lm^T gT) o triu(+1);  g_log_s from the diagonal of Lm^T gT
    invconv_mm<false, true>(B2, B1, C, [&](int r, int c, float v) { gl[r * C + c] = r > c ? v : 0.f; });
    invconv_mm<true, false>(B0, B2, C, [&](int r, int c, float v) {
        gu[r * C + c] = r < c ? v : 0.f;
        if (r == c) gs[r] = v * Um[r * CP + r] + gc;
    });
}
static int invconv_weights_fill(InvConvWeightsParams& q, const float* const* p, const float* const* lower,
                                const float* const* upper, const float* const* log_s, const float* const* sign_s, int K,
                                int C, int HW) {
    if (!(p && lower && upper && log_s && sign_s && K >= 1 && K <= RFN_INVCONV_MAX_STEPS && C >= 1 &&
          C <= RFN_INVCONV_MAX_CHANNELS && HW > 0))
        return -1;
    memset(&q, 0, sizeof(q));
    for (int k = 0; k < K; ++k) {
        if (!(p[k] && lower[k] && upper[k] && log_s[k] && sign_s[k])) return -2;
        q.p[k] = p[k]; q.lower[k] = lower[k]; q.upper[k] = upper[k]; q.log_s[k] = log_s[k]; q.sign_s[k] = sign_s[k];
    }
    q.C = C;
    q.hw = (float)HW;
    return 0;
}
extern "C" int rfn_invconv_weights_fwd_f32(const float* const* p, const float* const* lower, const float* const* upper,
                                           const float* const* log_s, const float* const* sign_s, float* W, float* logdet,
                                           int K, int C, int HW, rfn_stream_t stream) {
    InvConvWeightsParams q;
    int rc = invconv_weights_fill(q, p, lower, upper, log_s, sign_s, K, C, HW);
    if (rc || !W || !logdet) {
        rfn_set_error("rfn_invconv_weights_fwd_f32: argument check failed (%d)", rc ? rc : -3);
        return rc ? rc : -3;
    }
    q.W = W;
    q.logdet = logdet;
    if (C > 64)  // three padded C x C matrices: 112 KB at C = 96 (the LDS of a CU is 160 KB)
        (void)hipFuncSetAttribute((const void*)invconv_weights_fwd_kernel, hipFuncAttributeMaxDynamicSharedMemorySize,
                                  3 * C * (C + 1) * (int)sizeof(float));
    hipLaunchKernelGGL(invconv_weights_fwd_kernel, dim3(K), dim3(256), (size_t)3 * C * (C + 1) * sizeof(float),
                       (hipStream_t)stream, q);
    RFN_LAUNCH_CHECK();
    return 0;
}
extern "C" int rfn_invconv_weights_bwd_f32(const float* const* p, const float* const* lower, const float* const* upper,
                                           const float* const* log_s, const float* const* sign_s, const float* gW,
                                           const float* gc, float* g_lower, float* g_upper, float* g_log_s, int K, int C,
                                           int HW, rfn_stream_t stream) {
    InvConvWeightsParams q;
    int rc = invconv_weights_fill(q, p, lower, upper, log_s, sign_s, K, C, HW);
    if (rc || !gW || !g_lower || !g_upper || !g_log_s) {
        rfn_set_error("rfn_invconv_weights_bwd_f32: argument check failed (%d)", rc ? rc : -3);
        return rc ? rc : -3;
    }
    q.gW = gW; q.gc = gc; q.g_lower = g_lower; q.g_upper = g_upper; q.g_log_s = g_log_s;
    if (C > 64)
        (void)hipFuncSetAttribute((const void*)invconv_weights_bwd_kernel, hipFuncAttributeMaxDynamicSharedMemorySize,
                                  3 * C * (C + 1) * (int)sizeof(float));
    hipLaunchKernelGGL(invconv_weights_bwd_kernel, dim3(K), dim3(256), (size_t)3 * C * (C + 1) * sizeof(float),
                       (hipStream_t)stream, q);
    RFN_LAUNCH_CHECK();
    return 0;
}

// ------------------------------------------------------------------------------------------------ SRNN latent step
// RFN.loss per timestep (RFN_new.py:167-184,206-207): enc / pri are the outputs [B, 2*Z, HW] of the encoder / prior
// parameter convs (loc | raw scale, "chunk(2,1)" halves, SimpleParamNet.forward Utils/modules.py:240-244):
//   ps = softplus(pri_raw), es = softplus(enc_raw), pm = pri_loc, em = enc_loc (+ pm with res_q)
//   zt = pm + ps*eps_p ;  zxt = em + es*eps_q ;  kl = KL(N(em,es) || N(pm,ps)) element-wise
// One launch instead of ~25 tiny elementwise kernels per timestep (and ~50 in backward).
__device__ __forceinline__ float sigmoid_sp(float raw) { return raw > 20.f ? 1.f : 1.f / (1.f + expf(-raw)); }

__global__ void latent_step_fwd_kernel(const float* __restrict__ enc, const float* __restrict__ pri,
                                       const float* __restrict__ eps_p, const float* __restrict__ eps_q,
                                       float* __restrict__ zt, float* __restrict__ zxt, float* __restrict__ kl,
                                       float* __restrict__ em_o, float* __restrict__ es_o, int B, int ZHW, int res_q) {
    const long total = (long)B * ZHW;
    for (long idx = (long)blockIdx.x * blockDim.x + threadIdx.x; idx < total; idx += (long)gridDim.x * blockDim.x) {
        const long b = idx / ZHW, e = idx - b * ZHW;
        const float pm = pri[b * 2 * ZHW + e], ps = softplusf_(pri[b * 2 * ZHW + ZHW + e]);
        const float es = softplusf_(enc[b * 2 * ZHW + ZHW + e]);
        const float em = enc[b * 2 * ZHW + e] + (res_q ? pm : 0.f);
        zt[idx] = pm + ps * eps_p[idx];
        zxt[idx] = em + es * eps_q[idx];
        const float r = es / ps, d = (em - pm) / ps;
        kl[idx] = 0.5f * (r * r + d * d - 1.f - logf(r * r));
        em_o[idx] = em;
        es_o[idx] = es;
    }
}
__global__ void latent_step_bwd_kernel(const float* __restrict__ enc, const float* __restrict__ pri,
                                       const float* __restrict__ eps_p, const float* __restrict__ eps_q,
                                       const float* __restrict__ g_zt, long g_zt_ns,
                                       const float* __restrict__ g_zxt, long g_zxt_ns,
                                       const float* __restrict__ g_kl, const float* __restrict__ g_em,
                                       const float* __restrict__ g_es, float* __restrict__ g_enc,
                                       float* __restrict__ g_pri, int B, int ZHW, int res_q) {
    const long total = (long)B * ZHW;
    for (long idx = (long)blockIdx.x * blockDim.x + threadIdx.x; idx < total; idx += (long)gridDim.x * blockDim.x) {
        const long b = idx / ZHW, e = idx - b * ZHW;
        const float praw = pri[b * 2 * ZHW + ZHW + e], eraw = enc[b * 2 * ZHW + ZHW + e];
        const float pm = pri[b * 2 * ZHW + e], ps = softplusf_(praw), es = softplusf_(eraw);
        const float em = enc[b * 2 * ZHW + e] + (res_q ? pm : 0.f);
        const float gzt = g_zt ? g_zt[b * g_zt_ns + e] : 0.f, gzx = g_zxt ? g_zxt[b * g_zxt_ns + e] : 0.f;
        const float gk = g_kl ? g_kl[idx] : 0.f;
        const float ips = 1.f / ps, d = (em - pm) * ips * ips;  // (em-pm)/ps^2
        const float d_em = gzx + gk * d + (g_em ? g_em[idx] : 0.f);
        const float d_es = gzx * eps_q[idx] + gk * (es * ips * ips - 1.f / es) + (g_es ? g_es[idx] : 0.f);
        float d_pm = gzt - gk * d;
        const float d_ps = gzt * eps_p[idx] + gk * (ips - (es * es + (em - pm) * (em - pm)) * ips * ips * ips);
        if (res_q) d_pm += d_em;
        g_enc[b * 2 * ZHW + e] = d_em;
        g_enc[b * 2 * ZHW + ZHW + e] = d_es * sigmoid_sp(eraw);
        g_pri[b * 2 * ZHW + e] = d_pm;
        g_pri[b * 2 * ZHW + ZHW + e] = d_ps * sigmoid_sp(praw);
    }
}
extern "C" int rfn_latent_step_fwd_f32(const float* enc, const float* pri, const float* eps_p, const float* eps_q,
                                       float* zt, float* zxt, float* kl, float* em, float* es, int B, int ZHW,
                                       int res_q, rfn_stream_t stream) {
    RFN_CHECK_ARG(enc && pri && eps_p && eps_q && zt && zxt && kl && em && es && B >= 0 && ZHW > 0, -1);
    if (B == 0) return 0;
    long tot = (long)B * ZHW;
    int grid = (int)((tot + 255) / 256 < 1024 ? (tot + 255) / 256 : 1024);
    hipLaunchKernelGGL(latent_step_fwd_kernel, dim3(grid), dim3(256), 0, (hipStream_t)stream, enc, pri, eps_p, eps_q, zt,
                       zxt, kl, em, es, B, ZHW, res_q);
    RFN_LAUNCH_CHECK();
    return 0;
}
extern "C" int rfn_latent_step_bwd_f32(const float* enc, const float* pri, const float* eps_p, const float* eps_q,
                                       const float* g_zt, long g_zt_ns, const float* g_zxt, long g_zxt_ns,
                                       const float* g_kl, const float* g_em, const float* g_es, float* g_enc,
                                       float* g_pri, int B, int ZHW, int res_q, rfn_stream_t stream) {
    RFN_CHECK_ARG(enc && pri && eps_p && eps_q && g_enc && g_pri && B >= 0 && ZHW > 0, -1);
    RFN_CHECK_ARG((!g_zt || g_zt_ns >= ZHW) && (!g_zxt || g_zxt_ns >= ZHW), -1);
    if (B == 0) return 0;
    long tot = (long)B * ZHW;
    int grid = (int)((tot + 255) / 256 < 1024 ? (tot + 255) / 256 : 1024);
    hipLaunchKernelGGL(latent_step_bwd_kernel, dim3(grid), dim3(256), 0, (hipStream_t)stream, enc, pri, eps_p, eps_q,
                       g_zt, g_zt_ns, g_zxt, g_zxt_ns, g_kl, g_em, g_es, g_enc, g_pri, B, ZHW, res_q);
    RFN_LAUNCH_CHECK();
    return 0;
}

// ------------------------------------------------------------------------------------------------ ConvLSTM gates
__global__ void convlstm_gates_fwd_kernel(const float* __restrict__ cc, const float* __restrict__ c_prev, long c_ns,
                                          const float* __restrict__ Wci, const float* __restrict__ Wcf,
                                          const float* __restrict__ Wco, float* __restrict__ h_out, long h_ns,
                                          float* __restrict__ c_out, long co_ns, float* __restrict__ gates, int N,
                                          int Hc, int HW) {
    const long per = (long)Hc * HW;
    for (long idx = (long)blockIdx.x * blockDim.x + threadIdx.x; idx < (long)N * per;
         idx += (long)gridDim.x * blockDim.x) {
        int n = (int)(idx / per);
        long e = idx - n * per;  // channel*HW + p
        const float* ccn = cc + (long)n * 4 * per;
        float cp = c_prev[n * c_ns + e];
        float wi = Wci ? Wci[e] : 0.f, wf = Wcf ? Wcf[e] : 0.f, wo = Wco ? Wco[e] : 0.f;
        float i = 1.f / (1.f + expf(-(ccn[e] + wi * cp)));
        float f = 1.f / (1.f + expf(-(ccn[per + e] + wf * cp)));
        float g = tanhf(ccn[3 * per + e]);
        float cn = f * cp + i * g;
        float o = 1.f / (1.f + expf(-(ccn[2 * per + e] + wo * cn)));
        h_out[n * h_ns + e] = o * tanhf(cn);
        c_out[n * co_ns + e] = cn;
        if (gates) {
            float* gn = gates + (long)n * 4 * per;
            gn[e] = i;
            gn[per + e] = f;
            gn[2 * per + e] = o;
            gn[3 * per + e] = g;
        }
    }
}
extern "C" int rfn_convlstm_gates_fwd_f32(const float* cc, const float* c_prev, long c_ns, const float* Wci,
                                          const float* Wcf, const float* Wco, float* h_out, long h_ns, float* c_out,
                                          long co_ns, float* gates, int N, int Hc, int HW, rfn_stream_t stream) {
    RFN_CHECK_ARG(cc && c_prev && h_out && c_out && N >= 0 && Hc > 0 && HW > 0, -1);
    if (N == 0) return 0;
    long tot = (long)N * Hc * HW;
    int grid = (int)((tot + 255) / 256 < 2048 ? (tot + 255) / 256 : 2048);
    hipLaunchKernelGGL(convlstm_gates_fwd_kernel, dim3(grid), dim3(256), 0, (hipStream_t)stream, cc, c_prev, c_ns, Wci,
                       Wcf, Wco, h_out, h_ns, c_out, co_ns, gates, N, Hc, HW);
    RFN_LAUNCH_CHECK();
    return 0;
}

__global__ void convlstm_gates_bwd_kernel(const float* __restrict__ gates, const float* __restrict__ c_prev, long c_ns,
                                          const float* __restrict__ c_out, long co_ns, const float* __restrict__ gh,
                                          long gh_ns, const float* __restrict__ gc_next, long gcn_ns,
                                          const float* __restrict__ Wci, const float* __restrict__ Wcf,
                                          const float* __restrict__ Wco, float* __restrict__ gcc,
                                          float* __restrict__ gc_prev, long gcp_ns, int N, int Hc, int HW) {
    const long per = (long)Hc * HW;
    for (long idx = (long)blockIdx.x * blockDim.x + threadIdx.x; idx < (long)N * per;
         idx += (long)gridDim.x * blockDim.x) {
        int n = (int)(idx / per);
        long e = idx - n * per;
        const float* gn = gates + (long)n * 4 * per;
        float i = gn[e], f = gn[per + e], o = gn[2 * per + e], g = gn[3 * per + e];
        float cp = c_prev[n * c_ns + e];
        float cn = c_out[n * co_ns + e];
        float wi = Wci ? Wci[e] : 0.f, wf = Wcf ? Wcf[e] : 0.f, wo = Wco ? Wco[e] : 0.f;
        float ghv = gh ? gh[n * gh_ns + e] : 0.f;
        float gcn = gc_next ? gc_next[n * gcn_ns + e] : 0.f;
        float tc = tanhf(cn);
        float go_pre = ghv * tc * o * (1.f - o);  // grad wrt (cc_o + Wco*cn)
        float gc = gcn + ghv * o * (1.f - tc * tc) + go_pre * wo;
        float gi_pre = gc * g * i * (1.f - i);
        float gf_pre = gc * cp * f * (1.f - f);
        float gg_pre = gc * i * (1.f - g * g);
        float* gccn = gcc + (long)n * 4 * per;
        gccn[e] = gi_pre;
        gccn[per + e] = gf_pre;
        gccn[2 * per + e] = go_pre;
        gccn[3 * per + e] = gg_pre;
        gc_prev[n * gcp_ns + e] = gc * f + gi_pre * wi + gf_pre * wf;
    }
}
extern "C" int rfn_convlstm_gates_bwd_f32(const float* gates, const float* c_prev, long c_ns, const float* c_out,
                                          long co_ns, const float* gh, long gh_ns, const float* gc_next, long gcn_ns,
                                          const float* Wci, const float* Wcf, const float* Wco, float* gcc,
                                          float* gc_prev, long gcp_ns, int N, int Hc, int HW, rfn_stream_t stream) {
    RFN_CHECK_ARG(gates && c_prev && c_out && gcc && gc_prev && N >= 0 && Hc > 0 && HW > 0, -1);
    if (N == 0) return 0;
    long tot = (long)N * Hc * HW;
    int grid = (int)((tot + 255) / 256 < 2048 ? (tot + 255) / 256 : 2048);
    hipLaunchKernelGGL(convlstm_gates_bwd_kernel, dim3(grid), dim3(256), 0, (hipStream_t)stream, gates, c_prev, c_ns,
                       c_out, co_ns, gh, gh_ns, gc_next, gcn_ns, Wci, Wcf, Wco, gcc, gc_prev, gcp_ns, N, Hc, HW);
    RFN_LAUNCH_CHECK();
    return 0;
}
