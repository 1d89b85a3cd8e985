// bnorm.hip — train-mode BatchNorm (+ LeakyReLU / ReLU) on channels-last rows, forward and backward (training path, f1).
//
// The reference normalises with torch.nn.BatchNorm{2,3}d in train mode (models/fusion.py:49-58, models/encoder.py:16-40,
// models/volume_render.py:29-37, torchvision Bottleneck) followed by a separate activation: through MIOpen that is three kernels forward,
// three backward and an element-wise kernel each way - eight passes over the activation tensor. Here:
//   forward   bn_stats_kernel        per-channel sum / sum of squares over the M rows, float64 accumulation (E[x^2] - mean^2 is then exact to
//                                    ~1e-12); one partial per block -> ws[block][2][C] (no atomics: deterministic)
//             bn_finalize_kernel     sums the partials; forward: mean / invstd, running_mean / running_var (unbiased) updated in place
//             bn_apply_fwd_kernel    y = lrelu(x * scale + shift, slope), scale = gamma * invstd, shift = beta - mean * scale
//   backward  bn_reduce_bwd_kernel   g = dy * (pre > 0 ? 1 : slope) with pre recomputed by the forward's own expression (same sign);
//                                    per-channel sum g, sum g * xhat (float64 partials)
//             bn_finalize_kernel     dbeta / dgamma (also the two means the apply kernel needs, kept in ws[0])
//             bn_apply_bwd_kernel    dx = gamma * invstd * (g - mean(g) - xhat * mean(g xhat))
// Five passes over the activation instead of eight, six launches instead of eight. slope = 1: no activation, slope = 0: ReLU.
// Residual form (the bottleneck tail of the ResNet trunk, out = relu(bn3(conv3) + identity)): forward y = act(x scale + shift + res) in the
// same apply pass; backward takes the activation mask from the saved OUTPUT (y > 0 <=> pre-activation > 0 for any slope >= 0), and the
// apply pass also writes d res = g - what autograd otherwise runs as add, relu, threshold_backward and a gradient accumulation (4 passes).
// Rows [M][ld] fp32 with C % 4 == 0 channels used; thread = (4 channels, one row group).
#include "common.h"

namespace forge {

struct BnArgs {
    const float* x; int ldx;
    const float* dy; int lddy;            // backward only
    const float* res; int ldres;           // forward, nullable: y = act(bn(x) + res)
    const float* y; int ldy;               // backward, nullable: the forward's OUTPUT; non-null <=> the residual form (mask = y > 0)
    float* dres; int lddres;               // backward, nullable: d res = g (written)
    long long* nbt;                        // forward, nullable: num_batches_tracked, incremented by the finalize step
    float* out; int ldo;                   // forward: y; backward: dx
    const float* gamma; const float* beta; // nullable (affine=False): 1 / 0
    float* mean; float* invstd;            // [C]: written by the forward apply, read by the backward
    float* running_mean; float* running_var; float momentum;   // nullable: no running statistics
    float* dgamma; float* dbeta;           // backward outputs, nullable
    double* ws;                            // [nblk][2][C] float64 partial sums of the reduction kernel; the finalize kernel leaves the totals in ws[0]
    int nblk;                              // blocks (gridDim.x) of the reduction kernel = number of partials
    float eps, slope;
    long long M; int C;
    long long Mtot;                        // rows behind the totals the apply kernels normalise with (= M, or the all-rank count under SyncBatchNorm)
    const double* cnt;                     // non-null: that count read from device memory (the all-reduced row count: no host round trip)
};

constexpr int BN_THREADS = 256;

// channel quad / row group of a thread: CQ = min(C/4, 256) quads per block (blockIdx.y selects the slab of quads), RG = 256 / CQ row groups
__device__ __forceinline__ void bn_coords(int C, int& c, int& rg, int& RG) {
    const int C4 = C >> 2, CQ = C4 < BN_THREADS ? C4 : BN_THREADS;
    RG = BN_THREADS / CQ;
    const int q = (int)threadIdx.x % CQ + (int)blockIdx.y * CQ;
    rg = (int)threadIdx.x / CQ;
    c = (q < C4 && rg < RG) ? q << 2 : -1;
}

__device__ __forceinline__ void bn_block_sum(double (&a)[8], int c, int rg, int RG, int C, double* ws) {
    __shared__ double red[BN_THREADS][8];
#pragma unroll
    for (int k = 0; k < 8; ++k) red[threadIdx.x][k] = a[k];
    __syncthreads();
    if (c >= 0 && rg == 0) {
        const int C4 = C >> 2, CQ = C4 < BN_THREADS ? C4 : BN_THREADS;     // as bn_coords (BN_THREADS / RG only when C / 4 divides 256)
        for (int g = 1; g < RG; ++g)
#pragma unroll
            for (int k = 0; k < 8; ++k) a[k] += red[threadIdx.x + g * CQ][k];
        double* o = ws + (long long)blockIdx.x * 2 * C;
#pragma unroll
        for (int k = 0; k < 4; ++k) {
            o[c + k] = a[k];
            o[C + c + k] = a[4 + k];
        }
    }
}

// totals of the nblk partials: block = 4 channels x 64 slices of the partial list. Forward (bwd == 0): mean / invstd (+ running statistics);
// backward: dbeta / dgamma. Either way the two totals per channel are left in ws[0][2][C] for the apply kernel.
__global__ __launch_bounds__(BN_THREADS) void bn_finalize_kernel(const BnArgs a, int bwd) {
    __shared__ double red[2][BN_THREADS];
    const int cl = threadIdx.x & 3, sl = threadIdx.x >> 2, c = blockIdx.x * 4 + cl;
    double s0 = 0.0, s1 = 0.0;
    if (c < a.C) {
        // four independent partial sums per thread: the loads of four slices are in flight together (the loop is latency-bound: one 32-byte
        // read per channel quad and slice); fixed order, so the totals stay bit-reproducible
        double t0[4] = {0.0, 0.0, 0.0, 0.0}, t1[4] = {0.0, 0.0, 0.0, 0.0};
        int b = sl;
        for (; b + 192 < a.nblk; b += 256) {
#pragma unroll
            for (int u = 0; u < 4; ++u) {
                t0[u] += a.ws[(long long)(b + 64 * u) * 2 * a.C + c];
                t1[u] += a.ws[(long long)(b + 64 * u) * 2 * a.C + a.C + c];
            }
        }
        for (; b < a.nblk; b += 64) {
            t0[0] += a.ws[(long long)b * 2 * a.C + c];
            t1[0] += a.ws[(long long)b * 2 * a.C + a.C + c];
        }
        s0 = (t0[0] + t0[1]) + (t0[2] + t0[3]);
        s1 = (t1[0] + t1[1]) + (t1[2] + t1[3]);
    }
    red[0][threadIdx.x] = s0; red[1][threadIdx.x] = s1;
    __syncthreads();
    for (int s = 32; s > 0; s >>= 1) {
        if (sl < s) { red[0][threadIdx.x] += red[0][threadIdx.x + 4 * s]; red[1][threadIdx.x] += red[1][threadIdx.x + 4 * s]; }
        __syncthreads();
    }
    if (sl != 0 || c >= a.C) return;
    s0 = red[0][cl]; s1 = red[1][cl];
    a.ws[c] = s0; a.ws[a.C + c] = s1;                // safe: every partial of this channel has been read (barriers above), other channels untouched
    if (bwd == 2) return;                             // SyncBatchNorm forward: the local totals only (summed over ranks by the caller's all-reduce)
    if (bwd) {
        if (a.dbeta) a.dbeta[c] = (float)s0;
        if (a.dgamma) a.dgamma[c] = (float)s1;
        return;
    }
    if (a.nbt && c == 0) *a.nbt += 1;
    const double m = s0 / (double)a.M, var = fmax(s1 / (double)a.M - m * m, 0.0);
    a.mean[c] = (float)m;
    a.invstd[c] = (float)(1.0 / sqrt(var + (double)a.eps));
    if (a.running_mean) {
        const double unbiased = a.M > 1 ? var * (double)a.M / (double)(a.M - 1) : var;
        a.running_mean[c] = (1.f - a.momentum) * a.running_mean[c] + a.momentum * (float)m;
        a.running_var[c] = (1.f - a.momentum) * a.running_var[c] + a.momentum * (float)unbiased;
    }
}

// SyncBatchNorm forward, after the all-reduce: mean / invstd / running statistics from the all-rank totals (sum x, sum x^2 over Mtot rows).
__global__ __launch_bounds__(BN_THREADS) void bn_from_totals_kernel(const BnArgs a, const double* __restrict__ totals) {
    const int c = blockIdx.x * BN_THREADS + threadIdx.x;
    if (c >= a.C) return;
    if (a.nbt && c == 0) *a.nbt += 1;
    const double n = a.cnt ? *a.cnt : (double)a.Mtot, m = totals[c] / n, var = fmax(totals[a.C + c] / n - m * m, 0.0);
    a.mean[c] = (float)m;
    a.invstd[c] = (float)(1.0 / sqrt(var + (double)a.eps));
    if (a.running_mean) {
        const double unbiased = n > 1.0 ? var * n / (n - 1.0) : var;
        a.running_mean[c] = (1.f - a.momentum) * a.running_mean[c] + a.momentum * (float)m;
        a.running_var[c] = (1.f - a.momentum) * a.running_var[c] + a.momentum * (float)unbiased;
    }
}

// Eval-mode BatchNorm under autograd: the statistics ARE the running statistics (read, never updated) - mean / invstd in the form the apply
// kernels and the backward read them.
__global__ __launch_bounds__(BN_THREADS) void bn_from_running_kernel(const BnArgs a) {
    const int c = blockIdx.x * BN_THREADS + threadIdx.x;
    if (c >= a.C) return;
    a.mean[c] = a.running_mean[c];
    a.invstd[c] = (float)(1.0 / sqrt((double)a.running_var[c] + (double)a.eps));
}

__global__ __launch_bounds__(BN_THREADS) void bn_stats_kernel(const BnArgs a) {
    int c, rg, RG;
    bn_coords(a.C, c, rg, RG);
    double acc[8] = {0, 0, 0, 0, 0, 0, 0, 0};
    if (c >= 0) {
        const long long step = (long long)gridDim.x * RG;
        for (long long r0 = (long long)blockIdx.x * RG + rg; r0 < a.M; r0 += 4 * step) {      // 4 independent loads in flight per thread
            float4 v[4];
#pragma unroll
            for (int u = 0; u < 4; ++u) {
                const long long r = r0 + u * step;
                v[u] = r < a.M ? *reinterpret_cast<const float4*>(a.x + r * a.ldx + c) : make_float4(0.f, 0.f, 0.f, 0.f);
            }
#pragma unroll
            for (int u = 0; u < 4; ++u) {
                acc[0] += v[u].x; acc[1] += v[u].y; acc[2] += v[u].z; acc[3] += v[u].w;
                acc[4] += (double)v[u].x * v[u].x; acc[5] += (double)v[u].y * v[u].y; acc[6] += (double)v[u].z * v[u].z; acc[7] += (double)v[u].w * v[u].w;
            }
        }
    }
    bn_block_sum(acc, c, rg, RG, a.C, a.ws);
}

// scale / shift of 4 channels from the float64 sums (forward) - also what the backward recomputes, bit for bit, from mean / invstd
__device__ __forceinline__ void bn_affine4(const BnArgs& a, int c, const float* mean, const float* invstd, float (&sc)[4], float (&sh)[4]) {
#pragma unroll
    for (int k = 0; k < 4; ++k) {
        const float g = a.gamma ? a.gamma[c + k] : 1.f, b = a.beta ? a.beta[c + k] : 0.f;
        sc[k] = g * invstd[k];
        sh[k] = b - mean[k] * sc[k];
    }
}

__global__ __launch_bounds__(BN_THREADS) void bn_apply_fwd_kernel(const BnArgs a) {
    int c, rg, RG;
    bn_coords(a.C, c, rg, RG);
    if (c < 0) return;
    float mean[4], invstd[4], sc[4], sh[4];
#pragma unroll
    for (int k = 0; k < 4; ++k) { mean[k] = a.mean[c + k]; invstd[k] = a.invstd[c + k]; }
    bn_affine4(a, c, mean, invstd, sc, sh);
    for (long long r = (long long)blockIdx.x * RG + rg; r < a.M; r += (long long)gridDim.x * RG) {
        const float4 v = *reinterpret_cast<const float4*>(a.x + r * a.ldx + c);
        float y[4] = {fmaf(v.x, sc[0], sh[0]), fmaf(v.y, sc[1], sh[1]), fmaf(v.z, sc[2], sh[2]), fmaf(v.w, sc[3], sh[3])};
        if (a.res) {
            const float4 q = *reinterpret_cast<const float4*>(a.res + r * a.ldres + c);
            y[0] += q.x; y[1] += q.y; y[2] += q.z; y[3] += q.w;
        }
#pragma unroll
        for (int k = 0; k < 4; ++k) y[k] = y[k] > 0.f ? y[k] : y[k] * a.slope;
        *reinterpret_cast<float4*>(a.out + r * a.ldo + c) = make_float4(y[0], y[1], y[2], y[3]);
    }
}

__global__ __launch_bounds__(BN_THREADS) void bn_reduce_bwd_kernel(const BnArgs a) {
    int c, rg, RG;
    bn_coords(a.C, c, rg, RG);
    double acc[8] = {0, 0, 0, 0, 0, 0, 0, 0};
    if (c >= 0) {
        float mean[4], invstd[4], sc[4], sh[4];
#pragma unroll
        for (int k = 0; k < 4; ++k) { mean[k] = a.mean[c + k]; invstd[k] = a.invstd[c + k]; }
        bn_affine4(a, c, mean, invstd, sc, sh);
        const long long step = (long long)gridDim.x * RG;
        for (long long r0 = (long long)blockIdx.x * RG + rg; r0 < a.M; r0 += 2 * step) {      // 2 x 2 independent loads in flight per thread
            float4 v[2], d[2], yo[2];
#pragma unroll
            for (int u = 0; u < 2; ++u) {
                const long long r = r0 + u * step;
                const bool ok = r < a.M;
                v[u] = ok ? *reinterpret_cast<const float4*>(a.x + r * a.ldx + c) : make_float4(0.f, 0.f, 0.f, 0.f);
                d[u] = ok ? *reinterpret_cast<const float4*>(a.dy + r * a.lddy + c) : make_float4(0.f, 0.f, 0.f, 0.f);
                yo[u] = (ok && a.y) ? *reinterpret_cast<const float4*>(a.y + r * a.ldy + c) : make_float4(0.f, 0.f, 0.f, 0.f);
            }
#pragma unroll
            for (int u = 0; u < 2; ++u) {
                const float xv[4] = {v[u].x, v[u].y, v[u].z, v[u].w}, dv[4] = {d[u].x, d[u].y, d[u].z, d[u].w}, yv[4] = {yo[u].x, yo[u].y, yo[u].z, yo[u].w};
#pragma unroll
                for (int k = 0; k < 4; ++k) {
                    const float pre = a.y ? yv[k] : fmaf(xv[k], sc[k], sh[k]);                      // residual form: the sign of the saved output
                    const float g = pre > 0.f ? dv[k] : dv[k] * a.slope;                            // dy = 0 beyond M: contributes nothing
                    acc[k] += g;
                    acc[4 + k] += (double)g * ((xv[k] - mean[k]) * invstd[k]);
                }
            }
        }
    }
    bn_block_sum(acc, c, rg, RG, a.C, a.ws);
}

__global__ __launch_bounds__(BN_THREADS) void bn_apply_bwd_kernel(const BnArgs a) {
    int c, rg, RG;
    bn_coords(a.C, c, rg, RG);
    if (c < 0) return;
    float mean[4], invstd[4], sc[4], sh[4], k1[4], k2[4], k3[4];
#pragma unroll
    for (int k = 0; k < 4; ++k) { mean[k] = a.mean[c + k]; invstd[k] = a.invstd[c + k]; }
    bn_affine4(a, c, mean, invstd, sc, sh);
#pragma unroll
    for (int k = 0; k < 4; ++k) {
        const double db = a.ws[c + k], dg = a.ws[a.C + c + k];
        k1[k] = sc[k];                                  // gamma * invstd
        const double n = a.cnt ? *a.cnt : (double)a.Mtot;
        k2[k] = (float)(db / n);                        // mean(g)      over all rows the statistics were taken over (all ranks under SyncBN)
        k3[k] = (float)(dg / n);                        // mean(g * xhat)
    }
    for (long long r = (long long)blockIdx.x * RG + rg; r < a.M; r += (long long)gridDim.x * RG) {
        const float4 v = *reinterpret_cast<const float4*>(a.x + r * a.ldx + c);
        const float4 d = *reinterpret_cast<const float4*>(a.dy + r * a.lddy + c);
        const float4 yo = a.y ? *reinterpret_cast<const float4*>(a.y + r * a.ldy + c) : make_float4(0.f, 0.f, 0.f, 0.f);
        const float xv[4] = {v.x, v.y, v.z, v.w}, dv[4] = {d.x, d.y, d.z, d.w}, yv[4] = {yo.x, yo.y, yo.z, yo.w};
        float o[4], gg[4];
#pragma unroll
        for (int k = 0; k < 4; ++k) {
            const float pre = a.y ? yv[k] : fmaf(xv[k], sc[k], sh[k]);
            const float g = pre > 0.f ? dv[k] : dv[k] * a.slope;
            gg[k] = g;
            o[k] = k1[k] * (g - k2[k] - (xv[k] - mean[k]) * invstd[k] * k3[k]);
        }
        *reinterpret_cast<float4*>(a.out + r * a.ldo + c) = make_float4(o[0], o[1], o[2], o[3]);
        if (a.dres) *reinterpret_cast<float4*>(a.dres + r * a.lddres + c) = make_float4(gg[0], gg[1], gg[2], gg[3]);
    }
}

// Column sums of a DENSE [M][C] matrix for any small C (the 1- and 3-channel outputs of the density head / conv_rgb, whose bias gradients torch's
// generic reduction took 0.86 ms for at 40 x 256^2 x 3): the matrix is one flat array whose element i belongs to channel i % C; every thread
// walks it with a stride that is a multiple of C, so its channel is fixed and ONE float64 accumulator suffices. Partials -> ws[block][2][C]
// (second half zero) for bn_finalize_kernel. Deterministic.
__global__ __launch_bounds__(BN_THREADS) void colsum_flat_kernel(const float* __restrict__ x, double* __restrict__ ws, long long n, int C, long long stride) {
    __shared__ double red[BN_THREADS];
    const long long t0 = (long long)blockIdx.x * BN_THREADS + threadIdx.x;
    double acc = 0.0;
    if (t0 < stride)                                                 // stride = the thread count rounded DOWN to a multiple of C: the last few threads idle
        for (long long i = t0; i < n; i += stride) acc += (double)x[i];
    red[threadIdx.x] = acc;
    __syncthreads();
    // thread c < C sums the block's lanes of channel c in lane order; a lane's channel = (blockIdx.x * 256 + lane) % C
    if ((int)threadIdx.x < C) {
        const int c = (int)threadIdx.x;
        int first = (int)((c - ((long long)blockIdx.x * BN_THREADS) % C + C) % C);
        double s = 0.0;
        for (int l = first; l < BN_THREADS; l += C) s += red[l];
        ws[(long long)blockIdx.x * 2 * C + c] = s;
        ws[(long long)blockIdx.x * 2 * C + C + c] = 0.0;
    }
}

static int bn_check(const char* fn, const void* x, int ldx, long long M, int C, const void* ws) {
    FORGE_REQUIRE(x && ws, FORGE_EINVAL, "%s: null pointer argument", fn);
    FORGE_REQUIRE(M > 0 && C > 0 && C % 4 == 0 && ldx >= C && ldx % 4 == 0, FORGE_ESHAPE, "%s: M=%lld C=%d ld=%d (C, ld multiples of 4)", fn, M, C, ldx);
    return 0;
}

// reduce = true (the two kernels that leave one partial per block): at most BN_MAX_PARTIALS blocks, each streaming >= 4 rows per row group;
// the apply kernels take up to 2048 blocks.
constexpr int BN_MAX_PARTIALS = 1024;
static dim3 bn_grid(long long M, int C, bool reduce) {
    const int C4 = C / 4, CQ = C4 < BN_THREADS ? C4 : BN_THREADS, RG = BN_THREADS / CQ;
    const long long per = reduce ? 4 : 2, cap = reduce ? BN_MAX_PARTIALS : 2048;
    long long bx = (M + RG * per - 1) / (RG * per);
    if (bx > cap) bx = cap;
    if (bx < 1) bx = 1;
    return dim3((unsigned)bx, (unsigned)((C4 + CQ - 1) / CQ));
}

}  // namespace forge

using namespace forge;

extern "C" int forge_bn_ws_doubles(int C) { return 2 * C * BN_MAX_PARTIALS; }   // size of the float64 scratch the two calls below need

// Column sums of a row-major [M][C] matrix (row stride ldx): the bias gradient of a convolution, sum over the GEMM rows of dy
// (torch: dy.sum(dim = (0, 2, 3, 4)) as a generic reduce kernel of 20-28 us per call). The statistics pass of the BatchNorm kernels above -
// float64 partial sums, one partial per block, summed in a fixed order: deterministic - with the sum of squares discarded. ws: forge_bn_ws_doubles(C).
extern "C" int forge_colsum(const float* x, int ldx, float* out, double* ws, long long M, int C, forge_stream_t stream) {
    if (C > 0 && C <= 32 && (C % 4 != 0 || ldx % 4 != 0) && ldx == C) {
        // dense narrow matrix: the flat walk (any C <= 32)
        FORGE_REQUIRE(x && ws && out && M > 0, FORGE_EINVAL, "forge_colsum: null pointer argument / M <= 0");
        BnArgs a;
        memset(&a, 0, sizeof(a));
        a.ws = ws; a.M = M; a.Mtot = M; a.C = C; a.dbeta = out;
        const long long n = M * C;
        long long nb = (n + BN_THREADS * 16 - 1) / (BN_THREADS * 16);
        if (nb > BN_MAX_PARTIALS) nb = BN_MAX_PARTIALS;
        if (nb < 1) nb = 1;
        a.nblk = (int)nb;
        const long long stride = nb * BN_THREADS / C * C;                       // multiple of C: a thread never changes channel
        hipStream_t st = (hipStream_t)stream;
        hipLaunchKernelGGL(colsum_flat_kernel, dim3((unsigned)nb), dim3(BN_THREADS), 0, st, x, ws, n, C, stride);
        hipLaunchKernelGGL(bn_finalize_kernel, dim3((unsigned)((C + 3) / 4)), dim3(BN_THREADS), 0, st, a, 1);
        FORGE_LAUNCH_CHECK("forge_colsum");
        return 0;
    }
    if (int rc = bn_check("forge_colsum", x, ldx, M, C, ws)) return rc;
    FORGE_REQUIRE(out, FORGE_EINVAL, "forge_colsum: null output pointer");
    BnArgs a;
    memset(&a, 0, sizeof(a));
    a.x = x; a.ldx = ldx; a.ws = ws; a.M = M; a.Mtot = M; a.C = C; a.dbeta = out;
    const dim3 gr = bn_grid(M, C, true);
    a.nblk = (int)gr.x;
    hipStream_t st = (hipStream_t)stream;
    hipLaunchKernelGGL(bn_stats_kernel, gr, dim3(BN_THREADS), 0, st, a);
    hipLaunchKernelGGL(bn_finalize_kernel, dim3((unsigned)((C + 3) / 4)), dim3(BN_THREADS), 0, st, a, 1);       // "backward" flavour: dbeta = total of the first sum
    FORGE_LAUNCH_CHECK("forge_colsum");
    return 0;
}

static int bn_check_side(const char* fn, const char* what, const void* p, int ld, int C) {
    FORGE_REQUIRE(p == nullptr || (ld >= C && ld % 4 == 0), FORGE_ESHAPE, "%s: %s row stride %d (>= C = %d, multiple of 4)", fn, what, ld, C);
    return 0;
}

extern "C" int forge_bn_train_fwd(const float* x, int ldx, const float* gamma, const float* beta, float eps, float slope, float* y, int ldy,
                                  float* mean, float* invstd, float* running_mean, float* running_var, float momentum, double* ws,
                                  long long M, int C, const float* res, int ldres, long long* num_batches_tracked, int nblk_pre,
                                  forge_stream_t stream) {
    if (int rc = bn_check("forge_bn_train_fwd", x, ldx, M, C, ws)) return rc;
    FORGE_REQUIRE(y && mean && invstd && ldy >= C && ldy % 4 == 0 && (running_mean == nullptr) == (running_var == nullptr), FORGE_EINVAL,
                  "forge_bn_train_fwd: bad output / running-statistics arguments");
    BnArgs a;
    memset(&a, 0, sizeof(a));
    a.x = x; a.ldx = ldx; a.out = y; a.ldo = ldy; a.gamma = gamma; a.beta = beta; a.mean = mean; a.invstd = invstd;
    a.running_mean = running_mean; a.running_var = running_var; a.momentum = momentum; a.ws = ws; a.eps = eps; a.slope = slope; a.M = M; a.C = C; a.Mtot = M;
    if (int rc = bn_check_side("forge_bn_train_fwd", "residual", res, ldres, C)) return rc;
    a.res = res; a.ldres = ldres; a.nbt = num_batches_tracked;
    hipStream_t st = (hipStream_t)stream;
    FORGE_REQUIRE(nblk_pre >= 0, FORGE_EINVAL, "forge_bn_train_fwd: nblk_pre < 0");
    if (nblk_pre > 0) {
        a.nblk = nblk_pre;                                         // ws already holds nblk_pre partial rows [2][C] (the producing convolution's epilogue)
    } else {
        const dim3 gr = bn_grid(M, C, true);
        a.nblk = (int)gr.x;
        hipLaunchKernelGGL(bn_stats_kernel, gr, dim3(BN_THREADS), 0, st, a);
    }
    hipLaunchKernelGGL(bn_finalize_kernel, dim3((unsigned)((C + 3) / 4)), dim3(BN_THREADS), 0, st, a, 0);
    hipLaunchKernelGGL(bn_apply_fwd_kernel, bn_grid(M, C, false), dim3(BN_THREADS), 0, st, a);
    FORGE_LAUNCH_CHECK("forge_bn_train_fwd");
    return 0;
}

extern "C" int forge_bn_train_bwd(const float* dy, int lddy, const float* x, int ldx, const float* gamma, const float* beta, const float* mean,
                                  const float* invstd, float slope, float* dx, int lddx, float* dgamma, float* dbeta, double* ws, long long M, int C,
                                  const float* y, int ldy, float* dres, int lddres, forge_stream_t stream) {
    if (int rc = bn_check("forge_bn_train_bwd", x, ldx, M, C, ws)) return rc;
    if (int rc = bn_check_side("forge_bn_train_bwd", "y", y, ldy, C)) return rc;
    if (int rc = bn_check_side("forge_bn_train_bwd", "dres", dres, lddres, C)) return rc;
    FORGE_REQUIRE(dres == nullptr || y != nullptr, FORGE_EINVAL, "forge_bn_train_bwd: the residual form needs the forward's output y");
    FORGE_REQUIRE(dy && dx && mean && invstd && lddy >= C && lddy % 4 == 0 && lddx >= C && lddx % 4 == 0, FORGE_EINVAL, "forge_bn_train_bwd: bad arguments");
    BnArgs a;
    memset(&a, 0, sizeof(a));
    a.x = x; a.ldx = ldx; a.dy = dy; a.lddy = lddy; a.out = dx; a.ldo = lddx; a.gamma = gamma; a.beta = beta;
    a.mean = const_cast<float*>(mean); a.invstd = const_cast<float*>(invstd); a.dgamma = dgamma; a.dbeta = dbeta; a.ws = ws; a.slope = slope; a.M = M; a.C = C; a.Mtot = M;
    a.y = y; a.ldy = ldy; a.dres = dres; a.lddres = lddres;
    hipStream_t st = (hipStream_t)stream;
    const dim3 gr = bn_grid(M, C, true);
    a.nblk = (int)gr.x;
    hipLaunchKernelGGL(bn_reduce_bwd_kernel, gr, dim3(BN_THREADS), 0, st, a);
    hipLaunchKernelGGL(bn_finalize_kernel, dim3((unsigned)((C + 3) / 4)), dim3(BN_THREADS), 0, st, a, 1);
    hipLaunchKernelGGL(bn_apply_bwd_kernel, bn_grid(M, C, false), dim3(BN_THREADS), 0, st, a);
    FORGE_LAUNCH_CHECK("forge_bn_train_bwd");
    return 0;
}

// ---- SyncBatchNorm (torch.nn.SyncBatchNorm.convert_sync_batchnorm, kubric_train_pose_3D.py:119): the same kernels with ONE all-reduce of
// 2 C (+1) doubles between the reduction and the apply step, issued by the caller (forge_amd/fusion.py over torch.distributed = RCCL):
//   forward   forge_bn_sync_stats       -> ws[0 .. 2C) = this rank's (sum x, sum x^2)           [all-reduce SUM together with the row count]
//             forge_bn_sync_fwd_apply   mean / invstd / running statistics from the all-rank totals, y = lrelu(x scale + shift)
//   backward  forge_bn_sync_bwd_reduce  -> ws[0 .. 2C) = this rank's (sum g, sum g xhat); dgamma / dbeta = the LOCAL sums (as torch's
//                                          SyncBatchNorm: DDP averages parameter gradients afterwards)   [all-reduce SUM]
//             forge_bn_sync_bwd_apply   dx = gamma invstd (g - mean_all(g) - xhat mean_all(g xhat))
extern "C" int forge_bn_sync_stats(const float* x, int ldx, double* ws, long long M, int C, int nblk_pre, forge_stream_t stream) {
    if (int rc = bn_check("forge_bn_sync_stats", x, ldx, M, C, ws)) return rc;
    FORGE_REQUIRE(nblk_pre >= 0, FORGE_EINVAL, "forge_bn_sync_stats: nblk_pre < 0");
    BnArgs a;
    memset(&a, 0, sizeof(a));
    a.x = x; a.ldx = ldx; a.ws = ws; a.M = M; a.C = C; a.Mtot = M;
    hipStream_t st = (hipStream_t)stream;
    if (nblk_pre > 0) {
        a.nblk = nblk_pre;
    } else {
        const dim3 gr = bn_grid(M, C, true);
        a.nblk = (int)gr.x;
        hipLaunchKernelGGL(bn_stats_kernel, gr, dim3(BN_THREADS), 0, st, a);
    }
    hipLaunchKernelGGL(bn_finalize_kernel, dim3((unsigned)((C + 3) / 4)), dim3(BN_THREADS), 0, st, a, 2);
    FORGE_LAUNCH_CHECK("forge_bn_sync_stats");
    return 0;
}

extern "C" int forge_bn_sync_fwd_apply(const float* x, int ldx, const float* gamma, const float* beta, float eps, float slope, float* y, int ldy,
                                       float* mean, float* invstd, float* running_mean, float* running_var, float momentum, const double* totals,
                                       long long M_total, long long M, int C, const float* res, int ldres, long long* num_batches_tracked,
                                       forge_stream_t stream) {
    if (int rc = bn_check("forge_bn_sync_fwd_apply", x, ldx, M, C, totals)) return rc;
    if (int rc = bn_check_side("forge_bn_sync_fwd_apply", "residual", res, ldres, C)) return rc;
    FORGE_REQUIRE(y && mean && invstd && ldy >= C && ldy % 4 == 0 && (running_mean == nullptr) == (running_var == nullptr) && (M_total == 0 || M_total >= M), FORGE_EINVAL,
                  "forge_bn_sync_fwd_apply: bad output / running-statistics arguments or 0 < M_total < M");
    BnArgs a;
    memset(&a, 0, sizeof(a));
    a.x = x; a.ldx = ldx; a.out = y; a.ldo = ldy; a.gamma = gamma; a.beta = beta; a.mean = mean; a.invstd = invstd;
    a.running_mean = running_mean; a.running_var = running_var; a.momentum = momentum; a.eps = eps; a.slope = slope; a.M = M; a.C = C; a.Mtot = M_total;
    a.cnt = M_total == 0 ? totals + 2 * C : nullptr;
    a.res = res; a.ldres = ldres; a.nbt = num_batches_tracked;
    hipStream_t st = (hipStream_t)stream;
    hipLaunchKernelGGL(bn_from_totals_kernel, dim3((unsigned)((C + BN_THREADS - 1) / BN_THREADS)), dim3(BN_THREADS), 0, st, a, totals);
    hipLaunchKernelGGL(bn_apply_fwd_kernel, bn_grid(M, C, false), dim3(BN_THREADS), 0, st, a);
    FORGE_LAUNCH_CHECK("forge_bn_sync_fwd_apply");
    return 0;
}

extern "C" int forge_bn_eval_fwd(const float* x, int ldx, const float* gamma, const float* beta, const float* running_mean, const float* running_var,
                                 float eps, float slope, float* y, int ldy, float* mean, float* invstd, long long M, int C, const float* res, int ldres,
                                 forge_stream_t stream) {
    if (int rc = bn_check("forge_bn_eval_fwd", x, ldx, M, C, running_mean)) return rc;
    if (int rc = bn_check_side("forge_bn_eval_fwd", "residual", res, ldres, C)) return rc;
    FORGE_REQUIRE(y && mean && invstd && running_var && ldy >= C && ldy % 4 == 0, FORGE_EINVAL, "forge_bn_eval_fwd: null output / running-statistics pointer or bad ldy");
    BnArgs a;
    memset(&a, 0, sizeof(a));
    a.x = x; a.ldx = ldx; a.out = y; a.ldo = ldy; a.gamma = gamma; a.beta = beta; a.mean = mean; a.invstd = invstd;
    a.running_mean = const_cast<float*>(running_mean); a.running_var = const_cast<float*>(running_var);      // read only (bn_from_running_kernel)
    a.eps = eps; a.slope = slope; a.M = M; a.C = C; a.Mtot = M; a.res = res; a.ldres = ldres;
    hipStream_t st = (hipStream_t)stream;
    hipLaunchKernelGGL(bn_from_running_kernel, dim3((unsigned)((C + BN_THREADS - 1) / BN_THREADS)), dim3(BN_THREADS), 0, st, a);
    hipLaunchKernelGGL(bn_apply_fwd_kernel, bn_grid(M, C, false), dim3(BN_THREADS), 0, st, a);
    FORGE_LAUNCH_CHECK("forge_bn_eval_fwd");
    return 0;
}

extern "C" int forge_bn_sync_bwd_reduce(const float* dy, int lddy, const float* x, int ldx, const float* gamma, const float* beta, const float* mean,
                                        const float* invstd, float slope, float* dgamma, float* dbeta, double* ws, long long M, int C,
                                        const float* y, int ldy, forge_stream_t stream) {
    if (int rc = bn_check("forge_bn_sync_bwd_reduce", x, ldx, M, C, ws)) return rc;
    if (int rc = bn_check_side("forge_bn_sync_bwd_reduce", "y", y, ldy, C)) return rc;
    FORGE_REQUIRE(dy && mean && invstd && lddy >= C && lddy % 4 == 0, FORGE_EINVAL, "forge_bn_sync_bwd_reduce: bad arguments");
    BnArgs a;
    memset(&a, 0, sizeof(a));
    a.x = x; a.ldx = ldx; a.dy = dy; a.lddy = lddy; a.gamma = gamma; a.beta = beta; a.mean = const_cast<float*>(mean);
    a.invstd = const_cast<float*>(invstd); a.dgamma = dgamma; a.dbeta = dbeta; a.ws = ws; a.slope = slope; a.M = M; a.C = C; a.Mtot = M;
    a.y = y; a.ldy = ldy;
    hipStream_t st = (hipStream_t)stream;
    const dim3 gr = bn_grid(M, C, true);
    a.nblk = (int)gr.x;
    hipLaunchKernelGGL(bn_reduce_bwd_kernel, gr, dim3(BN_THREADS), 0, st, a);
    hipLaunchKernelGGL(bn_finalize_kernel, dim3((unsigned)((C + 3) / 4)), dim3(BN_THREADS), 0, st, a, 1);
    FORGE_LAUNCH_CHECK("forge_bn_sync_bwd_reduce");
    return 0;
}

extern "C" int forge_bn_sync_bwd_apply(const float* dy, int lddy, const float* x, int ldx, const float* gamma, const float* beta, const float* mean,
                                       const float* invstd, float slope, float* dx, int lddx, const double* totals, long long M_total, long long M, int C,
                                       const float* y, int ldy, float* dres, int lddres, forge_stream_t stream) {
    if (int rc = bn_check("forge_bn_sync_bwd_apply", x, ldx, M, C, totals)) return rc;
    if (int rc = bn_check_side("forge_bn_sync_bwd_apply", "y", y, ldy, C)) return rc;
    if (int rc = bn_check_side("forge_bn_sync_bwd_apply", "dres", dres, lddres, C)) return rc;
    FORGE_REQUIRE(dres == nullptr || y != nullptr, FORGE_EINVAL, "forge_bn_sync_bwd_apply: the residual form needs the forward's output y");
    FORGE_REQUIRE(dy && dx && mean && invstd && lddy >= C && lddy % 4 == 0 && lddx >= C && lddx % 4 == 0 && (M_total == 0 || M_total >= M), FORGE_EINVAL,
                  "forge_bn_sync_bwd_apply: bad arguments");
    BnArgs a;
    memset(&a, 0, sizeof(a));
    a.x = x; a.ldx = ldx; a.dy = dy; a.lddy = lddy; a.out = dx; a.ldo = lddx; a.gamma = gamma; a.beta = beta;
    a.mean = const_cast<float*>(mean); a.invstd = const_cast<float*>(invstd); a.ws = const_cast<double*>(totals); a.slope = slope; a.M = M; a.C = C;
    a.Mtot = M_total;
    a.cnt = M_total == 0 ? totals + 2 * C : nullptr;
    a.y = y; a.ldy = ldy; a.dres = dres; a.lddres = lddres;
    hipLaunchKernelGGL(bn_apply_bwd_kernel, bn_grid(M, C, false), dim3(BN_THREADS), 0, (hipStream_t)stream, a);
    FORGE_LAUNCH_CHECK("forge_bn_sync_bwd_apply");
    return 0;
}
