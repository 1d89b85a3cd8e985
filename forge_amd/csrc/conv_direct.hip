// conv_direct.hip — direct (non-matrix-core) convolution for TINY channel counts: Cin in {4, 8, 16}, Cout <= 4.
//
// The density head ends in Conv3d(8, 1, 3) (models/encoder.py:31) and conv_rgb in Conv2d(8, 3, 5) (models/volume_render.py:36).
// On the MFMA kernels such a layer is padded to a 32 x 32 (or 16 x 16) channel tile: the 8 -> 1 convolution then executes 128x its
// useful FLOPs, and in the training path its forward + data-gradient + weight-gradient cost 11 ms per step at 4 scenes / GPU for
// 0.7 GFLOP of real work. These layers are HBM/L2-streaming problems (27 x 8 MACs per voxel), so they run on the vector ALUs:
//   fwd    thread = output voxel: out[m][co] = b[co] + sum_t sum_ci w[t][co][ci] in[m + tap_t][ci]            (weights in LDS)
//   dgrad  thread = input voxel:  dx[m][ci] = sum_t sum_co w[t][co][ci] dy[m - tap_t][co]
//   wgrad  thread = voxel (grid-stride), one group of <= 8 taps per workgroup row: per-thread partial sums of
//          dw[t][co][ci] += dy[m][co] x[m + tap_t][ci] in registers, wave + workgroup reduction, one fp32 atomic per entry
// Rows are channels-last [M][ld]; stride-1 "same" geometry (taps are offsets inside the same (n, D, H, W) grid, zero outside).
#include "common.h"

namespace forge {

struct DirectArgs {
    const float* a; const float* b; float* o;     // fwd: in, -, out | dgrad: dy, -, dx | wgrad: dy, x, dw
    const float* w; const float* bias; float slope;          // fwd: out = lrelu(conv + bias, slope) (1 = none, 0 = ReLU)
    int lda, ldb, ldo;
    int n, D, H, W, ntaps;
    signed char tap[64][4];
};

constexpr int DMAX_W = 64 * 4 * 16;               // taps x Cout x Cin floats of weights in LDS

template <int CI4, int CO, bool DGRAD>
__global__ __launch_bounds__(256) void conv_direct_kernel(const DirectArgs a) {
    constexpr int CI = CI4 * 4;
    __shared__ float ws[DMAX_W];
    for (int i = threadIdx.x; i < a.ntaps * CO * CI; i += 256) ws[i] = a.w[i];
    __syncthreads();
    const long long M = (long long)a.n * a.D * a.H * a.W;
    const long long m = (long long)blockIdx.x * 256 + threadIdx.x;
    if (m >= M) return;
    long long v = m;
    const int x = (int)(v % a.W); v /= a.W;
    const int y = (int)(v % a.H); v /= a.H;
    const int z = (int)(v % a.D);
    if constexpr (!DGRAD) {
        float acc[CO];
#pragma unroll
        for (int c = 0; c < CO; ++c) acc[c] = a.bias ? a.bias[c] : 0.f;
        for (int t = 0; t < a.ntaps; ++t) {
            const int dz = a.tap[t][0], dy = a.tap[t][1], dx = a.tap[t][2];
            if ((unsigned)(z + dz) >= (unsigned)a.D || (unsigned)(y + dy) >= (unsigned)a.H || (unsigned)(x + dx) >= (unsigned)a.W) continue;
            const float* row = a.a + (m + ((long long)dz * a.H + dy) * a.W + dx) * a.lda;
            const float* wt = ws + t * CO * CI;
#pragma unroll
            for (int q = 0; q < CI4; ++q) {
                const float4 f = *reinterpret_cast<const float4*>(row + 4 * q);
#pragma unroll
                for (int c = 0; c < CO; ++c) {
                    const float* wc = wt + c * CI + 4 * q;
                    acc[c] = fmaf(f.x, wc[0], fmaf(f.y, wc[1], fmaf(f.z, wc[2], fmaf(f.w, wc[3], acc[c]))));
                }
            }
        }
#pragma unroll
        for (int c = 0; c < CO; ++c) a.o[m * a.ldo + c] = acc[c] > 0.f ? acc[c] : acc[c] * a.slope;
    } else {
        float4 acc[CI4];
#pragma unroll
        for (int q = 0; q < CI4; ++q) acc[q] = make_float4(0.f, 0.f, 0.f, 0.f);
        for (int t = 0; t < a.ntaps; ++t) {
            const int dz = -a.tap[t][0], dy = -a.tap[t][1], dx = -a.tap[t][2];     // dy of the OUTPUT voxel that read this input through tap t
            if ((unsigned)(z + dz) >= (unsigned)a.D || (unsigned)(y + dy) >= (unsigned)a.H || (unsigned)(x + dx) >= (unsigned)a.W) continue;
            const float* row = a.a + (m + ((long long)dz * a.H + dy) * a.W + dx) * a.lda;
            const float* wt = ws + t * CO * CI;
#pragma unroll
            for (int c = 0; c < CO; ++c) {
                const float g = row[c];
#pragma unroll
                for (int q = 0; q < CI4; ++q) {
                    const float* wc = wt + c * CI + 4 * q;
                    acc[q].x = fmaf(g, wc[0], acc[q].x); acc[q].y = fmaf(g, wc[1], acc[q].y);
                    acc[q].z = fmaf(g, wc[2], acc[q].z); acc[q].w = fmaf(g, wc[3], acc[q].w);
                }
            }
        }
#pragma unroll
        for (int q = 0; q < CI4; ++q) *reinterpret_cast<float4*>(a.o + m * a.ldo + 4 * q) = acc[q];
    }
}

// taps per workgroup row of the wgrad grid: ~48 partial sums (taps x Cout x Cin) per thread, at most 4 taps. Measured on conv_rgb's 8 -> 3 5x5
// layer / the density head's 8 -> 1 3x3x3 layer (4-scene training step): 8 taps (192 sums: 1 wave per SIMD with the batched loads) 1571 us / 245 us,
// 5 taps 630 / -, 4 taps 592 / 170, 3 taps - / 187, 2 taps 407 / -, 1 tap 477 / -: occupancy beats the re-reads of dy.
__host__ __device__ constexpr int wg_taps(int ci, int co) { return 48 / (ci * co) > 4 ? 4 : (48 / (ci * co) < 1 ? 1 : 48 / (ci * co)); }

template <int CI4, int CO>
__global__ __launch_bounds__(256) void conv_direct_wgrad_kernel(const DirectArgs a) {
    constexpr int CI = CI4 * 4, WG_TAPS = wg_taps(CI, CO), NACC = WG_TAPS * CO * CI;
    __shared__ float red[4][NACC];
    const int t0 = blockIdx.y * WG_TAPS;
    const int nt = min(WG_TAPS, a.ntaps - t0);
    const long long M = (long long)a.n * a.D * a.H * a.W;
    float acc[NACC];
#pragma unroll
    for (int i = 0; i < NACC; ++i) acc[i] = 0.f;
    for (long long m = (long long)blockIdx.x * 256 + threadIdx.x; m < M; m += (long long)gridDim.x * 256) {
        long long v = m;
        const int x = (int)(v % a.W); v /= a.W;
        const int y = (int)(v % a.H); v /= a.H;
        const int z = (int)(v % a.D);
        float g[CO];
#pragma unroll
        for (int c = 0; c < CO; ++c) g[c] = a.a[m * a.lda + c];
        // every tap's input row is loaded unconditionally (out-of-grid taps read this voxel's own row and are zeroed by a select): no
        // divergent branch sits between the loads, so all of a voxel's WG_TAPS x CI4 loads are in flight together (with a `continue` per
        // out-of-grid tap the kernel ran one load -> FMA chain at a time: 0.82 ms for conv_rgb's 8 -> 3 5x5 layer)
        float4 fx[WG_TAPS][CI4];
#pragma unroll
        for (int tt = 0; tt < WG_TAPS; ++tt) {
            const int ti = t0 + (tt < nt ? tt : 0);
            const int dz = a.tap[ti][0], dy = a.tap[ti][1], dx = a.tap[ti][2];
            const bool ok = tt < nt && (unsigned)(z + dz) < (unsigned)a.D && (unsigned)(y + dy) < (unsigned)a.H && (unsigned)(x + dx) < (unsigned)a.W;
            const float* row = a.b + (ok ? m + ((long long)dz * a.H + dy) * a.W + dx : m) * a.ldb;
#pragma unroll
            for (int q = 0; q < CI4; ++q) {
                const float4 f = *reinterpret_cast<const float4*>(row + 4 * q);
                fx[tt][q] = ok ? f : make_float4(0.f, 0.f, 0.f, 0.f);
            }
        }
#pragma unroll
        for (int tt = 0; tt < WG_TAPS; ++tt) {
#pragma unroll
            for (int q = 0; q < CI4; ++q) {
                const float4 f = fx[tt][q];
#pragma unroll
                for (int c = 0; c < CO; ++c) {
                    float* ac = acc + (tt * CO + c) * CI + 4 * q;
                    ac[0] = fmaf(g[c], f.x, ac[0]); ac[1] = fmaf(g[c], f.y, ac[1]);
                    ac[2] = fmaf(g[c], f.z, ac[2]); ac[3] = fmaf(g[c], f.w, ac[3]);
                }
            }
        }
    }
    const int lane = threadIdx.x & 63, wv = threadIdx.x >> 6;
#pragma unroll
    for (int i = 0; i < NACC; ++i) {
        float s = acc[i];
#pragma unroll
        for (int o = 32; o > 0; o >>= 1) s += __shfl_down(s, o, 64);
        if (lane == 0) red[wv][i] = s;
    }
    __syncthreads();
    for (int i = threadIdx.x; i < nt * CO * CI; i += 256) {
        const float s = red[0][i] + red[1][i] + red[2][i] + red[3][i];
        if (s != 0.f) atomic_add_f32(a.o + (long long)t0 * CO * CI + i, s);
    }
}

static int fill_direct(const char* fn, DirectArgs& a, int n, int D, int H, int W, int Cin, int Cout, const int* taps, int ntaps) {
    FORGE_REQUIRE(n > 0 && D > 0 && H > 0 && W > 0 && ntaps > 0 && ntaps <= 64 && taps, FORGE_EINVAL, "%s: bad dims", fn);
    FORGE_REQUIRE((Cin == 4 || Cin == 8 || Cin == 16) && Cout >= 1 && Cout <= 4, FORGE_ESHAPE,
                  "%s: Cin=%d Cout=%d outside the direct kernels' range (Cin 4/8/16, Cout 1..4)", fn, Cin, Cout);
    a.n = n; a.D = D; a.H = H; a.W = W; a.ntaps = ntaps;
    for (int t = 0; t < 64; ++t) {
        for (int k = 0; k < 3; ++k) a.tap[t][k] = (signed char)(t < ntaps ? taps[t * 3 + k] : 0);
        a.tap[t][3] = 0;
    }
    return 0;
}

#define FORGE_DIRECT_DISPATCH(Cin, Cout, ...)                                                            \
    switch (((Cin) / 4) * 8 + (Cout)) {                                                                  \
        case 1 * 8 + 1: { constexpr int CI4 = 1, CO = 1; __VA_ARGS__; } break;                           \
        case 1 * 8 + 2: { constexpr int CI4 = 1, CO = 2; __VA_ARGS__; } break;                           \
        case 1 * 8 + 3: { constexpr int CI4 = 1, CO = 3; __VA_ARGS__; } break;                           \
        case 1 * 8 + 4: { constexpr int CI4 = 1, CO = 4; __VA_ARGS__; } break;                           \
        case 2 * 8 + 1: { constexpr int CI4 = 2, CO = 1; __VA_ARGS__; } break;                           \
        case 2 * 8 + 2: { constexpr int CI4 = 2, CO = 2; __VA_ARGS__; } break;                           \
        case 2 * 8 + 3: { constexpr int CI4 = 2, CO = 3; __VA_ARGS__; } break;                           \
        case 2 * 8 + 4: { constexpr int CI4 = 2, CO = 4; __VA_ARGS__; } break;                           \
        case 4 * 8 + 1: { constexpr int CI4 = 4, CO = 1; __VA_ARGS__; } break;                           \
        case 4 * 8 + 2: { constexpr int CI4 = 4, CO = 2; __VA_ARGS__; } break;                           \
        case 4 * 8 + 3: { constexpr int CI4 = 4, CO = 3; __VA_ARGS__; } break;                           \
        default: { constexpr int CI4 = 4, CO = 4; __VA_ARGS__; } break;                                  \
    }

}  // namespace forge

using namespace forge;

extern "C" int forge_conv_direct_fwd(const float* in, int ld_in, const float* w, const float* bias, float slope, float* out, int ld_out,
                                     int n, int D, int H, int W, int Cin, int Cout, const int* taps, int ntaps, forge_stream_t stream) {
    FORGE_REQUIRE(in && w && out, FORGE_EINVAL, "forge_conv_direct_fwd: null pointer argument");
    DirectArgs a;
    if (int rc = fill_direct("forge_conv_direct_fwd", a, n, D, H, W, Cin, Cout, taps, ntaps)) return rc;
    FORGE_REQUIRE(ld_in >= Cin && ld_in % 4 == 0 && ld_out >= Cout, FORGE_ESHAPE, "forge_conv_direct_fwd: bad row strides");
    a.a = in; a.b = nullptr; a.o = out; a.w = w; a.bias = bias; a.slope = slope; a.lda = ld_in; a.ldb = 0; a.ldo = ld_out;
    const long long M = (long long)n * D * H * W;
    FORGE_DIRECT_DISPATCH(Cin, Cout, hipLaunchKernelGGL((conv_direct_kernel<CI4, CO, false>), dim3((unsigned)((M + 255) / 256)), dim3(256), 0,
                                                        (hipStream_t)stream, a));
    FORGE_LAUNCH_CHECK("forge_conv_direct_fwd");
    return 0;
}

extern "C" int forge_conv_direct_dgrad(const float* dy, int ld_dy, const float* w, float* dx, int ld_dx,
                                       int n, int D, int H, int W, int Cin, int Cout, const int* taps, int ntaps, forge_stream_t stream) {
    FORGE_REQUIRE(dy && w && dx, FORGE_EINVAL, "forge_conv_direct_dgrad: null pointer argument");
    DirectArgs a;
    if (int rc = fill_direct("forge_conv_direct_dgrad", a, n, D, H, W, Cin, Cout, taps, ntaps)) return rc;
    FORGE_REQUIRE(ld_dy >= Cout && ld_dx >= Cin && ld_dx % 4 == 0, FORGE_ESHAPE, "forge_conv_direct_dgrad: bad row strides");
    a.a = dy; a.b = nullptr; a.o = dx; a.w = w; a.bias = nullptr; a.slope = 1.f; a.lda = ld_dy; a.ldb = 0; a.ldo = ld_dx;
    const long long M = (long long)n * D * H * W;
    FORGE_DIRECT_DISPATCH(Cin, Cout, hipLaunchKernelGGL((conv_direct_kernel<CI4, CO, true>), dim3((unsigned)((M + 255) / 256)), dim3(256), 0,
                                                        (hipStream_t)stream, a));
    FORGE_LAUNCH_CHECK("forge_conv_direct_dgrad");
    return 0;
}

extern "C" int forge_conv_direct_wgrad(const float* dy, int ld_dy, const float* x, int ld_x, float* dw,
                                       int n, int D, int H, int W, int Cin, int Cout, const int* taps, int ntaps, forge_stream_t stream) {
    FORGE_REQUIRE(dy && x && dw, FORGE_EINVAL, "forge_conv_direct_wgrad: null pointer argument");
    DirectArgs a;
    if (int rc = fill_direct("forge_conv_direct_wgrad", a, n, D, H, W, Cin, Cout, taps, ntaps)) return rc;
    FORGE_REQUIRE(ld_dy >= Cout && ld_x >= Cin && ld_x % 4 == 0, FORGE_ESHAPE, "forge_conv_direct_wgrad: bad row strides");
    a.a = dy; a.b = x; a.o = dw; a.w = nullptr; a.bias = nullptr; a.slope = 1.f; a.lda = ld_dy; a.ldb = ld_x; a.ldo = 0;
    const long long M = (long long)n * D * H * W;
    long long gx = (M + 255) / 256;
    if (gx > 2048) gx = 2048;                                     // grid-stride: every workgroup ends with a reduction + atomics
    const int tg = wg_taps(Cin, Cout);
    const dim3 grid((unsigned)gx, (unsigned)((ntaps + tg - 1) / tg));
    FORGE_DIRECT_DISPATCH(Cin, Cout, hipLaunchKernelGGL((conv_direct_wgrad_kernel<CI4, CO>), grid, dim3(256), 0, (hipStream_t)stream, a));
    FORGE_LAUNCH_CHECK("forge_conv_direct_wgrad");
    return 0;
}
