// winograd.hip — the 3x3x3 convolutions of the ConvGRU fusion (models/fusion.py:29-35, 61-68, 88-95) with 2.25x fewer multiplies:
// Winograd F(2x2, 3x3) over (H, W), the three depth taps kept as a direct sum.
//
//   y[z, 2th + i, 2tw + j] = sum_kd  A^T [ (G w[kd] G^T) (.) (B^T d[z + kd - 1] B) ] A,    d = the 4x4 input patch at rows 2th - 1 .., cols 2tw - 1 ..
//
//   B^T = [1 0 -1 0; 0 1 1 0; 0 -1 1 0; 0 1 0 -1]    G = [1 0 0; .5 .5 .5; .5 -.5 .5; 0 0 1]    A^T = [1 1 1 0; 0 1 -1 -1]
//
// so that the channel contraction becomes 16 independent GEMMs (one per transformed point p = 4 i + j) over the TILE grid
// (n, D, H/2, W/2) with K = 3 C_in: 12 multiplies per output and channel pair instead of 27. Three launches per convolution:
//   wino_input_kernel    V[p][r][c]  = (B^T d B)[p]           HBM-bound: reads each input row ~once (4x through L2), writes 4x its size
//   forge_wino_gemm      Mm[p] = V[p] (x) U[p]                conv_igemm_kernel, 16 batched 3-tap problems (conv_igemm.hip); fp32 MFMA
//   wino_output_kernel   y = epilogue(A^T Mm A)               HBM-bound: reads 4x the output size, fuses the same element-wise tails
//                                                             as the direct kernel (bias, folded BN + LeakyReLU, GRU gates / state update)
// B^T and A^T hold only 0 / +-1: the transforms are exact additions; the only extra rounding relative to the direct kernel is in
// U = G w G^T (rounded once, on the host, from a float64 product) and in the order of the fp32 additions. Measured against a float64
// convolution the fp32 error is 1.4x that of a direct fp32 convolution (tests/test_gpu_winograd.py).
#include "common.h"

namespace forge {

__device__ __forceinline__ float4 f4_add(float4 a, float4 b) { return make_float4(a.x + b.x, a.y + b.y, a.z + b.z, a.w + b.w); }
__device__ __forceinline__ float4 f4_sub(float4 a, float4 b) { return make_float4(a.x - b.x, a.y - b.y, a.z - b.z, a.w - b.w); }

struct WinoInArgs {
    const float* in; int ld; long long bs;        // rows [n][D][H][W] x ld floats, batch stride bs rows
    float* V; int ldv; long long ptv;             // V[p] = V + p ptv, rows [n][D][H/2][W/2] x ldv floats
    int n, D, H, W, C;
    int nsum; long long ss;                       // nsum > 1: the input is the MEAN of nsum tensors ss rows apart (the view mean of models/encoder.py:62
                                                  // feeding fusion_conv: sum in view order, then x (1 / nsum), as torch.mean) - no separate reduction launch
    float* dM; long long ptm;                     // DY instantiation: also dM[p] = (A y A^T)[p] of the tile's own 2 x 2 pixels y (rows [n][D][H/2][W/2] x C)
};

// one thread = one tile x 4 channels: 16 float4 loads (zero outside the grid), 32 float4 additions, 16 float4 stores.
// DY: the tensor is an upstream gradient that the backward pass needs in BOTH transformed forms - B^T d B for the data-gradient GEMMs and
// A y A^T (wino_dy_kernel) for the weight-gradient GEMMs; the 2 x 2 pixels of the latter are the centre of the patch already in registers,
// so one launch writes both (9 instead of 10 passes of the tensor's size, one launch less per gradient).
template <bool DY>
__global__ __launch_bounds__(256) void wino_input_kernel(const WinoInArgs a) {
    const int C4 = a.C >> 2, Ht = a.H >> 1, Wt = a.W >> 1;
    // each XCD transforms one CONTIGUOUS slab of tiles: neighbouring tiles share half of their 4 x 4 input patches, and with the dispatcher's
    // round-robin order every XCD's private L2 fetched most of the input again (r03 PMC: 102.5 MB moved for 83.9 MB). Placement only.
    const long long idx = (long long)xcd_remap(blockIdx.x, gridDim.x) * 256 + threadIdx.x;
    const long long R = (long long)a.n * a.D * Ht * Wt;
    if (idx >= R * C4) return;
    const unsigned r = (unsigned)(idx / C4);
    const int c = (int)(idx - (long long)r * C4) << 2;
    unsigned q = r, t = q / (unsigned)Wt;
    const int tw = (int)(q - t * (unsigned)Wt); q = t; t = q / (unsigned)Ht;
    const int th = (int)(q - t * (unsigned)Ht); q = t; t = q / (unsigned)a.D;
    const int z = (int)(q - t * (unsigned)a.D);
    const float* base = a.in + ((long long)t * a.bs + (long long)z * a.H * a.W) * a.ld + c;
    float4 d[4][4];
#pragma unroll
    for (int i = 0; i < 4; ++i) {
        const int y = 2 * th - 1 + i;
#pragma unroll
        for (int j = 0; j < 4; ++j) {
            const int x = 2 * tw - 1 + j;
            const bool ok = (unsigned)y < (unsigned)a.H && (unsigned)x < (unsigned)a.W;
            float4 v = make_float4(0.f, 0.f, 0.f, 0.f);
            if (ok) {
                const float* p = base + ((long long)y * a.W + x) * a.ld;
                v = *reinterpret_cast<const float4*>(p);
                if (a.nsum > 1) {
                    for (int k = 1; k < a.nsum; ++k) v = f4_add(v, *reinterpret_cast<const float4*>(p + (long long)k * a.ss * a.ld));
                    const float inv = 1.f / (float)a.nsum;           // ATen divides by a scalar as a multiplication by its fp32 reciprocal
                    v = make_float4(v.x * inv, v.y * inv, v.z * inv, v.w * inv);
                }
            }
            d[i][j] = v;
        }
    }
    if constexpr (DY) {
        const float4 zero = make_float4(0.f, 0.f, 0.f, 0.f);
        float* mp = a.dM + (long long)r * a.C + c;
#pragma unroll
        for (int i = 0; i < 4; ++i) {                // rows of A y: y0, y0 + y1, y0 - y1, -y1 (y = d[1..2][1..2]); then (A y) A^T
            const float4 s0 = i == 0 ? d[1][1] : i == 1 ? f4_add(d[1][1], d[2][1]) : i == 2 ? f4_sub(d[1][1], d[2][1]) : f4_sub(zero, d[2][1]);
            const float4 s1 = i == 0 ? d[1][2] : i == 1 ? f4_add(d[1][2], d[2][2]) : i == 2 ? f4_sub(d[1][2], d[2][2]) : f4_sub(zero, d[2][2]);
            *reinterpret_cast<float4*>(mp + (4 * i + 0) * a.ptm) = s0;
            *reinterpret_cast<float4*>(mp + (4 * i + 1) * a.ptm) = f4_add(s0, s1);
            *reinterpret_cast<float4*>(mp + (4 * i + 2) * a.ptm) = f4_sub(s0, s1);
            *reinterpret_cast<float4*>(mp + (4 * i + 3) * a.ptm) = f4_sub(zero, s1);
        }
    }
    float4 w[4][4];                                  // rows: B^T d
#pragma unroll
    for (int j = 0; j < 4; ++j) {
        w[0][j] = f4_sub(d[0][j], d[2][j]);
        w[1][j] = f4_add(d[1][j], d[2][j]);
        w[2][j] = f4_sub(d[2][j], d[1][j]);
        w[3][j] = f4_sub(d[1][j], d[3][j]);
    }
    float* vp = a.V + (long long)r * a.ldv + c;
#pragma unroll
    for (int i = 0; i < 4; ++i) {                    // columns: (B^T d) B
        *reinterpret_cast<float4*>(vp + (4 * i + 0) * a.ptv) = f4_sub(w[i][0], w[i][2]);
        *reinterpret_cast<float4*>(vp + (4 * i + 1) * a.ptv) = f4_add(w[i][1], w[i][2]);
        *reinterpret_cast<float4*>(vp + (4 * i + 2) * a.ptv) = f4_sub(w[i][2], w[i][1]);
        *reinterpret_cast<float4*>(vp + (4 * i + 3) * a.ptv) = f4_sub(w[i][1], w[i][3]);
    }
}

// U[4i+j][kd][o][c] = (G w[kd] G^T)[i][j] from packed weights wp[(kd 3 + a) 3 + b][co][ci]: one thread per (kd, o, c). Evaluated in float64
// (exact: at most 9 fp32 terms with coefficients 1, 1/2, 1/4) and rounded once. transpose: the weights of the DATA GRADIENT - the
// correlation of dy with the flipped kernel and swapped channel roles: w'[kd][a][b][o = ci][c = co] = wp[(2-kd, 2-a, 2-b)][co][ci].
__global__ __launch_bounds__(256) void wino_weight_kernel(const float* __restrict__ wp, float* __restrict__ U, int Cout, int Cin, int KD, int transpose) {
    const int No = transpose ? Cin : Cout, Nc = transpose ? Cout : Cin;      // U's [o][c] extents
    const long long idx = (long long)blockIdx.x * 256 + threadIdx.x, per = (long long)No * Nc;
    if (idx >= KD * per) return;                      // KD = 3 depth taps (3x3x3 kernels) or 1 (3x3 kernels of a 2-D convolution)
    const int kd = (int)(idx / per);
    const long long oc = idx - kd * per;
    const int o = (int)(oc / Nc), c = (int)(oc - (long long)o * Nc);
    double w[3][3];
#pragma unroll
    for (int a = 0; a < 3; ++a)
#pragma unroll
        for (int b = 0; b < 3; ++b) {
            const int tap = transpose ? ((KD - 1 - kd) * 3 + (2 - a)) * 3 + (2 - b) : (kd * 3 + a) * 3 + b;
            w[a][b] = (double)wp[((long long)tap * Cout + (transpose ? c : o)) * Cin + (transpose ? o : c)];
        }
    double g[4][3];                                  // G w
#pragma unroll
    for (int b = 0; b < 3; ++b) {
        g[0][b] = w[0][b];
        g[1][b] = 0.5 * (w[0][b] + w[1][b] + w[2][b]);
        g[2][b] = 0.5 * (w[0][b] - w[1][b] + w[2][b]);
        g[3][b] = w[2][b];
    }
    float* up = U + ((long long)kd * No + o) * Nc + c;
    const long long pt = KD * per;
#pragma unroll
    for (int i = 0; i < 4; ++i) {                    // (G w) G^T
        up[(4 * i + 0) * pt] = (float)g[i][0];
        up[(4 * i + 1) * pt] = (float)(0.5 * (g[i][0] + g[i][1] + g[i][2]));
        up[(4 * i + 2) * pt] = (float)(0.5 * (g[i][0] - g[i][1] + g[i][2]));
        up[(4 * i + 3) * pt] = (float)g[i][2];
    }
}

enum WinoEpilogue : int { W_BIAS = 0, W_AFFINE_ACT = 1, W_GRU_GATES = 2, W_GRU_OUT = 3 };   // = ConvEpilogue of conv_igemm.hip

struct WinoOutArgs {
    const float* Mm; long long ptm;               // Mm[p] = Mm + p ptm, rows [n][D][H/2][W/2] x Cout floats
    const float* Mm2; long long ptm2, bs2;        // nullable second addend (the input half of conv([x, h], W), shared across fusions):
                                                  // Mm2[p] = Mm2 + p ptm2, batch element n starts at row n bs2 (rows [D][H/2][W/2] x Cout)
    const float* bias; const float* scale; const float* shift; float slope;
    const float* residual;                        // nullable, [rows][Cout]: added to the pre-activation
    const float* aux_h; const float* aux_z;
    float* out; float* out2; float* out3; int ldo;
    int n, D, H, W, Cout, epi;
};

// one thread = one tile x 4 output channels: 16 float4 loads, 24 float4 additions, then the element-wise tail of 2x2 output voxels
// HALF: Mm (and Mm2) hold 8 planes [2][4] - the row stage s = A^T m was applied by the GEMM's epilogue (forge_wino_gemm_half) - and this kernel runs the
// column stage only (8 instead of 16 float4 loads per operand).
template <int EPI, bool HALF = false>
__global__ __launch_bounds__(256) void wino_output_kernel(const WinoOutArgs a) {
    const int C4 = a.Cout >> 2, Ht = a.H >> 1, Wt = a.W >> 1;
    const long long idx = (long long)blockIdx.x * 256 + threadIdx.x;
    const long long R = (long long)a.n * a.D * Ht * Wt;
    if (idx >= R * C4) return;
    const unsigned r = (unsigned)(idx / C4);
    const int c = (int)(idx - (long long)r * C4) << 2;
    unsigned q = r, t = q / (unsigned)Wt;
    const int tw = (int)(q - t * (unsigned)Wt); q = t; t = q / (unsigned)Ht;
    const int th = (int)(q - t * (unsigned)Ht);                    // q / Ht = (n, z) plane index
    const float* mp = a.Mm + (long long)r * a.Cout + c;
    float4 s[2][4];                                  // A^T m
    if constexpr (HALF) {
#pragma unroll
        for (int i = 0; i < 2; ++i)
#pragma unroll
            for (int j = 0; j < 4; ++j) s[i][j] = *reinterpret_cast<const float4*>(mp + (4 * i + j) * a.ptm);
        if (a.Mm2) {                                  // the second addend in the same 8-plane form (the transform is linear)
            const unsigned R1 = (unsigned)(a.D * Ht * Wt), nn = r / R1;
            const float* mp2 = a.Mm2 + ((long long)nn * a.bs2 + (r - nn * R1)) * a.Cout + c;
#pragma unroll
            for (int i = 0; i < 2; ++i)
#pragma unroll
                for (int j = 0; j < 4; ++j) s[i][j] = f4_add(s[i][j], *reinterpret_cast<const float4*>(mp2 + (4 * i + j) * a.ptm2));
        }
    } else {
        float4 m[4][4];
#pragma unroll
        for (int i = 0; i < 4; ++i)
#pragma unroll
            for (int j = 0; j < 4; ++j) m[i][j] = *reinterpret_cast<const float4*>(mp + (4 * i + j) * a.ptm);
        if (a.Mm2) {
            const unsigned R1 = (unsigned)(a.D * Ht * Wt), nn = r / R1;
            const float* mp2 = a.Mm2 + ((long long)nn * a.bs2 + (r - nn * R1)) * a.Cout + c;
#pragma unroll
            for (int i = 0; i < 4; ++i)
#pragma unroll
                for (int j = 0; j < 4; ++j) m[i][j] = f4_add(m[i][j], *reinterpret_cast<const float4*>(mp2 + (4 * i + j) * a.ptm2));
        }
#pragma unroll
        for (int j = 0; j < 4; ++j) {
            s[0][j] = f4_add(f4_add(m[0][j], m[1][j]), m[2][j]);
            s[1][j] = f4_sub(f4_sub(m[1][j], m[2][j]), m[3][j]);
        }
    }
    float4 y[2][2];
#pragma unroll
    for (int i = 0; i < 2; ++i) {
        y[i][0] = f4_add(f4_add(s[i][0], s[i][1]), s[i][2]);
        y[i][1] = f4_sub(f4_sub(s[i][1], s[i][2]), s[i][3]);
    }
    float4 bias = make_float4(0.f, 0.f, 0.f, 0.f), sc = make_float4(1.f, 1.f, 1.f, 1.f), sh = bias;
    if (a.bias) bias = *reinterpret_cast<const float4*>(a.bias + c);
    if ((EPI == W_AFFINE_ACT || (EPI == W_GRU_OUT && a.out2)) && a.scale) {
        sc = *reinterpret_cast<const float4*>(a.scale + c);
        sh = *reinterpret_cast<const float4*>(a.shift + c);
    }
    const int Ch = a.Cout >> 1;
    const long long plane = (long long)t * a.H * a.W;               // first output row of plane (n, z)
#pragma unroll
    for (int i = 0; i < 2; ++i)
#pragma unroll
        for (int j = 0; j < 2; ++j) {
            const long long orow = plane + (long long)(2 * th + i) * a.W + (2 * tw + j);
            float v[4] = {y[i][j].x + bias.x, y[i][j].y + bias.y, y[i][j].z + bias.z, y[i][j].w + bias.w};
            if (a.residual) {
                const float4 rr = *reinterpret_cast<const float4*>(a.residual + orow * a.Cout + c);
                v[0] += rr.x; v[1] += rr.y; v[2] += rr.z; v[3] += rr.w;
            }
            const float scv[4] = {sc.x, sc.y, sc.z, sc.w}, shv[4] = {sh.x, sh.y, sh.z, sh.w};
            if constexpr (EPI == W_BIAS) {
                *reinterpret_cast<float4*>(a.out + orow * a.ldo + c) = make_float4(v[0], v[1], v[2], v[3]);
            } else if constexpr (EPI == W_AFFINE_ACT) {
#pragma unroll
                for (int k = 0; k < 4; ++k) {
                    const float u = fmaf(v[k], scv[k], shv[k]);
                    v[k] = u > 0.f ? u : u * a.slope;
                }
                *reinterpret_cast<float4*>(a.out + orow * a.ldo + c) = make_float4(v[0], v[1], v[2], v[3]);
            } else if constexpr (EPI == W_GRU_GATES) {
                float g[4];
#pragma unroll
                for (int k = 0; k < 4; ++k) g[k] = 1.f / (1.f + expf(-v[k]));      // full-precision exponential, as torch.sigmoid
                if (c < Ch) {
                    *reinterpret_cast<float4*>(a.out + orow * Ch + c) = make_float4(g[0], g[1], g[2], g[3]);
                } else {
                    const float4 h = *reinterpret_cast<const float4*>(a.aux_h + orow * Ch + (c - Ch));
                    *reinterpret_cast<float4*>(a.out2 + orow * Ch + (c - Ch)) = make_float4(h.x * g[0], h.y * g[1], h.z * g[2], h.w * g[3]);
                    if (a.out3) *reinterpret_cast<float4*>(a.out3 + orow * Ch + (c - Ch)) = make_float4(g[0], g[1], g[2], g[3]);
                }
            } else {   // W_GRU_OUT
                const float4 z4 = *reinterpret_cast<const float4*>(a.aux_z + orow * a.Cout + c);
                const float4 h4 = *reinterpret_cast<const float4*>(a.aux_h + orow * a.Cout + c);
                const float zv[4] = {z4.x, z4.y, z4.z, z4.w}, hv[4] = {h4.x, h4.y, h4.z, h4.w};
                float cand[4], hn[4];
#pragma unroll
                for (int k = 0; k < 4; ++k) {
                    cand[k] = tanhf(v[k]);
                    hn[k] = hv[k] * (1.f - zv[k]) + cand[k] * zv[k];
                }
                *reinterpret_cast<float4*>(a.out + orow * a.ldo + c) = make_float4(hn[0], hn[1], hn[2], hn[3]);
                if (a.out2)
                    *reinterpret_cast<float4*>(a.out2 + orow * a.ldo + c) =
                        make_float4(fmaf(hn[0], scv[0], shv[0]), fmaf(hn[1], scv[1], shv[1]), fmaf(hn[2], scv[2], shv[2]), fmaf(hn[3], scv[3], shv[3]));
                if (a.out3) *reinterpret_cast<float4*>(a.out3 + orow * a.ldo + c) = make_float4(cand[0], cand[1], cand[2], cand[3]);
            }
        }
}

// ---- weight gradient in the Winograd domain: dL/dU[p][kd] = sum_r dMm[p][r] (x) V[p][r + kd plane] (forge_wino_wgrad: conv_wgrad_kernel
// on 16 batched problems), with dMm = A dy A^T the adjoint of the inverse transform, then dL/dw[kd] = G^T dU[kd] G.
struct WinoDyArgs {
    const float* dy; int ldy;                     // upstream gradient rows [n][D][H][W] x ldy floats, Cout channels used
    float* dM; long long ptm;                     // dM[p] = dM + p ptm, rows [n][D][H/2][W/2] x Cout floats
    int n, D, H, W, Cout;
};

// one thread = one 2x2 output tile x 4 channels: 4 float4 loads, A = [1 0; 1 1; 1 -1; 0 -1] on both sides, 16 float4 stores
__global__ __launch_bounds__(256) void wino_dy_kernel(const WinoDyArgs a) {
    const int C4 = a.Cout >> 2, Ht = a.H >> 1, Wt = a.W >> 1;
    const long long idx = (long long)blockIdx.x * 256 + threadIdx.x;
    const long long R = (long long)a.n * a.D * Ht * Wt;
    if (idx >= R * C4) return;
    const unsigned r = (unsigned)(idx / C4);
    const int c = (int)(idx - (long long)r * C4) << 2;
    unsigned q = r, t = q / (unsigned)Wt;
    const int tw = (int)(q - t * (unsigned)Wt); q = t; t = q / (unsigned)Ht;
    const int th = (int)(q - t * (unsigned)Ht);                    // t = (n, z) plane index
    const float* yp = a.dy + ((long long)t * a.H * a.W + (long long)(2 * th) * a.W + 2 * tw) * a.ldy + c;
    float4 y[2][2];
#pragma unroll
    for (int i = 0; i < 2; ++i)
#pragma unroll
        for (int j = 0; j < 2; ++j) y[i][j] = *reinterpret_cast<const float4*>(yp + ((long long)i * a.W + j) * a.ldy);
    const float4 zero = make_float4(0.f, 0.f, 0.f, 0.f);
    float4 s[4][2];                                  // A y
#pragma unroll
    for (int j = 0; j < 2; ++j) {
        s[0][j] = y[0][j];
        s[1][j] = f4_add(y[0][j], y[1][j]);
        s[2][j] = f4_sub(y[0][j], y[1][j]);
        s[3][j] = f4_sub(zero, y[1][j]);
    }
    float* mp = a.dM + (long long)r * a.Cout + c;
#pragma unroll
    for (int i = 0; i < 4; ++i) {                    // (A y) A^T
        *reinterpret_cast<float4*>(mp + (4 * i + 0) * a.ptm) = s[i][0];
        *reinterpret_cast<float4*>(mp + (4 * i + 1) * a.ptm) = f4_add(s[i][0], s[i][1]);
        *reinterpret_cast<float4*>(mp + (4 * i + 2) * a.ptm) = f4_sub(s[i][0], s[i][1]);
        *reinterpret_cast<float4*>(mp + (4 * i + 3) * a.ptm) = f4_sub(zero, s[i][1]);
    }
}

// dw[(kd 3 + a) 3 + b][co][ci] += sum_ij G[i][a] G[j][b] dU[4i+j][kd][co][ci]   (G^T dU G): one thread per (kd, co, ci); accumulates into dw
// like forge_conv_wgrad (each element is touched by exactly one thread: deterministic)
__global__ __launch_bounds__(256) void wino_dw_kernel(const float* __restrict__ dU, float* __restrict__ dw, int Cout, int Cin, int KD) {
    const long long idx = (long long)blockIdx.x * 256 + threadIdx.x, per = (long long)Cout * Cin;
    if (idx >= KD * per) return;
    const int kd = (int)(idx / per);
    const long long oc = idx - kd * per;
    const long long pt = KD * per;
    float u[4][4];
#pragma unroll
    for (int i = 0; i < 4; ++i)
#pragma unroll
        for (int j = 0; j < 4; ++j) u[i][j] = dU[(4 * i + j) * pt + kd * per + oc];
    float g[3][4];                                   // G^T u: rows a = 0..2
#pragma unroll
    for (int j = 0; j < 4; ++j) {
        g[0][j] = u[0][j] + 0.5f * (u[1][j] + u[2][j]);
        g[1][j] = 0.5f * (u[1][j] - u[2][j]);
        g[2][j] = 0.5f * (u[1][j] + u[2][j]) + u[3][j];
    }
#pragma unroll
    for (int a_ = 0; a_ < 3; ++a_) {                 // (G^T u) G
        float* o = dw + ((long long)((kd * 3 + a_) * 3) * per) + oc;
        o[0 * per] += g[a_][0] + 0.5f * (g[a_][1] + g[a_][2]);
        o[1 * per] += 0.5f * (g[a_][1] - g[a_][2]);
        o[2 * per] += 0.5f * (g[a_][1] + g[a_][2]) + g[a_][3];
    }
}

}  // namespace forge

using namespace forge;

extern "C" int forge_wino_dy(const float* dy, int ldy, float* dM, int n, int D, int H, int W, int Cout, forge_stream_t stream) {
    FORGE_REQUIRE(dy && dM, FORGE_EINVAL, "forge_wino_dy: null pointer argument");
    FORGE_REQUIRE(n > 0 && D > 0 && H > 0 && W > 0 && H % 2 == 0 && W % 2 == 0 && Cout > 0 && Cout % 4 == 0 && ldy >= Cout && ldy % 4 == 0, FORGE_ESHAPE,
                  "forge_wino_dy: n=%d D=%d H=%d W=%d Cout=%d ldy=%d (H, W even; Cout, ldy multiples of 4)", n, D, H, W, Cout, ldy);
    WinoDyArgs a;
    const long long R = (long long)n * D * (H / 2) * (W / 2);
    a.dy = dy; a.ldy = ldy; a.dM = dM; a.ptm = R * Cout; a.n = n; a.D = D; a.H = H; a.W = W; a.Cout = Cout;
    FORGE_REQUIRE(R < (1ll << 31), FORGE_ESHAPE, "forge_wino_dy: more than 2^31 tiles; split the batch");
    const long long total = R * (Cout / 4), grid = (total + 255) / 256;
    FORGE_REQUIRE(grid < (1ll << 31), FORGE_ESHAPE, "forge_wino_dy: grid too large");
    hipLaunchKernelGGL(wino_dy_kernel, dim3((unsigned)grid), dim3(256), 0, (hipStream_t)stream, a);
    FORGE_LAUNCH_CHECK("forge_wino_dy");
    return 0;
}

extern "C" int forge_wino_dw(const float* dU, float* dw, int Cout, int Cin, int kd, forge_stream_t stream) {
    FORGE_REQUIRE(dU && dw && Cout > 0 && Cin > 0 && (kd == 1 || kd == 3), FORGE_EINVAL, "forge_wino_dw: bad argument (kd = 1 or 3)");
    const long long total = (long long)kd * Cout * Cin;
    hipLaunchKernelGGL(wino_dw_kernel, dim3((unsigned)((total + 255) / 256)), dim3(256), 0, (hipStream_t)stream, dU, dw, Cout, Cin, kd);
    FORGE_LAUNCH_CHECK("forge_wino_dw");
    return 0;
}

extern "C" int forge_wino_weights(const float* wp, float* U, int Cout, int Cin, int kd, int transpose, forge_stream_t stream) {
    FORGE_REQUIRE(wp && U && Cout > 0 && Cin > 0 && (kd == 1 || kd == 3), FORGE_EINVAL, "forge_wino_weights: bad argument (kd = 1 or 3)");
    const long long total = (long long)kd * Cout * Cin;
    hipLaunchKernelGGL(wino_weight_kernel, dim3((unsigned)((total + 255) / 256)), dim3(256), 0, (hipStream_t)stream, wp, U, Cout, Cin, kd, transpose);
    FORGE_LAUNCH_CHECK("forge_wino_weights");
    return 0;
}

extern "C" int forge_wino_input(const float* in, int ld, long long bs, float* V, int ldv, long long ptv, int n, int D, int H, int W, int C,
                                int nsum, long long sum_stride, forge_stream_t stream) {
    FORGE_REQUIRE(in && V, FORGE_EINVAL, "forge_wino_input: null pointer argument");
    FORGE_REQUIRE(nsum >= 1 && (nsum == 1 || sum_stride > 0), FORGE_EINVAL, "forge_wino_input: nsum >= 1, and a positive sum_stride (rows) with nsum > 1");
    FORGE_REQUIRE(n > 0 && D > 0 && H > 0 && W > 0 && H % 2 == 0 && W % 2 == 0 && C > 0 && C % 4 == 0 && ld >= C && ld % 4 == 0 && ldv >= C && ldv % 4 == 0,
                  FORGE_ESHAPE, "forge_wino_input: n=%d D=%d H=%d W=%d C=%d ld=%d ldv=%d (H, W even; C, ld, ldv multiples of 4)", n, D, H, W, C, ld, ldv);
    WinoInArgs a;
    a.in = in; a.ld = ld; a.bs = bs > 0 ? bs : (long long)D * H * W; a.V = V; a.ldv = ldv; a.n = n; a.D = D; a.H = H; a.W = W; a.C = C;
    a.nsum = nsum; a.ss = sum_stride;
    const long long R = (long long)n * D * (H / 2) * (W / 2);
    a.ptv = ptv > 0 ? ptv : R * ldv;
    FORGE_REQUIRE(R < (1ll << 31), FORGE_ESHAPE, "forge_wino_input: more than 2^31 tiles; split the batch");
    const long long total = R * (C / 4), grid = (total + 255) / 256;
    FORGE_REQUIRE(grid < (1ll << 31), FORGE_ESHAPE, "forge_wino_input: grid too large");
    a.dM = nullptr; a.ptm = 0;
    hipLaunchKernelGGL(wino_input_kernel<false>, dim3((unsigned)grid), dim3(256), 0, (hipStream_t)stream, a);
    FORGE_LAUNCH_CHECK("forge_wino_input");
    return 0;
}

extern "C" int forge_wino_input_dy(const float* dy, int ld, float* V, float* dM, int n, int D, int H, int W, int C, forge_stream_t stream) {
    FORGE_REQUIRE(dy && V && dM, FORGE_EINVAL, "forge_wino_input_dy: null pointer argument");
    FORGE_REQUIRE(n > 0 && D > 0 && H > 0 && W > 0 && H % 2 == 0 && W % 2 == 0 && C > 0 && C % 4 == 0 && ld >= C && ld % 4 == 0, FORGE_ESHAPE,
                  "forge_wino_input_dy: n=%d D=%d H=%d W=%d C=%d ld=%d (H, W even; C, ld multiples of 4)", n, D, H, W, C, ld);
    WinoInArgs a;
    a.in = dy; a.ld = ld; a.bs = (long long)D * H * W; a.V = V; a.ldv = C; a.n = n; a.D = D; a.H = H; a.W = W; a.C = C; a.nsum = 1; a.ss = 0;
    const long long R = (long long)n * D * (H / 2) * (W / 2);
    a.ptv = R * C; a.dM = dM; a.ptm = R * C;
    FORGE_REQUIRE(R < (1ll << 31), FORGE_ESHAPE, "forge_wino_input_dy: more than 2^31 tiles; split the batch");
    const long long total = R * (C / 4), grid = (total + 255) / 256;
    FORGE_REQUIRE(grid < (1ll << 31), FORGE_ESHAPE, "forge_wino_input_dy: grid too large");
    hipLaunchKernelGGL(wino_input_kernel<true>, dim3((unsigned)grid), dim3(256), 0, (hipStream_t)stream, a);
    FORGE_LAUNCH_CHECK("forge_wino_input_dy");
    return 0;
}

static int wino_output_impl(const float* Mm, const float* Mm2, long long bs2, long long pt2, const float* bias, const float* scale, const float* shift, float slope, const float* residual,
                            const float* aux_h, const float* aux_z, float* out, float* out2, float* out3, int n, int D, int H, int W, int Cout,
                            int ldo, int epilogue, bool half, forge_stream_t stream) {
    FORGE_REQUIRE(Mm && out, FORGE_EINVAL, "forge_wino_output: null pointer argument");
    FORGE_REQUIRE(n > 0 && D > 0 && H > 0 && W > 0 && H % 2 == 0 && W % 2 == 0 && Cout > 0 && Cout % 8 == 0 && ldo % 4 == 0, FORGE_ESHAPE,
                  "forge_wino_output: n=%d D=%d H=%d W=%d Cout=%d ldo=%d (H, W even; Cout multiple of 8; ldo of 4)", n, D, H, W, Cout, ldo);
    FORGE_REQUIRE(epilogue >= 0 && epilogue <= 3, FORGE_EINVAL, "forge_wino_output: unknown epilogue %d", epilogue);
    FORGE_REQUIRE(epilogue != W_AFFINE_ACT || (scale && shift), FORGE_EINVAL, "forge_wino_output: affine epilogue needs scale/shift");
    FORGE_REQUIRE(epilogue != W_GRU_GATES || (aux_h && out2), FORGE_EINVAL, "forge_wino_output: GRU gate epilogue needs aux_h, out2");
    FORGE_REQUIRE(epilogue != W_GRU_OUT || (aux_h && aux_z && (!out2 || (scale && shift))), FORGE_EINVAL,
                  "forge_wino_output: GRU out epilogue needs aux_h, aux_z (and scale/shift with out2)");
    FORGE_REQUIRE(out3 == nullptr || epilogue == W_GRU_GATES || epilogue == W_GRU_OUT, FORGE_EINVAL, "forge_wino_output: out3 is a GRU-epilogue output");
    WinoOutArgs a;
    const long long R = (long long)n * D * (H / 2) * (W / 2);
    a.Mm = Mm; a.ptm = R * Cout; a.Mm2 = Mm2; a.bs2 = bs2 > 0 ? bs2 : (long long)D * (H / 2) * (W / 2); a.ptm2 = pt2 > 0 ? pt2 : R * Cout; a.bias = bias; a.scale = scale; a.shift = shift; a.slope = slope; a.residual = residual; a.aux_h = aux_h; a.aux_z = aux_z;
    a.out = out; a.out2 = out2; a.out3 = out3; a.ldo = ldo; a.n = n; a.D = D; a.H = H; a.W = W; a.Cout = Cout; a.epi = epilogue;
    FORGE_REQUIRE(R < (1ll << 31), FORGE_ESHAPE, "forge_wino_output: more than 2^31 tiles; split the batch");
    const long long total = R * (Cout / 4), grid = (total + 255) / 256;
    FORGE_REQUIRE(grid < (1ll << 31), FORGE_ESHAPE, "forge_wino_output: grid too large");
    const dim3 g((unsigned)grid), b(256);
    hipStream_t st = (hipStream_t)stream;
    if (half) {
        switch (epilogue) {
            case W_BIAS: hipLaunchKernelGGL((wino_output_kernel<W_BIAS, true>), g, b, 0, st, a); break;
            case W_AFFINE_ACT: hipLaunchKernelGGL((wino_output_kernel<W_AFFINE_ACT, true>), g, b, 0, st, a); break;
            case W_GRU_GATES: hipLaunchKernelGGL((wino_output_kernel<W_GRU_GATES, true>), g, b, 0, st, a); break;
            default: hipLaunchKernelGGL((wino_output_kernel<W_GRU_OUT, true>), g, b, 0, st, a); break;
        }
    } else {
        switch (epilogue) {
            case W_BIAS: hipLaunchKernelGGL(wino_output_kernel<W_BIAS>, g, b, 0, st, a); break;
            case W_AFFINE_ACT: hipLaunchKernelGGL(wino_output_kernel<W_AFFINE_ACT>, g, b, 0, st, a); break;
            case W_GRU_GATES: hipLaunchKernelGGL(wino_output_kernel<W_GRU_GATES>, g, b, 0, st, a); break;
            default: hipLaunchKernelGGL(wino_output_kernel<W_GRU_OUT>, g, b, 0, st, a); break;
        }
    }
    FORGE_LAUNCH_CHECK("forge_wino_output");
    return 0;
}

extern "C" int forge_wino_output(const float* Mm, const float* Mm2, long long bs2, long long pt2, const float* bias, const float* scale, const float* shift, float slope, const float* residual,
                                 const float* aux_h, const float* aux_z, float* out, float* out2, float* out3, int n, int D, int H, int W, int Cout,
                                 int ldo, int epilogue, forge_stream_t stream) {
    return wino_output_impl(Mm, Mm2, bs2, pt2, bias, scale, shift, slope, residual, aux_h, aux_z, out, out2, out3, n, D, H, W, Cout, ldo, epilogue, false, stream);
}

// The column stage of the inverse transform + the fused tail on forge_wino_gemm_half's 8 planes Mm8 [2][4][R][Cout] (+ a second addend in the same form, as
// forge_wino_output's Mm2): without Mm2_8 bitwise forge_wino_output's result; with it the two operands are row-combined before they are added.
extern "C" int forge_wino_output_half(const float* Mm8, const float* Mm2_8, long long bs2, long long pt2, const float* bias, const float* scale, const float* shift, float slope,
                                      const float* residual, const float* aux_h, const float* aux_z, float* out, float* out2, float* out3, int n, int D, int H, int W,
                                      int Cout, int ldo, int epilogue, forge_stream_t stream) {
    return wino_output_impl(Mm8, Mm2_8, bs2, pt2, bias, scale, shift, slope, residual, aux_h, aux_z, out, out2, out3, n, D, H, W, Cout, ldo, epilogue, true, stream);
}
