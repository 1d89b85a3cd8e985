// rotate.hip — a2: fused voxel-grid pose warp for gfx950 (MI355X).
//
// Replaces models/rotate.py:127-141 of the reference: there PyTorch materialises a homogeneous
// grid [n, D*H*W, 4], multiplies it by T^T, divides, and calls F.grid_sample (trilinear, zeros,
// align_corners=False), then concatenates view 0. Here one launch does all of it: the 3x4 affine
// lives in SGPRs, every thread owns (output voxel, 4 channels), the 8 taps are 16-byte loads of
// channels-last rows (C contiguous floats per voxel => a voxel's 128 channels are one 512-B
// coalesced segment across 32 lanes), view 0 is copied by the same kernel.
//
// Roofline: HBM. Algorithmic bytes per warped view = 2 * C*D*H*W*4 (read once + write once);
// tap re-reads (each source voxel is touched by ~8 outputs) are meant to be served by L1/L2:
// workgroups are remapped so each XCD's L2 sees one contiguous z-slab of the output.
#include "common.h"

namespace forge {

struct TriTaps {
    int x0, y0, z0;          // floor of the pixel coordinate
    float wx0, wx1, wy0, wy1, wz0, wz1;
};

// align_corners=False un-normalisation + trilinear weights, in ATen's operation order
// (grid_sampler_unnormalize: ((coord + 1) * size - 1) / 2 ; weights as (x0+1 - x), (x - x0)).
__device__ __forceinline__ void taps_ac_false(float sx, float sy, float sz, int W, int H, int D, TriTaps& t) {
    float px = ((sx + 1.f) * (float)W - 1.f) * 0.5f;
    float py = ((sy + 1.f) * (float)H - 1.f) * 0.5f;
    float pz = ((sz + 1.f) * (float)D - 1.f) * 0.5f;
    // keep the int conversion defined for wild poses; anything beyond [-2, N+1] has no valid tap
    px = fminf(fmaxf(px, -2.f), (float)W + 1.f);
    py = fminf(fmaxf(py, -2.f), (float)H + 1.f);
    pz = fminf(fmaxf(pz, -2.f), (float)D + 1.f);
    const float fx = floorf(px), fy = floorf(py), fz = floorf(pz);
    t.x0 = (int)fx; t.y0 = (int)fy; t.z0 = (int)fz;
    t.wx1 = px - fx; t.wx0 = (fx + 1.f) - px;
    t.wy1 = py - fy; t.wy0 = (fy + 1.f) - py;
    t.wz1 = pz - fz; t.wz0 = (fz + 1.f) - pz;
}

// Normalised voxel-centre coordinate i -> [-1, 1] of a D-point axis, bit for bit torch.linspace(-1, 1, D)[i] in fp32 (what PyTorch3D's
// Volumes.get_coord_grid, i.e. models/rotate.py:50-51, builds its grid from): step = 2 / (D - 1) rounded once, the lower half counted up from -1,
// the upper half counted down from +1, each value ONE fused multiply-add (ATen's vectorised CPU kernel and the contracted `start + step * i` of its
// GPU kernel both round once; checked against torch.linspace for D = 7 .. 128 in tests/test_abi_and_surface.py). Replaces the IEEE division
// 2 i / (D - 1) - 1 of rounds 1-5, which agreed with linspace only to an ulp - at D = 128 an ulp of the coordinate is 1.5e-5 of a voxel after
// un-normalisation.
__device__ __forceinline__ float linspace_pm1(int i, int D) {
    const float step = 2.f / (float)(D - 1);
    return i < D / 2 ? fmaf(step, (float)i, -1.f) : fmaf(-step, (float)(D - 1 - i), 1.f);
}

// Voxel traversal order: linear index -> (x, y, z). Plain x-fastest order makes a workgroup's neighbours in time (same XCD, same
// L2) a full x-row / z-slice of the OUTPUT, whose SOURCE footprint under a rotation is a tilted slab spanning tens of z-slices
// (44 MB at 64^3 x 128 ch for 20 degrees) - every source row is then re-fetched from HBM for each of its y/z taps. Walking the output
// in 16^3-voxel cubes (8^3 tiles in 2x2x2 groups) keeps the footprint of the ~256 workgroups in flight on an XCD inside its 4 MiB L2.
__device__ __forceinline__ void voxel_of(unsigned v, int W, int H, int D, int& x, int& y, int& z) {
    if (((W | H | D) & 15) == 0) {
        // 8^3 tiles, themselves grouped 2 x 2 x 2: 4096 consecutive voxels = one 16^3 cube (2 MB of source at 128 channels)
        const unsigned w = v & 511u, sub = (v >> 9) & 7u;
        unsigned t = v >> 12;
        const unsigned nsx = (unsigned)W >> 4, nsy = (unsigned)H >> 4;
        const unsigned sx = t % nsx; t /= nsx;
        const unsigned sy = t % nsy;
        const unsigned sz = t / nsy;
        x = (int)((sx << 4) | ((sub & 1u) << 3) | (w & 7u));
        y = (int)((sy << 4) | (((sub >> 1) & 1u) << 3) | ((w >> 3) & 7u));
        z = (int)((sz << 4) | ((sub >> 2) << 3) | (w >> 6));
    } else {
        x = (int)(v % (unsigned)W); v /= (unsigned)W;
        y = (int)(v % (unsigned)H);
        z = (int)(v / (unsigned)H);
    }
}

// NQ = 16-byte channel groups per thread (c4, c4 + C4/NQ, ...): the tap / weight arithmetic (~150 VALU instructions in ATen's order) is the
// same for every channel of a voxel; two groups per thread halve that cost per byte (one group: 3.1 TB/s at 64^3 x 128 ch).
// What limits the kernel at 3.4-3.8 TB/s algorithmic on HBM-resident volumes (round 6, profiles/r06_rotate_limiter.md: timing of the same kernel
// under access patterns that differ only in the source walk + rocprofv3 SQ / TCP / TCC passes): NOT DRAM locality of the rotated gather (the
// identity warp, whose taps are neighbouring rows in output order, runs at the rotated rate: 3.56 vs 3.51 TB/s), NOT L2->fabric over-fetch
// (FETCH = algorithmic bytes), NOT VALU issue alone (VALU active 0.10 of wave-cycles at 2.7 resident waves per SIMD = 28 % of SIMD time) - the
// mode-0 copy through the same kernel streams at 5.1 TB/s, and every warped output costs 8 x the copy's load requests through TA / L1 (hit rate
// 0.83) / L2 (hit rate 0.60, mean TCP->TCC read latency 685 cycles): the warp is bound by the gather's request rate through the vector memory
// path, half of its wave-cycles waiting on it (SQ_WAIT_ANY 0.53) and a third stalled at issue (SQ_WAIT_INST_ANY 0.32). A persistent grid
// (6 workgroups per CU striding through XCD-contiguous chunk ranges) measured 12 % SLOWER (50.0 vs 44.7 us at the bench transform), nontemporal
// stores and 4 channel groups per thread changed nothing (TUNING_LOG r5-16, r6-3). All index math is 32-bit
// (a volume holds < 2^31 float4 elements; checked on the host side).
template <int NQ>
__global__ __launch_bounds__(256) void rotate_fwd_kernel(const float4* __restrict__ vox, const float* __restrict__ xf,
                                                         const int* __restrict__ mode, const int* __restrict__ dst_slot, float4* __restrict__ out,
                                                         int C4, int D, int H, int W, unsigned per_vol /* D*H*W*C4 */,
                                                         unsigned blocks_per_vol) {
    // grid = n * blocks_per_vol; a workgroup never straddles two volumes
    const unsigned bid = xcd_remap(blockIdx.x, gridDim.x);
    const unsigned n = bid / blocks_per_vol;
    const unsigned CQ = (unsigned)C4 / NQ;                           // threads per voxel
    const unsigned t = (bid % blocks_per_vol) * 256u + threadIdx.x;  // (voxel in tile order, channel group)
    if (t >= per_vol / NQ) return;
    const float4* src = vox + (size_t)n * per_vol;
    float4* dst = out + (size_t)(dst_slot ? (unsigned)dst_slot[n] : n) * per_vol;   // output volume index (view ordering fused into the store)
    const unsigned c4 = t % CQ;
    int x, y, z;
    voxel_of(t / CQ, W, H, D, x, y, z);
    const unsigned e = (((unsigned)z * H + y) * W + x) * C4 + c4;
    if (mode[n] == 0) {            // view 0: pass-through (models/rotate.py:141)
#pragma unroll
        for (int q = 0; q < NQ; ++q) dst[e + q * CQ] = src[e + q * CQ];
        return;
    }
    const float* A = xf + n * 12;  // uniform per workgroup -> scalar loads
    const float gx = linspace_pm1(x, W), gy = linspace_pm1(y, H), gz = linspace_pm1(z, D);
    const float sx = fmaf(A[0], gx, fmaf(A[1], gy, fmaf(A[2], gz, A[3])));
    const float sy = fmaf(A[4], gx, fmaf(A[5], gy, fmaf(A[6], gz, A[7])));
    const float sz = fmaf(A[8], gx, fmaf(A[9], gy, fmaf(A[10], gz, A[11])));
    TriTaps tp;
    taps_ac_false(sx, sy, sz, W, H, D, tp);

    const bool vx0 = (unsigned)tp.x0 < (unsigned)W, vx1 = (unsigned)(tp.x0 + 1) < (unsigned)W;
    const bool vy0 = (unsigned)tp.y0 < (unsigned)H, vy1 = (unsigned)(tp.y0 + 1) < (unsigned)H;
    const bool vz0 = (unsigned)tp.z0 < (unsigned)D, vz1 = (unsigned)(tp.z0 + 1) < (unsigned)D;
    if (!((vx0 | vx1) & (vy0 | vy1) & (vz0 | vz1))) {   // fully outside: zeros padding
#pragma unroll
        for (int q = 0; q < NQ; ++q) dst[e + q * CQ] = make_float4(0.f, 0.f, 0.f, 0.f);
        return;
    }
    // clamped indices + zeroed weights: every load is in-bounds and unconditional (8 NQ in flight)
    const unsigned xa = (unsigned)min(max(tp.x0, 0), W - 1), xb = (unsigned)min(max(tp.x0 + 1, 0), W - 1);
    const unsigned ya = (unsigned)min(max(tp.y0, 0), H - 1), yb = (unsigned)min(max(tp.y0 + 1, 0), H - 1);
    const unsigned za = (unsigned)min(max(tp.z0, 0), D - 1), zb = (unsigned)min(max(tp.z0 + 1, 0), D - 1);
    const float wxa = vx0 ? tp.wx0 : 0.f, wxb = vx1 ? tp.wx1 : 0.f;
    const float wya = vy0 ? tp.wy0 : 0.f, wyb = vy1 ? tp.wy1 : 0.f;
    const float wza = vz0 ? tp.wz0 : 0.f, wzb = vz1 ? tp.wz1 : 0.f;
    const unsigned sW = (unsigned)C4, sH = (unsigned)W * C4, sD = (unsigned)H * W * C4;
    const unsigned o000 = za * sD + ya * sH + xa * sW + c4, o001 = za * sD + ya * sH + xb * sW + c4;
    const unsigned o010 = za * sD + yb * sH + xa * sW + c4, o011 = za * sD + yb * sH + xb * sW + c4;
    const unsigned o100 = zb * sD + ya * sH + xa * sW + c4, o101 = zb * sD + ya * sH + xb * sW + c4;
    const unsigned o110 = zb * sD + yb * sH + xa * sW + c4, o111 = zb * sD + yb * sH + xb * sW + c4;
    float4 v[NQ][8];
#pragma unroll
    for (int q = 0; q < NQ; ++q) {
        const float4* p = src + q * CQ;
        v[q][0] = p[o000]; v[q][1] = p[o001]; v[q][2] = p[o010]; v[q][3] = p[o011];
        v[q][4] = p[o100]; v[q][5] = p[o101]; v[q][6] = p[o110]; v[q][7] = p[o111];
    }
    // ATen accumulation order: tnw, tne, tsw, tse, bnw, bne, bsw, bse (t = z0, n = y0, w = x0)
    const float w8[8] = {wxa * wya * wza, wxb * wya * wza, wxa * wyb * wza, wxb * wyb * wza,
                         wxa * wya * wzb, wxb * wya * wzb, wxa * wyb * wzb, wxb * wyb * wzb};
#pragma unroll
    for (int q = 0; q < NQ; ++q) {
        float4 acc = make_float4(0.f, 0.f, 0.f, 0.f);
#pragma unroll
        for (int k = 0; k < 8; ++k) acc = f4_fma(w8[k], v[q][k], acc);
        dst[e + q * CQ] = acc;
    }
}

// Gradient w.r.t. the 3x4 affine (pose refinement): per OUTPUT voxel, the upstream gradient dotted with the 8 source taps
// and chained through the trilinear weights; block reduction + 12 atomics per workgroup. (The volume gradient is the gather
// kernel below; its first version scatter-added 8 x C fp32 atomics per voxel from here.)
constexpr int AFF_ITER = 8;            // (4: 82 us, 8: 78, 16: 80, 32: 129 - too few workgroups)

__global__ __launch_bounds__(256) void rotate_bwd_affine_kernel(const float4* __restrict__ dout, const float4* __restrict__ vox,
                                                         const float* __restrict__ xf, const int* __restrict__ mode,
                                                         const int* __restrict__ src_slot, float* __restrict__ dxf,
                                                         int C4, int D, int H, int W, long long per_vol,
                                                         unsigned blocks_per_vol) {
    __shared__ float red[12][4];   // per-wave partials of d xf
    const unsigned bid = xcd_remap(blockIdx.x, gridDim.x);
    const int n = bid / blocks_per_vol;
    const int md = mode[n];
    float dA[12];
#pragma unroll
    for (int i = 0; i < 12; ++i) dA[i] = 0.f;
    // AFF_ITER elements per thread (a workgroup covers 256 AFF_ITER consecutive float4 elements) before ONE block reduction: with one element per
    // thread the 72 shuffles + 12 atomics per workgroup outweighed the gather (173 us per refinement iteration for 4 warped 32^3 x 128 views)
#pragma unroll 1
    for (int it = 0; it < AFF_ITER; ++it) {
    const long long e = ((long long)(bid % blocks_per_vol) * AFF_ITER + it) * 256 + threadIdx.x;
    const bool active = e < per_vol;
    if (active && md != 0) {                       // mode-0 volumes are copied, not warped: no dependence on the affine
        const int c4 = (int)(e % C4);
        long long v = e / C4;
        const int x = (int)(v % W); v /= W;
        const int y = (int)(v % H);
        const int z = (int)(v / H);
        const float* A = xf + n * 12;
        const float gx = linspace_pm1(x, W), gy = linspace_pm1(y, H), gz = linspace_pm1(z, D);
        const float sx = fmaf(A[0], gx, fmaf(A[1], gy, fmaf(A[2], gz, A[3])));
        const float sy = fmaf(A[4], gx, fmaf(A[5], gy, fmaf(A[6], gz, A[7])));
        const float sz = fmaf(A[8], gx, fmaf(A[9], gy, fmaf(A[10], gz, A[11])));
        TriTaps t;
        taps_ac_false(sx, sy, sz, W, H, D, t);
        const float4 g = dout[(long long)(src_slot ? src_slot[n] : n) * per_vol + e];     // forge_rotate_fwd_slots stored view n at volume slot[n]
        const long long sW = C4, sH = (long long)W * C4, sD = (long long)H * W * C4;
        float gsx = 0.f, gsy = 0.f, gsz = 0.f;   // d loss / d pixel coordinate
        // all eight corners are loaded unconditionally (out-of-grid corners read a clamped address and are zeroed by a select): no branch sits
        // between the gathers, so they are in flight together
        float4 sv[8];
#pragma unroll
        for (int k = 0; k < 8; ++k) {
            const int dx = k & 1, dy = (k >> 1) & 1, dz = k >> 2;
            const int xi = t.x0 + dx, yi = t.y0 + dy, zi = t.z0 + dz;
            const bool ok = (unsigned)xi < (unsigned)W && (unsigned)yi < (unsigned)H && (unsigned)zi < (unsigned)D;
            const int xc = min(max(xi, 0), W - 1), yc = min(max(yi, 0), H - 1), zc = min(max(zi, 0), D - 1);
            const float4 s = vox[(long long)n * per_vol + zc * sD + yc * sH + xc * sW + c4];
            sv[k] = ok ? s : make_float4(0.f, 0.f, 0.f, 0.f);
        }
#pragma unroll
        for (int k = 0; k < 8; ++k) {
            const int dx = k & 1, dy = (k >> 1) & 1, dz = k >> 2;
            const float wx = dx ? t.wx1 : t.wx0, wy = dy ? t.wy1 : t.wy0, wz = dz ? t.wz1 : t.wz0;
            const float dot = sv[k].x * g.x + sv[k].y * g.y + sv[k].z * g.z + sv[k].w * g.w;
            gsx += (dx ? 1.f : -1.f) * wy * wz * dot;
            gsy += (dy ? 1.f : -1.f) * wx * wz * dot;
            gsz += (dz ? 1.f : -1.f) * wx * wy * dot;
        }
        // d pixel / d s = N/2 per axis (align_corners=False)
        gsx *= 0.5f * (float)W; gsy *= 0.5f * (float)H; gsz *= 0.5f * (float)D;
        dA[0] = fmaf(gsx, gx, dA[0]); dA[1] = fmaf(gsx, gy, dA[1]); dA[2] = fmaf(gsx, gz, dA[2]); dA[3] += gsx;
        dA[4] = fmaf(gsy, gx, dA[4]); dA[5] = fmaf(gsy, gy, dA[5]); dA[6] = fmaf(gsy, gz, dA[6]); dA[7] += gsy;
        dA[8] = fmaf(gsz, gx, dA[8]); dA[9] = fmaf(gsz, gy, dA[9]); dA[10] = fmaf(gsz, gz, dA[10]); dA[11] += gsz;
    }
    }
    if (md != 0) {   // md is workgroup-uniform: whole-block reduction then 12 atomics
        const int lane = threadIdx.x & 63, wv = threadIdx.x >> 6;
#pragma unroll
        for (int i = 0; i < 12; ++i) {
            float s = dA[i];
#pragma unroll
            for (int o = 32; o > 0; o >>= 1) s += __shfl_down(s, o, 64);
            if (lane == 0) red[i][wv] = s;
        }
        __syncthreads();
        if (threadIdx.x < 12) {
            const float s = red[threadIdx.x][0] + red[threadIdx.x][1] + red[threadIdx.x][2] + red[threadIdx.x][3];
            atomic_add_f32(dxf + n * 12 + threadIdx.x, s);
        }
    }
}

// Volume gradient as a GATHER (the adjoint of the warp without atomics): thread = (SOURCE voxel q, 4 channels). The warp maps an
// output voxel x to the source pixel coordinate p(x) = P x + p0 (affine: normalisation, 3x4 transform, align_corners=False
// un-normalisation), and q receives w(q, p(x)) dout[x] from every x with |p(x) - q| < 1 on all axes, i.e. from the integer points of
// the box P^-1 (q - p0 + (-1, 1)^3): centre P^-1 (q - p0), half extents sum_b |P^-1_ab| (<= sqrt 3 for a rigid pose), ~8 hits among
// <= 5^3 candidates. Each candidate is re-evaluated with the forward pass's own expressions (same taps, same weights), so the result is
// the exact transpose of rotate_fwd_kernel; it is deterministic (fixed summation order) and dvox is written, not accumulated.
template <int NQ>
__global__ __launch_bounds__(256) void rotate_bwd_gather_kernel(const float4* __restrict__ dout, const float* __restrict__ xf,
                                                                const int* __restrict__ mode, const int* __restrict__ src_slot, float4* __restrict__ dvox,
                                                                int C4, int D, int H, int W, unsigned per_vol, unsigned blocks_per_vol) {
    const unsigned bid = xcd_remap(blockIdx.x, gridDim.x);
    const unsigned n = bid / blocks_per_vol;
    const unsigned CQ = (unsigned)C4 / NQ;                           // threads per voxel, NQ channel groups each (as in rotate_fwd_kernel)
    const unsigned tq = (bid % blocks_per_vol) * 256u + threadIdx.x;
    if (tq >= per_vol / NQ) return;
    const float4* g = dout + (size_t)(src_slot ? (unsigned)src_slot[n] : n) * per_vol;
    float4* dv = dvox + (size_t)n * per_vol;
    const unsigned c4 = tq % CQ;
    int qx, qy, qz;
    voxel_of(tq / CQ, W, H, D, qx, qy, qz);
    const unsigned eq = (((unsigned)qz * H + qy) * W + qx) * C4 + c4;
    if (mode[n] == 0) {
#pragma unroll
        for (int q = 0; q < NQ; ++q) dv[eq + q * CQ] = g[eq + q * CQ];
        return;
    }
    const float* A = xf + n * 12;
    const float Nf[3] = {(float)W, (float)H, (float)D};
    float P[3][3], p0[3];
#pragma unroll
    for (int a = 0; a < 3; ++a) {
#pragma unroll
        for (int b = 0; b < 3; ++b) P[a][b] = A[a * 4 + b] * Nf[a] / (Nf[b] - 1.f);
        p0[a] = ((A[a * 4 + 3] - A[a * 4] - A[a * 4 + 1] - A[a * 4 + 2] + 1.f) * Nf[a] - 1.f) * 0.5f;
    }
    const float c00 = P[1][1] * P[2][2] - P[1][2] * P[2][1], c01 = P[0][2] * P[2][1] - P[0][1] * P[2][2], c02 = P[0][1] * P[1][2] - P[0][2] * P[1][1];
    const float c10 = P[1][2] * P[2][0] - P[1][0] * P[2][2], c11 = P[0][0] * P[2][2] - P[0][2] * P[2][0], c12 = P[0][2] * P[1][0] - P[0][0] * P[1][2];
    const float c20 = P[1][0] * P[2][1] - P[1][1] * P[2][0], c21 = P[0][1] * P[2][0] - P[0][0] * P[2][1], c22 = P[0][0] * P[1][1] - P[0][1] * P[1][0];
    const float det = P[0][0] * c00 + P[0][1] * c10 + P[0][2] * c20;
    int lo[3] = {0, 0, 0}, hi[3] = {W - 1, H - 1, D - 1};           // singular transform: scan the whole grid (never a camera pose)
    if (fabsf(det) > 1e-20f) {
        const float id = 1.f / det;
        const float I[3][3] = {{c00 * id, c01 * id, c02 * id}, {c10 * id, c11 * id, c12 * id}, {c20 * id, c21 * id, c22 * id}};
        const float r[3] = {(float)qx - p0[0], (float)qy - p0[1], (float)qz - p0[2]};
#pragma unroll
        for (int a = 0; a < 3; ++a) {
            const float cc = I[a][0] * r[0] + I[a][1] * r[1] + I[a][2] * r[2];
            const float hh = fabsf(I[a][0]) + fabsf(I[a][1]) + fabsf(I[a][2]) + 0.02f + 1e-5f * fabsf(cc);
            lo[a] = max((int)ceilf(fmaxf(cc - hh, -1.f)), 0);
            hi[a] = min((int)floorf(fminf(cc + hh, Nf[a])), (int)Nf[a] - 1);
        }
    }
    float4 acc[NQ];
#pragma unroll
    for (int q = 0; q < NQ; ++q) acc[q] = make_float4(0.f, 0.f, 0.f, 0.f);
    const unsigned sW = (unsigned)C4, sH = (unsigned)W * C4, sD = (unsigned)H * W * C4;
    for (int z = lo[2]; z <= hi[2]; ++z) {
        const float gz = linspace_pm1(z, D);
        for (int y = lo[1]; y <= hi[1]; ++y) {
            const float gy = linspace_pm1(y, H);
            for (int x = lo[0]; x <= hi[0]; ++x) {
                const float gx = linspace_pm1(x, W);
                const float sx = fmaf(A[0], gx, fmaf(A[1], gy, fmaf(A[2], gz, A[3])));
                const float sy = fmaf(A[4], gx, fmaf(A[5], gy, fmaf(A[6], gz, A[7])));
                const float sz = fmaf(A[8], gx, fmaf(A[9], gy, fmaf(A[10], gz, A[11])));
                TriTaps t;
                taps_ac_false(sx, sy, sz, W, H, D, t);
                const unsigned ux = (unsigned)(qx - t.x0), uy = (unsigned)(qy - t.y0), uz = (unsigned)(qz - t.z0);
                if (ux <= 1u && uy <= 1u && uz <= 1u) {
                    const float w = (ux ? t.wx1 : t.wx0) * (uy ? t.wy1 : t.wy0) * (uz ? t.wz1 : t.wz0);
                    const unsigned off = (unsigned)z * sD + (unsigned)y * sH + (unsigned)x * sW + c4;
#pragma unroll
                    for (int q = 0; q < NQ; ++q) acc[q] = f4_fma(w, g[off + q * CQ], acc[q]);
                }
            }
        }
    }
#pragma unroll
    for (int q = 0; q < NQ; ++q) dv[eq + q * CQ] = acc[q];
}

// poses [B][t][16] row-major camera poses -> xf [B*t][12], mode [B*t]: T = P_0 P_i^-1 (models/rotate.py:64-89, general
// 4x4 inverse by cofactors — the reference calls torch.inverse), xf = [R_T | t_T / e] (rotate.py:132-135). One thread per view.
__global__ void pose_xf_kernel(const float* __restrict__ poses, float* __restrict__ xf, int* __restrict__ mode, int* __restrict__ slot,
                               const float* __restrict__ dist_in, int B, int t, float e) {
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= B * t) return;
    const int v = i % t;
    float* o = xf + i * 12;
    mode[i] = v == 0 ? 0 : 1;
    if (slot) {
        // view order of models/model.py:152-158 (sort by squared distance of the camera position to view 0's): slot = rank of this view
        // (ties: lower view index first), offset by the scene's first volume - the warp then stores view i at volume slot[i]
        // keys: the caller's distances when given (torch's own reduction: near-ties of symmetric camera rigs must rank exactly as
        // sequence_from_distance ranks them), else computed here without fused multiply-adds
        const float* pb = poses + (long long)(i - v) * 16;
        auto dist = [&](int j) {
            if (dist_in) return dist_in[(i - v) + j];
            const float dx = pb[j * 16 + 3] - pb[3], dy = pb[j * 16 + 7] - pb[7], dz = pb[j * 16 + 11] - pb[11];
            return __fadd_rn(__fadd_rn(__fmul_rn(dx, dx), __fmul_rn(dy, dy)), __fmul_rn(dz, dz));
        };
        const float di = dist(v);
        int rank = 0;
        for (int j = 0; j < t; ++j) {
            const float dj = dist(j);
            rank += (dj < di || (dj == di && j < v)) ? 1 : 0;
        }
        slot[i] = (i - v) + rank;
    }
    if (v == 0) {
#pragma unroll
        for (int k = 0; k < 12; ++k) o[k] = 0.f;
        return;
    }
    const float* m = poses + (long long)i * 16;
    const float* p0 = poses + (long long)(i - v) * 16;
    float inv[16];
    const float s0 = m[0] * m[5] - m[4] * m[1], s1 = m[0] * m[6] - m[4] * m[2], s2 = m[0] * m[7] - m[4] * m[3];
    const float s3 = m[1] * m[6] - m[5] * m[2], s4 = m[1] * m[7] - m[5] * m[3], s5 = m[2] * m[7] - m[6] * m[3];
    const float c5 = m[10] * m[15] - m[14] * m[11], c4 = m[9] * m[15] - m[13] * m[11], c3 = m[9] * m[14] - m[13] * m[10];
    const float c2 = m[8] * m[15] - m[12] * m[11], c1 = m[8] * m[14] - m[12] * m[10], c0 = m[8] * m[13] - m[12] * m[9];
    const float det = s0 * c5 - s1 * c4 + s2 * c3 + s3 * c2 - s4 * c1 + s5 * c0;
    const float id = 1.f / det;
    inv[0] = (m[5] * c5 - m[6] * c4 + m[7] * c3) * id;   inv[1] = (-m[1] * c5 + m[2] * c4 - m[3] * c3) * id;
    inv[2] = (m[13] * s5 - m[14] * s4 + m[15] * s3) * id; inv[3] = (-m[9] * s5 + m[10] * s4 - m[11] * s3) * id;
    inv[4] = (-m[4] * c5 + m[6] * c2 - m[7] * c1) * id;  inv[5] = (m[0] * c5 - m[2] * c2 + m[3] * c1) * id;
    inv[6] = (-m[12] * s5 + m[14] * s2 - m[15] * s1) * id; inv[7] = (m[8] * s5 - m[10] * s2 + m[11] * s1) * id;
    inv[8] = (m[4] * c4 - m[5] * c2 + m[7] * c0) * id;   inv[9] = (-m[0] * c4 + m[1] * c2 - m[3] * c0) * id;
    inv[10] = (m[12] * s4 - m[13] * s2 + m[15] * s0) * id; inv[11] = (-m[8] * s4 + m[9] * s2 - m[11] * s0) * id;
    inv[12] = (-m[4] * c3 + m[5] * c1 - m[6] * c0) * id; inv[13] = (m[0] * c3 - m[1] * c1 + m[2] * c0) * id;
    inv[14] = (-m[12] * s3 + m[13] * s1 - m[14] * s0) * id; inv[15] = (m[8] * s3 - m[9] * s1 + m[10] * s0) * id;
#pragma unroll
    for (int r = 0; r < 3; ++r) {
#pragma unroll
        for (int c = 0; c < 4; ++c) {
            float acc = 0.f;
#pragma unroll
            for (int k = 0; k < 4; ++k) acc = fmaf(p0[r * 4 + k], inv[k * 4 + c], acc);
            o[r * 4 + c] = c == 3 ? acc / e : acc;
        }
    }
}

// ---- pose chain of the refinement loop (kubric_eval.py:412-530, demo.py:115-188) -----------------------------------------------------------
// From the optimised 7-D relative poses (raw quaternion, translation) of the non-reference views to BOTH kernel operands - the warp's affine
// xf = [R_T | t_T / e] with T = P_0 P_i^-1 (models/rotate.py:64-89,132-135) and the ray-marcher's packed cameras [R | T | fx fy cx cy] from the
// extrinsics P_i^-1 (models/volume_render.py:40-51) - in ONE launch, with the Jacobian of the 24 pose-dependent outputs w.r.t. the 7 inputs by
// forward-mode differentiation (dual numbers), so that the backward is one tiny matrix-vector kernel. The torch algebra it replaces
// (F.normalize, quat2mat, canonical @ rel, inverse_affine twice, P_0 @ inverse, cat / stack / slicing and all their autograd nodes) was ~250
// launches of 3-8 us per iteration inside the captured graph. Same formulas in the same order as forge_amd/geo_utils.py.
struct Dual7 {
    float v, d[7];
};
__device__ __forceinline__ Dual7 dconst(float c) { Dual7 r; r.v = c; for (int i = 0; i < 7; ++i) r.d[i] = 0.f; return r; }
__device__ __forceinline__ Dual7 dvar(float x, int k) { Dual7 r = dconst(x); r.d[k] = 1.f; return r; }
__device__ __forceinline__ Dual7 operator+(const Dual7& a, const Dual7& b) { Dual7 r; r.v = a.v + b.v; for (int i = 0; i < 7; ++i) r.d[i] = a.d[i] + b.d[i]; return r; }
__device__ __forceinline__ Dual7 operator-(const Dual7& a, const Dual7& b) { Dual7 r; r.v = a.v - b.v; for (int i = 0; i < 7; ++i) r.d[i] = a.d[i] - b.d[i]; return r; }
__device__ __forceinline__ Dual7 operator-(const Dual7& a) { Dual7 r; r.v = -a.v; for (int i = 0; i < 7; ++i) r.d[i] = -a.d[i]; return r; }
__device__ __forceinline__ Dual7 operator*(const Dual7& a, const Dual7& b) { Dual7 r; r.v = a.v * b.v; for (int i = 0; i < 7; ++i) r.d[i] = a.d[i] * b.v + a.v * b.d[i]; return r; }
__device__ __forceinline__ Dual7 operator*(float c, const Dual7& a) { Dual7 r; r.v = c * a.v; for (int i = 0; i < 7; ++i) r.d[i] = c * a.d[i]; return r; }
__device__ __forceinline__ Dual7 operator/(const Dual7& a, const Dual7& b) {
    Dual7 r; r.v = a.v / b.v;
    const float ib = 1.f / b.v;
    for (int i = 0; i < 7; ++i) r.d[i] = (a.d[i] - r.v * b.d[i]) * ib;
    return r;
}
__device__ __forceinline__ Dual7 dsqrt(const Dual7& a) { Dual7 r; r.v = sqrtf(a.v); const float h = 0.5f / r.v; for (int i = 0; i < 7; ++i) r.d[i] = a.d[i] * h; return r; }

// translation row i of can_p @ [R | tr]: the one expression both the pose output and the view ranking use
__device__ __forceinline__ float pose_translation(const float* __restrict__ can_p, const float* __restrict__ tr, int i) {
    return fmaf(can_p[i * 4], tr[0], fmaf(can_p[i * 4 + 1], tr[1], fmaf(can_p[i * 4 + 2], tr[2], can_p[i * 4 + 3])));
}

// One thread per view (b t of them). rot [b (t-1)][4], trans [b (t-1)][3]; can_p / can_e: pose and extrinsics of the reference view (row-major 4x4);
// K [b t][9] full-resolution intrinsics. Outputs: xf [b t][12], mode [b t], cam [b t][16], poses [b t][16], origin [b t][2] (nullable),
// jac [b (t-1)][24][7] (nullable): d (xf[0..11], cam[0..11]) / d (quaternion, translation).
__global__ __launch_bounds__(64) void pose_chain_fwd_kernel(const float* __restrict__ rot, const float* __restrict__ trans, const float* __restrict__ can_p,
                                      const float* __restrict__ can_e, const float* __restrict__ K, float e, int b, int t, float* __restrict__ xf,
                                      int* __restrict__ mode, int* __restrict__ slot, float* __restrict__ cam, float* __restrict__ poses,
                                      float* __restrict__ origin, float* __restrict__ jac) {
    const int idx = blockIdx.x * blockDim.x + threadIdx.x;
    if (idx >= b * t) return;
    const int v = idx % t;
    if (slot) {
        // view order of models/model.py:152-158 (stable sort by squared distance of the camera position to view 0's): slot = scene's first volume +
        // rank of this view. The camera position of view j is the translation of its pose, can_p_A trans_j + can_p_t - no quaternion involved -,
        // evaluated here for every view of the scene with the SAME fmaf chain that produces this kernel's `poses` output, and ranked on
        // ((dx^2 + dy^2) + dz^2) without fused multiply-adds like sequence_from_distance's reduction.
        auto cam_pos = [&](int j, float* o) {
            if (j == 0) { o[0] = can_p[3]; o[1] = can_p[7]; o[2] = can_p[11]; return; }
            const float* tj = trans + ((long long)(idx / t) * (t - 1) + j - 1) * 3;
            for (int i = 0; i < 3; ++i) o[i] = pose_translation(can_p, tj, i);
        };
        auto dist = [&](int j) {
            float p0[3], pj[3];
            cam_pos(0, p0); cam_pos(j, pj);
            const float dx = pj[0] - p0[0], dy = pj[1] - p0[1], dz = pj[2] - p0[2];
            return __fadd_rn(__fadd_rn(__fmul_rn(dx, dx), __fmul_rn(dy, dy)), __fmul_rn(dz, dz));
        };
        const float di = dist(v);
        int rank = 0;
        for (int j = 0; j < t; ++j) {
            const float dj = dist(j);
            rank += (dj < di || (dj == di && j < v)) ? 1 : 0;
        }
        slot[idx] = (idx - v) + rank;
    }
    const float* Kv = K + (long long)idx * 9;
    const float fx = Kv[0] / 2.f, fy = Kv[4] / 2.f, cx = Kv[2] / 2.f, cy = Kv[5] / 2.f;     // models/volume_render.py:50-51
    float* xo = xf + (long long)idx * 12;
    float* co = cam + (long long)idx * 16;
    float* po = poses + (long long)idx * 16;
    mode[idx] = v == 0 ? 0 : 1;
    co[12] = fx; co[13] = fy; co[14] = cx; co[15] = cy;
    if (v == 0) {                                                    // the reference view: identity warp, canonical camera
        for (int k = 0; k < 12; ++k) xo[k] = 0.f;
        for (int k = 0; k < 16; ++k) po[k] = can_p[k];
        for (int r = 0; r < 3; ++r) {
            for (int c = 0; c < 3; ++c) co[r * 3 + c] = can_e[r * 4 + c];
            co[9 + r] = can_e[r * 4 + 3];
        }
        if (origin) { origin[idx * 2] = fx * co[9] / co[11] + cx; origin[idx * 2 + 1] = fy * co[10] / co[11] + cy; }
        return;
    }
    const int k = (idx / t) * (t - 1) + v - 1;
    Dual7 q[4], tr[3];
    for (int i = 0; i < 4; ++i) q[i] = dvar(rot[k * 4 + i], i);
    for (int i = 0; i < 3; ++i) tr[i] = dvar(trans[k * 3 + i], 4 + i);
    // F.normalize(rot) (refine.py), then quat2mat_transform's own q / |q| (geo_utils.py:54)
    Dual7 n1 = dsqrt(q[0] * q[0] + q[1] * q[1] + q[2] * q[2] + q[3] * q[3]);
    if (n1.v < 1e-12f) n1 = dconst(1e-12f);
    for (int i = 0; i < 4; ++i) q[i] = q[i] / n1;
    const Dual7 n2 = dsqrt(q[0] * q[0] + q[1] * q[1] + q[2] * q[2] + q[3] * q[3]);
    const Dual7 w = q[0] / n2, x = q[1] / n2, y = q[2] / n2, z = q[3] / n2;
    Dual7 R[3][3];
    R[0][0] = w * w + x * x - y * y - z * z; R[0][1] = 2.f * x * y - 2.f * w * z;     R[0][2] = 2.f * w * y + 2.f * x * z;
    R[1][0] = 2.f * w * z + 2.f * x * y;     R[1][1] = w * w - x * x + y * y - z * z; R[1][2] = 2.f * y * z - 2.f * w * x;
    R[2][0] = 2.f * x * z - 2.f * w * y;     R[2][1] = 2.f * w * x + 2.f * y * z;     R[2][2] = w * w - x * x - y * y + z * z;
    // pose = can_p @ [R | tr]
    Dual7 A[3][3], tp[3];
    for (int i = 0; i < 3; ++i) {
        for (int j = 0; j < 3; ++j) A[i][j] = can_p[i * 4] * R[0][j] + can_p[i * 4 + 1] * R[1][j] + can_p[i * 4 + 2] * R[2][j];
        tp[i] = can_p[i * 4] * tr[0] + can_p[i * 4 + 1] * tr[1] + can_p[i * 4 + 2] * tr[2] + dconst(can_p[i * 4 + 3]);
        tp[i].v = pose_translation(can_p, trans + k * 3, i);       // the value with the pinned fmaf chain (the derivatives do not depend on it)
    }
    for (int i = 0; i < 3; ++i) { for (int j = 0; j < 3; ++j) po[i * 4 + j] = A[i][j].v; po[i * 4 + 3] = tp[i].v; }
    po[12] = 0.f; po[13] = 0.f; po[14] = 0.f; po[15] = 1.f;
    // extrinsics = inverse_affine(pose): A^-1 = [b x c, c x a, a x b] / det, t' = -A^-1 t (geo_utils.py:108-122)
    auto cross = [](const Dual7* u, const Dual7* w2, Dual7* o) {
        o[0] = u[1] * w2[2] - u[2] * w2[1]; o[1] = u[2] * w2[0] - u[0] * w2[2]; o[2] = u[0] * w2[1] - u[1] * w2[0];
    };
    Dual7 bc[3], ca[3], ab[3];
    cross(A[1], A[2], bc); cross(A[2], A[0], ca); cross(A[0], A[1], ab);
    const Dual7 det = A[0][0] * bc[0] + A[0][1] * bc[1] + A[0][2] * bc[2];
    Dual7 Ai[3][3], ti[3];
    for (int i = 0; i < 3; ++i) { Ai[i][0] = bc[i] / det; Ai[i][1] = ca[i] / det; Ai[i][2] = ab[i] / det; }
    for (int i = 0; i < 3; ++i) ti[i] = -(Ai[i][0] * tp[0] + Ai[i][1] * tp[1] + Ai[i][2] * tp[2]);
    // T = P_0 @ extrinsics with P_0 = can_p; xf = [T_A | T_t / e]
    Dual7 out[24];
    for (int i = 0; i < 3; ++i) {
        for (int j = 0; j < 3; ++j) out[i * 4 + j] = can_p[i * 4] * Ai[0][j] + can_p[i * 4 + 1] * Ai[1][j] + can_p[i * 4 + 2] * Ai[2][j];
        out[i * 4 + 3] = (1.f / e) * (can_p[i * 4] * ti[0] + can_p[i * 4 + 1] * ti[1] + can_p[i * 4 + 2] * ti[2] + dconst(can_p[i * 4 + 3]));
    }
    for (int i = 0; i < 3; ++i) { for (int j = 0; j < 3; ++j) out[12 + i * 3 + j] = Ai[i][j]; out[21 + i] = ti[i]; }
    for (int o = 0; o < 12; ++o) { xo[o] = out[o].v; co[o] = out[12 + o].v; }
    if (origin) { origin[idx * 2] = fx * ti[0].v / ti[2].v + cx; origin[idx * 2 + 1] = fy * ti[1].v / ti[2].v + cy; }
    if (jac) {
        float* jo = jac + (long long)k * 24 * 7;
        for (int o = 0; o < 24; ++o)
            for (int i = 0; i < 7; ++i) jo[o * 7 + i] = out[o].d[i];
    }
}

// drot [n][4], dtrans [n][3] = J^T (dxf row | dcam row[0..11]) per non-reference view; one thread per (view, input).
__global__ void pose_chain_bwd_kernel(const float* __restrict__ jac, const float* __restrict__ dxf, const float* __restrict__ dcam, float* __restrict__ drot,
                                      float* __restrict__ dtrans, int b, int t) {
    const int idx = blockIdx.x * blockDim.x + threadIdx.x;
    const int n = b * (t - 1);
    if (idx >= n * 7) return;
    const int k = idx / 7, i = idx - k * 7;
    const int row = (k / (t - 1)) * t + (k % (t - 1)) + 1;
    const float* jo = jac + (long long)k * 24 * 7;
    float acc = 0.f;
    for (int o = 0; o < 12; ++o) acc = fmaf(dxf ? dxf[(long long)row * 12 + o] : 0.f, jo[o * 7 + i], acc);
    for (int o = 0; o < 12; ++o) acc = fmaf(dcam ? dcam[(long long)row * 16 + o] : 0.f, jo[(12 + o) * 7 + i], acc);
    if (i < 4) drot[k * 4 + i] = acc; else dtrans[k * 3 + (i - 4)] = acc;
}

// cameras_from_opencv_projection's inputs -> the ray-marcher's packed cameras (models/volume_render.py:40-51: K / 2 with K[2][2] = 1 on a
// copy) and, optionally, the projected world origin (models/volume_render.py:77-79). One thread per camera; inputs may be strided views.
__global__ void pack_cameras_kernel(const float* __restrict__ R, long long r0, long long r1, long long r2, const float* __restrict__ T, long long t0,
                                    long long t1, const float* __restrict__ K, long long k0, long long k1, long long k2, float* __restrict__ cam,
                                    float* __restrict__ origin, int V) {
    const int v = blockIdx.x * blockDim.x + threadIdx.x;
    if (v >= V) return;
    float* c = cam + v * 16;
#pragma unroll
    for (int i = 0; i < 3; ++i)
#pragma unroll
        for (int j = 0; j < 3; ++j) c[i * 3 + j] = R[v * r0 + i * r1 + j * r2];
    const float tx = T[v * t0], ty = T[v * t0 + t1], tz = T[v * t0 + 2 * t1];
    c[9] = tx; c[10] = ty; c[11] = tz;
    const float fx = K[v * k0] / 2.0f, fy = K[v * k0 + k1 + k2] / 2.0f, cx = K[v * k0 + 2 * k2] / 2.0f, cy = K[v * k0 + k1 + 2 * k2] / 2.0f;
    c[12] = fx; c[13] = fy; c[14] = cx; c[15] = cy;
    if (origin) {
        origin[v * 2] = fx * tx / tz + cx;
        origin[v * 2 + 1] = fy * ty / tz + cy;
    }
}

static int check_rotate_args(const void* a, const void* b, const void* c, const void* d, int n, int C, int D, int H, int W) {
    FORGE_REQUIRE(a && b && c && d, FORGE_EINVAL, "forge_rotate: null pointer argument");
    FORGE_REQUIRE(n > 0 && C > 0 && D > 1 && H > 1 && W > 1, FORGE_EINVAL,
                  "forge_rotate: need n>0, C>0 and D,H,W>1 (got n=%d C=%d D=%d H=%d W=%d)", n, C, D, H, W);
    FORGE_REQUIRE(C % 4 == 0, FORGE_ESHAPE, "forge_rotate: C=%d must be a multiple of 4 (16-byte channel groups)", C);
    return 0;
}

}  // namespace forge

using namespace forge;

static int rotate_fwd_launch(const float* vox, const float* xf, const int* mode, const int* dst_slot, float* out,
                             int n, int C, int D, int H, int W, forge_stream_t stream) {
    if (int rc = check_rotate_args(vox, xf, mode, out, n, C, D, H, W)) return rc;
    const int C4 = C / 4;
    const long long per_vol = (long long)D * H * W * C4;
    FORGE_REQUIRE(per_vol < (1ll << 31), FORGE_ESHAPE, "forge_rotate_fwd: a volume holds >= 2^31 float4 elements");
    const int nq = (C4 >= 16 && C4 % 2 == 0) ? 2 : 1;                  // channel groups per thread (narrow volumes: one, for coalescing)
    const unsigned bpv = (unsigned)((per_vol / nq + 255) / 256);
    FORGE_REQUIRE((long long)bpv * n < (1ll << 31), FORGE_ESHAPE, "forge_rotate_fwd: grid too large");
    if (nq == 2)
        hipLaunchKernelGGL(rotate_fwd_kernel<2>, dim3(bpv * (unsigned)n), dim3(256), 0, (hipStream_t)stream,
                           (const float4*)vox, xf, mode, dst_slot, (float4*)out, C4, D, H, W, (unsigned)per_vol, bpv);
    else
        hipLaunchKernelGGL(rotate_fwd_kernel<1>, dim3(bpv * (unsigned)n), dim3(256), 0, (hipStream_t)stream,
                           (const float4*)vox, xf, mode, dst_slot, (float4*)out, C4, D, H, W, (unsigned)per_vol, bpv);
    FORGE_LAUNCH_CHECK("forge_rotate_fwd");
    return 0;
}

extern "C" int forge_rotate_fwd(const float* vox, const float* xf, const int* mode, float* out,
                                int n, int C, int D, int H, int W, forge_stream_t stream) {
    return rotate_fwd_launch(vox, xf, mode, nullptr, out, n, C, D, H, W, stream);
}

extern "C" int forge_rotate_fwd_slots(const float* vox, const float* xf, const int* mode, const int* dst_slot, float* out,
                                      int n, int C, int D, int H, int W, forge_stream_t stream) {
    FORGE_REQUIRE(dst_slot, FORGE_EINVAL, "forge_rotate_fwd_slots: null dst_slot");
    return rotate_fwd_launch(vox, xf, mode, dst_slot, out, n, C, D, H, W, stream);
}

extern "C" int forge_rotate_xf_from_poses(const float* poses, float* xf, int* mode, int* slot, const float* dist, int B, int t,
                                         float half_extent, forge_stream_t stream) {
    FORGE_REQUIRE(poses && xf && mode, FORGE_EINVAL, "forge_rotate_xf_from_poses: null pointer argument");
    FORGE_REQUIRE(B > 0 && t > 0 && half_extent > 0.f, FORGE_EINVAL, "forge_rotate_xf_from_poses: bad B=%d t=%d e=%g", B, t, half_extent);
    const int n = B * t;
    hipLaunchKernelGGL(pose_xf_kernel, dim3((n + 63) / 64), dim3(64), 0, (hipStream_t)stream, poses, xf, mode, slot, dist, B, t, half_extent);
    FORGE_LAUNCH_CHECK("forge_rotate_xf_from_poses");
    return 0;
}

extern "C" int forge_pack_cameras(const float* R, long long r0, long long r1, long long r2, const float* T, long long t0, long long t1,
                                  const float* K, long long k0, long long k1, long long k2, float* cam16, float* origin, int V, forge_stream_t stream) {
    FORGE_REQUIRE(R && T && K && cam16 && V > 0, FORGE_EINVAL, "forge_pack_cameras: null pointer argument or V <= 0");
    hipLaunchKernelGGL(pack_cameras_kernel, dim3((V + 63) / 64), dim3(64), 0, (hipStream_t)stream, R, r0, r1, r2, T, t0, t1, K, k0, k1, k2, cam16, origin, V);
    FORGE_LAUNCH_CHECK("forge_pack_cameras");
    return 0;
}

extern "C" int forge_pose_chain_fwd(const float* rot, const float* trans, const float* can_pose, const float* can_extr, const float* K, float half_extent,
                                    int b, int t, float* xf, int* mode, int* slot, float* cam16, float* poses, float* origin, float* jac, forge_stream_t stream) {
    FORGE_REQUIRE(can_pose && can_extr && K && xf && mode && cam16 && poses, FORGE_EINVAL, "forge_pose_chain_fwd: null pointer argument");
    FORGE_REQUIRE(b > 0 && t > 0 && half_extent > 0.f && (t == 1 || (rot && trans)), FORGE_EINVAL, "forge_pose_chain_fwd: bad b=%d t=%d e=%g", b, t, half_extent);
    const int n = b * t;
    hipLaunchKernelGGL(pose_chain_fwd_kernel, dim3((n + 63) / 64), dim3(64), 0, (hipStream_t)stream, rot, trans, can_pose, can_extr, K, half_extent, b, t, xf,
                       mode, slot, cam16, poses, origin, jac);
    FORGE_LAUNCH_CHECK("forge_pose_chain_fwd");
    return 0;
}

extern "C" int forge_pose_chain_bwd(const float* jac, const float* dxf, const float* dcam, float* drot, float* dtrans, int b, int t, forge_stream_t stream) {
    FORGE_REQUIRE(jac && drot && dtrans && (dxf || dcam), FORGE_EINVAL, "forge_pose_chain_bwd: null pointer argument");
    FORGE_REQUIRE(b > 0 && t > 1, FORGE_EINVAL, "forge_pose_chain_bwd: bad b=%d t=%d (t > 1)", b, t);
    const int n = b * (t - 1) * 7;
    hipLaunchKernelGGL(pose_chain_bwd_kernel, dim3((n + 63) / 64), dim3(64), 0, (hipStream_t)stream, jac, dxf, dcam, drot, dtrans, b, t);
    FORGE_LAUNCH_CHECK("forge_pose_chain_bwd");
    return 0;
}

static int rotate_bwd_launch(const float* dout, const float* vox, const float* xf, const int* mode, const int* src_slot,
                             float* dvox, float* dxf, int n, int C, int D, int H, int W, forge_stream_t stream) {
    if (int rc = check_rotate_args(dout, xf, mode, dvox ? (const void*)dvox : (const void*)dxf, n, C, D, H, W)) return rc;     // at least one of dvox / dxf
    FORGE_REQUIRE(!dxf || vox, FORGE_EINVAL, "forge_rotate_bwd: dxf requested but vox is NULL");
    const int C4 = C / 4;
    const long long per_vol = (long long)D * H * W * C4;
    const unsigned bpv = (unsigned)((per_vol + 255) / 256);
    FORGE_REQUIRE((long long)bpv * n < (1ll << 31), FORGE_ESHAPE, "forge_rotate_bwd: grid too large");
    FORGE_REQUIRE(per_vol < (1ll << 31), FORGE_ESHAPE, "forge_rotate_bwd: a volume holds >= 2^31 float4 elements");
    const int nq = (C4 >= 32 && C4 % 4 == 0) ? 4 : (C4 >= 16 && C4 % 2 == 0) ? 2 : 1;   // the candidate tests are per voxel: amortise them
    const unsigned bpg = (unsigned)((per_vol / nq + 255) / 256);
#define FORGE_LAUNCH_GATHER(NQv)                                                                                            \
    hipLaunchKernelGGL(rotate_bwd_gather_kernel<NQv>, dim3(bpg * (unsigned)n), dim3(256), 0, (hipStream_t)stream,           \
                       (const float4*)dout, xf, mode, src_slot, (float4*)dvox, C4, D, H, W, (unsigned)per_vol, bpg)
    if (!dvox) { }                                                   // frozen features (pose refinement): only the affine gradient is wanted
    else if (nq == 4) FORGE_LAUNCH_GATHER(4);
    else if (nq == 2) FORGE_LAUNCH_GATHER(2);
    else FORGE_LAUNCH_GATHER(1);
#undef FORGE_LAUNCH_GATHER
    if (dxf) {
        const unsigned bpa = (unsigned)((per_vol + 256 * AFF_ITER - 1) / (256 * AFF_ITER));
        hipLaunchKernelGGL(rotate_bwd_affine_kernel, dim3(bpa * (unsigned)n), dim3(256), 0, (hipStream_t)stream,
                           (const float4*)dout, (const float4*)vox, xf, mode, src_slot, dxf, C4, D, H, W, per_vol, bpa);
    }
    FORGE_LAUNCH_CHECK("forge_rotate_bwd");
    return 0;
}

extern "C" int forge_rotate_bwd(const float* dout, const float* vox, const float* xf, const int* mode,
                                float* dvox, float* dxf, int n, int C, int D, int H, int W,
                                forge_stream_t stream) {
    return rotate_bwd_launch(dout, vox, xf, mode, nullptr, dvox, dxf, n, C, D, H, W, stream);
}

extern "C" int forge_rotate_bwd_slots(const float* dout, const float* vox, const float* xf, const int* mode, const int* src_slot,
                                      float* dvox, float* dxf, int n, int C, int D, int H, int W, forge_stream_t stream) {
    FORGE_REQUIRE(src_slot, FORGE_EINVAL, "forge_rotate_bwd_slots: null slot array");
    return rotate_bwd_launch(dout, vox, xf, mode, src_slot, dvox, dxf, n, C, D, H, W, stream);
}
