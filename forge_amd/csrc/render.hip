// render.hip — a6: fused ray sampler + trilinear volume sampler + emission-absorption ray-marcher.
//
// Replaces models/volume_render.py:53-63 of the reference, i.e. PyTorch3D's
// cameras_from_opencv_projection -> NDCGridRaysampler -> VolumeSampler (2x grid_sample,
// align_corners=True, zeros) -> EmissionAbsorptionRaymarcher (+ README.md:26-33 depth patch),
// which materialise ray points (12.6 MB/view) and sampled tensors (71 MB/view) in HBM.
// Here nothing is materialised: each ray is marched in registers.
//
// Work decomposition (gfx950, wave = 64): C/4 lanes per ray, each lane owning 4 feature channels
// (one 16-byte channels-last load per tap) and all lanes of a ray sharing the density taps. A wave
// covers a 4x4 pixel quad (C=16), a 256-thread workgroup an 8x8 pixel tile, so the lanes of a
// wave walk neighbouring rays in lock-step and their taps hit the same L1 lines. The march is
// sequential per ray (transmittance carried in a register, same multiplication order as
// torch.cumprod), but tap addresses do not depend on loaded data, so the compiler keeps the next
// sample's 16 loads in flight under the current sample's FMAs.
//
// Exact early-outs only (densities are unclamped): (i) samples whose 8 taps are
// all outside the grid contribute d = 0 exactly -> the march is restricted to the conservative
// ray/AABB sample interval; (ii) T == 0 exactly -> every later weight is exactly 0.
//
// Roofline: compulsory HBM traffic is tiny (one read of the 17-channel volume per scene + the
// output planes); the kernel is bound by L1/TA gather rate — see DESIGN.md.
#include "common.h"

namespace forge {

struct RayCam {
    float ox, oy, oz;      // camera centre  c = -R^T t
    float dx, dy, dz;      // un-normalised world direction R^T ((w+.5-cx)/fx, (h+.5-cy)/fy, 1)
};

__device__ __forceinline__ RayCam make_ray(const float* __restrict__ cam, int w, int h) {
    // cam: R[9] row-major, T[3], fx, fy, cx, cy
    const float dxc = ((float)w + 0.5f - cam[14]) / cam[12];
    const float dyc = ((float)h + 0.5f - cam[15]) / cam[13];
    RayCam r;
    r.dx = cam[0] * dxc + cam[3] * dyc + cam[6];
    r.dy = cam[1] * dxc + cam[4] * dyc + cam[7];
    r.dz = cam[2] * dxc + cam[5] * dyc + cam[8];
    r.ox = -(cam[0] * cam[9] + cam[3] * cam[10] + cam[6] * cam[11]);
    r.oy = -(cam[1] * cam[9] + cam[4] * cam[10] + cam[7] * cam[11]);
    r.oz = -(cam[2] * cam[9] + cam[5] * cam[10] + cam[8] * cam[11]);
    return r;
}

// torch.linspace(zmin, zmax, S)[s]: ATen fills the first half as start + step*i and the second
// half as end - step*(S-1-i).
__device__ __forceinline__ float sample_depth(int s, int S, float zmin, float zmax, float step) {
    return (s < S / 2) ? (zmin + step * (float)s) : (zmax - step * (float)(S - 1 - s));
}

// Conservative sample-index interval [s0, s1] in which a ray can have any in-range tap.
// pix_a(z) = alpha_a + beta_a z must lie in (-1, N_a) on all three axes.
__device__ __forceinline__ void ray_interval(const RayCam& r, float hx, float hy, float hz, int W, int H, int D,
                                             int S, float zmin, float step, int& s0, int& s1) {
    float lo = -INFINITY, hi = INFINITY;
    const float o[3] = {r.ox, r.oy, r.oz}, d[3] = {r.dx, r.dy, r.dz}, hh[3] = {hx, hy, hz};
    const int N[3] = {W, H, D};
    bool empty = false;
#pragma unroll
    for (int a = 0; a < 3; ++a) {
        const float sc = 0.5f * (float)(N[a] - 1);
        const float alpha = (o[a] / hh[a] + 1.f) * sc, beta = (d[a] / hh[a]) * sc;
        const float pmin = -1.001f, pmax = (float)N[a] + 0.001f;
        if (fabsf(beta) < 1e-20f) {
            empty |= !(alpha > pmin && alpha < pmax);
        } else {
            const float t1 = (pmin - alpha) / beta, t2 = (pmax - alpha) / beta;
            lo = fmaxf(lo, fminf(t1, t2));
            hi = fminf(hi, fmaxf(t1, t2));
        }
    }
    if (empty || !(lo <= hi)) { s0 = 0; s1 = -1; return; }
    const float fs0 = floorf((lo - zmin) / step) - 1.f, fs1 = ceilf((hi - zmin) / step) + 1.f;
    s0 = (int)fmaxf(fs0, 0.f);
    s1 = (int)fminf(fs1, (float)(S - 1));
    if (fs1 < 0.f || fs0 > (float)(S - 1)) { s0 = 0; s1 = -1; }
}

struct Taps {
    long long i000;                 // voxel index of the clamped (z0,y0,x0) tap
    int ox, oy, oz;                 // voxel-index offsets to the +x/+y/+z taps (0 when clamped)
    int xa, ya, za, sx, sy, sz;     // coordinates of the clamped (z0,y0,x0) tap and 0/1 coordinate steps to the + taps
    float w[8];                     // trilinear weights, 0 for out-of-range taps
    float ax[2], ay[2], az[2];      // per-axis weights (0 when that index is out of range)
    float bx[2], by[2], bz[2];      // d(axis weight)/d(pixel coord): -1 / +1 for in-range indices, else 0
    bool any;
};

// align_corners=True un-normalisation ((l + 1) / 2) * (N - 1) in ATen's operation order.
__device__ __forceinline__ void taps_ac_true(float px, float py, float pz, int W, int H, int D, Taps& t) {
    px = fminf(fmaxf(px, -2.f), (float)W + 1.f);
    py = fminf(fmaxf(py, -2.f), (float)H + 1.f);
    pz = fminf(fmaxf(pz, -2.f), (float)D + 1.f);
    const float fx = floorf(px), fy = floorf(py), fz = floorf(pz);
    const int x0 = (int)fx, y0 = (int)fy, z0 = (int)fz;
    const bool vx0 = (unsigned)x0 < (unsigned)W, vx1 = (unsigned)(x0 + 1) < (unsigned)W;
    const bool vy0 = (unsigned)y0 < (unsigned)H, vy1 = (unsigned)(y0 + 1) < (unsigned)H;
    const bool vz0 = (unsigned)z0 < (unsigned)D, vz1 = (unsigned)(z0 + 1) < (unsigned)D;
    t.any = (vx0 | vx1) & (vy0 | vy1) & (vz0 | vz1);
    const float wxa = vx0 ? (fx + 1.f) - px : 0.f, wxb = vx1 ? px - fx : 0.f;
    const float wya = vy0 ? (fy + 1.f) - py : 0.f, wyb = vy1 ? py - fy : 0.f;
    const float wza = vz0 ? (fz + 1.f) - pz : 0.f, wzb = vz1 ? pz - fz : 0.f;
    const int xa = min(max(x0, 0), W - 1), xb = min(max(x0 + 1, 0), W - 1);
    const int ya = min(max(y0, 0), H - 1), yb = min(max(y0 + 1, 0), H - 1);
    const int za = min(max(z0, 0), D - 1), zb = min(max(z0 + 1, 0), D - 1);
    t.i000 = ((long long)za * H + ya) * W + xa;
    t.ox = xb - xa; t.oy = (yb - ya) * W; t.oz = (zb - za) * H * W;
    t.xa = xa; t.ya = ya; t.za = za; t.sx = xb - xa; t.sy = yb - ya; t.sz = zb - za;
    t.ax[0] = wxa; t.ax[1] = wxb; t.ay[0] = wya; t.ay[1] = wyb; t.az[0] = wza; t.az[1] = wzb;
    t.bx[0] = vx0 ? -1.f : 0.f; t.bx[1] = vx1 ? 1.f : 0.f;
    t.by[0] = vy0 ? -1.f : 0.f; t.by[1] = vy1 ? 1.f : 0.f;
    t.bz[0] = vz0 ? -1.f : 0.f; t.bz[1] = vz1 ? 1.f : 0.f;
    t.w[0] = wxa * wya * wza; t.w[1] = wxb * wya * wza; t.w[2] = wxa * wyb * wza; t.w[3] = wxb * wyb * wza;
    t.w[4] = wxa * wya * wzb; t.w[5] = wxb * wya * wzb; t.w[6] = wxa * wyb * wzb; t.w[7] = wxb * wyb * wzb;
}

__device__ __forceinline__ long long tap_off(const Taps& t, int k) {
    return t.i000 + ((k & 1) ? t.ox : 0) + ((k & 2) ? t.oy : 0) + ((k & 4) ? t.oz : 0);
}

// ray r_local of a workgroup -> pixel inside the 8 x (RPB/8) tile, 4x4 quads per 16 consecutive rays
__device__ __forceinline__ void tile_pixel(int r, int& lx, int& ly) {
    lx = (r & 3) | (((r >> 4) & 1) << 2);
    ly = ((r >> 2) & 3) | ((r >> 5) << 2);
}

template <int C4>
__global__ __launch_bounds__(256) void render_fwd_kernel(const float4* __restrict__ feat, const float* __restrict__ dens,
                                                         const float* __restrict__ cams, const int* __restrict__ view2vol,
                                                         float* __restrict__ out_feat, float* __restrict__ out_opac,
                                                         float* __restrict__ out_depth, int D, int H, int W, int Hr, int Wr,
                                                         int S, float zmin, float zmax, float hx, float hy, float hz, int V, int band_order) {
    constexpr int RPB = 256 / C4, TH = RPB / 8;
    // Workgroup -> (view, pixel tile). 1-D grid of V x ny x nx workgroups; the dispatcher deals consecutive ids round-robin to the 8 XCDs.
    const unsigned nx = (unsigned)(Wr + 7) / 8, ny = (unsigned)(Hr + TH - 1) / TH;
    unsigned bx, by, bz;
    if (band_order) {
        // XCD x marches tile rows [x ny/8, (x+1) ny/8) of EVERY view, the views of one tile position back to back: the rays of one image-row
        // band of cameras around the object stay inside one slab of the volume, so an XCD's 4 MiB L2 serves the views from the slab it has
        // already fetched. Measured (round 3, tools/pmc_render.sh, L2->fabric bytes per launch / ms): 64^3 x 5 views 111 -> 38 MB (algorithmic
        // 23 MB), 0.0975 -> 0.0928; 64^3 x 28 views 444 -> 51 MB (algorithmic 49 MB), 0.490 -> 0.463; 128^3 x 5 750 -> 599 MB, 0.132 ->
        // 0.115; 128^3 x 28 4532 -> 3304 MB, 0.725 -> 0.644 (at 128^3 the resident workgroups' frusta alone exceed the L2: Infinity-Cache
        // served). Placement only: results are bit-identical to launch order.
        const unsigned g = blockIdx.x, xcd = g % NUM_XCD, k = g / NUM_XCD, rows_per = ny / NUM_XCD;
        bz = k % (unsigned)V;
        const unsigned t = k / (unsigned)V;
        by = xcd * rows_per + t / nx;
        bx = t % nx;
    } else {
        bx = blockIdx.x % nx;
        by = (blockIdx.x / nx) % ny;
        bz = blockIdx.x / (nx * ny);
    }
    const int v = (int)bz;
    const int cg = threadIdx.x % C4, r = threadIdx.x / C4;
    int lx, ly;
    tile_pixel(r, lx, ly);
    const int w = (int)bx * 8 + lx, h = (int)by * TH + ly;
    if (w >= Wr || h >= Hr) return;
    const float* cam = cams + v * 16;
    const long long nvox = (long long)D * H * W;
    const float4* F = feat + (long long)view2vol[v] * nvox * C4 + cg;
    const float* Dn = dens + (long long)view2vol[v] * nvox;

    const RayCam ray = make_ray(cam, w, h);
    const float step = (zmax - zmin) / (float)(S - 1);
    int s0, s1;
    ray_interval(ray, hx, hy, hz, W, H, D, S, zmin, step, s0, s1);
    const float scx = (float)(W - 1), scy = (float)(H - 1), scz = (float)(D - 1);

    float4 acc = make_float4(0.f, 0.f, 0.f, 0.f);
    float T = 1.f, depth = 0.f;
#pragma unroll 2
    for (int s = s0; s <= s1; ++s) {
        const float z = sample_depth(s, S, zmin, zmax, step);
        const float px = (((ray.ox + ray.dx * z) / hx + 1.f) / 2.f) * scx;
        const float py = (((ray.oy + ray.dy * z) / hy + 1.f) / 2.f) * scy;
        const float pz = (((ray.oz + ray.dz * z) / hz + 1.f) / 2.f) * scz;
        Taps t;
        taps_ac_true(px, py, pz, W, H, D, t);
        if (!t.any) continue;                     // d = 0 exactly: weight 0, T unchanged
        float d = 0.f;
        float4 f = make_float4(0.f, 0.f, 0.f, 0.f);
#pragma unroll
        for (int k = 0; k < 8; ++k) {
            const long long o = tap_off(t, k);
            d = fmaf(t.w[k], Dn[o], d);
            f = f4_fma(t.w[k], F[o * C4], f);
        }
        const float wgt = d * T;
        acc = f4_fma(wgt, f, acc);
        depth = fmaf(wgt, z, depth);
        T *= (1.f - d);
        if (T == 0.f) break;                      // all later weights are exactly 0
    }
    const long long plane = (long long)Hr * Wr, pix = (long long)h * Wr + w;
    // channels-last feature map [V][Hr][Wr][C]: one 16-byte store per lane, what the conv_rgb GEMM consumes
    reinterpret_cast<float4*>(out_feat)[((long long)v * plane + pix) * C4 + cg] = acc;
    if (cg == 0) {
        out_opac[(long long)v * plane + pix] = 1.f - T;
        if (out_depth) out_depth[(long long)v * plane + pix] = depth;
    }
}

// Backward = two deterministic launches (no atomics: bit-identical gradients run to run).
//
//   dL/dd_s = T_s (a_s - Q_s),  Q_{s-1} = a_s d_s + (1 - d_s) Q_s,  Q_{S-1} = -g_opacity,
//   a_s = sum_c g_c f_sc + g_depth z_s          (no division by (1 - d_s): densities may be 1)
//   dL/df_sc = T_s d_s g_c
//
// (1) render_bwd_rays_kernel (ray-parallel, the forward's decomposition: C/4 lanes per ray, 4x4 pixel quads per wave): re-march, park
//     (d_s, a_s) of every sample in LDS, run the Q recurrence backwards and the transmittance forwards over the parked values, and
//     write the two per-sample SCALARS every volume gradient is made of - (dL/dd_s, T_s d_s) - to the workspace G [V][S][Hr][Wr][2]
//     (8.4 MB per 128^2 x 64 view). With CAM the same pass re-samples the taps for d loss / d (ray origin, direction) and leaves one
//     16-float partial per workgroup; render_bwd_cam_reduce_kernel sums them in a fixed order.
// (2) render_bwd_voxels_kernel (voxel-parallel GATHER): one lane per voxel of every volume. For each view of its volume it finds the
//     depth planes s whose samples can lie inside the voxel's (-1, 1)^3 neighbourhood (|z_s - z_voxel| < sum_a |R_2a| voxel_a) and,
//     per plane, the small pixel rectangle whose samples can (positions are affine in the pixel index: a 2x2 solve on the best-
//     conditioned pair of axes, as the forward's taps see them), re-evaluates those rays' sample positions with the FORWARD's
//     expressions, and sums w(q, p) (dL/dd_s | T_s d_s g_c) with w(q, p) = prod_a (1 - |p_a - q_a|)+ = the forward's trilinear tap
//     weight of voxel q. Every voxel-channel is WRITTEN once (no zero-fill, no atomics): the scatter of 143 M tap contributions per
//     view becomes ~20 gathered samples per voxel and view (0.55 voxel / pixel, 1.5 voxels / sample at 64^3).
// Round 3's in-workgroup gather + one atomic per touched voxel-channel and step took 0.16 ms / view and was order-dependent in the
// last bits (the noise floor of every training-gradient tolerance); this form costs the forward's march once more plus ~0.02 ms / view.
template <int C4, bool CAM>
__global__ __launch_bounds__(256) void render_bwd_rays_kernel(const float4* __restrict__ feat, const float* __restrict__ dens,
                                                              const float* __restrict__ cams, const int* __restrict__ view2vol,
                                                              const float* __restrict__ g_feat, const float* __restrict__ g_opac,
                                                              const float* __restrict__ g_depth, float2* __restrict__ G,
                                                              float* __restrict__ cam_part, float2* __restrict__ sab, int D, int H, int W, int Hr, int Wr,
                                                              int S, float zmin, float zmax, float hx, float hy, float hz, int V, int band_order) {
    constexpr int RPB = 256 / C4, TH = RPB / 8;
    extern __shared__ __attribute__((aligned(16))) float lds[];   // [2][RPB][S + 1]: (d_s, a_s) -> (T_s d_s, dL/dd_s); row pad: a wave's rays hit distinct banks
    const int SP = S + 1;
    float* lds_d = lds;
    float* lds_a = lds + (size_t)RPB * SP;
    // workgroup -> (view, pixel tile) as in render_fwd_kernel: with band_order an XCD marches one band of tile rows of EVERY view, so that its
    // L2 serves all views from the slab of the volume it has already fetched (PMC, 10 views of one 64^3 volume: 342 MB -> see profiles/ r04)
    const unsigned nx = (unsigned)(Wr + 7) / 8, ny = (unsigned)(Hr + TH - 1) / TH;
    unsigned bxu, byu, bzu;
    if (band_order) {
        const unsigned g = blockIdx.x, xcd = g % NUM_XCD, k = g / NUM_XCD, rows_per = ny / NUM_XCD;
        bzu = k % (unsigned)V;
        const unsigned t = k / (unsigned)V;
        byu = xcd * rows_per + t / nx;
        bxu = t % nx;
    } else {
        bxu = blockIdx.x % nx;
        byu = (blockIdx.x / nx) % ny;
        bzu = blockIdx.x / (nx * ny);
    }
    const int v = (int)bzu;
    const int cg = threadIdx.x % C4, r = threadIdx.x / C4;
    int lx, ly;
    tile_pixel(r, lx, ly);
    const int w = (int)bxu * 8 + lx, h = (int)byu * TH + ly;
    const bool inside = (w < Wr) && (h < Hr);
    const float* cam = cams + v * 16;
    const long long nvox = (long long)D * H * W;
    const long long vbase = (long long)view2vol[v] * nvox;
    const float4* F = feat + vbase * C4 + cg;
    const float* Dn = dens + vbase;

    const RayCam ray = make_ray(cam, min(w, Wr - 1), min(h, Hr - 1));
    const float step = (zmax - zmin) / (float)(S - 1);
    int s0 = 0, s1 = -1;
    if (inside) ray_interval(ray, hx, hy, hz, W, H, D, S, zmin, step, s0, s1);
    const float scx = (float)(W - 1), scy = (float)(H - 1), scz = (float)(D - 1);
    const long long plane = (long long)Hr * Wr, pix = (long long)min(h, Hr - 1) * Wr + min(w, Wr - 1);
    float4 g = make_float4(0.f, 0.f, 0.f, 0.f);
    float gop = 0.f, gdep = 0.f;
    if (inside) {
        g = reinterpret_cast<const float4*>(g_feat)[((long long)v * plane + pix) * C4 + cg];
        gop = g_opac[(long long)v * plane + pix];
        if (g_depth) gdep = g_depth[(long long)v * plane + pix];
    }
    float* row_d = lds_d + r * SP;
    float* row_a = lds_a + r * SP;
    for (int s = cg; s < S; s += C4) { row_d[s] = 0.f; row_a[s] = 0.f; }     // samples outside the ray's interval: exact zeros in G
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "workgroup");
    __builtin_amdgcn_wave_barrier();
    // pass A (forward march over the ray's sample interval): d_s and a_s. All C4 lanes of a ray run the same trip count, so the
    // xor-shuffles are executed by whole ray groups. No early break at T == 0: the tail's (d, a) still enter Q.
#pragma unroll 2
    for (int s = s0; s <= s1; ++s) {
        const float z = sample_depth(s, S, zmin, zmax, step);
        const float px = (((ray.ox + ray.dx * z) / hx + 1.f) / 2.f) * scx;
        const float py = (((ray.oy + ray.dy * z) / hy + 1.f) / 2.f) * scy;
        const float pz = (((ray.oz + ray.dz * z) / hz + 1.f) / 2.f) * scz;
        Taps t;
        taps_ac_true(px, py, pz, W, H, D, t);
        float d = 0.f;
        float4 f = make_float4(0.f, 0.f, 0.f, 0.f);
        // CAM: the sample's value is trilinear in its position p, so d loss / d p = T_s d_s SA + (dL/dd_s) SB with SA = sum_k (d w_k / d p)
        // (F_k . g) and SB = sum_k (d w_k / d p) Dn_k - six scalars made HERE from the corners this pass loads anyway and parked per sample
        // (`sab`), so that pass C, which learns T_s d_s and dL/dd_s, does not gather the eight corners a second time (0.27 of the 0.36 ms this
        // launch took per refinement iteration)
        float sax = 0.f, say = 0.f, saz = 0.f, sbx = 0.f, sby = 0.f, sbz = 0.f;
        if (t.any) {
#pragma unroll
            for (int k = 0; k < 8; ++k) {
                const long long o = tap_off(t, k);
                const float dn = Dn[o];
                const float4 fv = F[o * C4];
                d = fmaf(t.w[k], dn, d);
                f = f4_fma(t.w[k], fv, f);
                if constexpr (CAM) {
                    const int dx = k & 1, dy = (k >> 1) & 1, dz = k >> 2;
                    const float cx = t.bx[dx] * t.ay[dy] * t.az[dz], cy = t.ax[dx] * t.by[dy] * t.az[dz], cz = t.ax[dx] * t.ay[dy] * t.bz[dz];
                    const float q = fv.x * g.x + fv.y * g.y + fv.z * g.z + fv.w * g.w;
                    sax = fmaf(cx, q, sax); say = fmaf(cy, q, say); saz = fmaf(cz, q, saz);
                    sbx = fmaf(cx, dn, sbx); sby = fmaf(cy, dn, sby); sbz = fmaf(cz, dn, sbz);
                }
            }
        }
        float a = g.x * f.x + g.y * f.y + g.z * f.z + g.w * f.w;
#pragma unroll
        for (int o = 1; o < C4; o <<= 1) a += __shfl_xor(a, o, 64);
        a = fmaf(gdep, z, a);
        if (cg == 0) { row_d[s] = d; row_a[s] = a; }
        if constexpr (CAM) {
#pragma unroll
            for (int o = 1; o < C4; o <<= 1) { sax += __shfl_xor(sax, o, 64); say += __shfl_xor(say, o, 64); saz += __shfl_xor(saz, o, 64); }
            if (cg == 0) {
                float2* sp = sab + (((long long)v * plane + pix) * S + s) * 3;
                sp[0] = make_float2(sax, say); sp[1] = make_float2(saz, sbx); sp[2] = make_float2(sby, sbz);
            }
        }
    }
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "workgroup");        // each ray's rows are written and read by its own C4 lanes (same wave for C4 <= 64)
    __builtin_amdgcn_wave_barrier();
    // pass B (reverse, LDS only): X_s = a_s - Q_s parked over a_s
    float Q = -gop;
    for (int s = s1; s >= s0; --s) {
        const float d = row_d[s], a = row_a[s];
        __builtin_amdgcn_wave_barrier();                          // all lanes of the ray have read a_s before lane 0 overwrites it
        if (cg == 0) row_a[s] = a - Q;
        Q = fmaf(a, d, (1.f - d) * Q);
    }
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "workgroup");
    __builtin_amdgcn_wave_barrier();
    // pass C (forward): T_s in the forward's multiplication order; (dL/dd_s, T_s d_s) parked, camera gradients accumulated
    float Go[3] = {0.f, 0.f, 0.f}, Gd[3] = {0.f, 0.f, 0.f};     // d loss / d (ray origin, ray direction), this lane's share
    float T = 1.f;
    for (int s = s0; s <= s1; ++s) {
        const float d = row_d[s], X = row_a[s];
        const float dLdd = T * X, wgt = d * T;
        __builtin_amdgcn_wave_barrier();
        if (cg == 0) { row_d[s] = wgt; row_a[s] = dLdd; }
        T *= (1.f - d);
        if (CAM && cg == 0 && (wgt != 0.f || dLdd != 0.f)) {
            // d loss / d sample position from the scalars pass A parked (summed over the ray's channel lanes there); everything downstream is
            // linear, so rays are summed once at the end
            const float2* sp = sab + (((long long)v * plane + pix) * S + s) * 3;
            const float2 q0 = sp[0], q1 = sp[1], q2 = sp[2];
            const float gpx = fmaf(wgt, q0.x, dLdd * q1.y), gpy = fmaf(wgt, q0.y, dLdd * q2.x), gpz = fmaf(wgt, q1.x, dLdd * q2.y);
            const float z = sample_depth(s, S, zmin, zmax, step);
            const float kx = 0.5f * scx / hx, ky = 0.5f * scy / hy, kz = 0.5f * scz / hz;
            Go[0] = fmaf(kx, gpx, Go[0]); Go[1] = fmaf(ky, gpy, Go[1]); Go[2] = fmaf(kz, gpz, Go[2]);
            Gd[0] = fmaf(kx * z, gpx, Gd[0]); Gd[1] = fmaf(ky * z, gpy, Gd[1]); Gd[2] = fmaf(kz * z, gpz, Gd[2]);
        }
    }
    __syncthreads();
    // write-out, plane-major G [V][S][Hr][Wr] (what the voxel gather wants: the lanes of a wave - neighbouring voxels - read neighbouring
    // pixels of one depth plane): per (s, tile row) 8 pixels = 64 contiguous bytes; samples outside a ray's interval are exact zeros
    {
        const int x0 = (int)bxu * 8, y0 = (int)byu * TH;
        for (int idx = threadIdx.x; idx < RPB * S; idx += 256) {
            const int s = idx / RPB, pp = idx - s * RPB;            // pp = py * 8 + px inside the 8 x TH tile
            const int px_ = pp & 7, py_ = pp >> 3;
            const int ww = x0 + px_, hh = y0 + py_;
            if (ww >= Wr || hh >= Hr) continue;
            const int rr = (px_ & 3) | ((py_ & 3) << 2) | ((px_ >> 2) << 4) | ((py_ >> 2) << 5);   // inverse of tile_pixel
            G[(((long long)v * S + s) * Hr + hh) * Wr + ww] = make_float2(lds_a[rr * SP + s], lds_d[rr * SP + s]);
        }
    }
    if (CAM) {
        // chain to the packed camera (R[9], T[3], fx, fy, cx, cy): o = -R^T T, dir = R^T (dxc, dyc, 1),
        // dxc = (w + .5 - cx)/fx, dyc = (h + .5 - cy)/fy; then one block reduction in a fixed order -> this workgroup's partial
        float dc[16];
        const float dxc = ((float)min(w, Wr - 1) + 0.5f - cam[14]) / cam[12], dyc = ((float)min(h, Hr - 1) + 0.5f - cam[15]) / cam[13];
        const float dcv[3] = {dxc, dyc, 1.f};
#pragma unroll
        for (int j = 0; j < 3; ++j)
#pragma unroll
            for (int i = 0; i < 3; ++i) dc[j * 3 + i] = inside ? (-Go[i] * cam[9 + j] + Gd[i] * dcv[j]) : 0.f;
#pragma unroll
        for (int j = 0; j < 3; ++j) dc[9 + j] = inside ? -(Go[0] * cam[j * 3] + Go[1] * cam[j * 3 + 1] + Go[2] * cam[j * 3 + 2]) : 0.f;
        const float gdx = Gd[0] * cam[0] + Gd[1] * cam[1] + Gd[2] * cam[2], gdy = Gd[0] * cam[3] + Gd[1] * cam[4] + Gd[2] * cam[5];
        dc[12] = inside ? -gdx * dxc / cam[12] : 0.f;
        dc[13] = inside ? -gdy * dyc / cam[13] : 0.f;
        dc[14] = inside ? -gdx / cam[12] : 0.f;
        dc[15] = inside ? -gdy / cam[13] : 0.f;
        __shared__ float red[16][4];
        const int lane = threadIdx.x & 63, wv = threadIdx.x >> 6;
#pragma unroll
        for (int i = 0; i < 16; ++i) {
            float sacc = dc[i];
#pragma unroll
            for (int o = 32; o > 0; o >>= 1) sacc += __shfl_down(sacc, o, 64);
            if (lane == 0) red[i][wv] = sacc;
        }
        __syncthreads();
        const long long blk = ((long long)v * ny + byu) * nx + bxu;
        if (threadIdx.x < 16) cam_part[blk * 16 + threadIdx.x] = (red[threadIdx.x][0] + red[threadIdx.x][1]) + (red[threadIdx.x][2] + red[threadIdx.x][3]);
    }
}

// dcam [V][16] = the per-workgroup partials of one view summed in a fixed order (deterministic): workgroup = view, thread = (slice j of the
// partial list, camera entry k); 16 slices are summed in parallel, then in slice order
__global__ __launch_bounds__(256) void render_bwd_cam_reduce_kernel(const float* __restrict__ cam_part, float* __restrict__ dcam, int V, int nblk) {
    __shared__ float red[16][16];
    const int v = blockIdx.x, k = threadIdx.x & 15, j = threadIdx.x >> 4;
    const float* p = cam_part + (long long)v * nblk * 16 + k;
    float acc = 0.f;
    for (int b = j; b < nblk; b += 16) acc += p[(long long)b * 16];
    red[j][k] = acc;
    __syncthreads();
    if (threadIdx.x < 16) {
        float s = 0.f;
#pragma unroll
        for (int i = 0; i < 16; ++i) s += red[i][threadIdx.x];
        dcam[v * 16 + threadIdx.x] = s;
    }
}

template <int C4>
__global__ __launch_bounds__(256) void render_bwd_voxels_kernel(const float* __restrict__ cams, const int* __restrict__ view2vol,
                                                                const float* __restrict__ g_feat, const float2* __restrict__ G,
                                                                float* __restrict__ dfeat, float* __restrict__ ddens,
                                                                int V, int D, int H, int W, int Hr, int Wr, int S, float zmin,
                                                                float zmax, float hx, float hy, float hz) {
    const long long nvox = (long long)D * H * W;
    const int n = blockIdx.y;
    const long long o = (long long)blockIdx.x * 256 + threadIdx.x;       // voxel of volume n
    const bool live = o < nvox;
    const long long oc = live ? o : nvox - 1;
    const int qx = (int)(oc % W), qy = (int)((oc / W) % H), qz = (int)(oc / ((long long)W * H));
    const float fqx = (float)qx, fqy = (float)qy, fqz = (float)qz;
    const float scx = (float)(W - 1), scy = (float)(H - 1), scz = (float)(D - 1);
    const float step = (zmax - zmin) / (float)(S - 1);
    // world position of the voxel centre (the inverse of the forward's pix = ((x / h + 1) / 2) (N - 1)) and the voxel pitch
    const float xw = (2.f * fqx / scx - 1.f) * hx, yw = (2.f * fqy / scy - 1.f) * hy, zw = (2.f * fqz / scz - 1.f) * hz;
    const float vsx = 2.f * hx / scx, vsy = 2.f * hy / scy, vsz = 2.f * hz / scz;
    const float kx = 0.5f * scx / hx, ky = 0.5f * scy / hy, kz = 0.5f * scz / hz;     // voxel coordinates per world unit
    float4 acc[C4];
#pragma unroll
    for (int c = 0; c < C4; ++c) acc[c] = make_float4(0.f, 0.f, 0.f, 0.f);
    float accd = 0.f;
    __shared__ int vlist[64];                                            // views of volume n, 64 at a time
    __shared__ int vcount, vnext;
    int vscan = 0;
    while (vscan < V) {
        __syncthreads();
        if (threadIdx.x == 0) {
            int c = 0, v = vscan;
            for (; v < V && c < 64; ++v)
                if (view2vol[v] == n) vlist[c++] = v;
            vcount = c; vnext = v;
        }
        __syncthreads();
        const int nviews = vcount;
        vscan = vnext;
        for (int j = 0; j < nviews; ++j) {
            const int v = vlist[j];
            const float* cam = cams + v * 16;
            // camera-z of the voxel centre and the half extent in camera-z of its (-1, 1)^3-voxel neighbourhood: samples of depth plane s
            // all have camera-z = z_s exactly, so only planes with |z_s - zq| < dzq can hold a sample with a tap on this voxel
            const float zq = fmaf(cam[6], xw, fmaf(cam[7], yw, fmaf(cam[8], zw, cam[11])));
            const float dzq = (fabsf(cam[6]) * vsx + fabsf(cam[7]) * vsy + fabsf(cam[8]) * vsz) * 1.001f + 1e-6f;
            const float fs_lo = (zq - dzq - zmin) / step, fs_hi = (zq + dzq - zmin) / step;
            if (!(fs_hi >= -0.02f) || !(fs_lo <= (float)(S - 1) + 0.02f)) continue;
            const int s_lo = max(0, (int)ceilf(fs_lo - 0.02f)), s_hi = min(S - 1, (int)floorf(fs_hi + 0.02f));
            const RayCam c00 = make_ray(cam, 0, 0);
            const float dUx = cam[0] / cam[12], dUy = cam[1] / cam[12], dUz = cam[2] / cam[12];     // d direction / d pixel column
            const float dVx = cam[3] / cam[13], dVy = cam[4] / cam[13], dVz = cam[5] / cam[13];     // d direction / d pixel row
            const float2* Gv = G + (long long)v * S * Hr * Wr;
            const float4* gv = reinterpret_cast<const float4*>(g_feat) + (long long)v * Hr * Wr * C4;
            for (int s = s_lo; s <= s_hi; ++s) {
                const float z = sample_depth(s, S, zmin, zmax, step);
                // sample positions of plane s in voxel coordinates are affine in the pixel index: p(x, y) = P0 + x U + y V (rounding aside);
                // relative to the voxel: e(x, y) = p - q
                const float E0x = fmaf(fmaf(c00.dx, z, c00.ox), kx, 0.5f * scx) - fqx, E0y = fmaf(fmaf(c00.dy, z, c00.oy), ky, 0.5f * scy) - fqy,
                            E0z = fmaf(fmaf(c00.dz, z, c00.oz), kz, 0.5f * scz) - fqz;
                const float Ux = z * dUx * kx, Uy = z * dUy * ky, Uz = z * dUz * kz;
                const float Vx = z * dVx * kx, Vy = z * dVy * ky, Vz = z * dVz * kz;
                // |e_a| < 1 on the two axes (a, b) with the best-conditioned 2x2 system confines (x, y) to a small rectangle
                const float det01 = Ux * Vy - Uy * Vx, det02 = Ux * Vz - Uz * Vx, det12 = Uy * Vz - Uz * Vy;
                float Ua = Ux, Va = Vx, ra = -E0x, Ub = Uy, Vb = Vy, rb = -E0y, det = det01;
                if (fabsf(det02) > fabsf(det)) { Ub = Uz; Vb = Vz; rb = -E0z; det = det02; }
                if (fabsf(det12) > fabsf(det)) { Ua = Uy; Va = Vy; ra = -E0y; Ub = Uz; Vb = Vz; rb = -E0z; det = det12; }
                int x0 = 0, x1 = Wr - 1, y0 = 0, y1 = Hr - 1;
                if (fabsf(det) > 1e-12f) {
                    const float idet = 1.f / det;
                    const float cxp = (Vb * ra - Va * rb) * idet, cyp = (Ua * rb - Ub * ra) * idet;
                    const float ext_x = (fabsf(Vb) + fabsf(Va)) * fabsf(idet) + 0.05f, ext_y = (fabsf(Ub) + fabsf(Ua)) * fabsf(idet) + 0.05f;
                    if (!(cxp + ext_x >= 0.f) || !(cyp + ext_y >= 0.f)) continue;                 // left / above the image (also catches NaN)
                    // clamp in float first: a far-away voxel's centre may not fit an int
                    x0 = (int)fmaxf(ceilf(cxp - ext_x), 0.f); x1 = (int)fminf(floorf(cxp + ext_x), (float)(Wr - 1));
                    y0 = (int)fmaxf(ceilf(cyp - ext_y), 0.f); y1 = (int)fminf(floorf(cyp + ext_y), (float)(Hr - 1));
                }
                for (int yy = y0; yy <= y1; ++yy) {
                    const float rx = fmaf((float)yy, Vx, E0x), ry = fmaf((float)yy, Vy, E0y), rz = fmaf((float)yy, Vz, E0z);
                    for (int xx = x0; xx <= x1; ++xx) {
                        // cheap reject on the affine form (its rounding is ~1e-5 voxel: a 1e-3 guard band), then the forward's own expressions
                        const float ax_ = fmaf((float)xx, Ux, rx), ay_ = fmaf((float)xx, Uy, ry), az_ = fmaf((float)xx, Uz, rz);
                        if (!(fabsf(ax_) < 1.001f && fabsf(ay_) < 1.001f && fabsf(az_) < 1.001f)) continue;
                        const float2 gs = Gv[((long long)s * Hr + yy) * Wr + xx];            // (dL/dd_s, T_s d_s)
                        if (gs.x == 0.f && gs.y == 0.f) continue;
                        const RayCam rc = make_ray(cam, xx, yy);
                        float px = (((rc.ox + rc.dx * z) / hx + 1.f) / 2.f) * scx;
                        float py = (((rc.oy + rc.dy * z) / hy + 1.f) / 2.f) * scy;
                        float pz = (((rc.oz + rc.dz * z) / hz + 1.f) / 2.f) * scz;
                        px = fminf(fmaxf(px, -2.f), (float)W + 1.f);                        // taps_ac_true's clamp
                        py = fminf(fmaxf(py, -2.f), (float)H + 1.f);
                        pz = fminf(fmaxf(pz, -2.f), (float)D + 1.f);
                        const float ddx = px - fqx, ddy = py - fqy, ddz = pz - fqz;
                        if (fabsf(ddx) < 1.f && fabsf(ddy) < 1.f && fabsf(ddz) < 1.f) {
                            // the forward's weight expressions: lower tap (q = floor p) (q + 1) - p, upper tap (q = floor p + 1) p - (q - 1)
                            const float wx = ddx >= 0.f ? (fqx + 1.f) - px : px - (fqx - 1.f);
                            const float wy = ddy >= 0.f ? (fqy + 1.f) - py : py - (fqy - 1.f);
                            const float wz = ddz >= 0.f ? (fqz + 1.f) - pz : pz - (fqz - 1.f);
                            const float wq = wx * wy * wz;
                            accd = fmaf(wq, gs.x, accd);
                            if (gs.y != 0.f) {
                                const float wg = wq * gs.y;
                                const float4* gp = gv + ((long long)yy * Wr + xx) * C4;
#pragma unroll
                                for (int c = 0; c < C4; ++c) acc[c] = f4_fma(wg, gp[c], acc[c]);
                            }
                        }
                    }
                }
            }
        }
    }
    if (live) {
        float4* df = reinterpret_cast<float4*>(dfeat) + ((long long)n * nvox + o) * C4;
#pragma unroll
        for (int c = 0; c < C4; ++c) df[c] = acc[c];
        ddens[(long long)n * nvox + o] = accd;
    }
}

// ---- bilinear resize of the opacity / depth planes to the image size (models/volume_render.py:69,74: F.upsample(size=img_size, mode='bilinear'),
// i.e. align_corners=False). ATen's arithmetic, expression for expression (area_pixel_compute_source_index + upsample_bilinear2d_out_frame):
//   src = max(scale (dst + 0.5) - 0.5, 0), scale = in / out;  i1 = (int)src, i1p = i1 < in - 1, l1 = src - i1, l0 = 1 - l1
//   out = h0 (w0 v[h1][w1] + w1l v[h1][w1 + w1p]) + h1l (w0 v[h1 + h1p][w1] + w1l v[h1 + h1p][w1 + w1p])
__device__ __forceinline__ void bilinear_src(int dst, float scale, int n_in, int& i1, int& ip, float& l0, float& l1) {
    const float src = fmaxf(scale * ((float)dst + 0.5f) - 0.5f, 0.f);
    i1 = (int)src;
    ip = (i1 < n_in - 1) ? 1 : 0;
    l1 = src - (float)i1;
    l0 = 1.f - l1;
}

__global__ __launch_bounds__(256) void resize_bilinear_fwd_kernel(const float* __restrict__ in, float* __restrict__ out, int P, int Hi, int Wi, int Ho, int Wo,
                                                                  float sh, float sw) {
    const long long idx = (long long)blockIdx.x * 256 + threadIdx.x;
    if (idx >= (long long)P * Ho * Wo) return;
    const int X = (int)(idx % Wo), Y = (int)((idx / Wo) % Ho);
    const long long p = idx / ((long long)Wo * Ho);
    int h1, hp, w1, wp;
    float h0l, h1l, w0l, w1l;
    bilinear_src(Y, sh, Hi, h1, hp, h0l, h1l);
    bilinear_src(X, sw, Wi, w1, wp, w0l, w1l);
    const float* v = in + p * Hi * Wi;
    out[idx] = h0l * (w0l * v[h1 * Wi + w1] + w1l * v[h1 * Wi + w1 + wp]) + h1l * (w0l * v[(h1 + hp) * Wi + w1] + w1l * v[(h1 + hp) * Wi + w1 + wp]);
}

// adjoint as a gather per INPUT pixel (deterministic, no atomics): the output rows / columns that can reference input index i lie in
// ( (i - 0.5) / scale - 0.5, (i + 1.5) / scale - 0.5 ) for any scale; each candidate's taps / weights are re-evaluated with the forward's
// own expressions.
__global__ __launch_bounds__(256) void resize_bilinear_bwd_kernel(const float* __restrict__ g, float* __restrict__ din, int P, int Hi, int Wi, int Ho, int Wo,
                                                                  float sh, float sw) {
    const long long idx = (long long)blockIdx.x * 256 + threadIdx.x;
    if (idx >= (long long)P * Hi * Wi) return;
    const int x = (int)(idx % Wi), y = (int)((idx / Wi) % Hi);
    const long long p = idx / ((long long)Wi * Hi);
    // src(Y) = sh (Y + 0.5) - 0.5 taps rows floor(src) and floor(src) + 1: row y is referenced iff y - 1 < src(Y) < y + 1, i.e.
    // (y - 0.5) / sh - 0.5 < Y < (y + 1.5) / sh - 0.5 (+- 1 against rounding; the clamp of src at 0 only adds outputs below that range
    // for y = 0, which the max(0, .) covers)
    const int Y0 = max(0, (int)ceilf(((float)y - 0.5f) / sh - 0.5f) - 1), Y1 = min(Ho - 1, (int)floorf(((float)y + 1.5f) / sh - 0.5f) + 1);
    const int X0 = max(0, (int)ceilf(((float)x - 0.5f) / sw - 0.5f) - 1), X1 = min(Wo - 1, (int)floorf(((float)x + 1.5f) / sw - 0.5f) + 1);
    const float* gp = g + p * Ho * Wo;
    float acc = 0.f;
    for (int Y = Y0; Y <= Y1; ++Y) {
        int h1, hp; float h0l, h1l;
        bilinear_src(Y, sh, Hi, h1, hp, h0l, h1l);
        const float wy = (h1 == y ? h0l : 0.f) + (h1 + hp == y ? h1l : 0.f);       // hp = 0: both taps are row h1 (weights h0l + h1l = 1)
        if (wy == 0.f) continue;
        for (int X = X0; X <= X1; ++X) {
            int w1, wp; float w0l, w1l;
            bilinear_src(X, sw, Wi, w1, wp, w0l, w1l);
            const float wx = (w1 == x ? w0l : 0.f) + (w1 + wp == x ? w1l : 0.f);
            if (wx != 0.f) acc = fmaf(wy * wx, gp[Y * Wo + X], acc);
        }
    }
    din[idx] = acc;
}

static int check_render_args(const char* fn, const void* feat, const void* dens, const void* cam, const void* v2v,
                             int V, int nvol, int C, int D, int H, int W, int Hr, int Wr, int S, float hx, float hy, float hz) {
    FORGE_REQUIRE(feat && dens && cam && v2v, FORGE_EINVAL, "%s: null pointer argument", fn);
    FORGE_REQUIRE(V > 0 && nvol > 0 && D > 1 && H > 1 && W > 1 && Hr > 0 && Wr > 0 && S > 1, FORGE_EINVAL,
                  "%s: bad dims V=%d nvol=%d D=%d H=%d W=%d Hr=%d Wr=%d S=%d", fn, V, nvol, D, H, W, Hr, Wr, S);
    FORGE_REQUIRE(C == 4 || C == 8 || C == 16 || C == 32, FORGE_ESHAPE, "%s: C=%d unsupported (4, 8, 16 or 32)", fn, C);
    FORGE_REQUIRE(hx > 0.f && hy > 0.f && hz > 0.f, FORGE_EINVAL, "%s: half extents must be > 0", fn);
    FORGE_REQUIRE(V <= 65535, FORGE_ESHAPE, "%s: V=%d exceeds gridDim.z", fn, V);
    return 0;
}

}  // namespace forge

using namespace forge;

#define FORGE_DISPATCH_C4(C, ...)                      \
    switch ((C) / 4) {                                 \
        case 1: { constexpr int C4 = 1; __VA_ARGS__; } break; \
        case 2: { constexpr int C4 = 2; __VA_ARGS__; } break; \
        case 4: { constexpr int C4 = 4; __VA_ARGS__; } break; \
        case 8: { constexpr int C4 = 8; __VA_ARGS__; } break; \
    }

extern "C" int forge_render_fwd(const float* feat, const float* dens, const float* cam, const int* view2vol,
                                float* out_feat, float* out_opac, float* out_depth,
                                int V, int nvol, int C, int D, int H, int W, int Hr, int Wr, int S,
                                float zmin, float zmax, float hx, float hy, float hz, forge_stream_t stream) {
    if (int rc = check_render_args("forge_render_fwd", feat, dens, cam, view2vol, V, nvol, C, D, H, W, Hr, Wr, S, hx, hy, hz)) return rc;
    FORGE_REQUIRE(out_feat && out_opac, FORGE_EINVAL, "forge_render_fwd: null output pointer");
    FORGE_DISPATCH_C4(C, {
        constexpr int TH = (256 / C4) / 8;
        const unsigned nx = (unsigned)(Wr + 7) / 8, ny = (unsigned)(Hr + TH - 1) / TH;
        const int band_order = (ny % NUM_XCD == 0) ? 1 : 0;      // tile rows split evenly over the XCDs (128 rows: 16 tile rows); else launch order
        hipLaunchKernelGGL(render_fwd_kernel<C4>, dim3(nx * ny * (unsigned)V), dim3(256), 0, (hipStream_t)stream, (const float4*)feat, dens, cam,
                           view2vol, out_feat, out_opac, out_depth, D, H, W, Hr, Wr, S, zmin, zmax, hx, hy, hz, V, band_order);
    });
    FORGE_LAUNCH_CHECK("forge_render_fwd");
    return 0;
}

// bytes of workspace forge_render_bwd needs: G [V][S][Hr][Wr] float2 (+ per-workgroup camera partials and the per-sample position-gradient scalars when dcam is requested)
static size_t render_bwd_ws_layout(int V, int C, int Hr, int Wr, int S, int want_cam, size_t* cam_off, size_t* sab_off = nullptr) {
    const int C4 = C / 4, RPB = 256 / C4, TH = RPB / 8;
    const size_t g_bytes = (size_t)V * Hr * Wr * S * sizeof(float2);
    const size_t nblk = (size_t)((Wr + 7) / 8) * ((Hr + TH - 1) / TH);
    const size_t cam_bytes = want_cam ? ((size_t)V * nblk * 16 * sizeof(float) + 15) / 16 * 16 : 0;
    if (cam_off) *cam_off = g_bytes;
    if (sab_off) *sab_off = g_bytes + cam_bytes;
    // + with camera gradients: six position-gradient scalars per sample, [V][Hr][Wr][S][6] (render_bwd_rays_kernel, pass A -> pass C)
    return g_bytes + cam_bytes + (want_cam ? (size_t)V * Hr * Wr * S * 6 * sizeof(float) : 0);
}

extern "C" long long forge_render_bwd_ws_bytes(int V, int C, int Hr, int Wr, int S, int want_cam) {
    if (V <= 0 || Hr <= 0 || Wr <= 0 || S <= 1 || !(C == 4 || C == 8 || C == 16 || C == 32)) return -1;
    return (long long)render_bwd_ws_layout(V, C, Hr, Wr, S, want_cam, nullptr);
}

extern "C" int forge_render_bwd(const float* feat, const float* dens, const float* cam, const int* view2vol,
                                const float* g_feat, const float* g_opac, const float* g_depth,
                                float* dfeat, float* ddens, float* dcam,
                                int V, int nvol, int C, int D, int H, int W, int Hr, int Wr, int S,
                                float zmin, float zmax, float hx, float hy, float hz, void* ws, long long ws_bytes, forge_stream_t stream) {
    if (int rc = check_render_args("forge_render_bwd", feat, dens, cam, view2vol, V, nvol, C, D, H, W, Hr, Wr, S, hx, hy, hz)) return rc;
    FORGE_REQUIRE(g_feat && g_opac && dfeat && ddens, FORGE_EINVAL, "forge_render_bwd: null gradient pointer");
    FORGE_REQUIRE(nvol <= 65535, FORGE_ESHAPE, "forge_render_bwd: nvol=%d exceeds gridDim.y", nvol);
    size_t cam_off = 0, sab_off = 0;
    const size_t need = render_bwd_ws_layout(V, C, Hr, Wr, S, dcam != nullptr, &cam_off, &sab_off);
    FORGE_REQUIRE(ws && ws_bytes >= (long long)need, FORGE_EINVAL, "forge_render_bwd: workspace of %lld B given, %zu B needed (forge_render_bwd_ws_bytes)",
                  ws_bytes, need);
    FORGE_REQUIRE(((size_t)ws & 15) == 0, FORGE_EINVAL, "forge_render_bwd: workspace must be 16-byte aligned");
    const size_t lds_bytes = (size_t)2 * (256 / (C / 4)) * (S + 1) * sizeof(float);
    FORGE_REQUIRE(lds_bytes <= 160 * 1024 - 2048, FORGE_ESHAPE, "forge_render_bwd: S=%d needs %zu B of LDS (> 160 KiB)", S, lds_bytes);
    float2* G = (float2*)ws;
    float* cam_part = (float*)((char*)ws + cam_off);
    float2* sab = (float2*)((char*)ws + sab_off);
    const long long nvox = (long long)D * H * W;
    FORGE_REQUIRE((nvox + 255) / 256 < (1ll << 31), FORGE_ESHAPE, "forge_render_bwd: grid too large");
    FORGE_DISPATCH_C4(C, {
        constexpr int TH = (256 / C4) / 8;
        const unsigned nxg = (unsigned)(Wr + 7) / 8, nyg = (unsigned)(Hr + TH - 1) / TH;
        const int band_order = (nyg % NUM_XCD == 0) ? 1 : 0;     // as forge_render_fwd
        dim3 grid(nxg * nyg * (unsigned)V);
        if (dcam) {
            FORGE_SET_MAX_LDS_ONCE((render_bwd_rays_kernel<C4, true>), 160 * 1024 - 2048);
            hipLaunchKernelGGL((render_bwd_rays_kernel<C4, true>), grid, dim3(256), lds_bytes, (hipStream_t)stream, (const float4*)feat, dens, cam,
                               view2vol, g_feat, g_opac, g_depth, G, cam_part, sab, D, H, W, Hr, Wr, S, zmin, zmax, hx, hy, hz, V, band_order);
            hipLaunchKernelGGL(render_bwd_cam_reduce_kernel, dim3(V), dim3(256), 0, (hipStream_t)stream, (const float*)cam_part, dcam, V,
                               (int)(nxg * nyg));
        } else {
            FORGE_SET_MAX_LDS_ONCE((render_bwd_rays_kernel<C4, false>), 160 * 1024 - 2048);
            hipLaunchKernelGGL((render_bwd_rays_kernel<C4, false>), grid, dim3(256), lds_bytes, (hipStream_t)stream, (const float4*)feat, dens, cam,
                               view2vol, g_feat, g_opac, g_depth, G, cam_part, sab, D, H, W, Hr, Wr, S, zmin, zmax, hx, hy, hz, V, band_order);
        }
        hipLaunchKernelGGL(render_bwd_voxels_kernel<C4>, dim3((unsigned)((nvox + 255) / 256), nvol), dim3(256), 0, (hipStream_t)stream, cam, view2vol,
                           g_feat, (const float2*)G, dfeat, ddens, V, D, H, W, Hr, Wr, S, zmin, zmax, hx, hy, hz);
    });
    FORGE_LAUNCH_CHECK("forge_render_bwd");
    return 0;
}

// P planes [Hi][Wi] -> [Ho][Wo], bilinear, align_corners = False (the mask / depth up-sampling of models/volume_render.py:69,74).
extern "C" int forge_resize_bilinear_fwd(const float* in, float* out, int P, int Hi, int Wi, int Ho, int Wo, forge_stream_t stream) {
    FORGE_REQUIRE(in && out && P > 0 && Hi > 0 && Wi > 0 && Ho > 0 && Wo > 0, FORGE_EINVAL, "forge_resize_bilinear_fwd: bad argument");
    const long long total = (long long)P * Ho * Wo;
    FORGE_REQUIRE((total + 255) / 256 < (1ll << 31), FORGE_ESHAPE, "forge_resize_bilinear_fwd: grid too large");
    hipLaunchKernelGGL(resize_bilinear_fwd_kernel, dim3((unsigned)((total + 255) / 256)), dim3(256), 0, (hipStream_t)stream, in, out, P, Hi, Wi, Ho, Wo,
                       (float)Hi / (float)Ho, (float)Wi / (float)Wo);
    FORGE_LAUNCH_CHECK("forge_resize_bilinear_fwd");
    return 0;
}

// adjoint of the above: g [P][Ho][Wo] -> din [P][Hi][Wi] (written, not accumulated; deterministic)
extern "C" int forge_resize_bilinear_bwd(const float* g, float* din, int P, int Hi, int Wi, int Ho, int Wo, forge_stream_t stream) {
    FORGE_REQUIRE(g && din && P > 0 && Hi > 0 && Wi > 0 && Ho > 0 && Wo > 0, FORGE_EINVAL, "forge_resize_bilinear_bwd: bad argument");
    const long long total = (long long)P * Hi * Wi;
    FORGE_REQUIRE((total + 255) / 256 < (1ll << 31), FORGE_ESHAPE, "forge_resize_bilinear_bwd: grid too large");
    hipLaunchKernelGGL(resize_bilinear_bwd_kernel, dim3((unsigned)((total + 255) / 256)), dim3(256), 0, (hipStream_t)stream, g, din, P, Hi, Wi, Ho, Wo,
                       (float)Hi / (float)Ho, (float)Wi / (float)Wo);
    FORGE_LAUNCH_CHECK("forge_resize_bilinear_bwd");
    return 0;
}
