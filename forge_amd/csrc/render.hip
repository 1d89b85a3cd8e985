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

// Backward. Pass 1 re-marches the densities and parks (d_s, T_s) of every sample in LDS
// ([s][ray] so a wave's lanes hit consecutive banks); pass 2 walks the ray backwards with
//     dL/dd_s = T_s (a_s - Q_s),  Q_{s-1} = a_s d_s + (1 - d_s) Q_s,  Q_{S-1} = -g_opacity,
//     a_s = sum_c g_c f_sc + g_depth z_s          (no division by (1 - d_s): densities may be 1)
// and scatter-adds the volume gradients through the same 8 taps.
//
// The scatter is the cost: 8 taps x (C + 1) floats per sample per ray = 143 M fp32 atomics per 128^2 x 64 view when every lane
// adds its own taps to HBM, although the 64 rays of a workgroup's 8x8 pixel tile land on the same ~5x5x2 voxels at every sample
// (adjacent pixels are ~0.5 voxel apart on a 64^3 grid). LDS float atomics do not help: ds_add_f32 serialises to ~3 clocks per
// lane under these same-address collisions (measured: 5 ms of 5.8 for 10 views). So each sample step is turned from a scatter
// into a GATHER inside the workgroup:
//   phase A (ray-parallel, lane = ray x 4 channels): re-sample, a_s, dL/dd_s, Q update; park the sample's pixel-space position,
//           dL/dd_s and the vector T_s d_s g_c in LDS (double-buffered, one barrier per step);
//   phase B (voxel-parallel, lane = voxel x 4 channels): the voxels of the step's bounding box (block-uniform, from the four
//           corner rays of the tile: positions are affine in the pixel coordinates) each sum w(q, p_r) * parked vector over the
//           rays - w(q, p) = prod_a (1 - |p_a - q_a|)+ IS the trilinear tap weight, evaluated with the forward pass's expressions -
//           in registers, and issue ONE fp32 atomic per non-zero (voxel, channel).
// HBM atomics drop ~10x (one per touched voxel-channel per step), no LDS atomics, no camera-dependent fallback path.
template <int C4, bool CAM>
__global__ __launch_bounds__(256) void render_bwd_kernel(const float4* __restrict__ feat, const float* __restrict__ dens,
                                                         const float* __restrict__ cams, const int* __restrict__ view2vol,
                                                         const float* __restrict__ g_feat, const float* __restrict__ g_opac,
                                                         const float* __restrict__ g_depth, float* __restrict__ dfeat,
                                                         float* __restrict__ ddens, float* __restrict__ dcam,
                                                         int D, int H, int W, int Hr, int Wr,
                                                         int S, float zmin, float zmax, float hx, float hy, float hz) {
    constexpr int RPB = 256 / C4, TH = RPB / 8;
    extern __shared__ __attribute__((aligned(16))) float lds[];   // [2][S][RPB] (d, T)  +  2 x { [RPB] float4 (p, dL/dd), [RPB][C4] float4 T d g }
    float* lds_d = lds;
    float* lds_T = lds + (size_t)S * RPB;
    float4* stage = reinterpret_cast<float4*>(lds + (size_t)2 * S * RPB);
    constexpr int STAGE4 = RPB * (1 + C4);                         // float4 per staging buffer
    __shared__ int s_range[2];                                     // block-wide [min s0, max s_last] of the marched samples
    if (threadIdx.x == 0) { s_range[0] = 0x7fffffff; s_range[1] = -1; }
    __syncthreads();
    const int v = blockIdx.z;
    const int cg = threadIdx.x % C4, r = threadIdx.x / C4;
    int lx, ly;
    tile_pixel(r, lx, ly);
    const int w = blockIdx.x * 8 + lx, h = blockIdx.y * TH + ly;
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

    // pass 1: densities + transmittance (every lane of the ray computes the same values; lane cg==0 stores)
    float T = 1.f;
    int s_last = s0 - 1;          // last sample marched in the forward pass
    for (int s = s0; s <= s1; ++s) {
        const float z = sample_depth(s, S, zmin, zmax, step);
        const float px = (((ray.ox + ray.dx * z) / hx + 1.f) / 2.f) * scx;
        const float py = (((ray.oy + ray.dy * z) / hy + 1.f) / 2.f) * scy;
        const float pz = (((ray.oz + ray.dz * z) / hz + 1.f) / 2.f) * scz;
        Taps t;
        taps_ac_true(px, py, pz, W, H, D, t);
        float d = 0.f;
        if (t.any) {
#pragma unroll
            for (int k = 0; k < 8; ++k) d = fmaf(t.w[k], Dn[tap_off(t, k)], d);
        }
        if (cg == 0) { lds_d[s * RPB + r] = d; lds_T[s * RPB + r] = T; }
        T *= (1.f - d);
        s_last = s;
        if (T == 0.f) break;
    }
    if (cg == 0 && s_last >= s0) { atomicMin(&s_range[0], s0); atomicMax(&s_range[1], s_last); }
    __syncthreads();
    const int s_lo = s_range[0], s_hi = s_range[1];

    const long long plane = (long long)Hr * Wr, pix = (long long)min(h, Hr - 1) * Wr + min(w, Wr - 1);
    float4 g = make_float4(0.f, 0.f, 0.f, 0.f);
    float gop = 0.f, gdep = 0.f;
    if (inside) {
        g = reinterpret_cast<const float4*>(g_feat)[((long long)v * plane + pix) * C4 + cg];
        gop = g_opac[(long long)v * plane + pix];
        if (g_depth) gdep = g_depth[(long long)v * plane + pix];
    }
    // Samples after an exact T == 0 break have weight 0 and T_s = 0: dL/dd_s = 0, and Q only matters
    // multiplied by T_s = 0 further down... except through (1-d) factors of *earlier* samples, for
    // which the forward value of later samples is irrelevant once T hit 0 exactly only if the zero
    // came from the last marched sample (d = 1): Q_{s_last} would need later terms. They are all
    // multiplied by T_j = 0 in the true gradient, so starting the recurrence at s_last with
    // Q = -g_op * prod_{i > s_last}(1 - d_i) is required; that product is not known without marching
    // on. Keep it exact: when the forward pass broke early, finish marching densities here.
    float Q = -gop;
    if (inside && s_last < s1) {
        // rare path (T == 0 exactly): fold the tail's (1 - d) factors and a_j d_j terms into Q
        for (int s = s1; s > s_last; --s) {
            const float z = sample_depth(s, S, zmin, zmax, step);
            const float px = (((ray.ox + ray.dx * z) / hx + 1.f) / 2.f) * scx;
            const float py = (((ray.oy + ray.dy * z) / hy + 1.f) / 2.f) * scy;
            const float pz = (((ray.oz + ray.dz * z) / hz + 1.f) / 2.f) * scz;
            Taps t;
            taps_ac_true(px, py, pz, W, H, D, t);
            float d = 0.f;
            float4 f = make_float4(0.f, 0.f, 0.f, 0.f);
            if (t.any) {
#pragma unroll
                for (int k = 0; k < 8; ++k) {
                    const long long o = tap_off(t, k);
                    d = fmaf(t.w[k], Dn[o], d);
                    f = f4_fma(t.w[k], F[o * C4], f);
                }
            }
            float a = g.x * f.x + g.y * f.y + g.z * f.z + g.w * f.w;
#pragma unroll
            for (int o = 1; o < C4; o <<= 1) a += __shfl_xor(a, o, 64);
            a = fmaf(gdep, z, a);
            Q = fmaf(a, d, (1.f - d) * Q);
        }
    }
    float Go[3] = {0.f, 0.f, 0.f}, Gd[3] = {0.f, 0.f, 0.f};   // d loss / d (ray origin, ray direction), this lane's share
    // pass 2: reverse march, block-uniform over s (rays outside their own [s0, s_last] idle in phase A).
    // NOTE: all C4 lanes of a ray have identical (s0, s_last), so the xor-shuffles below are
    // executed by all lanes of each ray group together.
    // the four corner rays of the tile (identical in every thread): their sample positions bound those of all rays of the tile
    float cdir[4][3];
    {
        const int x0 = min((int)blockIdx.x * 8, Wr - 1), x1 = min((int)blockIdx.x * 8 + 7, Wr - 1);
        const int y0 = min((int)blockIdx.y * TH, Hr - 1), y1 = min((int)blockIdx.y * TH + TH - 1, Hr - 1);
#pragma unroll
        for (int q = 0; q < 4; ++q) {
            const RayCam c = make_ray(cam, (q & 1) ? x1 : x0, (q & 2) ? y1 : y0);
            cdir[q][0] = c.dx; cdir[q][1] = c.dy; cdir[q][2] = c.dz;
        }
    }
    float d00[3], dU[3], dV[3];                                    // direction of the tile's pixel (0, 0) and its per-pixel increments
    {
        const RayCam c = make_ray(cam, (int)blockIdx.x * 8, (int)blockIdx.y * TH);
        d00[0] = c.dx; d00[1] = c.dy; d00[2] = c.dz;
        dU[0] = cam[0] / cam[12]; dU[1] = cam[1] / cam[12]; dU[2] = cam[2] / cam[12];
        dV[0] = cam[3] / cam[13]; dV[1] = cam[4] / cam[13]; dV[2] = cam[5] / cam[13];
    }
    const int vslot = threadIdx.x / C4;                            // phase B: voxel slot of this lane (RPB slots x C4 channel groups)
    int buf = 0;
    for (int s = s_hi; s >= s_lo; --s, buf ^= 1) {
        const float z = sample_depth(s, S, zmin, zmax, step);
        float4* st_p = stage + buf * STAGE4;                       // [RPB] (px, py, pz, dL/dd)
        float4* st_g = st_p + RPB;                                 // [RPB][C4] T d g
        // ---- phase A
        int step_active = 0;
        {
            float4 park_p = make_float4(-1e30f, -1e30f, -1e30f, 0.f), park_g = make_float4(0.f, 0.f, 0.f, 0.f);
            if (s <= s_last && s >= s0) {
                float px = (((ray.ox + ray.dx * z) / hx + 1.f) / 2.f) * scx;
                float py = (((ray.oy + ray.dy * z) / hy + 1.f) / 2.f) * scy;
                float pz = (((ray.oz + ray.dz * z) / hz + 1.f) / 2.f) * scz;
                Taps t;
                taps_ac_true(px, py, pz, W, H, D, t);
                const float d = lds_d[s * RPB + r], Ts = lds_T[s * RPB + r];
                float4 f = make_float4(0.f, 0.f, 0.f, 0.f);
                if (t.any) {
#pragma unroll
                    for (int k = 0; k < 8; ++k) f = f4_fma(t.w[k], F[tap_off(t, k) * C4], f);
                }
                float a = g.x * f.x + g.y * f.y + g.z * f.z + g.w * f.w;
#pragma unroll
                for (int o = 1; o < C4; o <<= 1) a += __shfl_xor(a, o, 64);
                a = fmaf(gdep, z, a);
                const float dLdd = Ts * (a - Q);
                Q = fmaf(a, d, (1.f - d) * Q);
                const float wgt = d * Ts;
                if (CAM && t.any) {
                    // d loss / d pixel coordinate of this sample, this lane's share (its 4 channels; lane cg==0 adds the density
                    // term). Everything downstream is linear, so lanes and rays are summed once at the end.
                    float gpx = 0.f, gpy = 0.f, gpz = 0.f;
#pragma unroll
                    for (int k = 0; k < 8; ++k) {
                        const int dx = k & 1, dy = (k >> 1) & 1, dz = k >> 2;
                        const long long o = tap_off(t, k);
                        const float4 fv = F[o * C4];
                        float q = wgt * (fv.x * g.x + fv.y * g.y + fv.z * g.z + fv.w * g.w);
                        if (cg == 0) q = fmaf(Dn[o], dLdd, q);
                        gpx = fmaf(t.bx[dx] * t.ay[dy] * t.az[dz], q, gpx);
                        gpy = fmaf(t.ax[dx] * t.by[dy] * t.az[dz], q, gpy);
                        gpz = fmaf(t.ax[dx] * t.ay[dy] * t.bz[dz], q, gpz);
                    }
                    const float kx = 0.5f * scx / hx, ky = 0.5f * scy / hy, kz = 0.5f * scz / hz;
                    Go[0] = fmaf(kx, gpx, Go[0]); Go[1] = fmaf(ky, gpy, Go[1]); Go[2] = fmaf(kz, gpz, Go[2]);
                    Gd[0] = fmaf(kx * z, gpx, Gd[0]); Gd[1] = fmaf(ky * z, gpy, Gd[1]); Gd[2] = fmaf(kz * z, gpz, Gd[2]);
                }
                if (t.any && (wgt != 0.f || dLdd != 0.f)) {       // else: this sample's volume gradients are exactly 0
                    // the position the taps were built from (taps_ac_true clamps it to [-2, N + 1])
                    park_p = make_float4(fminf(fmaxf(px, -2.f), (float)W + 1.f), fminf(fmaxf(py, -2.f), (float)H + 1.f),
                                         fminf(fmaxf(pz, -2.f), (float)D + 1.f), dLdd);
                    park_g = make_float4(wgt * g.x, wgt * g.y, wgt * g.z, wgt * g.w);
                }
            }
            if (cg == 0) st_p[r] = park_p;
            st_g[r * C4 + cg] = park_g;
            step_active = park_p.x > -1e29f;
        }
        if (!__syncthreads_or(step_active)) continue;              // no ray of the tile scatters at this sample (empty space)
        // ---- phase B: bounding box of the step's taps from the corner rays (+- a guard against rounding), clamped to the grid
        float lo[3] = {1e30f, 1e30f, 1e30f}, hi[3] = {-1e30f, -1e30f, -1e30f};
        {
            const float o3[3] = {ray.ox, ray.oy, ray.oz}, h3[3] = {hx, hy, hz}, sc3[3] = {scx, scy, scz};
#pragma unroll
            for (int q = 0; q < 4; ++q)
#pragma unroll
                for (int ax = 0; ax < 3; ++ax) {
                    const float pc = (((o3[ax] + cdir[q][ax] * z) / h3[ax] + 1.f) / 2.f) * sc3[ax];
                    lo[ax] = fminf(lo[ax], pc); hi[ax] = fmaxf(hi[ax], pc);
                }
        }
        const int bx0 = max((int)floorf(fmaxf(lo[0], -4.f) - 0.01f), 0), bx1 = min((int)floorf(fminf(hi[0], (float)W + 4.f) + 0.01f) + 1, W - 1);
        const int by0 = max((int)floorf(fmaxf(lo[1], -4.f) - 0.01f), 0), by1 = min((int)floorf(fminf(hi[1], (float)H + 4.f) + 0.01f) + 1, H - 1);
        const int bz0 = max((int)floorf(fmaxf(lo[2], -4.f) - 0.01f), 0), bz1 = min((int)floorf(fminf(hi[2], (float)D + 4.f) + 0.01f) + 1, D - 1);
        const int ex = bx1 - bx0 + 1, ey = by1 - by0 + 1, ez = bz1 - bz0 + 1;
        const int nbox = (ex > 0 && ey > 0 && ez > 0) ? ex * ey * ez : 0;
        // rays that can touch a voxel: positions are affine in the tile-local pixel index, p(lx, ly) = P00 + lx U + ly V, so
        // |p_a - q_a| < 1 on the two axes (a, b) with the best-conditioned 2x2 system confines (lx, ly) to a small rectangle
        // around M^-1 (q - P00)_ab with block-uniform half extents (~2 pixels at 0.55 voxel / pixel instead of the whole 8 x 8 tile)
        float P00[3], U[3], V[3];
        {
            const float o3[3] = {ray.ox, ray.oy, ray.oz}, h3[3] = {hx, hy, hz}, sc3[3] = {scx, scy, scz};
#pragma unroll
            for (int ax = 0; ax < 3; ++ax) {
                P00[ax] = (((o3[ax] + d00[ax] * z) / h3[ax] + 1.f) / 2.f) * sc3[ax];
                U[ax] = z * dU[ax] * (0.5f * sc3[ax] / h3[ax]);
                V[ax] = z * dV[ax] * (0.5f * sc3[ax] / h3[ax]);
            }
        }
        const float det01 = U[0] * V[1] - U[1] * V[0], det02 = U[0] * V[2] - U[2] * V[0], det12 = U[1] * V[2] - U[2] * V[1];
        int aa = 0, ab = 1;
        float det = det01;
        if (fabsf(det02) > fabsf(det)) { aa = 0; ab = 2; det = det02; }
        if (fabsf(det12) > fabsf(det)) { aa = 1; ab = 2; det = det12; }
        const bool solvable = fabsf(det) > 1e-12f;
        const float idet = solvable ? 1.f / det : 0.f;
        const float Ua = aa == 0 ? U[0] : U[1], Va = aa == 0 ? V[0] : V[1], Pa = aa == 0 ? P00[0] : P00[1];
        const float Ub = ab == 1 ? U[1] : U[2], Vb = ab == 1 ? V[1] : V[2], Pb = ab == 1 ? P00[1] : P00[2];
        const float ext_x = (fabsf(Vb) + fabsf(Va)) * fabsf(idet) + 0.05f, ext_y = (fabsf(Ub) + fabsf(Ua)) * fabsf(idet) + 0.05f;
        for (int idx = vslot; idx < nbox; idx += RPB) {
            const int qx = bx0 + idx % ex, qy = by0 + (idx / ex) % ey, qz = bz0 + idx / (ex * ey);
            const float fqx = (float)qx, fqy = (float)qy, fqz = (float)qz;
            int lx0 = 0, lx1 = 7, ly0 = 0, ly1 = TH - 1;
            if (solvable) {
                const float ra = (aa == 0 ? fqx : fqy) - Pa, rb = (ab == 1 ? fqy : fqz) - Pb;
                const float cxp = (Vb * ra - Va * rb) * idet, cyp = (Ua * rb - Ub * ra) * idet;
                lx0 = max((int)ceilf(cxp - ext_x), 0); lx1 = min((int)floorf(cxp + ext_x), 7);
                ly0 = max((int)ceilf(cyp - ext_y), 0); ly1 = min((int)floorf(cyp + ext_y), TH - 1);
            }
            float4 accf = make_float4(0.f, 0.f, 0.f, 0.f);
            float accd = 0.f;
            for (int yy = ly0; yy <= ly1; ++yy)
                for (int xx = lx0; xx <= lx1; ++xx) {
                    const int rr = (xx & 3) | ((yy & 3) << 2) | ((xx >> 2) << 4) | ((yy >> 2) << 5);   // inverse of tile_pixel
                    const float4 pp = st_p[rr];
                    const float ddx = pp.x - fqx, ddy = pp.y - fqy, ddz = pp.z - fqz;
                    if (fabsf(ddx) < 1.f && fabsf(ddy) < 1.f && fabsf(ddz) < 1.f) {
                        // the forward pass's weight expressions: lower tap (q = floor p) (q + 1) - p, upper tap (q = floor p + 1) p - (q - 1)
                        const float wx = ddx >= 0.f ? (fqx + 1.f) - pp.x : pp.x - (fqx - 1.f);
                        const float wy = ddy >= 0.f ? (fqy + 1.f) - pp.y : pp.y - (fqy - 1.f);
                        const float wz = ddz >= 0.f ? (fqz + 1.f) - pp.z : pp.z - (fqz - 1.f);
                        const float wq = wx * wy * wz;
                        accf = f4_fma(wq, st_g[rr * C4 + cg], accf);
                        accd = fmaf(wq, pp.w, accd);
                    }
                }
            const long long o = ((long long)qz * H + qy) * W + qx;
            float* df = dfeat + ((vbase + o) * C4 + cg) * 4;
            if (accf.x != 0.f) atomic_add_f32(df + 0, accf.x);
            if (accf.y != 0.f) atomic_add_f32(df + 1, accf.y);
            if (accf.z != 0.f) atomic_add_f32(df + 2, accf.z);
            if (accf.w != 0.f) atomic_add_f32(df + 3, accf.w);
            if (cg == 0 && accd != 0.f) atomic_add_f32(ddens + vbase + o, accd);
        }
    }
    if (CAM) {
        // chain to the packed camera (R[9], T[3], fx, fy, cx, cy): o = -R^T T, dir = R^T (dxc, dyc, 1),
        // dxc = (w + .5 - cx)/fx, dyc = (h + .5 - cy)/fy; then one block reduction and 16 atomics per workgroup.
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
        if (threadIdx.x < 16) atomic_add_f32(dcam + v * 16 + threadIdx.x, red[threadIdx.x][0] + red[threadIdx.x][1] + red[threadIdx.x][2] + red[threadIdx.x][3]);
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
// [ (i - 1) / scale - 1, (i + 1) / scale + 1 ]; each candidate's taps / weights are re-evaluated with the forward's own expressions.
__global__ __launch_bounds__(256) void resize_bilinear_bwd_kernel(const float* __restrict__ g, float* __restrict__ din, int P, int Hi, int Wi, int Ho, int Wo,
                                                                  float sh, float sw) {
    const long long idx = (long long)blockIdx.x * 256 + threadIdx.x;
    if (idx >= (long long)P * Hi * Wi) return;
    const int x = (int)(idx % Wi), y = (int)((idx / Wi) % Hi);
    const long long p = idx / ((long long)Wi * Hi);
    const int Y0 = max(0, (int)floorf((float)(y - 1) / sh) - 1), Y1 = min(Ho - 1, (int)ceilf((float)(y + 1) / sh) + 1);
    const int X0 = max(0, (int)floorf((float)(x - 1) / sw) - 1), X1 = min(Wo - 1, (int)ceilf((float)(x + 1) / sw) + 1);
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

extern "C" int forge_render_bwd(const float* feat, const float* dens, const float* cam, const int* view2vol,
                                const float* g_feat, const float* g_opac, const float* g_depth,
                                float* dfeat, float* ddens, float* dcam,
                                int V, int nvol, int C, int D, int H, int W, int Hr, int Wr, int S,
                                float zmin, float zmax, float hx, float hy, float hz, forge_stream_t stream) {
    if (int rc = check_render_args("forge_render_bwd", feat, dens, cam, view2vol, V, nvol, C, D, H, W, Hr, Wr, S, hx, hy, hz)) return rc;
    FORGE_REQUIRE(g_feat && g_opac && dfeat && ddens, FORGE_EINVAL, "forge_render_bwd: null gradient pointer");
    const size_t lds_bytes = ((size_t)2 * S * (256 / (C / 4)) + (size_t)2 * (256 / (C / 4)) * (4 + C)) * sizeof(float);
    FORGE_REQUIRE(lds_bytes <= 160 * 1024 - 2048, FORGE_ESHAPE, "forge_render_bwd: S=%d needs %zu B of LDS (> 160 KiB)", S, lds_bytes);
    FORGE_DISPATCH_C4(C, {
        constexpr int TH = (256 / C4) / 8;
        dim3 grid((Wr + 7) / 8, (Hr + TH - 1) / TH, V);
        if (dcam) {
            FORGE_SET_MAX_LDS_ONCE((render_bwd_kernel<C4, true>), 160 * 1024 - 2048);
            hipLaunchKernelGGL((render_bwd_kernel<C4, true>), grid, dim3(256), lds_bytes, (hipStream_t)stream, (const float4*)feat, dens, cam,
                               view2vol, g_feat, g_opac, g_depth, dfeat, ddens, dcam, D, H, W, Hr, Wr, S, zmin, zmax, hx, hy, hz);
        } else {
            FORGE_SET_MAX_LDS_ONCE((render_bwd_kernel<C4, false>), 160 * 1024 - 2048);
            hipLaunchKernelGGL((render_bwd_kernel<C4, false>), grid, dim3(256), lds_bytes, (hipStream_t)stream, (const float4*)feat, dens, cam,
                               view2vol, g_feat, g_opac, g_depth, dfeat, ddens, dcam, D, H, W, Hr, Wr, S, zmin, zmax, hx, hy, hz);
        }
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
