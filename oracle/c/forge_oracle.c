/* TEST INFRASTRUCTURE — plain-C restatement of the two gather kernels of the FORGE hot path.
 *
 * Independent of torch's F.grid_sample: scalar loops over NCDHW fp32 arrays, written from the
 * algorithm statements in SURVEY.md Appendix A (models/rotate.py:92-141; models/volume_render.py:40-88
 * + PyTorch3D 0.7.0 NDCGridRaysampler / VolumeSampler / EmissionAbsorptionRaymarcher).
 * Used only by tests/ (cross-check of oracle/forge_oracle.py against the golden vectors) — never by
 * forge_amd/. Build: `make -C oracle/c` -> oracle/_build/liboracle_c.so
 */
#include <math.h>
#include <stddef.h>

static float tap(const float* v, int D, int H, int W, int z, int y, int x) {
    if (z < 0 || z >= D || y < 0 || y >= H || x < 0 || x >= W) return 0.f;   /* zeros padding, per tap */
    return v[((size_t)z * H + y) * W + x];
}

static float trilinear(const float* v, int D, int H, int W, float px, float py, float pz) {
    if (!(px > -2.f && px < W + 1.f && py > -2.f && py < H + 1.f && pz > -2.f && pz < D + 1.f)) return 0.f;
    const float fx = floorf(px), fy = floorf(py), fz = floorf(pz);
    const int x0 = (int)fx, y0 = (int)fy, z0 = (int)fz;
    const float tx = px - fx, ty = py - fy, tz = pz - fz;
    float acc = 0.f;
    for (int k = 0; k < 8; ++k) {
        const int dx = k & 1, dy = (k >> 1) & 1, dz = k >> 2;
        const float w = (dx ? tx : 1.f - tx) * (dy ? ty : 1.f - ty) * (dz ? tz : 1.f - tz);
        acc += w * tap(v, D, H, W, z0 + dz, y0 + dy, x0 + dx);
    }
    return acc;
}

/* models/rotate.py:127-141. vox/out [n][C][D][H][W]; T [n][16] row-major 4x4 (= P_0 P_i^-1);
 * mode[n] 0 = copy. World voxel centre = g * e, g = linspace(-1,1,N); sample s = (R (g e) + t) / e;
 * align_corners=False: pix = ((s + 1) N - 1) / 2. */
void oracle_rotate(const float* vox, const float* T, const int* mode, float* out, int n, int C, int D, int H, int W, float e) {
    const size_t vol = (size_t)D * H * W;
    for (int i = 0; i < n; ++i) {
        const float* t = T + 16 * i;
        for (int c = 0; c < C; ++c) {
            const float* v = vox + ((size_t)i * C + c) * vol;
            float* o = out + ((size_t)i * C + c) * vol;
            for (int z = 0; z < D; ++z) for (int y = 0; y < H; ++y) for (int x = 0; x < W; ++x) {
                if (mode[i] == 0) { o[((size_t)z * H + y) * W + x] = v[((size_t)z * H + y) * W + x]; continue; }
                const float gx = (2.f * x / (W - 1) - 1.f) * e, gy = (2.f * y / (H - 1) - 1.f) * e, gz = (2.f * z / (D - 1) - 1.f) * e;
                const float sx = (t[0] * gx + t[1] * gy + t[2] * gz + t[3]) / e;
                const float sy = (t[4] * gx + t[5] * gy + t[6] * gz + t[7]) / e;
                const float sz = (t[8] * gx + t[9] * gy + t[10] * gz + t[11]) / e;
                o[((size_t)z * H + y) * W + x] = trilinear(v, D, H, W, ((sx + 1.f) * W - 1.f) * .5f, ((sy + 1.f) * H - 1.f) * .5f,
                                                             ((sz + 1.f) * D - 1.f) * .5f);
            }
        }
    }
}

/* models/volume_render.py:53-67. feat [V][C][D][H][W], dens [V][D][H][W] (one volume per view, as the reference),
 * R [V][9], T [V][3], Kh [V][4] = fx,fy,cx,cy at half resolution; out [V][Hr][Wr][C+2] = features, opacity, depth. */
void oracle_render(const float* feat, const float* dens, const float* R, const float* T, const float* Kh, float* out,
                   int V, int C, int D, int H, int W, int Hr, int Wr, int S, float zmin, float zmax, float hx, float hy, float hz) {
    const size_t vol = (size_t)D * H * W;
    for (int v = 0; v < V; ++v) {
        const float* r = R + 9 * v; const float* t = T + 3 * v; const float* k = Kh + 4 * v;
        const float ox = -(r[0] * t[0] + r[3] * t[1] + r[6] * t[2]);
        const float oy = -(r[1] * t[0] + r[4] * t[1] + r[7] * t[2]);
        const float oz = -(r[2] * t[0] + r[5] * t[1] + r[8] * t[2]);
        for (int h = 0; h < Hr; ++h) for (int w = 0; w < Wr; ++w) {
            const float dxc = (w + .5f - k[2]) / k[0], dyc = (h + .5f - k[3]) / k[1];
            const float dx = r[0] * dxc + r[3] * dyc + r[6], dy = r[1] * dxc + r[4] * dyc + r[7], dz = r[2] * dxc + r[5] * dyc + r[8];
            float* o = out + (((size_t)v * Hr + h) * Wr + w) * (C + 2);
            for (int c = 0; c < C + 2; ++c) o[c] = 0.f;
            float Tr = 1.f;
            for (int s = 0; s < S; ++s) {
                const float z = zmin + (zmax - zmin) * (float)s / (float)(S - 1);
                const float px = ((ox + dx * z) / hx + 1.f) * .5f * (W - 1);       /* align_corners=True */
                const float py = ((oy + dy * z) / hy + 1.f) * .5f * (H - 1);
                const float pz = ((oz + dz * z) / hz + 1.f) * .5f * (D - 1);
                const float d = trilinear(dens + (size_t)v * vol, D, H, W, px, py, pz);
                const float wgt = d * Tr;
                for (int c = 0; c < C; ++c) o[c] += wgt * trilinear(feat + ((size_t)v * C + c) * vol, D, H, W, px, py, pz);
                o[C + 1] += wgt * z;
                Tr *= (1.f - d);
            }
            o[C] = 1.f - Tr;
        }
    }
}
