/* A host with no Python and no torch: plain C against include/forge_hip.h + the HIP runtime (tests/test_gpu_c_host.py builds
 * and runs it). Exercises the C-ABI the way a foreign-language binding would: device buffers from hipMalloc, an explicit
 * stream, integer return codes, forge_last_error.
 *   1. forge_rotate_fwd: a mode-0 volume comes back bit-identical; an identity affine reproduces the reference's
 *      align_corners mismatch (interior values shrink towards the centre) - checked against a scalar
 *      re-computation of the trilinear sample for one voxel;
 *   2. forge_render_fwd: an empty density volume renders exact zeros; a uniform density d renders opacity 1 - (1 - d)^k;
 *   3. forge_conv_igemm (identity 1x1 convolution on the matrix cores, planned inside and with the caller's explicit plan), forge_conv_igemm_plan,
 *      forge_resize_bilinear_fwd;
 *   4. error path: a NULL pointer / an unknown tile returns FORGE_EINVAL and sets forge_last_error.
 */
#include <hip/hip_runtime_api.h>
#include <math.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>

#include "forge_hip.h"

#define CHECK_HIP(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { fprintf(stderr, "HIP error %s at %s:%d\n", hipGetErrorString(e_), __FILE__, __LINE__); return 2; } } while (0)
#define CHECK_FORGE(x) do { int rc_ = (x); if (rc_ != 0) { fprintf(stderr, "forge error %d: %s (%s:%d)\n", rc_, forge_last_error(), __FILE__, __LINE__); return 3; } } while (0)
#define EXPECT(cond, msg) do { if (!(cond)) { fprintf(stderr, "FAILED: %s (%s:%d)\n", msg, __FILE__, __LINE__); return 1; } } while (0)

int main(void) {
    printf("forge_version = %d\n", forge_version());
    hipStream_t st;
    CHECK_HIP(hipStreamCreate(&st));

    /* ---------------- rotate: 2 volumes of 16^3 x 8 channels, volume 0 = pass-through, volume 1 = identity affine */
    const int n = 2, C = 8, D = 16;
    const size_t vol = (size_t)D * D * D * C, bytes = n * vol * sizeof(float);
    float* h_in = (float*)malloc(bytes);
    float* h_out = (float*)malloc(bytes);
    for (size_t i = 0; i < n * vol; ++i) h_in[i] = (float)((i * 2654435761u) % 1000u) / 1000.0f - 0.5f;
    float h_xf[24] = {1, 0, 0, 0, 0, 1, 0, 0, 0, 0, 1, 0, 1, 0, 0, 0, 0, 1, 0, 0, 0, 0, 1, 0};
    int h_mode[2] = {0, 1};
    float *d_in, *d_out, *d_xf;
    int* d_mode;
    CHECK_HIP(hipMalloc((void**)&d_in, bytes));
    CHECK_HIP(hipMalloc((void**)&d_out, bytes));
    CHECK_HIP(hipMalloc((void**)&d_xf, sizeof(h_xf)));
    CHECK_HIP(hipMalloc((void**)&d_mode, sizeof(h_mode)));
    CHECK_HIP(hipMemcpy(d_in, h_in, bytes, hipMemcpyHostToDevice));
    CHECK_HIP(hipMemcpy(d_xf, h_xf, sizeof(h_xf), hipMemcpyHostToDevice));
    CHECK_HIP(hipMemcpy(d_mode, h_mode, sizeof(h_mode), hipMemcpyHostToDevice));
    CHECK_FORGE(forge_rotate_fwd(d_in, d_xf, d_mode, d_out, n, C, D, D, D, (forge_stream_t)st));
    CHECK_HIP(hipStreamSynchronize(st));
    CHECK_HIP(hipMemcpy(h_out, d_out, bytes, hipMemcpyDeviceToHost));
    EXPECT(memcmp(h_in, h_out, vol * sizeof(float)) == 0, "mode-0 volume must be copied bit-exactly");
    {   /* voxel (z,y,x) = (5,9,3) of volume 1 under the identity affine: s = 2 i / (D-1) - 1, p = ((s + 1) D - 1) / 2 */
        const int q[3] = {3, 9, 5};                            /* x, y, z */
        float p[3]; int i0[3]; float w1[3];
        for (int a = 0; a < 3; ++a) {
            const float s = 2.f * (float)q[a] / (float)(D - 1) - 1.f;
            p[a] = ((s + 1.f) * (float)D - 1.f) * 0.5f;
            i0[a] = (int)floorf(p[a]);
            w1[a] = p[a] - (float)i0[a];
        }
        for (int c = 0; c < C; ++c) {
            float acc = 0.f;
            for (int k = 0; k < 8; ++k) {
                const int dx = k & 1, dy = (k >> 1) & 1, dz = k >> 2;
                const int x = i0[0] + dx, y = i0[1] + dy, z = i0[2] + dz;
                if (x < 0 || x >= D || y < 0 || y >= D || z < 0 || z >= D) continue;
                const float w = (dx ? w1[0] : 1.f - w1[0]) * (dy ? w1[1] : 1.f - w1[1]) * (dz ? w1[2] : 1.f - w1[2]);
                acc += w * h_in[vol + (((size_t)z * D + y) * D + x) * C + c];
            }
            const float got = h_out[vol + (((size_t)q[2] * D + q[1]) * D + q[0]) * C + c];
            EXPECT(fabsf(got - acc) < 1e-5f, "identity-affine warp does not match the scalar trilinear re-computation");
        }
        EXPECT(fabsf(p[0] - (float)q[0]) > 1e-3f, "align_corners mismatch: the identity affine must NOT be an identity resample");
    }

    /* ---------------- render: 1 view of 16 x 16 rays x 24 samples through a 16^3 x 4-channel volume */
    const int V = 1, Cr = 4, Hr = 16, Wr = 16, S = 24;
    const size_t nv = (size_t)D * D * D;
    float* h_feat = (float*)malloc(nv * Cr * sizeof(float));
    float* h_dens = (float*)calloc(nv, sizeof(float));
    for (size_t i = 0; i < nv * Cr; ++i) h_feat[i] = 1.0f;
    /* camera on the +z axis looking at the origin: R = I, T = (0,0,1.5); fx = fy = 16, cx = cy = 8 */
    float h_cam[16] = {1, 0, 0, 0, 1, 0, 0, 0, 1, 0, 0, 1.5f, 16, 16, 8, 8};
    int h_v2v[1] = {0};
    float *d_feat, *d_dens, *d_cam, *d_of, *d_oo;
    int* d_v2v;
    CHECK_HIP(hipMalloc((void**)&d_feat, nv * Cr * sizeof(float)));
    CHECK_HIP(hipMalloc((void**)&d_dens, nv * sizeof(float)));
    CHECK_HIP(hipMalloc((void**)&d_cam, sizeof(h_cam)));
    CHECK_HIP(hipMalloc((void**)&d_v2v, sizeof(h_v2v)));
    CHECK_HIP(hipMalloc((void**)&d_of, (size_t)V * Hr * Wr * Cr * sizeof(float)));
    CHECK_HIP(hipMalloc((void**)&d_oo, (size_t)V * Hr * Wr * sizeof(float)));
    CHECK_HIP(hipMemcpy(d_feat, h_feat, nv * Cr * sizeof(float), hipMemcpyHostToDevice));
    CHECK_HIP(hipMemcpy(d_dens, h_dens, nv * sizeof(float), hipMemcpyHostToDevice));
    CHECK_HIP(hipMemcpy(d_cam, h_cam, sizeof(h_cam), hipMemcpyHostToDevice));
    CHECK_HIP(hipMemcpy(d_v2v, h_v2v, sizeof(h_v2v), hipMemcpyHostToDevice));
    const float half = 0.5f * (float)(D - 1) / (float)D;
    float* h_of = (float*)malloc((size_t)Hr * Wr * Cr * sizeof(float));
    float* h_oo = (float*)malloc((size_t)Hr * Wr * sizeof(float));
    CHECK_FORGE(forge_render_fwd(d_feat, d_dens, d_cam, d_v2v, d_of, d_oo, NULL, V, 1, Cr, D, D, D, Hr, Wr, S, 0.5f, 2.5f, half, half, half, (forge_stream_t)st));
    CHECK_HIP(hipStreamSynchronize(st));
    CHECK_HIP(hipMemcpy(h_of, d_of, (size_t)Hr * Wr * Cr * sizeof(float), hipMemcpyDeviceToHost));
    CHECK_HIP(hipMemcpy(h_oo, d_oo, (size_t)Hr * Wr * sizeof(float), hipMemcpyDeviceToHost));
    for (int i = 0; i < Hr * Wr; ++i) EXPECT(h_oo[i] == 0.0f && h_of[i * Cr] == 0.0f, "empty volume must render exact zeros");
    /* uniform density 0.1, unit features: every sample inside the cube has d = 0.1, f = 1, so feature == opacity == 1 - 0.9^k */
    for (size_t i = 0; i < nv; ++i) h_dens[i] = 0.1f;
    CHECK_HIP(hipMemcpy(d_dens, h_dens, nv * sizeof(float), hipMemcpyHostToDevice));
    CHECK_FORGE(forge_render_fwd(d_feat, d_dens, d_cam, d_v2v, d_of, d_oo, NULL, V, 1, Cr, D, D, D, Hr, Wr, S, 0.5f, 2.5f, half, half, half, (forge_stream_t)st));
    CHECK_HIP(hipStreamSynchronize(st));
    CHECK_HIP(hipMemcpy(h_of, d_of, (size_t)Hr * Wr * Cr * sizeof(float), hipMemcpyDeviceToHost));
    CHECK_HIP(hipMemcpy(h_oo, d_oo, (size_t)Hr * Wr * sizeof(float), hipMemcpyDeviceToHost));
    {
        const int centre = (Hr / 2) * Wr + Wr / 2;
        const float op = h_oo[centre];
        EXPECT(op > 0.3f && op < 1.0f, "central ray through a uniform fog must be partially opaque");
        /* samples strictly inside the cube all have d = 0.1; the two boundary samples may be partially interpolated against the zero
         * padding, so 1 - 0.9^k holds for some integer k within one sample */
        const float k = logf(1.f - op) / logf(0.9f);
        EXPECT(k > 5.f && k < (float)S, "opacity is not of the form 1 - (1 - d)^k");
        /* unit features composite to the opacity, except that the boundary samples interpolate features AND density against the zero
         * padding (f_s < 1 there): feature <= opacity, within the weight of one boundary sample */
        for (int c = 0; c < Cr; ++c)
            EXPECT(h_of[centre * Cr + c] <= op + 1e-6f && h_of[centre * Cr + c] > op - 0.1f && h_of[centre * Cr + c] == h_of[centre * Cr],
                   "unit features must composite to (just below) the opacity, identically in every channel");
    }

    /* ---------------- matrix-core convolution through the ABI: 1x1 conv, 64 -> 64 channels, identity weights, bias 0.25 on a 1 x 4 x 8 x 8 grid,
     * once with the library's own launch plan (tile = 0) and once with the caller's plan ('D', no split-K); then the plan query */
    {
        const int Cc = 64, Mr = 4 * 8 * 8;
        float* h_x = (float*)malloc((size_t)Mr * Cc * sizeof(float));
        float* h_w = (float*)calloc((size_t)Cc * Cc, sizeof(float));
        float* h_b = (float*)malloc(Cc * sizeof(float));
        float* h_y = (float*)malloc((size_t)Mr * Cc * sizeof(float));
        for (int i = 0; i < Mr * Cc; ++i) h_x[i] = (float)((i * 40503u) % 997u) / 997.0f - 0.5f;
        for (int c = 0; c < Cc; ++c) { h_w[c * Cc + c] = 1.0f; h_b[c] = 0.25f; }
        float *d_x, *d_w, *d_b, *d_y;
        CHECK_HIP(hipMalloc((void**)&d_x, (size_t)Mr * Cc * sizeof(float)));
        CHECK_HIP(hipMalloc((void**)&d_w, (size_t)Cc * Cc * sizeof(float)));
        CHECK_HIP(hipMalloc((void**)&d_b, Cc * sizeof(float)));
        CHECK_HIP(hipMalloc((void**)&d_y, (size_t)Mr * Cc * sizeof(float)));
        CHECK_HIP(hipMemcpy(d_x, h_x, (size_t)Mr * Cc * sizeof(float), hipMemcpyHostToDevice));
        CHECK_HIP(hipMemcpy(d_w, h_w, (size_t)Cc * Cc * sizeof(float), hipMemcpyHostToDevice));
        CHECK_HIP(hipMemcpy(d_b, h_b, Cc * sizeof(float), hipMemcpyHostToDevice));
        const int taps[3] = {0, 0, 0};
        for (int pass = 0; pass < 2; ++pass) {
            CHECK_HIP(hipMemset(d_y, 0xff, (size_t)Mr * Cc * sizeof(float)));
            CHECK_FORGE(forge_conv_igemm(d_x, Cc, Cc, 0, NULL, 0, 0, 0, d_w, d_b, NULL, NULL, 1.0f, NULL, NULL, NULL, d_y, NULL, NULL,
                                         1, 4, 8, 8, 1, 4, 8, 8, Cc, Cc, taps, 1, 1, 0, 0, 0, 4, 8, 8, /*epilogue*/ 0, /*lift*/ 0,
                                         pass ? 'D' : 0, 1, NULL, 0, /*stats*/ NULL, (forge_stream_t)st));
            CHECK_HIP(hipStreamSynchronize(st));
            CHECK_HIP(hipMemcpy(h_y, d_y, (size_t)Mr * Cc * sizeof(float), hipMemcpyDeviceToHost));
            for (int i = 0; i < Mr * Cc; ++i) EXPECT(h_y[i] == h_x[i] + 0.25f, "identity 1x1 convolution + bias must reproduce x + 0.25 exactly");
        }
        int tile = 0, ks = 0;
        CHECK_FORGE(forge_conv_igemm_plan(32768, 256, 256, 27, 1, 2, 128, 0, &tile, &ks));
        EXPECT(tile >= 'A' && tile <= 'E' && ks == 1, "the plan query must name a tile A..E and no split-K without a workspace");
        EXPECT(forge_conv_igemm(d_x, Cc, Cc, 0, NULL, 0, 0, 0, d_w, d_b, NULL, NULL, 1.0f, NULL, NULL, NULL, d_y, NULL, NULL, 1, 4, 8, 8, 1, 4, 8, 8, Cc, Cc,
                                taps, 1, 1, 0, 0, 0, 4, 8, 8, 0, 0, 'Z', 1, NULL, 0, NULL, (forge_stream_t)st) == FORGE_EINVAL, "an unknown tile letter must be refused");
        /* output statistics need the caller's explicit plan: tile = 0 with stats != NULL is refused before any launch */
        EXPECT(forge_conv_igemm(d_x, Cc, Cc, 0, NULL, 0, 0, 0, d_w, d_b, NULL, NULL, 1.0f, NULL, NULL, NULL, d_y, NULL, NULL, 1, 4, 8, 8, 1, 4, 8, 8, Cc, Cc,
                                taps, 1, 1, 0, 0, 0, 4, 8, 8, 0, 0, 0, 1, NULL, 0, (double*)d_y, (forge_stream_t)st) == FORGE_EINVAL,
               "stats without an explicit tile must be refused");
    }
    /* ---------------- bilinear x2 of a constant plane and of a column ramp (align_corners = False: interior values are the 0.25 / 0.75 blends) */
    {
        const int Hi = 4, Wi = 4, Ho = 8, Wo = 8;
        float h_p[16], h_q[64];
        for (int i = 0; i < 16; ++i) h_p[i] = (float)(i % Wi);
        float *d_p, *d_q;
        CHECK_HIP(hipMalloc((void**)&d_p, sizeof(h_p)));
        CHECK_HIP(hipMalloc((void**)&d_q, sizeof(h_q)));
        CHECK_HIP(hipMemcpy(d_p, h_p, sizeof(h_p), hipMemcpyHostToDevice));
        CHECK_FORGE(forge_resize_bilinear_fwd(d_p, d_q, 1, Hi, Wi, Ho, Wo, (forge_stream_t)st));
        CHECK_HIP(hipStreamSynchronize(st));
        CHECK_HIP(hipMemcpy(h_q, d_q, sizeof(h_q), hipMemcpyDeviceToHost));
        const float want[8] = {0.f, 0.25f, 0.75f, 1.25f, 1.75f, 2.25f, 2.75f, 3.f};
        for (int y = 0; y < Ho; ++y)
            for (int x = 0; x < Wo; ++x) EXPECT(fabsf(h_q[y * Wo + x] - want[x]) < 1e-6f, "bilinear x2 of a column ramp");
    }

    /* ---------------- attention: softmax(q k^T) v, 64 queries x 128 keys x one 64-channel head, against a double-precision host evaluation;
     * identical keys make the softmax uniform (out = mean of v) - a closed form next to the general case */
    {
        enum { NQ = 64, NK = 128, DH = 64 };
        static float h_q[NQ * DH], h_k[NK * DH], h_v[NK * DH], h_o[NQ * DH];
        unsigned rng = 12345u;
        for (int i = 0; i < NQ * DH; ++i) { rng = rng * 1664525u + 1013904223u; h_q[i] = ((rng >> 8) & 0xffff) / 65536.f - 0.5f; }
        for (int i = 0; i < NK * DH; ++i) { rng = rng * 1664525u + 1013904223u; h_k[i] = ((rng >> 8) & 0xffff) / 65536.f - 0.5f; }
        for (int i = 0; i < NK * DH; ++i) { rng = rng * 1664525u + 1013904223u; h_v[i] = ((rng >> 8) & 0xffff) / 32768.f - 1.f; }
        float *d_q, *d_k, *d_v, *d_o;
        CHECK_HIP(hipMalloc((void**)&d_q, sizeof(h_q)));
        CHECK_HIP(hipMalloc((void**)&d_k, sizeof(h_k)));
        CHECK_HIP(hipMalloc((void**)&d_v, sizeof(h_v)));
        CHECK_HIP(hipMalloc((void**)&d_o, sizeof(h_o)));
        CHECK_HIP(hipMemcpy(d_q, h_q, sizeof(h_q), hipMemcpyHostToDevice));
        CHECK_HIP(hipMemcpy(d_v, h_v, sizeof(h_v), hipMemcpyHostToDevice));
        for (int pass = 0; pass < 2; ++pass) {
            if (pass == 1)
                for (int j = 1; j < NK; ++j) memcpy(h_k + j * DH, h_k, DH * sizeof(float));       /* all keys equal: uniform weights */
            CHECK_HIP(hipMemcpy(d_k, h_k, sizeof(h_k), hipMemcpyHostToDevice));
            CHECK_FORGE(forge_attention_fwd(d_q, d_k, d_v, NK, d_o, 1, NQ, NK, DH, (forge_stream_t)st));
            CHECK_HIP(hipStreamSynchronize(st));
            CHECK_HIP(hipMemcpy(h_o, d_o, sizeof(h_o), hipMemcpyDeviceToHost));
            double worst = 0.0;
            for (int i = 0; i < NQ; ++i) {
                double lg[NK], mx = -1e300, den = 0.0;
                for (int j = 0; j < NK; ++j) {
                    double a = 0.0;
                    for (int c = 0; c < DH; ++c) a += (double)h_q[i * DH + c] * h_k[j * DH + c];
                    lg[j] = a;
                    if (a > mx) mx = a;
                }
                for (int j = 0; j < NK; ++j) { lg[j] = exp(lg[j] - mx); den += lg[j]; }
                for (int c = 0; c < DH; ++c) {
                    double o = 0.0;
                    for (int j = 0; j < NK; ++j) o += lg[j] * h_v[j * DH + c];
                    const double e = fabs(o / den - h_o[i * DH + c]);
                    if (e > worst) worst = e;
                }
            }
            EXPECT(worst < 2e-6, pass ? "attention with identical keys = mean of v" : "attention vs the double-precision host evaluation");
        }
        EXPECT(forge_attention_fwd(d_q, d_k, d_v, NK, d_o, 1, NQ - 1, NK, DH, (forge_stream_t)st) == FORGE_ESHAPE, "63 queries must be refused");
    }

    /* ---------------- error path */
    EXPECT(forge_rotate_fwd(NULL, d_xf, d_mode, d_out, n, C, D, D, D, (forge_stream_t)st) == FORGE_EINVAL, "NULL input must return FORGE_EINVAL");
    EXPECT(strlen(forge_last_error()) > 0, "forge_last_error must describe the failure");
    printf("C host: rotate + render + conv_igemm (explicit plan) + resize + attention + error path OK (central opacity %.6f)\n", h_oo[(Hr / 2) * Wr + Wr / 2]);
    return 0;
}
