// loss.hip — f1: the squared-error sums of the reconstruction losses (scripts/kubric_compute_loss.py:26-29, 136-139) in ONE pass.
//
// The reference evaluates four F.mse_loss terms per iteration (rendered rgb / mask of the first and of the second half of the 2t
// views of a scene, each against the SAME t target views) on reshaped / repeated copies: ~30 small element-wise and reduction
// launches forward + backward. Here one kernel reads every rendered pixel once, looks its target up through the view-group mapping
// (rendered view v of scene b <-> target view v mod Vt, group v / gsize) and accumulates one sum of squared errors per group;
// the backward is one kernel writing d pred = coef[group] * (pred - target) in the rendered tensor's own memory layout.
// Deterministic: per-workgroup partial sums (fixed reduction tree), summed by the caller in a fixed order - no atomics.
//
//   pred    [B][Vp][C][H][W] with arbitrary element strides (sn, sc, sh, sw) per view: the rgb maps are channels-last memory
//           behind an NCHW view (conv_rgb's GEMM output), masks are plain NCHW
//   target  [B][Vt][C][H][W] contiguous;   groups G = Vp / gsize  (GT-pose model: Vp = 2t, Vt = gsize = t; joint model: Vp = Vt = 2t, gsize = t)
#include "common.h"

namespace forge {

struct SseArgs {
    const float* pred; const float* target;
    long long sn, sc, sh, sw;            // element strides of pred per (view, channel, row, column)
    int B, Vp, Vt, gsize, C, H, W;       // target view = v % Vt, group = v / gsize
};

constexpr int SSE_MAX_GROUPS = 4;

__global__ __launch_bounds__(256) void sse_groups_fwd_kernel(const SseArgs a, float* __restrict__ partial /* [gridDim.x][G] */, int G) {
    const long long HW = (long long)a.H * a.W, per_view = (long long)a.C * HW, total = (long long)a.B * a.Vp * per_view;
    float acc[SSE_MAX_GROUPS] = {0.f, 0.f, 0.f, 0.f};
    for (long long i = (long long)blockIdx.x * 256 + threadIdx.x; i < total; i += (long long)gridDim.x * 256) {
        // i enumerates the TARGET-ordered element (view, c, h, w) so that target reads are coalesced
        const long long n = i / per_view, e = i - n * per_view;
        const int c = (int)(e / HW);
        const long long hw = e - c * HW;
        const int h = (int)(hw / a.W), w = (int)(hw - (long long)h * a.W);
        const int b = (int)(n / a.Vp), v = (int)(n - (long long)b * a.Vp), g = v / a.gsize, vt = v % a.Vt;
        const float p = a.pred[n * a.sn + c * a.sc + h * a.sh + w * a.sw];
        const float q = a.target[((long long)b * a.Vt + vt) * per_view + e];
        const float d = p - q;
#pragma unroll
        for (int k = 0; k < SSE_MAX_GROUPS; ++k) acc[k] += (k == g) ? d * d : 0.f;
    }
    __shared__ float red[SSE_MAX_GROUPS][256];
#pragma unroll
    for (int k = 0; k < SSE_MAX_GROUPS; ++k) red[k][threadIdx.x] = acc[k];
    __syncthreads();
    for (int s = 128; s > 0; s >>= 1) {
        if ((int)threadIdx.x < s) {
#pragma unroll
            for (int k = 0; k < SSE_MAX_GROUPS; ++k) red[k][threadIdx.x] += red[k][threadIdx.x + s];
        }
        __syncthreads();
    }
    if ((int)threadIdx.x < G) partial[(long long)blockIdx.x * G + threadIdx.x] = red[threadIdx.x][0];
}

// dpred[...] = coef[g] * (pred - target), written with pred's strides (dpred has the same layout as pred)
__global__ __launch_bounds__(256) void sse_groups_bwd_kernel(const SseArgs a, const float* __restrict__ coef /* [G] */, float* __restrict__ dpred) {
    const long long HW = (long long)a.H * a.W, per_view = (long long)a.C * HW, total = (long long)a.B * a.Vp * per_view;
    for (long long i = (long long)blockIdx.x * 256 + threadIdx.x; i < total; i += (long long)gridDim.x * 256) {
        const long long n = i / per_view, e = i - n * per_view;
        const int c = (int)(e / HW);
        const long long hw = e - c * HW;
        const int h = (int)(hw / a.W), w = (int)(hw - (long long)h * a.W);
        const int b = (int)(n / a.Vp), v = (int)(n - (long long)b * a.Vp), g = v / a.gsize, vt = v % a.Vt;
        const long long po = n * a.sn + c * a.sc + h * a.sh + w * a.sw;
        dpred[po] = coef[g] * (a.pred[po] - a.target[((long long)b * a.Vt + vt) * per_view + e]);
    }
}

static int fill_args(SseArgs& a, const char* fn, const float* pred, const float* target, long long sn, long long sc, long long sh, long long sw,
                     int B, int Vp, int Vt, int gsize, int C, int H, int W) {
    FORGE_REQUIRE(pred && target, FORGE_EINVAL, "%s: null pointer argument", fn);
    FORGE_REQUIRE(B > 0 && Vp > 0 && Vt > 0 && gsize > 0 && C > 0 && H > 0 && W > 0 && Vp % Vt == 0 && Vp % gsize == 0 && Vp / gsize <= SSE_MAX_GROUPS,
                  FORGE_ESHAPE, "%s: B=%d Vp=%d Vt=%d gsize=%d C=%d H=%d W=%d (Vp must be a multiple of Vt and of gsize, at most %d groups)", fn, B, Vp,
                  Vt, gsize, C, H, W, SSE_MAX_GROUPS);
    a.pred = pred; a.target = target; a.sn = sn; a.sc = sc; a.sh = sh; a.sw = sw; a.B = B; a.Vp = Vp; a.Vt = Vt; a.gsize = gsize; a.C = C; a.H = H; a.W = W;
    return 0;
}

// torch.optim.Adam (no weight decay, no amsgrad) on ONE small tensor: p -= lr / (1 - b1^k) * m / (sqrt(v) / sqrt(1 - b2^k) + eps) after
// m = lerp(m, g, 1 - b1), v = b2 v + (1 - b2) g^2, with the step count k kept on the device (the captured refinement iteration advances
// it itself). One workgroup: the refinement loop optimises b (t-1) x 7 numbers, for which the capturable torch optimiser issues ~30 launches.
__global__ __launch_bounds__(256) void adam_small_kernel(float* __restrict__ p, const float* __restrict__ g, float* __restrict__ m, float* __restrict__ v,
                                                         float* __restrict__ step, int n, float lr, float b1, float b2, float eps) {
    const float k = step[0] + 1.f;
    __syncthreads();
    if (threadIdx.x == 0) step[0] = k;
    const float bc1 = 1.f - powf(b1, k), bc2 = 1.f - powf(b2, k);
    const float step_size = lr / bc1, bc2_sqrt = sqrtf(bc2);
    for (int i = threadIdx.x; i < n; i += 256) {
        const float gi = g[i];
        const float mi = m[i] + (gi - m[i]) * (1.f - b1);          // Tensor.lerp_(grad, 1 - beta1)
        const float vi = v[i] * b2 + (1.f - b2) * gi * gi;          // mul_(beta2).addcmul_(grad, grad, value = 1 - beta2)
        m[i] = mi; v[i] = vi;
        p[i] -= step_size * (mi / (sqrtf(vi) / bc2_sqrt + eps));
    }
}

}  // namespace forge

using namespace forge;

extern "C" int forge_adam_small(float* param, const float* grad, float* exp_avg, float* exp_avg_sq, float* step, int n, float lr, float beta1, float beta2,
                                float eps, forge_stream_t stream) {
    FORGE_REQUIRE(param && grad && exp_avg && exp_avg_sq && step && n > 0, FORGE_EINVAL, "forge_adam_small: null pointer argument or n <= 0");
    FORGE_REQUIRE(n <= (1 << 20), FORGE_ESHAPE, "forge_adam_small: n=%d - a one-workgroup kernel for small parameter sets", n);
    hipLaunchKernelGGL(adam_small_kernel, dim3(1), dim3(256), 0, (hipStream_t)stream, param, grad, exp_avg, exp_avg_sq, step, n, lr, beta1, beta2, eps);
    FORGE_LAUNCH_CHECK("forge_adam_small");
    return 0;
}

extern "C" int forge_sse_groups_blocks(void) { return 1024; }

extern "C" int forge_sse_groups_fwd(const float* pred, long long sn, long long sc, long long sh, long long sw, const float* target,
                                    float* partial, int B, int Vp, int Vt, int gsize, int C, int H, int W, forge_stream_t stream) {
    SseArgs a;
    if (int rc = fill_args(a, "forge_sse_groups_fwd", pred, target, sn, sc, sh, sw, B, Vp, Vt, gsize, C, H, W)) return rc;
    FORGE_REQUIRE(partial, FORGE_EINVAL, "forge_sse_groups_fwd: null partial buffer");
    hipLaunchKernelGGL(sse_groups_fwd_kernel, dim3(forge_sse_groups_blocks()), dim3(256), 0, (hipStream_t)stream, a, partial, Vp / gsize);
    FORGE_LAUNCH_CHECK("forge_sse_groups_fwd");
    return 0;
}

extern "C" int forge_sse_groups_bwd(const float* pred, long long sn, long long sc, long long sh, long long sw, const float* target,
                                    const float* coef, float* dpred, int B, int Vp, int Vt, int gsize, int C, int H, int W, forge_stream_t stream) {
    SseArgs a;
    if (int rc = fill_args(a, "forge_sse_groups_bwd", pred, target, sn, sc, sh, sw, B, Vp, Vt, gsize, C, H, W)) return rc;
    FORGE_REQUIRE(coef && dpred, FORGE_EINVAL, "forge_sse_groups_bwd: null pointer argument");
    hipLaunchKernelGGL(sse_groups_bwd_kernel, dim3(forge_sse_groups_blocks()), dim3(256), 0, (hipStream_t)stream, a, coef, dpred);
    FORGE_LAUNCH_CHECK("forge_sse_groups_bwd");
    return 0;
}
