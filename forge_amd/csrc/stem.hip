// stem.hip — the two non-GEMM pieces of the ResNet-50 stem (models/encoder.py:71-73 -> torchvision conv1/bn1/relu/maxpool):
//   * patch gather for the 7x7 stride-2 conv on the 3-channel image: rows [n*Ho*Wo][Kpad], k = (ky*kw + kx)*C + c,
//     zero outside the image and for k >= kh*kw*C. The conv itself then runs on the MFMA GEMM kernel (taps = 1, K = Kpad)
//     with BN + ReLU folded into its epilogue. (The 3-channel input cannot feed the 32-wide K-step tap-by-tap.)
//   * 3x3 stride-2 max-pool on channels-last activations (-inf padding semantics, as nn.MaxPool2d).
// Both are HBM-bound streaming kernels: 0.8 MB image -> 52 MB patch matrix per 5 views; 21 MB -> 5 MB pool.
#include "common.h"

namespace forge {

__global__ __launch_bounds__(256) void im2col_nchw_kernel(const float* __restrict__ img, float* __restrict__ out, int N, int C, int H, int W,
                                                          int kh, int kw, int stride, int pad, int Ho, int Wo, int Kpad) {
    const long long total = (long long)N * Ho * Wo * Kpad;
    const int K = kh * kw * C;
    for (long long i = (long long)blockIdx.x * 256 + threadIdx.x; i < total; i += (long long)gridDim.x * 256) {
        const int k = (int)(i % Kpad);
        long long r = i / Kpad;
        const int ox = (int)(r % Wo); r /= Wo;
        const int oy = (int)(r % Ho);
        const int n = (int)(r / Ho);
        float v = 0.f;
        if (k < K) {
            const int c = k % C, t = k / C, kx = t % kw, ky = t / kw;
            const int y = oy * stride - pad + ky, x = ox * stride - pad + kx;
            if ((unsigned)y < (unsigned)H && (unsigned)x < (unsigned)W) v = img[(((long long)n * C + c) * H + y) * W + x];
        }
        out[i] = v;
    }
}

__global__ __launch_bounds__(256) void maxpool2d_nhwc_kernel(const float4* __restrict__ in, float4* __restrict__ out, int N, int H, int W, int C4,
                                                             int k, int stride, int pad, int Ho, int Wo) {
    const long long total = (long long)N * Ho * Wo * C4;
    for (long long i = (long long)blockIdx.x * 256 + threadIdx.x; i < total; i += (long long)gridDim.x * 256) {
        const int c = (int)(i % C4);
        long long r = i / C4;
        const int ox = (int)(r % Wo); r /= Wo;
        const int oy = (int)(r % Ho);
        const int n = (int)(r / Ho);
        float4 m = make_float4(-INFINITY, -INFINITY, -INFINITY, -INFINITY);
        for (int ky = 0; ky < k; ++ky) {
            const int y = oy * stride - pad + ky;
            if ((unsigned)y >= (unsigned)H) continue;
            for (int kx = 0; kx < k; ++kx) {
                const int x = ox * stride - pad + kx;
                if ((unsigned)x >= (unsigned)W) continue;
                const float4 v = in[(((long long)n * H + y) * W + x) * C4 + c];
                m.x = fmaxf(m.x, v.x); m.y = fmaxf(m.y, v.y); m.z = fmaxf(m.z, v.z); m.w = fmaxf(m.w, v.w);
            }
        }
        out[i] = m;
    }
}

}  // namespace forge

using namespace forge;

extern "C" int forge_im2col_nchw(const float* img, float* out, int N, int C, int H, int W, int kh, int kw, int stride, int pad,
                                 int Kpad, forge_stream_t stream) {
    FORGE_REQUIRE(img && out, FORGE_EINVAL, "forge_im2col_nchw: null pointer argument");
    FORGE_REQUIRE(N > 0 && C > 0 && H > 0 && W > 0 && kh > 0 && kw > 0 && stride > 0 && pad >= 0 && Kpad >= kh * kw * C, FORGE_EINVAL,
                  "forge_im2col_nchw: bad dims (Kpad=%d must be >= kh*kw*C=%d)", Kpad, kh * kw * C);
    const int Ho = (H + 2 * pad - kh) / stride + 1, Wo = (W + 2 * pad - kw) / stride + 1;
    const long long total = (long long)N * Ho * Wo * Kpad;
    const unsigned grid = (unsigned)((total + 255) / 256 < 256 * 64 ? (total + 255) / 256 : 256 * 64);
    hipLaunchKernelGGL(im2col_nchw_kernel, dim3(grid), dim3(256), 0, (hipStream_t)stream, img, out, N, C, H, W, kh, kw, stride, pad, Ho, Wo, Kpad);
    FORGE_LAUNCH_CHECK("forge_im2col_nchw");
    return 0;
}

extern "C" int forge_maxpool2d_nhwc(const float* in, float* out, int N, int H, int W, int C, int k, int stride, int pad, forge_stream_t stream) {
    FORGE_REQUIRE(in && out, FORGE_EINVAL, "forge_maxpool2d_nhwc: null pointer argument");
    FORGE_REQUIRE(N > 0 && H > 0 && W > 0 && C > 0 && C % 4 == 0 && k > 0 && stride > 0 && pad >= 0 && 2 * pad <= k, FORGE_EINVAL,
                  "forge_maxpool2d_nhwc: bad dims (C %% 4 == 0 required)");
    const int Ho = (H + 2 * pad - k) / stride + 1, Wo = (W + 2 * pad - k) / stride + 1;
    const long long total = (long long)N * Ho * Wo * (C / 4);
    const unsigned grid = (unsigned)((total + 255) / 256 < 256 * 64 ? (total + 255) / 256 : 256 * 64);
    hipLaunchKernelGGL(maxpool2d_nhwc_kernel, dim3(grid), dim3(256), 0, (hipStream_t)stream, (const float4*)in, (float4*)out, N, H, W, C / 4,
                       k, stride, pad, Ho, Wo);
    FORGE_LAUNCH_CHECK("forge_maxpool2d_nhwc");
    return 0;
}
