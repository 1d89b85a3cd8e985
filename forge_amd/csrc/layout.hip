// layout.hip — NCDHW <-> channels-last (NDHWC) repack for callers that hold plain-contiguous
// volumes (the torch host code uses torch.channels_last_3d tensors and never needs these).
// 32x32 LDS-tiled transpose: coalesced on both sides, +1 padding against bank conflicts.
#include "common.h"

namespace forge {

// src [n][R][Cc] -> dst [n][Cc][R]
__global__ __launch_bounds__(256) void transpose_kernel(const float* __restrict__ src, float* __restrict__ dst,
                                                        long long R, long long Cc) {
    __shared__ float tile[32][33];
    const long long n = blockIdx.z;
    const long long c0 = (long long)blockIdx.x * 32, r0 = (long long)blockIdx.y * 32;
    const int tx = threadIdx.x & 31, ty = threadIdx.x >> 5;   // 32 x 8
    const float* s = src + n * R * Cc;
    float* d = dst + n * R * Cc;
#pragma unroll
    for (int j = 0; j < 32; j += 8) {
        const long long r = r0 + ty + j, c = c0 + tx;
        if (r < R && c < Cc) tile[ty + j][tx] = s[r * Cc + c];
    }
    __syncthreads();
#pragma unroll
    for (int j = 0; j < 32; j += 8) {
        const long long c = c0 + ty + j, r = r0 + tx;
        if (r < R && c < Cc) d[c * R + r] = tile[tx][ty + j];
    }
}

static int launch_transpose(const char* fn, const float* src, float* dst, int n, long long R, long long Cc, hipStream_t st) {
    FORGE_REQUIRE(src && dst, FORGE_EINVAL, "%s: null pointer argument", fn);
    FORGE_REQUIRE(n > 0 && R > 0 && Cc > 0, FORGE_EINVAL, "%s: bad dims", fn);
    const long long gx = (Cc + 31) / 32, gy = (R + 31) / 32;
    FORGE_REQUIRE(gy <= 65535 || gx <= 65535, FORGE_ESHAPE, "%s: grid too large", fn);
    FORGE_REQUIRE(n <= 65535, FORGE_ESHAPE, "%s: n too large", fn);
    dim3 grid((unsigned)gx, (unsigned)gy, (unsigned)n);
    FORGE_REQUIRE(gy <= 65535, FORGE_ESHAPE, "%s: more than 65535 row tiles", fn);
    hipLaunchKernelGGL(transpose_kernel, grid, dim3(256), 0, st, src, dst, R, Cc);
    FORGE_LAUNCH_CHECK(fn);
    return 0;
}

}  // namespace forge

using namespace forge;

extern "C" int forge_ncdhw_to_ndhwc(const float* src, float* dst, int n, int C, long long P, forge_stream_t stream) {
    // src [n][C][P] -> dst [n][P][C]: rows = C, cols = P; put the long axis on grid.x
    return launch_transpose("forge_ncdhw_to_ndhwc", src, dst, n, C, P, (hipStream_t)stream);
}

extern "C" int forge_ndhwc_to_ncdhw(const float* src, float* dst, int n, int C, long long P, forge_stream_t stream) {
    // src [n][P][C] -> dst [n][C][P]: rows = P, cols = C
    FORGE_REQUIRE(P / 32 < 65535, FORGE_ESHAPE, "forge_ndhwc_to_ncdhw: P too large for one launch");
    return launch_transpose("forge_ndhwc_to_ncdhw", src, dst, n, P, C, (hipStream_t)stream);
}
