// attention.hip — single-head dot-product attention O = softmax(Q K^T) V in fp32 on the matrix cores, for the N = 4096-token attentions of the
// 3-D pose estimator in predicted-pose INFERENCE (models/model_utils.py:207-229 `Attention`, unscaled, one head of 64 channels; called by
// models/pose_estimator_3d.py:116-144: cross attention whose N x N matrix multiplies the positional table, then a self-attention block).
//
// Stock torch materialises the [B, N, N] matrix three times (rocBLAS QK^T, softmax, rocBLAS PV: 268 MB per pass at B = 4, N = 4096); here it
// never leaves registers (online softmax over key tiles, Milakov & Gimelshein / FlashAttention recurrence):
//   workgroup = 4 waves; KS = 2: 64 queries, wave w -> queries 32 (w & 1) .. +31 and key half (w >> 1); KS = 4 (few queries: B Nq / 64 workgroups
//   would leave SIMDs empty): 32 queries, wave w -> key quarter w. The key parts of a query are merged through LDS at the end.
//   per 32-key tile and wave: S^T = K Q^T   (32 v_mfma_f32_32x32x2_f32: lanes = queries, accumulator registers = keys)
//                             running max / sum per query = per lane (+ one exchange between the two half-waves), P = exp(S - max) in place
//                             (Q is pre-multiplied by log2 e, so exp is one v_exp_f32 per element: 2^(s' - max'))
//                             O^T += V^T P^T (32 MFMAs): the S^T accumulator registers ARE the B operand - the contraction runs over the keys in
//                             the order the accumulator holds them, and the A operand (V) is loaded in that order
// Q stays in registers for the whole loop; K / V tiles come straight from global memory (2 MB per batch element: L2-resident, every wave of a
// workgroup and 63 other workgroups read the same tiles), the next K tile is requested before the current tile's MFMAs.
// Bound: MFMA fp32. FLOPs = 4 B Nq Nk 64. The result differs from softmax-then-matmul only in the order of the fp32 additions.
#include "common.h"

namespace forge {

typedef float f16v __attribute__((ext_vector_type(16)));

constexpr int ATT_D = 64;          // channels of q / k and of v (one head)
constexpr float LOG2E = 1.44269504088896340736f;

template <int KS>
__global__ __launch_bounds__(256) void attention_fwd_kernel(const float* __restrict__ q, const float* __restrict__ k, const float* __restrict__ v,
                                                            long long v_batch_rows, float* __restrict__ out, int Nq, int Nk) {
    constexpr int QW = 4 / KS;                             // query groups (of 32) per workgroup
    __shared__ float mrg[3][64][35];                       // key parts 1.. of a query group: (O^T column: 32 floats, max, sum) per lane, padded
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const int h = lane >> 5, n = lane & 31;
    const int qtiles = Nq / (32 * QW);
    const int b = blockIdx.x / qtiles, qt = blockIdx.x - b * qtiles;
    const int qw = wave % QW, kh = wave / QW;
    const int q0 = (qt * QW + qw) << 5;
    const int kbeg = kh * (Nk / KS), kend = kbeg + Nk / KS;
    const float* Kb = k + (size_t)b * Nk * ATT_D;
    const float* Vb = v + (size_t)b * v_batch_rows * ATT_D;

    // B operand of S^T = K Q^T: lane (query n, half h) holds Q[q0 + n][32 h + s] for MFMA step s (the two channels one step contracts are
    // s and 32 + s: any pairing of the 64 channels is the same sum up to the order of the additions)
    float qr[32], kr[32], kn[32];
    {
        const float4* p = reinterpret_cast<const float4*>(q + ((size_t)b * Nq + q0 + n) * ATT_D + 32 * h);
#pragma unroll
        for (int i = 0; i < 8; ++i) { const float4 t = p[i]; qr[4 * i] = t.x * LOG2E; qr[4 * i + 1] = t.y * LOG2E; qr[4 * i + 2] = t.z * LOG2E; qr[4 * i + 3] = t.w * LOG2E; }
        const float4* pk = reinterpret_cast<const float4*>(Kb + (size_t)(kbeg + n) * ATT_D + 32 * h);
#pragma unroll
        for (int i = 0; i < 8; ++i) { const float4 t = pk[i]; kr[4 * i] = t.x; kr[4 * i + 1] = t.y; kr[4 * i + 2] = t.z; kr[4 * i + 3] = t.w; }
    }
    f16v o0, o1;
#pragma unroll
    for (int r = 0; r < 16; ++r) { o0[r] = 0.f; o1[r] = 0.f; }
    float m = -INFINITY, l = 0.f;

    for (int kt = kbeg; kt < kend; kt += 32) {
        // A operand of O^T += V^T P^T, in the key order of the S^T accumulator: register r of half h holds key 8 (r / 4) + 4 h + r % 4
        float v0[16], v1[16];
#pragma unroll
        for (int r = 0; r < 16; ++r) {
            const float* pv = Vb + (size_t)(kt + 8 * (r >> 2) + 4 * h + (r & 3)) * ATT_D + n;
            v0[r] = pv[0];
            v1[r] = pv[32];
        }
        const int ktn = kt + 32 < kend ? kt + 32 : kbeg;            // (the last iteration re-reads the first tile: no branch around the loads)
        {
            const float4* pk = reinterpret_cast<const float4*>(Kb + (size_t)(ktn + n) * ATT_D + 32 * h);
#pragma unroll
            for (int i = 0; i < 8; ++i) { const float4 t = pk[i]; kn[4 * i] = t.x; kn[4 * i + 1] = t.y; kn[4 * i + 2] = t.z; kn[4 * i + 3] = t.w; }
        }
        f16v s;
#pragma unroll
        for (int r = 0; r < 16; ++r) s[r] = 0.f;
#pragma unroll
        for (int i = 0; i < 32; ++i) s = __builtin_amdgcn_mfma_f32_32x32x2f32(kr[i], qr[i], s, 0, 0, 0);
        float tmax = s[0];
#pragma unroll
        for (int r = 1; r < 16; ++r) tmax = fmaxf(tmax, s[r]);
        tmax = fmaxf(tmax, __shfl_xor(tmax, 32));
        const float mn = fmaxf(m, tmax);
        const float sc = __builtin_amdgcn_exp2f(m - mn);                             // first tile: 2^(-inf) = 0
        float ps = 0.f;
#pragma unroll
        for (int r = 0; r < 16; ++r) { s[r] = __builtin_amdgcn_exp2f(s[r] - mn); ps += s[r]; }
        ps += __shfl_xor(ps, 32);
        l = l * sc + ps;
        m = mn;
#pragma unroll
        for (int r = 0; r < 16; ++r) { o0[r] *= sc; o1[r] *= sc; }
#pragma unroll
        for (int r = 0; r < 16; ++r) {
            o0 = __builtin_amdgcn_mfma_f32_32x32x2f32(v0[r], s[r], o0, 0, 0, 0);
            o1 = __builtin_amdgcn_mfma_f32_32x32x2f32(v1[r], s[r], o1, 0, 0, 0);
        }
#pragma unroll
        for (int i = 0; i < 32; ++i) kr[i] = kn[i];
    }

    // merge the key parts of each query: O = sum_p O_p 2^(m_p - M) / sum_p l_p 2^(m_p - M), M = max_p m_p
    if (kh > 0) {
        float* dst = mrg[(kh - 1) * QW + qw][lane];
#pragma unroll
        for (int r = 0; r < 16; ++r) { dst[r] = o0[r]; dst[16 + r] = o1[r]; }
        dst[32] = m;
        dst[33] = l;
    }
    __syncthreads();
    if (kh == 0) {
        float M = m;
#pragma unroll
        for (int p = 1; p < KS; ++p) M = fmaxf(M, mrg[(p - 1) * QW + qw][lane][32]);
        const float a0 = __builtin_amdgcn_exp2f(m - M);
        float den = l * a0;
#pragma unroll
        for (int r = 0; r < 16; ++r) { o0[r] *= a0; o1[r] *= a0; }
#pragma unroll
        for (int p = 1; p < KS; ++p) {
            const float* src = mrg[(p - 1) * QW + qw][lane];
            const float ap = __builtin_amdgcn_exp2f(src[32] - M);
            den += src[33] * ap;
#pragma unroll
            for (int r = 0; r < 16; ++r) { o0[r] += src[r] * ap; o1[r] += src[16 + r] * ap; }
        }
        const float inv = 1.f / den;
        // accumulator register r of half h = channel 8 (r / 4) + 4 h + r % 4 (o1: + 32) of query n: four consecutive channels per float4
        float* po = out + ((size_t)b * Nq + q0 + n) * ATT_D + 4 * h;
#pragma unroll
        for (int g = 0; g < 4; ++g) {
            *reinterpret_cast<float4*>(po + 8 * g) = make_float4(o0[4 * g] * inv, o0[4 * g + 1] * inv, o0[4 * g + 2] * inv, o0[4 * g + 3] * inv);
            *reinterpret_cast<float4*>(po + 32 + 8 * g) = make_float4(o1[4 * g] * inv, o1[4 * g + 1] * inv, o1[4 * g + 2] * inv, o1[4 * g + 3] * inv);
        }
    }
}

}  // namespace forge

using namespace forge;

extern "C" int forge_attention_fwd(const float* q, const float* k, const float* v, long long v_batch_rows, float* out, int B, int Nq, int Nk, int d,
                                   forge_stream_t stream) {
    FORGE_REQUIRE(q && k && v && out, FORGE_EINVAL, "forge_attention_fwd: null pointer argument");
    FORGE_REQUIRE(d == ATT_D, FORGE_ESHAPE, "forge_attention_fwd: one head of %d channels (got d=%d)", ATT_D, d);
    FORGE_REQUIRE(B > 0 && Nq > 0 && Nk > 0 && Nq % 64 == 0 && Nk % 64 == 0, FORGE_ESHAPE,
                  "forge_attention_fwd: B=%d Nq=%d Nk=%d (Nq and Nk must be multiples of 64)", B, Nq, Nk);
    // key parts per query: 4 when two-part workgroups (64 queries each) would not give every CU two workgroups (MI355X in SPX mode: 256 CUs ->
    // fewer than 512 query tiles). MI355X only, as the whole library: the constant is not derived from the device properties.
    const bool ks4 = Nk % 128 == 0 && (long long)B * (Nq / 64) < 2 * 256;
    FORGE_REQUIRE(v_batch_rows == 0 || v_batch_rows >= Nk, FORGE_EINVAL, "forge_attention_fwd: v batch stride %lld rows (0 = one v for every batch element, else >= Nk)",
                  v_batch_rows);
    FORGE_REQUIRE((long long)B * (Nq / 64) < (1ll << 31), FORGE_ESHAPE, "forge_attention_fwd: too many query tiles");
    if (ks4)
        hipLaunchKernelGGL(attention_fwd_kernel<4>, dim3((unsigned)(B * (Nq / 32))), dim3(256), 0, (hipStream_t)stream, q, k, v, v_batch_rows, out, Nq, Nk);
    else
        hipLaunchKernelGGL(attention_fwd_kernel<2>, dim3((unsigned)(B * (Nq / 64))), dim3(256), 0, (hipStream_t)stream, q, k, v, v_batch_rows, out, Nq, Nk);
    FORGE_LAUNCH_CHECK("forge_attention_fwd");
    return 0;
}
