// gru.hip — element-wise halves of the ConvGRU cell for the TRAINING path (models/fusion.py:29-35 under autograd).
//
// Inference fuses these into the GEMM epilogues (conv_igemm.hip: EPI_GRU_GATES / EPI_GRU_OUT). With an autograd graph the
// backward needs the gate values themselves, so the two convolutions run with the plain bias epilogue and the cell's
// element-wise math is ONE kernel per half and direction instead of ~23 generic tensor ops per step:
//   gates  fwd: z = sigma(g[:, :C]), r = sigma(g[:, C:]), hr = h r            bwd: dg = (dz z (1-z) | dhr h r (1-r)), dh += dhr r
//   state  fwd: cand = tanh(c), hn = h (1 - z) + cand z                        bwd: dh = dhn (1-z), dz = dhn (cand - h), dc = dhn z (1 - cand^2)
// All tensors are channels-last rows [M][C] (C % 4 == 0), float4 per thread, HBM-bound (5-7 arrays of M x C floats per launch).
#include "common.h"

namespace forge {

__device__ __forceinline__ float sigmoidf_(float v) { return 1.f / (1.f + expf(-v)); }


__global__ __launch_bounds__(256) void gru_gates_fwd_kernel(const float* __restrict__ g, const float* __restrict__ h, float* __restrict__ z,
                                                            float* __restrict__ r, float* __restrict__ hr, long long M, int C) {
    const int C4 = C >> 2;
    for (long long i = (long long)blockIdx.x * 256 + threadIdx.x; i < M * C4; i += (long long)gridDim.x * 256) {
        const long long m = i / C4;
        const int c = (int)(i - m * C4) << 2;
        const float4 gz = *reinterpret_cast<const float4*>(g + m * 2 * C + c), gr = *reinterpret_cast<const float4*>(g + m * 2 * C + C + c);
        const float4 hv = *reinterpret_cast<const float4*>(h + m * C + c);
        float4 zv, rv, hrv;
        zv.x = sigmoidf_(gz.x); zv.y = sigmoidf_(gz.y); zv.z = sigmoidf_(gz.z); zv.w = sigmoidf_(gz.w);
        rv.x = sigmoidf_(gr.x); rv.y = sigmoidf_(gr.y); rv.z = sigmoidf_(gr.z); rv.w = sigmoidf_(gr.w);
        hrv.x = hv.x * rv.x; hrv.y = hv.y * rv.y; hrv.z = hv.z * rv.z; hrv.w = hv.w * rv.w;
        *reinterpret_cast<float4*>(z + m * C + c) = zv;
        *reinterpret_cast<float4*>(r + m * C + c) = rv;
        *reinterpret_cast<float4*>(hr + m * C + c) = hrv;
    }
}

__global__ __launch_bounds__(256) void gru_state_fwd_kernel(float* __restrict__ c_cand, const float* __restrict__ h, const float* __restrict__ z,
                                                            float* __restrict__ hn, long long n4) {
    for (long long i = (long long)blockIdx.x * 256 + threadIdx.x; i < n4; i += (long long)gridDim.x * 256) {
        const float4 cv = reinterpret_cast<const float4*>(c_cand)[i], hv = reinterpret_cast<const float4*>(h)[i], zv = reinterpret_cast<const float4*>(z)[i];
        float4 t, o;
        t.x = tanhf(cv.x); t.y = tanhf(cv.y); t.z = tanhf(cv.z); t.w = tanhf(cv.w);
        o.x = hv.x * (1.f - zv.x) + t.x * zv.x; o.y = hv.y * (1.f - zv.y) + t.y * zv.y;
        o.z = hv.z * (1.f - zv.z) + t.z * zv.z; o.w = hv.w * (1.f - zv.w) + t.w * zv.w;
        reinterpret_cast<float4*>(c_cand)[i] = t;                  // the pre-activation is not needed again: keep tanh(c) for the backward
        reinterpret_cast<float4*>(hn)[i] = o;
    }
}

__global__ __launch_bounds__(256) void gru_state_bwd_kernel(const float* __restrict__ dhn, int ld_dhn, const float* __restrict__ h, const float* __restrict__ z,
                                                            const float* __restrict__ cand, float* __restrict__ dh, float* __restrict__ dz,
                                                            float* __restrict__ dc, long long n4, int C4, float* __restrict__ acc, long long acc_bs,
                                                            long long vol, int acc_mode) {
    for (long long i = (long long)blockIdx.x * 256 + threadIdx.x; i < n4; i += (long long)gridDim.x * 256) {
        const long long m = i / C4;
        const float4 g = *reinterpret_cast<const float4*>(dhn + m * ld_dhn + ((i - m * C4) << 2)), hv = reinterpret_cast<const float4*>(h)[i];
        const float4 zv = reinterpret_cast<const float4*>(z)[i], t = reinterpret_cast<const float4*>(cand)[i];
        float4 a, b, c;
        a.x = g.x * (1.f - zv.x); a.y = g.y * (1.f - zv.y); a.z = g.z * (1.f - zv.z); a.w = g.w * (1.f - zv.w);
        b.x = g.x * (t.x - hv.x); b.y = g.y * (t.y - hv.y); b.z = g.z * (t.z - hv.z); b.w = g.w * (t.w - hv.w);
        c.x = g.x * zv.x * (1.f - t.x * t.x); c.y = g.y * zv.y * (1.f - t.y * t.y);
        c.z = g.z * zv.z * (1.f - t.z * t.z); c.w = g.w * zv.w * (1.f - t.w * t.w);
        reinterpret_cast<float4*>(dh)[i] = a;
        reinterpret_cast<float4*>(dz)[i] = b;
        reinterpret_cast<float4*>(dc)[i] = c;
        if (acc_mode) {                                               // the candidate conv's input half is shared by several fusions: its gradient is summed per view
            const long long bi = m / vol;
            float4* ap = reinterpret_cast<float4*>(acc + (bi * acc_bs + (m - bi * vol)) * (long long)(C4 << 2)) + (i - m * C4);
            if (acc_mode == 2) { const float4 o = *ap; c.x += o.x; c.y += o.y; c.z += o.z; c.w += o.w; }
            *ap = c;
        }
    }
}

__global__ __launch_bounds__(256) void gru_gates_bwd_kernel(const float* __restrict__ dz, const float* dhr /* may alias dh_out */, int ld_dhr,
                                                            const float* __restrict__ h, const float* __restrict__ z, const float* __restrict__ r,
                                                            float* __restrict__ dg, const float* __restrict__ dh, float* dh_out, int ld_dh_out, long long M, int C,
                                                            float* __restrict__ acc, long long acc_bs, long long vol, int acc_mode) {
    const int C4 = C >> 2;
    for (long long i = (long long)blockIdx.x * 256 + threadIdx.x; i < M * C4; i += (long long)gridDim.x * 256) {
        const long long m = i / C4;
        const int c = (int)(i - m * C4) << 2;
        const float4 dzv = *reinterpret_cast<const float4*>(dz + m * C + c), dhrv = *reinterpret_cast<const float4*>(dhr + m * ld_dhr + c);
        const float4 hv = *reinterpret_cast<const float4*>(h + m * C + c), zv = *reinterpret_cast<const float4*>(z + m * C + c);
        const float4 rv = *reinterpret_cast<const float4*>(r + m * C + c);
        float4 gz, gr, dhv = *reinterpret_cast<const float4*>(dh + m * C + c);
        gz.x = dzv.x * zv.x * (1.f - zv.x); gz.y = dzv.y * zv.y * (1.f - zv.y); gz.z = dzv.z * zv.z * (1.f - zv.z); gz.w = dzv.w * zv.w * (1.f - zv.w);
        gr.x = dhrv.x * hv.x * rv.x * (1.f - rv.x); gr.y = dhrv.y * hv.y * rv.y * (1.f - rv.y);
        gr.z = dhrv.z * hv.z * rv.z * (1.f - rv.z); gr.w = dhrv.w * hv.w * rv.w * (1.f - rv.w);
        dhv.x = fmaf(dhrv.x, rv.x, dhv.x); dhv.y = fmaf(dhrv.y, rv.y, dhv.y); dhv.z = fmaf(dhrv.z, rv.z, dhv.z); dhv.w = fmaf(dhrv.w, rv.w, dhv.w);
        *reinterpret_cast<float4*>(dg + m * 2 * C + c) = gz;
        *reinterpret_cast<float4*>(dg + m * 2 * C + C + c) = gr;
        *reinterpret_cast<float4*>(dh_out + m * ld_dh_out + c) = dhv;
        if (acc_mode) {                                               // gradient of the gate conv's shared input half, summed per view over the fusions
            const long long bi = m / vol;
            float* ap = acc + (bi * acc_bs + (m - bi * vol)) * 2 * C + c;
            if (acc_mode == 2) {
                const float4 oz = *reinterpret_cast<const float4*>(ap), orr = *reinterpret_cast<const float4*>(ap + C);
                gz.x += oz.x; gz.y += oz.y; gz.z += oz.z; gz.w += oz.w; gr.x += orr.x; gr.y += orr.y; gr.z += orr.z; gr.w += orr.w;
            }
            *reinterpret_cast<float4*>(ap) = gz;
            *reinterpret_cast<float4*>(ap + C) = gr;
        }
    }
}

// dx = dy * (y > 0 ? 1 : slope) * scale[c]: backward of the folded eval-BatchNorm + LeakyReLU / ReLU epilogue (y = the forward OUTPUT;
// its sign is the pre-activation's sign for slope >= 0). dy / dx rows may be strided; scale nullable (= 1).
__global__ __launch_bounds__(256) void affine_act_bwd_kernel(const float* __restrict__ dy, int ld_dy, const float* __restrict__ y, int ld_y,
                                                             const float* __restrict__ scale, float slope, float* __restrict__ dx, int ld_dx,
                                                             long long M, int C) {
    for (long long i = (long long)blockIdx.x * 256 + threadIdx.x; i < M * C; i += (long long)gridDim.x * 256) {
        const long long m = i / C;
        const int c = (int)(i - m * C);
        const float g = dy[m * ld_dy + c] * (scale ? scale[c] : 1.f);
        dx[m * ld_dx + c] = y[m * ld_y + c] > 0.f ? g : g * slope;
    }
}

static unsigned ew_grid(long long n4) {
    long long b = (n4 + 255) / 256;
    return (unsigned)(b < 1 ? 1 : b > 256 * 16 ? 256 * 16 : b);          // grid-stride: at most 16 workgroups per CU
}

}  // namespace forge

using namespace forge;

extern "C" int forge_gru_gates_fwd(const float* g, const float* h, float* z, float* r, float* hr, long long M, int C, forge_stream_t stream) {
    FORGE_REQUIRE(g && h && z && r && hr, FORGE_EINVAL, "forge_gru_gates_fwd: null pointer argument");
    FORGE_REQUIRE(M > 0 && C > 0 && C % 4 == 0, FORGE_ESHAPE, "forge_gru_gates_fwd: M=%lld C=%d (C must be a multiple of 4)", M, C);
    hipLaunchKernelGGL(gru_gates_fwd_kernel, dim3(ew_grid(M * (C / 4))), dim3(256), 0, (hipStream_t)stream, g, h, z, r, hr, M, C);
    FORGE_LAUNCH_CHECK("forge_gru_gates_fwd");
    return 0;
}

extern "C" int forge_gru_state_fwd(float* c_cand, const float* h, const float* z, float* hn, long long M, int C, forge_stream_t stream) {
    FORGE_REQUIRE(c_cand && h && z && hn, FORGE_EINVAL, "forge_gru_state_fwd: null pointer argument");
    FORGE_REQUIRE(M > 0 && C > 0 && C % 4 == 0, FORGE_ESHAPE, "forge_gru_state_fwd: M=%lld C=%d (C must be a multiple of 4)", M, C);
    hipLaunchKernelGGL(gru_state_fwd_kernel, dim3(ew_grid(M * (C / 4))), dim3(256), 0, (hipStream_t)stream, c_cand, h, z, hn, M * (C / 4));
    FORGE_LAUNCH_CHECK("forge_gru_state_fwd");
    return 0;
}

static int gru_acc_check(const char* fn, const float* acc, long long acc_bs, long long vol, int acc_mode, long long M) {
    FORGE_REQUIRE(acc_mode >= 0 && acc_mode <= 2 && (acc_mode == 0 || (acc && vol > 0 && M % vol == 0 && acc_bs >= vol)), FORGE_EINVAL,
                  "%s: accumulator mode %d needs a buffer, vol > 0 dividing M and a batch stride >= vol rows", fn, acc_mode);
    return 0;
}

extern "C" int forge_gru_state_bwd(const float* dhn, int ld_dhn, const float* h, const float* z, const float* cand, float* dh, float* dz, float* dc,
                                   long long M, int C, float* dc_acc, long long acc_bs, long long vol, int acc_mode, forge_stream_t stream) {
    FORGE_REQUIRE(dhn && h && z && cand && dh && dz && dc, FORGE_EINVAL, "forge_gru_state_bwd: null pointer argument");
    if (int rc = gru_acc_check("forge_gru_state_bwd", dc_acc, acc_bs, vol, acc_mode, M)) return rc;
    FORGE_REQUIRE(M > 0 && C > 0 && C % 4 == 0 && ld_dhn >= C && ld_dhn % 4 == 0, FORGE_ESHAPE,
                  "forge_gru_state_bwd: M=%lld C=%d ld_dhn=%d (multiples of 4, ld_dhn >= C)", M, C, ld_dhn);
    hipLaunchKernelGGL(gru_state_bwd_kernel, dim3(ew_grid(M * (C / 4))), dim3(256), 0, (hipStream_t)stream, dhn, ld_dhn, h, z, cand, dh, dz, dc,
                       M * (C / 4), C / 4, dc_acc, acc_bs, vol > 0 ? vol : 1, acc_mode);
    FORGE_LAUNCH_CHECK("forge_gru_state_bwd");
    return 0;
}

extern "C" int forge_gru_gates_bwd(const float* dz, const float* dhr, int ld_dhr, const float* h, const float* z, const float* r,
                                   float* dg, float* dh, float* dh_out, int ld_dh_out, long long M, int C, float* dg_acc, long long acc_bs,
                                   long long vol, int acc_mode, forge_stream_t stream) {
    FORGE_REQUIRE(dz && dhr && h && z && r && dg && dh, FORGE_EINVAL, "forge_gru_gates_bwd: null pointer argument");
    if (int rc = gru_acc_check("forge_gru_gates_bwd", dg_acc, acc_bs, vol, acc_mode, M)) return rc;
    FORGE_REQUIRE(M > 0 && C > 0 && C % 4 == 0 && ld_dhr >= C && ld_dhr % 4 == 0, FORGE_ESHAPE,
                  "forge_gru_gates_bwd: M=%lld C=%d ld_dhr=%d (multiples of 4, ld_dhr >= C)", M, C, ld_dhr);
    FORGE_REQUIRE(dh_out == nullptr || (ld_dh_out >= C && ld_dh_out % 4 == 0), FORGE_ESHAPE, "forge_gru_gates_bwd: bad ld_dh_out=%d", ld_dh_out);
    hipLaunchKernelGGL(gru_gates_bwd_kernel, dim3(ew_grid(M * (C / 4))), dim3(256), 0, (hipStream_t)stream, dz, dhr, ld_dhr, h, z, r, dg, dh,
                       dh_out ? dh_out : dh, dh_out ? ld_dh_out : C, M, C, dg_acc, acc_bs, vol > 0 ? vol : 1, acc_mode);
    FORGE_LAUNCH_CHECK("forge_gru_gates_bwd");
    return 0;
}

extern "C" int forge_affine_act_bwd(const float* dy, int ld_dy, const float* y, int ld_y, const float* scale, float slope, float* dx, int ld_dx,
                                    long long M, int C, forge_stream_t stream) {
    FORGE_REQUIRE(dy && y && dx, FORGE_EINVAL, "forge_affine_act_bwd: null pointer argument");
    FORGE_REQUIRE(M > 0 && C > 0 && ld_dy >= C && ld_y >= C && ld_dx >= C, FORGE_ESHAPE, "forge_affine_act_bwd: M=%lld C=%d ld=%d/%d/%d", M, C, ld_dy, ld_y, ld_dx);
    hipLaunchKernelGGL(affine_act_bwd_kernel, dim3(ew_grid((M * C + 3) / 4)), dim3(256), 0, (hipStream_t)stream, dy, ld_dy, y, ld_y, scale, slope, dx, ld_dx, M, C);
    FORGE_LAUNCH_CHECK("forge_affine_act_bwd");
    return 0;
}
