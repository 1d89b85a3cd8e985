// wino_in_bw.hip - which form of the Winograd input transform (csrc/winograd.hip wino_input_kernel) gets closest to the HBM rate of its own
// traffic pattern (read 1 x, write 4 x into 16 planes)? Standalone: hipcc --offload-arch=gfx950 -O3 -o wino_in_bw wino_in_bw.hip; ./wino_in_bw [n]
//   cur      one thread = one tile x 4 channels, 16 loads / 16 stores (the product kernel)
//   nt       the same with nontemporal V stores
//   pair     one thread = two tiles adjacent in W x 4 channels (24 loads instead of 32, half the waves)
//   loop2/4  the product kernel walking 2 / 4 tiles per thread (grid / 2, / 4): fewer, longer waves
//   ref      the same store pattern fed by each tile's OWN 2 x 2 pixels only (4 loads, no halo): the ceiling of the pattern
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <vector>
#include <algorithm>
#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e_)); exit(1); } } while (0)
typedef float v4 __attribute__((ext_vector_type(4)));
struct A { const float* in; float* V; long long ptv; int n, D, H, W, C; };
__device__ __forceinline__ unsigned xcd_remap(unsigned bid, unsigned nwg) {
    const unsigned q = nwg / 8, r = nwg % 8, xcd = bid % 8, k = bid / 8;
    return ((xcd < r) ? xcd * (q + 1) : r * (q + 1) + (xcd - r) * q) + k;
}
template <bool NT> __device__ __forceinline__ void st(float* p, v4 v) {
    if (NT) __builtin_nontemporal_store(v, reinterpret_cast<v4*>(p)); else *reinterpret_cast<v4*>(p) = v;
}
template <bool NT>
__device__ __forceinline__ void tile(const A& a, unsigned r, int c) {
    const int Ht = a.H >> 1, Wt = a.W >> 1;
    unsigned q = r, t = q / (unsigned)Wt;
    const int tw = (int)(q - t * (unsigned)Wt); q = t; t = q / (unsigned)Ht;
    const int th = (int)(q - t * (unsigned)Ht); q = t; t = q / (unsigned)a.D;
    const int z = (int)(q - t * (unsigned)a.D);
    const float* base = a.in + (((long long)t * a.D + z) * a.H * a.W) * a.C + c;
    v4 d[4][4];
#pragma unroll
    for (int i = 0; i < 4; ++i) {
        const int y = 2 * th - 1 + i;
#pragma unroll
        for (int j = 0; j < 4; ++j) {
            const int x = 2 * tw - 1 + j;
            const bool ok = (unsigned)y < (unsigned)a.H && (unsigned)x < (unsigned)a.W;
            d[i][j] = ok ? *reinterpret_cast<const v4*>(base + ((long long)y * a.W + x) * a.C) : (v4)(0.f);
        }
    }
    v4 w[4][4];
#pragma unroll
    for (int j = 0; j < 4; ++j) { w[0][j] = d[0][j] - d[2][j]; w[1][j] = d[1][j] + d[2][j]; w[2][j] = d[2][j] - d[1][j]; w[3][j] = d[1][j] - d[3][j]; }
    float* vp = a.V + (long long)r * a.C + c;
#pragma unroll
    for (int i = 0; i < 4; ++i) {
        st<NT>(vp + (4 * i + 0) * a.ptv, w[i][0] - w[i][2]);
        st<NT>(vp + (4 * i + 1) * a.ptv, w[i][1] + w[i][2]);
        st<NT>(vp + (4 * i + 2) * a.ptv, w[i][2] - w[i][1]);
        st<NT>(vp + (4 * i + 3) * a.ptv, w[i][1] - w[i][3]);
    }
}
template <bool NT, int LOOP>
__global__ __launch_bounds__(256) void k_cur(const A a) {
    const int C4 = a.C >> 2;
    const long long R = (long long)a.n * a.D * (a.H >> 1) * (a.W >> 1);
    const unsigned b = xcd_remap(blockIdx.x, gridDim.x);
#pragma unroll 1
    for (int l = 0; l < LOOP; ++l) {
        const long long idx = ((long long)b * LOOP + l) * 256 + threadIdx.x;
        if (idx >= R * C4) return;
        const unsigned r = (unsigned)(idx / C4);
        tile<NT>(a, r, (int)(idx - (long long)r * C4) << 2);
    }
}
template <bool NT>
__global__ __launch_bounds__(256) void k_pair(const A a) {       // two tiles (tw, tw + 1) per thread
    const int C4 = a.C >> 2, Ht = a.H >> 1, Wt = a.W >> 1, Wp = Wt >> 1;
    const long long idx = (long long)xcd_remap(blockIdx.x, gridDim.x) * 256 + threadIdx.x;
    const long long RP = (long long)a.n * a.D * Ht * Wp;
    if (idx >= RP * C4) return;
    const unsigned rp = (unsigned)(idx / C4);
    const int c = (int)(idx - (long long)rp * C4) << 2;
    unsigned q = rp, t = q / (unsigned)Wp;
    const int tp = (int)(q - t * (unsigned)Wp); q = t; t = q / (unsigned)Ht;
    const int th = (int)(q - t * (unsigned)Ht); q = t; t = q / (unsigned)a.D;
    const int z = (int)(q - t * (unsigned)a.D);
    const float* base = a.in + (((long long)t * a.D + z) * a.H * a.W) * a.C + c;
    v4 w[4][6];
    {
        v4 d[4][6];
#pragma unroll
        for (int i = 0; i < 4; ++i) {
            const int y = 2 * th - 1 + i;
#pragma unroll
            for (int j = 0; j < 6; ++j) {
                const int x = 4 * tp - 1 + j;
                const bool ok = (unsigned)y < (unsigned)a.H && (unsigned)x < (unsigned)a.W;
                d[i][j] = ok ? *reinterpret_cast<const v4*>(base + ((long long)y * a.W + x) * a.C) : (v4)(0.f);
            }
        }
#pragma unroll
        for (int j = 0; j < 6; ++j) { w[0][j] = d[0][j] - d[2][j]; w[1][j] = d[1][j] + d[2][j]; w[2][j] = d[2][j] - d[1][j]; w[3][j] = d[1][j] - d[3][j]; }
    }
    const unsigned r0 = ((t * a.D + z) * Ht + th) * Wt + 2 * tp;
#pragma unroll
    for (int u = 0; u < 2; ++u) {
        float* vp = a.V + (long long)(r0 + u) * a.C + c;
#pragma unroll
        for (int i = 0; i < 4; ++i) {
            st<NT>(vp + (4 * i + 0) * a.ptv, w[i][2 * u + 0] - w[i][2 * u + 2]);
            st<NT>(vp + (4 * i + 1) * a.ptv, w[i][2 * u + 1] + w[i][2 * u + 2]);
            st<NT>(vp + (4 * i + 2) * a.ptv, w[i][2 * u + 2] - w[i][2 * u + 1]);
            st<NT>(vp + (4 * i + 3) * a.ptv, w[i][2 * u + 1] - w[i][2 * u + 3]);
        }
    }
}
template <bool NT>
__global__ __launch_bounds__(256) void k_ref(const A a) {        // own 2 x 2 pixels only: compulsory traffic, same store pattern
    const int C4 = a.C >> 2, Ht = a.H >> 1, Wt = a.W >> 1;
    const long long idx = (long long)xcd_remap(blockIdx.x, gridDim.x) * 256 + threadIdx.x;
    const long long R = (long long)a.n * a.D * Ht * Wt;
    if (idx >= R * C4) return;
    const unsigned r = (unsigned)(idx / C4);
    const int c = (int)(idx - (long long)r * C4) << 2;
    unsigned q = r, t = q / (unsigned)Wt;
    const int tw = (int)(q - t * (unsigned)Wt); q = t; t = q / (unsigned)Ht;
    const int th = (int)(q - t * (unsigned)Ht); q = t; t = q / (unsigned)a.D;
    const int z = (int)(q - t * (unsigned)a.D);
    const float* base = a.in + ((((long long)t * a.D + z) * a.H + 2 * th) * a.W + 2 * tw) * a.C + c;
    v4 d[2][2];
#pragma unroll
    for (int i = 0; i < 2; ++i)
#pragma unroll
        for (int j = 0; j < 2; ++j) d[i][j] = *reinterpret_cast<const v4*>(base + ((long long)i * a.W + j) * a.C);
    float* vp = a.V + (long long)r * a.C + c;
#pragma unroll
    for (int p = 0; p < 16; ++p) st<NT>(vp + p * a.ptv, d[p & 1][(p >> 1) & 1] + (float)p * d[(p >> 2) & 1][(p >> 3) & 1]);
}
int main(int argc, char** argv) {
    const int n = argc > 1 ? atoi(argv[1]) : 1, D = 32, H = 32, W = 32, C = 128, NB = 6, IT = 60;
    const long long in_f = (long long)n * D * H * W * C, R = (long long)n * D * (H / 2) * (W / 2), v_f = 16 * R * C;
    std::vector<float*> ins(NB), Vs(NB);
    for (int i = 0; i < NB; ++i) { CK(hipMalloc(&ins[i], in_f * 4)); CK(hipMalloc(&Vs[i], v_f * 4)); CK(hipMemset(ins[i], 0, in_f * 4)); }
    hipEvent_t e0, e1; CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
    const long long total = R * (C / 4);
    const double bytes = (double)(in_f + v_f) * 4;
    auto run = [&](const char* name, auto launch) {
        std::vector<float> ts;
        for (int it = 0; it < IT + 5; ++it) {
            A a{ins[it % NB], Vs[it % NB], R * C, n, D, H, W, C};
            CK(hipEventRecord(e0, 0)); launch(a); CK(hipEventRecord(e1, 0)); CK(hipEventSynchronize(e1));
            float ms; CK(hipEventElapsedTime(&ms, e0, e1)); if (it >= 5) ts.push_back(ms * 1e3f);
        }
        std::sort(ts.begin(), ts.end());
        printf("%-10s n=%d  median %.2f us  min %.2f us  -> %.2f TB/s algorithmic (median)\n", name, n, ts[ts.size() / 2], ts[0], bytes / ts[ts.size() / 2] * 1e-6);
    };
    const unsigned g1 = (unsigned)((total + 255) / 256);
    run("cur", [&](A a) { hipLaunchKernelGGL((k_cur<false, 1>), dim3(g1), dim3(256), 0, 0, a); });
    run("nt", [&](A a) { hipLaunchKernelGGL((k_cur<true, 1>), dim3(g1), dim3(256), 0, 0, a); });
    run("loop2", [&](A a) { hipLaunchKernelGGL((k_cur<false, 2>), dim3((g1 + 1) / 2), dim3(256), 0, 0, a); });
    run("loop4", [&](A a) { hipLaunchKernelGGL((k_cur<false, 4>), dim3((g1 + 3) / 4), dim3(256), 0, 0, a); });
    run("loop4nt", [&](A a) { hipLaunchKernelGGL((k_cur<true, 4>), dim3((g1 + 3) / 4), dim3(256), 0, 0, a); });
    run("pair", [&](A a) { hipLaunchKernelGGL((k_pair<false>), dim3((g1 + 1) / 2), dim3(256), 0, 0, a); });
    run("pair_nt", [&](A a) { hipLaunchKernelGGL((k_pair<true>), dim3((g1 + 1) / 2), dim3(256), 0, 0, a); });
    run("ref", [&](A a) { hipLaunchKernelGGL((k_ref<false>), dim3(g1), dim3(256), 0, 0, a); });
    run("ref_nt", [&](A a) { hipLaunchKernelGGL((k_ref<true>), dim3(g1), dim3(256), 0, 0, a); });
    return 0;
}
