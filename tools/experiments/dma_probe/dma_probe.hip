// Feasibility probe (round 3): buffer_load_dwordx4 ... lds (LDS-DMA, 16 bytes per lane) on gfx950 through inline assembly, so that the compiler
// inserts no vmcnt(0) drain and the waits can be placed by hand. Checks the LDS image order (lane-linear, 16 B per lane) and OOB-zero behaviour.
#include <hip/hip_runtime.h>
#include <cstdio>
typedef int v4i __attribute__((ext_vector_type(4)));
__global__ void k_asm(const float* g, float* out, int nbytes) {
    __shared__ __attribute__((aligned(16))) float lds[2048];
    const unsigned long long p = (unsigned long long)g;
    v4i r; r.x = (int)(p & 0xffffffffu); r.y = (int)((p >> 32) & 0xffff); r.z = nbytes; r.w = 0x00020000;
    const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63;
    const unsigned ldsbase = __builtin_amdgcn_readfirstlane((unsigned)(unsigned long long)(__attribute__((address_space(3))) float*)lds + wave * 2048);   // 2 loads x 1 KB per wave (wave-uniform -> SGPR)
    unsigned voff = (threadIdx.x ^ 1) * 16;                 // swap neighbouring 16-byte chunks: LDS lane L gets global chunk L ^ 1
    if (threadIdx.x == 5) voff = 0x80000000u;              // out of range -> zeros
    asm volatile("s_mov_b32 m0, %0\n\ts_nop 0\n\tbuffer_load_dwordx4 %1, %2, 0 offen lds\n" :: "s"(ldsbase), "v"(voff), "s"(r) : "memory");
    const unsigned voff2 = threadIdx.x * 16 + 4096;
    asm volatile("s_mov_b32 m0, %0\n\ts_nop 0\n\tbuffer_load_dwordx4 %1, %2, 0 offen lds\n" :: "s"(ldsbase + 1024), "v"(voff2), "s"(r) : "memory");
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __syncthreads();
    for (int i = 0; i < 8; ++i) out[threadIdx.x * 8 + i] = lds[wave * 512 + (i >> 2) * 256 + lane * 4 + (i & 3)];
}
int main() {
    const int N = 4096;
    float h[N], o[2048];
    for (int i = 0; i < N; ++i) h[i] = (float)i;
    float *d, *dout;
    hipMalloc(&d, sizeof(h)); hipMalloc(&dout, sizeof(o));
    hipMemcpy(d, h, sizeof(h), hipMemcpyHostToDevice);
    hipLaunchKernelGGL(k_asm, dim3(1), dim3(128), 0, 0, d, dout, (int)sizeof(h));
    hipMemcpy(o, dout, 128 * 8 * sizeof(float), hipMemcpyDeviceToHost);
    int bad = 0;
    for (int t = 0; t < 128; ++t)
        for (int i = 0; i < 8; ++i) {
            float want = i < 4 ? (t == 5 ? 0.f : (float)((t ^ 1) * 4 + i)) : (float)(1024 + t * 4 + (i - 4));
            if (o[t * 8 + i] != want) { if (bad < 8) printf("mismatch t=%d i=%d got %g want %g\n", t, i, o[t * 8 + i], want); ++bad; }
        }
    printf("dma_probe: %s (%d mismatches)\n", bad ? "FAILED" : "OK: lane-linear 16-byte LDS image, OOB lanes read zeros", bad);
    return bad != 0;
}
