// conv_igemm.hip — implicit-GEMM convolution on the gfx950 fp32 matrix cores.
//
// Replaces the dense convolutions of the hot path that the reference runs through cuDNN:
// the ConvGRU fusion (models/fusion.py:29-35, 61-68: 71 % of the hot-path FLOPs), conv1
// (models/encoder.py:36-40) and the heads (models/encoder.py:16-34), with the element-wise tails
// (bias, eval-mode BN, LeakyReLU, GRU sigmoid/tanh/lerp, the cat([x, h])) fused into the GEMM.
//
//   GEMM view:  M = output voxels, N = C_out, K = taps * C_in
//   A[m][k]     = in[voxel(m) + tap offset][ci]      gathered on the fly (zero outside the grid)
//   B[k][n]     = Wp[tap][n][ci]                      weights pre-packed [tap][C_out][C_in]
//
// Arithmetic is exact fp32: v_mfma_f32_32x32x2_f32 (f32 in, f32 accumulate; bitwise an fmaf
// chain; 157 TFLOP/s chip peak = 1/16 of bf16) — the reference is fp32 and parity is stated in
// fp32. There is no xf32/TF32 on gfx950.
//
// Tiling: conv_igemm_kernel<BM, BN, NW, MT>: NW waves (8 or 4) arranged WM(M) x WN(N), workgroup
// tile BM x BN x 32(K) (128x128, 64x128, 128x64 with 8 waves; 64x64, 128x32 with 4), wave tile
// (32 MT) x (BN / WN) made of 32x32 MFMA tiles; the tile and a split-K factor are chosen per launch
// by plan_conv(). Both operands are staged as [row][32 k] images in LDS (128-byte rows, 16-byte
// chunks XOR-swizzled with (row>>1)&7 so that the 16-lane groups of ds_read_b128 hit 16 distinct
// slots), filled by LDS-DMA (buffer_load_dwordx4 ... lds: global memory straight into LDS, an
// out-of-range offset lands as zeros — the zero padding of out-of-grid taps / ragged tiles costs
// neither a branch nor a select; round 3, see the kernel's comment and common.h), double-buffered
// with one barrier per K-step and the next K-step's loads in flight under the current MFMAs. The K order inside a 32-chunk is permuted identically for A
// and B: a lane's 16-byte read supplies operand k = 4c+j (lanes 0-31) / 4c+4+j (lanes 32-63) of
// MFMA j.
//
// Round 2: the same kernel also runs the 16 point GEMMs of a Winograd F(2x2, 3x3) x depth-tap convolution in ONE launch (forge_wino_gemm,
// `nbat` problems: workgroup ranges select the point and add a per-point offset to the operand / weight / output pointers; transforms in
// winograd.hip) - the stride-1 3x3x3 convolutions of the fusion, conv1 and the ResNet layer3/4 3x3 convolutions take that route.
#include "common.h"
#include <cmath>
#include <type_traits>

namespace forge {

typedef __attribute__((ext_vector_type(16))) float f32x16;

enum ConvEpilogue : int {
    EPI_BIAS = 0,        // y = acc + bias
    EPI_AFFINE_ACT = 1,  // y = lrelu((acc + bias) * scale + shift, slope)   (eval BN folded; slope 1 = none, 0 = ReLU)
    EPI_GRU_GATES = 2,   // cols [0,Ch): out[m][c] = sigmoid(v) (update z); cols [Ch,2Ch): out2[m][c-Ch] = h * sigmoid(v)
    EPI_GRU_OUT = 3,     // cand = tanh(v); hn = h (1 - z) + cand z; out = hn; out2 (nullable) = hn * scale + shift
};

constexpr int MAX_TAPS = 64;

struct ConvArgs {
    const float* in1; const float* in2;   // channel-concatenated inputs, channels-last; in2 nullable
    int C1, C2, ld1, ld2;                  // channels taken from each input and their row strides (floats)
    long long bs1r, bs2r;                  // batch strides of in1/in2 in rows (voxels)
    long long span1, span2;                // byte spans of in1/in2 (buffer-descriptor bounds, < 2 GiB)
    const float* wp;                       // [ntaps][Cout][C1 + C2]
    const float* bias;                     // [Cout] nullable
    const float* scale; const float* shift; float slope;
    const float* aux_h; const float* aux_z;
    float* out; float* out2; float* out3;   // out3 (nullable): GRU gates -> reset gate r; GRU out -> cand = tanh(v)  (saved for a hand-written backward)
    int n, D, H, W;                        // GEMM-row grid (M = n D H W rows)
    int is, Di, Hi, Wi;                    // input voxel = (z is + dz, y is + dy, x is + dx) in an (n,Di,Hi,Wi) grid
    const float* residual; int ldr;        // EPI_AFFINE_ACT: added before the activation (nullable), [rows][ldr]
    int Cout, ldo;                         // output channels, output row stride (floats)
    int ntaps;
    int os, pz, py, px, Do, Ho, Wo;        // output voxel = (z os + pz, y os + py, x os + px) in an (Do,Ho,Wo) grid
    int nphase, tpp;                       // > 1: all output phases of a stride-2 transposed conv in ONE launch: phase p = (pz,py,px) bits uses taps [p tpp, (p+1) tpp)
    int epi;
    int lift;                              // > 0: 2D->3D lift of the output (models/encoder.py:49), see forge_hip.h
    int nbat;                              // > 1: nbat independent GEMMs in one launch (the 16 Winograd points, forge_wino_gemm): problem p adds
    long long pt1, pt2, ptw, pto;          //      p * pt1 / pt2 / ptw / pto floats to in1 / in2 / wp / out (EPI_BIAS, no split-K, no phases)
    float* ws; int ksplit;                 // split-K: raw partial tiles go to ws[ks][M][Cout], a second kernel reduces + applies the epilogue
    double* stats;                         // EPI_BIAS, nullable: per 32-row block of M the column sums / sums of squares of the OUTPUT, [ceil(M/32)..][2][Cout] float64 -
                                           // the batch statistics of the BatchNorm behind this convolution as a by-product of its epilogue (csrc/bnorm.hip finalizes them)
    signed char tap[MAX_TAPS][4];          // (dz, dy, dx, 0)
};

constexpr int BK = 32, NTHREADS = 512;

#ifdef FORGE_CONV_TIMING   // debug build (tools/debug/conv_timing.py): per-workgroup clock stamps at entry / first barrier / loop end / exit
__device__ long long g_conv_stamp[8192 * 4];
#define FORGE_STAMP(k) do { if (threadIdx.x == 0 && blockIdx.x < 8192) g_conv_stamp[blockIdx.x * 4 + (k)] = wall_clock64(); } while (0)
extern "C" int forge_debug_conv_stamps(long long* host, int n) {
    return (int)hipMemcpyFromSymbol(host, HIP_SYMBOL(g_conv_stamp), (size_t)n * 4 * sizeof(long long));
}
#else
#define FORGE_STAMP(k) do { } while (0)
#endif

constexpr unsigned OOB = 0x80000000u;      // byte offset beyond any buffer (< 2 GiB spans enforced on the host side): such a lane of an LDS-DMA load
                                           // lands as zeros - the zero padding of out-of-grid taps and of rows / columns beyond M / Cout costs neither
                                           // a branch nor a select

__device__ __forceinline__ int lds_off(int row, int chunk) {   // float offset of a 16-byte chunk
    return row * BK + ((chunk ^ ((row >> 1) & 7)) << 2);
}

// Operand staging is LDS-DMA: `buffer_load_dwordx4 ... offen lds` (M0 = the wave's LDS byte address; lane L lands at + 16 L, out-of-range
// lanes write zeros) puts a K-step's A / B rows straight into the other LDS stage - no staging VGPRs, no ds_write. The loads are inline
// assembly because the compiler's own waitcnt insertion would drain them (vmcnt(0)) BEFORE the step's MFMAs (round 2's builtin-based variant
// lost 2-6 % to that); here the loads of step s+1 are issued at the top of step s, fly under its 16 MFMAs per accumulator, and the hand-placed
// `s_waitcnt vmcnt(0)` sits directly in front of the step's closing barrier. Against the register-staged loop this kernel replaced (same K
// order, bitwise the same results): +5..21 % per launch (profiles/TUNING_LOG.md, round 3: 64x128 tile 106 -> 129 TF on the Winograd gates
// GEMM, 128x128 tile 112 -> 135 TF on the K = 6912 direct launch). A third LDS stage (loads two steps ahead) was slower than two: the extra
// LDS costs a resident workgroup per CU.
//
// RS = 1 (forge_wino_gemm_half, 64x128 tile): one workgroup runs the FOUR Winograd points i = 0..3 of a point column j on its tile, one K loop after
// the other into four accumulator sets (the first stage of the next point is loaded under the last step of the current one), and its epilogue
// applies the ROW stage of the inverse transform A^T M A in registers - the same lane holds element (row, col) of all four points:
// s0 = (m0 + m1) + m2, s1 = (m1 - m2) - m3, wino_output_kernel's own operations in its own order - and stores 2 planes instead of 4:
// Mm8 [2][4][R][Cout]. The point products then cross HBM as 2x instead of 4x the output tensor, written here and read by the inverse transform.
template <int BM, int BN, int NW, int MT = 1, int RS = 0>
__global__ __launch_bounds__(NW * 64) void conv_igemm_kernel(const ConvArgs a) {
    constexpr int NP = RS ? 4 : 1;                   // points per workgroup
    constexpr int WM = BM / (32 * MT), WN = NW / WM; // NW waves as WM(M) x WN(N); wave tile (32 MT) x (BN / WN)
    constexpr int NT = BN / (32 * WN);              // 32-col MFMA tiles per wave
    static_assert(NT >= 1 && WM * WN == NW && NT * 32 * WN == BN && WM * 32 * MT == BM, "unsupported tile");
    constexpr int A_FLOATS = BM * BK, B_FLOATS = BN * BK;
    constexpr int RPP = NW * 8;                     // tile rows staged per pass (8 threads per 128-byte row)
    constexpr int ACH = BM / RPP, BCH = BN / RPP;   // 16-byte chunks per thread per K-step
    static_assert(ACH >= 1 && BCH >= 1, "tile too small for the workgroup");
    extern __shared__ __attribute__((aligned(16))) float smem[];   // [2][A_FLOATS + B_FLOATS]

    FORGE_STAMP(0);
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int wm = wave / WN, wn = wave % WN;
    const long long M = (long long)a.n * a.D * a.H * a.W;
    const int ntile_n = (a.Cout + BN - 1) / BN;
    const unsigned bid_all = xcd_remap(blockIdx.x, gridDim.x);
    const int ks = (int)(bid_all % (unsigned)a.ksplit);           // K-slice of this workgroup (split-K for small M x N problems)
    unsigned bid = bid_all / (unsigned)a.ksplit;
    int phase = 0;
    if (a.nphase > 1) {
        // merged transposed-conv phases, phase fastest: the 2^nd output phases of one row tile read the same 3^nd neighbourhood of input rows -
        // dispatched back to back on one XCD (xcd_remap) they share them through its L2 (phase-major, every phase streamed the whole input
        // again: 8 GB of L2->fabric traffic per heads launch of the 4-scene training step, profiles/r04_train_pmc_traffic_b4.txt; 883 -> 867 us)
        phase = (int)(bid % (unsigned)a.nphase);
        bid /= (unsigned)a.nphase;
    }
    const int pz = a.nphase > 1 ? (a.nphase == 8 ? (phase >> 2) & 1 : 0) : a.pz;
    const int py = a.nphase > 1 ? (phase >> 1) & 1 : a.py, px = a.nphase > 1 ? phase & 1 : a.px;
    const int t_lo = phase * a.tpp;
    long long pb = 0;                                               // batched problems: workgroups [p tiles, (p+1) tiles) do problem p, so that (with
    if (a.nbat > 1) {                                               // xcd_remap's contiguous chunks) an XCD's L2 holds the weights of its own problems only
        const unsigned tiles = (unsigned)((M + BM - 1) / BM) * (unsigned)ntile_n;
        pb = bid / tiles;
        bid -= (unsigned)pb * tiles;
    }
    float* const outp = a.out + pb * a.pto;
    const long long m0 = (long long)(bid / ntile_n) * BM;
    const int n0 = (bid % ntile_n) * BN;
    const int Cin = a.C1 + a.C2;
    const int kchunks = Cin / BK;
    const int nsteps_all = a.tpp * kchunks;
    const int s_begin = (int)((long long)ks * nsteps_all / a.ksplit), s_end = (int)((long long)(ks + 1) * nsteps_all / a.ksplit);
    const int nsteps = s_end - s_begin;

    // RS: pb = the point column j; the workgroup's points are pb, pb + 4, pb + 8, pb + 12 (descriptors re-made at every point switch)
    forge_v4i32 w1 = make_rsrc_words(a.in1 + pb * a.pt1, a.span1), w2 = make_rsrc_words(a.in2 ? a.in2 + pb * a.pt2 : a.in1, a.in2 ? a.span2 : 0),
                ww = make_rsrc_words(a.wp + pb * a.ptw, (long long)a.ntaps * a.Cout * Cin * 4);
    // LDS byte address of this WAVE's first 1 KB block of a stage (lane L lands at + 16 L: the row-major [row][32 k] image, 8 lanes per row)
    const unsigned lds_wave = lds_addr(smem) + (unsigned)__builtin_amdgcn_readfirstlane(wave) * 1024u;
    // ---- per-thread staging geometry: ACH A rows, BCH B rows, one 16-byte chunk each
    const int cp = tid & 7;                                      // physical chunk in the 128-byte LDS row
    int ar[ACH], az[ACH], ay[ACH], ax[ACH], an[ACH], asrc[ACH];
    bool aval[ACH];
#pragma unroll
    for (int j = 0; j < ACH; ++j) {
        ar[j] = (tid >> 3) + RPP * j;
        const long long vl = m0 + ar[j];
        aval[j] = vl < M;
        unsigned v = aval[j] ? (unsigned)vl : 0u;                // M < 2^31 (checked on the host): 32-bit divisions (64-bit ones are ~100 instructions each)
        unsigned q = v / (unsigned)a.W;
        ax[j] = (int)(v - q * (unsigned)a.W) * a.is; v = q;
        q = v / (unsigned)a.H;
        ay[j] = (int)(v - q * (unsigned)a.H) * a.is; v = q;
        q = v / (unsigned)a.D;
        az[j] = (int)(v - q * (unsigned)a.D) * a.is;
        an[j] = (int)q;
        asrc[j] = (cp ^ ((ar[j] >> 1) & 7)) << 2;                // logical channel offset inside the 32-chunk
    }
    int br[BCH]; unsigned boff[BCH];
#pragma unroll
    for (int j = 0; j < BCH; ++j) {
        br[j] = (tid >> 3) + RPP * j;
        const int bsrc = (cp ^ ((br[j] >> 1) & 7)) << 2;
        boff[j] = (n0 + br[j]) < a.Cout ? (unsigned)(((n0 + br[j]) * Cin + bsrc) * 4) : OOB;
    }

    // 128x128 tile (tap-inner K order, below): switching taps every K-step must be cheap. Per A row: bit t of vmask = tap t lands
    // inside the input grid; the input row of tap t is base + delta(t) with the uniform delta(t) = (dz Hi + dy) Wi + dx (LDS table),
    // so a tap switch costs a shift, a test and an add per row. The other tiles switch taps once per Cin / 32 K-steps and compute it directly.
    constexpr bool KC_OUTER = (BM == 128 && BN == 128);
    static_assert(!RS || (!KC_OUTER && MT == 1), "the row-stage form exists on the tap-outer tiles");
    __shared__ int sdelta[KC_OUTER ? 2 * MAX_TAPS : 1];             // byte deltas of the taps for in1 / in2
    unsigned long long vmask[KC_OUTER ? ACH : 1];
    unsigned base1[KC_OUTER ? ACH : 1], base2[KC_OUTER ? ACH : 1];  // byte offset of (row of tap (0,0,0), this thread's 16-byte chunk) in in1 / in2
    if constexpr (KC_OUTER) {
#pragma unroll
        for (int j = 0; j < ACH; ++j) {
            const int sp = (az[j] * a.Hi + ay[j]) * a.Wi + ax[j];
            base1[j] = (unsigned)(((an[j] * (int)a.bs1r + sp) * a.ld1 + asrc[j]) * 4);
            base2[j] = (unsigned)(((an[j] * (int)a.bs2r + sp) * a.ld2 + asrc[j]) * 4);
            vmask[j] = 0ull;
        }
        for (int tt = 0; tt < a.ntaps; ++tt) {
            const int dz = a.tap[tt][0], dy = a.tap[tt][1], dx = a.tap[tt][2];
#pragma unroll
            for (int j = 0; j < ACH; ++j) {
                const int zi = az[j] + dz, yi = ay[j] + dy, xi = ax[j] + dx;
                const bool ok = aval[j] && (unsigned)zi < (unsigned)a.Di && (unsigned)yi < (unsigned)a.Hi && (unsigned)xi < (unsigned)a.Wi;
                vmask[j] |= (unsigned long long)ok << tt;
            }
        }
        if (tid < a.ntaps) {
            const int d = (a.tap[tid][0] * a.Hi + a.tap[tid][1]) * a.Wi + a.tap[tid][2];
            sdelta[tid] = d * a.ld1 * 4;
            sdelta[MAX_TAPS + tid] = d * a.ld2 * 4;
        }
        __syncthreads();
    }
    // Byte offset of each A row's chunk for the current tap, OOB (>= 2^31) when the tap falls outside the grid: a K-step's load address is then
    // ONE add (offset + 4 c0) per chunk - no multiply, no select (OOB + anything < 2^31 stays out of the buffer's range and reads 0).
    unsigned eoff[ACH], eoff2[ACH];
    auto prep_tap = [&](int t) {
        if constexpr (KC_OUTER) {
            const unsigned d1 = (unsigned)sdelta[t], d2 = (unsigned)sdelta[MAX_TAPS + t];
#pragma unroll
            for (int j = 0; j < ACH; ++j) {
                const bool ok = (vmask[j] >> t) & 1ull;
                eoff[j] = ok ? base1[j] + d1 : OOB;
                eoff2[j] = ok ? base2[j] + d2 : OOB;
            }
        } else {
            const int dz = a.tap[t][0], dy = a.tap[t][1], dx = a.tap[t][2];
#pragma unroll
            for (int j = 0; j < ACH; ++j) {
                const int zi = az[j] + dz, yi = ay[j] + dy, xi = ax[j] + dx;
                const bool ok = aval[j] && (unsigned)zi < (unsigned)a.Di && (unsigned)yi < (unsigned)a.Hi && (unsigned)xi < (unsigned)a.Wi;
                const int sp = (zi * a.Hi + yi) * a.Wi + xi;
                eoff[j] = ok ? (unsigned)(((an[j] * (int)a.bs1r + sp) * a.ld1 + asrc[j]) * 4) : OOB;
                eoff2[j] = ok ? (unsigned)(((an[j] * (int)a.bs2r + sp) * a.ld2 + asrc[j]) * 4) : OOB;
            }
        }
    };
    // LDS-DMA: chunk j of this thread is row (tid >> 3) + RPP j = LDS bytes 16 tid + j NW 1024 -> per wave a lane-linear 1 KB block
    auto issue_step = [&](int t, int kc, int buf) {
        const int c0 = kc * BK;
        const unsigned stage = lds_wave + (unsigned)buf * (unsigned)((A_FLOATS + B_FLOATS) * 4);
        if (c0 < a.C1) {
#pragma unroll
            for (int j = 0; j < ACH; ++j) lds_dma16(w1, eoff[j] + (unsigned)(c0 * 4), stage + (unsigned)(j * NW * 1024));
        } else {
#pragma unroll
            for (int j = 0; j < ACH; ++j) lds_dma16(w2, eoff2[j] + (unsigned)((c0 - a.C1) * 4), stage + (unsigned)(j * NW * 1024));
        }
        const unsigned wbase = (unsigned)((t * a.Cout * Cin + c0) * 4);
#pragma unroll
        for (int j = 0; j < BCH; ++j) lds_dma16(ww, boff[j] + wbase, stage + (unsigned)(A_FLOATS * 4 + j * NW * 1024));
    };
    f32x16 accs[NP][MT][NT];
#pragma unroll
    for (int q = 0; q < NP; ++q)
#pragma unroll
        for (int i = 0; i < MT; ++i)
#pragma unroll
            for (int j = 0; j < NT; ++j)
#pragma unroll
                for (int r = 0; r < 16; ++r) accs[q][i][j][r] = 0.f;
    f32x16 (&acc)[MT][NT] = accs[0];                                // the single-point form's accumulators (epilogues below)

    const int half = lane >> 5, l31 = lane & 31;
    // K order of the 128x128 tile: channel chunk OUTER, tap INNER. All workgroups of an XCD walk the taps of one 32-channel slice of
    // their rows before moving to the next slice, so the slice (128 B per row, ~0.7 MB per XCD with its halo) stays in the 4 MiB L2
    // across the 27 taps; with the tap outer the per-tap working set is the full 1 KB rows = the whole L2 and every tap re-fetches them
    // from the fabric (ConvGRU gates launch: 502 -> 132 MB of L2 fills per launch, FETCH_SIZE). The 64-row tiles keep the tap outer:
    // they measured 3-4 % slower with the chunk outer (tools/conv_ab.py).
    int kc, t;
    if constexpr (KC_OUTER) { kc = s_begin / a.tpp; t = s_begin - kc * a.tpp; }
    else { t = s_begin / kchunks; kc = s_begin - t * kchunks; }
    t += t_lo;
    auto advance = [&]() {
        if constexpr (KC_OUTER) {
            if (++t == t_lo + a.tpp) { t = t_lo; ++kc; }
            prep_tap(t);
        } else {
            if (++kc == kchunks) { kc = 0; ++t; prep_tap(t); }
        }
    };
    prep_tap(t);
    issue_step(t, kc, 0);
    lds_dma_wait();
    __syncthreads();
    FORGE_STAMP(1);
    // One K-step = 4 MFMA groups of 8 k-values. (A/B in round 1: issuing the next tile's global loads after group 0 and its LDS
    // writes after group 2, pinned with sched_barrier, changed nothing: 127.3 vs 126.3 TF on the ConvGRU gates shape.)
    auto mfma_group = [&](f32x16 (&acc)[MT][NT], const float* sa, const float* sb, int g) {
        float4 fa[MT], fb[NT];
#pragma unroll
        for (int i = 0; i < MT; ++i) fa[i] = *reinterpret_cast<const float4*>(sa + lds_off(wm * (32 * MT) + i * 32 + l31, 2 * g + half));
#pragma unroll
        for (int j = 0; j < NT; ++j) fb[j] = *reinterpret_cast<const float4*>(sb + lds_off(wn * (BN / WN) + j * 32 + l31, 2 * g + half));
#pragma unroll
        for (int i = 0; i < MT; ++i)
#pragma unroll
            for (int j = 0; j < NT; ++j) acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x2f32(fa[i].x, fb[j].x, acc[i][j], 0, 0, 0);
#pragma unroll
        for (int i = 0; i < MT; ++i)
#pragma unroll
            for (int j = 0; j < NT; ++j) acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x2f32(fa[i].y, fb[j].y, acc[i][j], 0, 0, 0);
#pragma unroll
        for (int i = 0; i < MT; ++i)
#pragma unroll
            for (int j = 0; j < NT; ++j) acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x2f32(fa[i].z, fb[j].z, acc[i][j], 0, 0, 0);
#pragma unroll
        for (int i = 0; i < MT; ++i)
#pragma unroll
            for (int j = 0; j < NT; ++j) acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x2f32(fa[i].w, fb[j].w, acc[i][j], 0, 0, 0);
    };
    int buf = 0;
#pragma unroll
    for (int q = 0; q < NP; ++q) {
        for (int s = 0; s < nsteps; ++s) {
            const float* sa = smem + buf * (A_FLOATS + B_FLOATS);
            const float* sb = sa + A_FLOATS;
            if (s + 1 < nsteps) {                                 // the other stage was last read in step s - 1: every wave is past that barrier
                advance();
                issue_step(t, kc, buf ^ 1);                       // in flight under this step's MFMAs
            } else if (q + 1 < NP) {                              // RS: the next point's first stage, in flight under this point's last step
                const long long pn = pb + 4 * (q + 1);
                w1 = make_rsrc_words(a.in1 + pn * a.pt1, a.span1);
                w2 = make_rsrc_words(a.in2 ? a.in2 + pn * a.pt2 : a.in1, a.in2 ? a.span2 : 0);
                ww = make_rsrc_words(a.wp + pn * a.ptw, (long long)a.ntaps * a.Cout * Cin * 4);
                t = s_begin / kchunks; kc = s_begin - t * kchunks; t += t_lo;
                prep_tap(t);
                issue_step(t, kc, buf ^ 1);
            }
#pragma unroll
            for (int g = 0; g < 4; ++g) mfma_group(accs[q], sa, sb, g);
            lds_dma_wait();                                       // this wave's part of the next step is in LDS ...
            __syncthreads();                                      // ... and so is everybody else's
            buf ^= 1;
        }
    }

    FORGE_STAMP(2);
    // ---- epilogue: C/D layout of 32x32 MFMA: col = lane & 31, row = (r & 3) + 8 (r >> 2) + 4 (lane >> 5)
    // The output-row mapping (identity, strided / phase remap of a transposed convolution, or the 2D->3D lift) is computed ONCE per
    // tile row by one thread (32-bit divisions) into an LDS table; the fully unrolled per-accumulator code below (the accumulators
    // must stay in registers) then holds no division and no remap branch. (Round 1 inlined three 64-bit divisions per accumulator
    // element: 45-90 k instructions per instantiation, far beyond the instruction cache, which cost every workgroup a chain of
    // instruction-fetch misses - the reason 2 us of MFMA work took 20 us per launch in the ResNet trunk.)
    const int Ch = a.Cout / 2;
    const bool remap = (a.os != 1) || (a.Do != a.D) || (a.Ho != a.H) || (a.Wo != a.W);
    long long* s_row = reinterpret_cast<long long*>(smem);          // [BM] output row (or lifted base offset) of each tile row, -1 = beyond M
    if (a.ksplit == 1 && tid < BM) {
        const long long m = m0 + tid;
        long long o = -1;
        if (m < M) {
            if (a.lift > 0) {                                        // row (n, hw) -> base of out[n][0][hw][0] in the [n][lift][HW][Cl] volume
                const unsigned HW = (unsigned)(a.H * a.W), nn = (unsigned)m / HW, hw = (unsigned)m - nn * HW;
                o = ((long long)nn * a.lift * HW + hw) * (a.Cout / a.lift);
            } else if (remap) {
                unsigned q = (unsigned)m, t2 = q / (unsigned)a.W;
                const int x = (int)(q - t2 * (unsigned)a.W); q = t2; t2 = q / (unsigned)a.H;
                const int y = (int)(q - t2 * (unsigned)a.H); q = t2; t2 = q / (unsigned)a.D;
                const int z = (int)(q - t2 * (unsigned)a.D);
                o = (((long long)t2 * a.Do + (z * a.os + pz)) * a.Ho + (y * a.os + py)) * a.Wo + (x * a.os + px);
            } else {
                o = m;
            }
        }
        s_row[tid] = o;
    }
    __syncthreads();
    auto epilogue = [&](auto EPI_TAG) {
        constexpr int EPI = decltype(EPI_TAG)::value;
#pragma unroll
        for (int j = 0; j < NT; ++j) {
            const int col = n0 + wn * (BN / WN) + j * 32 + l31;
            const bool cok = col < a.Cout;
            const int colc = cok ? col : 0;
            const float bias = a.bias ? a.bias[colc] : 0.f;
            float sc = 1.f, sh = 0.f;
            if ((EPI == EPI_AFFINE_ACT || (EPI == EPI_GRU_OUT && a.out2)) && a.scale) { sc = a.scale[colc]; sh = a.shift[colc]; }
            long long liftoff = 0;
            if (EPI == EPI_AFFINE_ACT && a.lift > 0) {               // column (z, c), c fastest -> + z HW Cl + c
                const int Cl = a.Cout / a.lift, zc = colc / Cl;
                liftoff = (long long)zc * a.H * a.W * Cl + (colc - zc * Cl);
            }
#pragma unroll
            for (int i = 0; i < MT; ++i) {
#pragma unroll
                for (int r = 0; r < 16; ++r) {
                    const int rl = wm * (32 * MT) + i * 32 + (r & 3) + 8 * (r >> 2) + 4 * half;
                    const long long orow = s_row[rl];
                    float v = acc[i][j][r] + bias;
                    if (cok && orow >= 0) {
                        if constexpr (EPI == EPI_BIAS) {
                            outp[orow * a.ldo + col] = v;
                        } else if constexpr (EPI == EPI_AFFINE_ACT) {
                            v = fmaf(v, sc, sh);
                            if (a.lift > 0) {
                                if (a.residual) v += a.residual[(m0 + rl) * a.ldr + col];      // residual rows are the un-lifted GEMM rows
                                v = v > 0.f ? v : v * a.slope;
                                a.out[orow + liftoff] = v;
                            } else {
                                if (a.residual) v += a.residual[orow * a.ldr + col];
                                v = v > 0.f ? v : v * a.slope;
                                a.out[orow * a.ldo + col] = v;
                            }
                        } else if constexpr (EPI == EPI_GRU_GATES) {
                            if (a.residual) v += a.residual[orow * a.ldr + col];       // precomputed input half conv(x, W_x) (shared across fusions)
                            const float g = 1.f / (1.f + expf(-v));          // torch.sigmoid's formula with the full-precision exponential (models/fusion.py:31-32)
                            if (col < Ch) a.out[orow * Ch + col] = g;
                            else {
                                a.out2[orow * Ch + (col - Ch)] = a.aux_h[orow * Ch + (col - Ch)] * g;
                                if (a.out3) a.out3[orow * Ch + (col - Ch)] = g;
                            }
                        } else {   // EPI_GRU_OUT
                            if (a.residual) v += a.residual[orow * a.ldr + col];
                            const float cand = tanhf(v);
                            const float z = a.aux_z[orow * a.Cout + col], h = a.aux_h[orow * a.Cout + col];
                            const float hn = h * (1.f - z) + cand * z;
                            a.out[orow * a.ldo + col] = hn;
                            if (a.out2) a.out2[orow * a.ldo + col] = fmaf(hn, sc, sh);
                            if (a.out3) a.out3[orow * a.ldo + col] = cand;
                        }
                    }
                }
            }
            if constexpr (EPI == EPI_BIAS) {
                if (a.stats) {                                       // workgroup-uniform, off the plain path: a second walk over this column's accumulators
#pragma unroll
                    for (int i = 0; i < MT; ++i) {
                        double t1 = 0.0, t2 = 0.0;                   // this lane's share of column `col` in the 32-row block (rows beyond M excluded)
#pragma unroll
                        for (int r = 0; r < 16; ++r) {
                            const int rl = wm * (32 * MT) + i * 32 + (r & 3) + 8 * (r >> 2) + 4 * half;
                            if (s_row[rl] >= 0) {
                                const double v = (double)(acc[i][j][r] + bias);
                                t1 += v; t2 += v * v;
                            }
                        }
                        // the two half-waves hold the rows 4 (lane >> 5) + ... of the block: add them; lanes 0..31 then own one column each
                        t1 += __shfl_xor(t1, 32, 64); t2 += __shfl_xor(t2, 32, 64);
                        const long long blk = (m0 + wm * (32 * MT) + i * 32) >> 5;
                        if (half == 0 && cok) {
                            a.stats[(blk * 2) * a.Cout + col] = t1;
                            a.stats[(blk * 2 + 1) * a.Cout + col] = t2;
                        }
                    }
                }
            }
        }
    };
    if constexpr (RS) {                                             // row stage of A^T M A over the four points, two planes stored (identity rows, no bias)
        float* const o0 = a.out + pb * a.pto;                        // plane (i' = 0, j)
        float* const o1 = a.out + (pb + 4) * a.pto;                  // plane (i' = 1, j)
#pragma unroll
        for (int j = 0; j < NT; ++j) {
            const int col = n0 + wn * (BN / WN) + j * 32 + l31;
            if (col < a.Cout) {
#pragma unroll
                for (int r = 0; r < 16; ++r) {
                    const long long orow = s_row[wm * 32 + (r & 3) + 8 * (r >> 2) + 4 * half];
                    if (orow >= 0) {
                        const float m0 = accs[0][0][j][r], m1 = accs[1][0][j][r], m2 = accs[NP - 2][0][j][r], m3 = accs[NP - 1][0][j][r];
                        o0[orow * a.ldo + col] = (m0 + m1) + m2;
                        o1[orow * a.ldo + col] = (m1 - m2) - m3;
                    }
                }
            }
        }
        return;
    }
    if (a.ksplit > 1) {                                             // raw partial sums; conv_splitk_epilogue_kernel finishes the job
        float* wsl = a.ws + (long long)ks * M * a.Cout;
#pragma unroll
        for (int j = 0; j < NT; ++j) {
            const int col = n0 + wn * (BN / WN) + j * 32 + l31;
#pragma unroll
            for (int i = 0; i < MT; ++i)
#pragma unroll
                for (int r = 0; r < 16; ++r) {
                    const long long m = m0 + wm * (32 * MT) + i * 32 + (r & 3) + 8 * (r >> 2) + 4 * half;
                    if (col < a.Cout && m < M) wsl[m * a.Cout + col] = acc[i][j][r];
                }
        }
        return;
    }
    switch (a.epi) {
        case EPI_BIAS: epilogue(std::integral_constant<int, EPI_BIAS>{}); break;
        case EPI_AFFINE_ACT: epilogue(std::integral_constant<int, EPI_AFFINE_ACT>{}); break;
        case EPI_GRU_GATES: epilogue(std::integral_constant<int, EPI_GRU_GATES>{}); break;
        default: epilogue(std::integral_constant<int, EPI_GRU_OUT>{}); break;
    }
    FORGE_STAMP(3);
}



// Second half of a split-K launch: out = epilogue(sum_ks ws[ks][m][col]). One thread per 4 output channels; deterministic
// (fixed summation order), memory-bound (ksplit x M x Cout floats read once).
__global__ __launch_bounds__(256) void conv_splitk_epilogue_kernel(const ConvArgs a) {
    const long long M = (long long)a.n * a.D * a.H * a.W;
    const int C4 = a.Cout >> 2;
    const long long idx = (long long)blockIdx.x * 256 + threadIdx.x;
    if (idx >= M * C4) return;
    const long long m = idx / C4;
    const int col = (int)(idx - m * C4) << 2;
    float4 v = *reinterpret_cast<const float4*>(a.ws + m * a.Cout + col);
    for (int k = 1; k < a.ksplit; ++k) {
        const float4 p = *reinterpret_cast<const float4*>(a.ws + ((long long)k * M + m) * a.Cout + col);
        v.x += p.x; v.y += p.y; v.z += p.z; v.w += p.w;
    }
    long long orow = m;
    if ((a.os != 1) || (a.Do != a.D) || (a.Ho != a.H) || (a.Wo != a.W)) {
        long long q = m;
        const int x = (int)(q % a.W); q /= a.W;
        const int y = (int)(q % a.H); q /= a.H;
        const int z = (int)(q % a.D); q /= a.D;
        orow = ((q * a.Do + (z * a.os + a.pz)) * a.Ho + (y * a.os + a.py)) * a.Wo + (x * a.os + a.px);
    }
    float r[4] = {v.x, v.y, v.z, v.w};
#pragma unroll
    for (int c = 0; c < 4; ++c) {
        const int cc = col + c;
        float t = r[c] + (a.bias ? a.bias[cc] : 0.f);
        if (a.epi == EPI_AFFINE_ACT) {
            t = fmaf(t, a.scale[cc], a.shift[cc]);
            if (a.residual) t += a.residual[orow * a.ldr + cc];
            t = t > 0.f ? t : t * a.slope;
        }
        if (a.lift > 0) {
            const int Cl = a.Cout / a.lift, zc = cc / Cl, c2 = cc - zc * Cl;
            const long long HW = (long long)a.H * a.W, nn = m / HW, hw = m - nn * HW;
            a.out[((nn * a.lift + zc) * HW + hw) * Cl + c2] = t;
        } else {
            a.out[orow * a.ldo + cc] = t;
        }
    }
}

// ------------------------------------------------------------------------------------------------
// Narrow-N variant for Cout <= 16 (render-feature / density convs of the heads, conv_rgb): a 128-wide
// N tile would waste >= 75 % of the matrix-core work, so this one uses v_mfma_f32_16x16x4_f32
// (16 output channels per MFMA, same 64 FLOP/clk/SIMD), tile 256(M) x 16(N) x 16(K), 8 waves x 32 rows.
// LDS images are [row][16 k] (64-byte rows); chunk c of row r is stored at c ^ g[(r>>2)&3], g = {0,3,2,1},
// which makes the 16-lane groups of ds_read_b128 (rows r..r+15, chunk = lane>>4) conflict-free.
typedef __attribute__((ext_vector_type(4))) float f32x4;
constexpr int BM16 = 256, BK16 = 16;

__device__ __forceinline__ int lds_off16(int row, int chunk) {
    return row * BK16 + ((chunk ^ ((4 - ((row >> 2) & 3)) & 3)) << 2);
}

__global__ __launch_bounds__(NTHREADS) void conv_igemm_n16_kernel(const ConvArgs a) {
    constexpr int A_FLOATS = BM16 * BK16, B_FLOATS = 16 * BK16;
    __shared__ __attribute__((aligned(16))) float smem[2 * (A_FLOATS + B_FLOATS)];

    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const long long M = (long long)a.n * a.D * a.H * a.W;
    unsigned bid = xcd_remap(blockIdx.x, gridDim.x);
    int phase = 0;
    if (a.nphase > 1) {                                             // merged transposed-conv phases (see conv_igemm_kernel)
        const unsigned tiles = (unsigned)((M + BM16 - 1) / BM16);
        phase = (int)(bid / tiles);
        bid -= (unsigned)phase * tiles;
    }
    const int pz = a.nphase > 1 ? (a.nphase == 8 ? (phase >> 2) & 1 : 0) : a.pz;
    const int py = a.nphase > 1 ? (phase >> 1) & 1 : a.py, px = a.nphase > 1 ? phase & 1 : a.px;
    const int t_lo = phase * a.tpp;
    const long long m0 = (long long)bid * BM16;
    const int Cin = a.C1 + a.C2;
    const int kchunks = Cin / BK16;
    const int nsteps = a.tpp * kchunks;

    const forge_v4i32 w1 = make_rsrc_words(a.in1, a.span1), w2 = make_rsrc_words(a.in2 ? a.in2 : a.in1, a.in2 ? a.span2 : 0),
                      ww = make_rsrc_words(a.wp, (long long)a.ntaps * a.Cout * Cin * 4);
    // LDS-DMA staging (common.h): A chunk j of this thread is LDS bytes 16 tid + 8192 j of the stage, the weight chunk (wave 0) bytes 16 tid of the B image
    const unsigned lds_wave = lds_addr(smem) + (unsigned)__builtin_amdgcn_readfirstlane(wave) * 1024u;

    const int cp = tid & 3;                                      // physical chunk in the 64-byte LDS row
    int ar[2], az[2], ay[2], ax[2], an[2], asrc[2];
    bool aval[2];
#pragma unroll
    for (int j = 0; j < 2; ++j) {
        ar[j] = (tid >> 2) + 128 * j;
        const long long vl = m0 + ar[j];
        aval[j] = vl < M;
        unsigned v = aval[j] ? (unsigned)vl : 0u;
        unsigned q = v / (unsigned)a.W;
        ax[j] = (int)(v - q * (unsigned)a.W) * a.is; v = q;
        q = v / (unsigned)a.H;
        ay[j] = (int)(v - q * (unsigned)a.H) * a.is; v = q;
        q = v / (unsigned)a.D;
        az[j] = (int)(v - q * (unsigned)a.D) * a.is;
        an[j] = (int)q;
        asrc[j] = (cp ^ ((4 - ((ar[j] >> 2) & 3)) & 3)) << 2;
    }
    const int brow = tid >> 2;                                   // threads 0..63 stage the 16 x 16 weight tile
    const unsigned boff = (tid < 64 && brow < a.Cout) ? (unsigned)((brow * Cin + ((cp ^ ((4 - ((brow >> 2) & 3)) & 3)) << 2)) * 4) : OOB;

    int erow[2], erow2[2];
    auto prep_tap = [&](int t) {
        const int dz = a.tap[t][0], dy = a.tap[t][1], dx = a.tap[t][2];
#pragma unroll
        for (int j = 0; j < 2; ++j) {
            const int zi = az[j] + dz, yi = ay[j] + dy, xi = ax[j] + dx;
            const bool ok = aval[j] && (unsigned)zi < (unsigned)a.Di && (unsigned)yi < (unsigned)a.Hi && (unsigned)xi < (unsigned)a.Wi;
            const int sp = (zi * a.Hi + yi) * a.Wi + xi;
            erow[j] = ok ? an[j] * (int)a.bs1r + sp : -1;
            erow2[j] = ok ? an[j] * (int)a.bs2r + sp : -1;
        }
    };
    auto issue_step = [&](int t, int kc, int buf) {
        const int c0 = kc * BK16;
        const unsigned stage = lds_wave + (unsigned)buf * (unsigned)((A_FLOATS + B_FLOATS) * 4);
        if (c0 < a.C1) {
#pragma unroll
            for (int j = 0; j < 2; ++j) lds_dma16(w1, erow[j] < 0 ? OOB : (unsigned)((erow[j] * a.ld1 + c0 + asrc[j]) * 4), stage + (unsigned)(j * 8192));
        } else {
#pragma unroll
            for (int j = 0; j < 2; ++j) lds_dma16(w2, erow2[j] < 0 ? OOB : (unsigned)((erow2[j] * a.ld2 + (c0 - a.C1) + asrc[j]) * 4), stage + (unsigned)(j * 8192));
        }
        if (wave == 0) lds_dma16(ww, boff == OOB ? OOB : boff + (unsigned)((t * a.Cout * Cin + c0) * 4), stage + (unsigned)(A_FLOATS * 4));
    };

    f32x4 acc[2];
#pragma unroll
    for (int i = 0; i < 2; ++i)
#pragma unroll
        for (int r = 0; r < 4; ++r) acc[i][r] = 0.f;

    const int kq = lane >> 4, l15 = lane & 15;
    int t = t_lo, kc = 0;
    prep_tap(t);
    issue_step(t, 0, 0);
    lds_dma_wait();
    __syncthreads();
    for (int s = 0; s < nsteps; ++s) {
        const int buf = s & 1;
        if (s + 1 < nsteps) {
            if (++kc == kchunks) { kc = 0; ++t; prep_tap(t); }
            issue_step(t, kc, buf ^ 1);                            // in flight under this step's MFMAs
        }
        const float* sa = smem + buf * (A_FLOATS + B_FLOATS);
        const float* sb = sa + A_FLOATS;
        const float4 fb = *reinterpret_cast<const float4*>(sb + lds_off16(l15, kq));
        float4 fa[2];
#pragma unroll
        for (int i = 0; i < 2; ++i) fa[i] = *reinterpret_cast<const float4*>(sa + lds_off16(wave * 32 + i * 16 + l15, kq));
#pragma unroll
        for (int i = 0; i < 2; ++i) acc[i] = __builtin_amdgcn_mfma_f32_16x16x4f32(fa[i].x, fb.x, acc[i], 0, 0, 0);
#pragma unroll
        for (int i = 0; i < 2; ++i) acc[i] = __builtin_amdgcn_mfma_f32_16x16x4f32(fa[i].y, fb.y, acc[i], 0, 0, 0);
#pragma unroll
        for (int i = 0; i < 2; ++i) acc[i] = __builtin_amdgcn_mfma_f32_16x16x4f32(fa[i].z, fb.z, acc[i], 0, 0, 0);
#pragma unroll
        for (int i = 0; i < 2; ++i) acc[i] = __builtin_amdgcn_mfma_f32_16x16x4f32(fa[i].w, fb.w, acc[i], 0, 0, 0);
        lds_dma_wait();
        __syncthreads();
    }

    // ---- epilogue (bias / folded-BN + LeakyReLU + residual only). 16x16 C layout: col = lane & 15, row = (lane >> 4) * 4 + r
    // output rows through an LDS table (one 32-bit decomposition per tile row), as in conv_igemm_kernel
    long long* s_row = reinterpret_cast<long long*>(smem);          // [BM16]
    {
        const bool remap = (a.os != 1) || (a.Do != a.D) || (a.Ho != a.H) || (a.Wo != a.W);
        if (tid < BM16) {
            const long long m = m0 + tid;
            long long o = -1;
            if (m < M) {
                o = m;
                if (remap) {
                    unsigned q = (unsigned)m, t2 = q / (unsigned)a.W;
                    const int x = (int)(q - t2 * (unsigned)a.W); q = t2; t2 = q / (unsigned)a.H;
                    const int y = (int)(q - t2 * (unsigned)a.H); q = t2; t2 = q / (unsigned)a.D;
                    const int z = (int)(q - t2 * (unsigned)a.D);
                    o = (((long long)t2 * a.Do + (z * a.os + pz)) * a.Ho + (y * a.os + py)) * a.Wo + (x * a.os + px);
                }
            }
            s_row[tid] = o;
        }
        __syncthreads();
    }
    const int col = l15;
    if (col < a.Cout) {
        const float bias = a.bias ? a.bias[col] : 0.f;
        float sc = 1.f, sh = 0.f;
        if (a.epi == EPI_AFFINE_ACT) { sc = a.scale[col]; sh = a.shift[col]; }
#pragma unroll
        for (int i = 0; i < 2; ++i)
#pragma unroll
            for (int r = 0; r < 4; ++r) {
                const long long orow = s_row[wave * 32 + i * 16 + kq * 4 + r];
                if (orow >= 0) {
                    float v = acc[i][r] + bias;
                    if (a.epi == EPI_AFFINE_ACT) {
                        v = fmaf(v, sc, sh);
                        if (a.residual) v += a.residual[orow * a.ldr + col];
                        v = v > 0.f ? v : v * a.slope;
                    }
                    a.out[orow * a.ldo + col] = v;
                }
            }
    }
}


// Narrow-N kernel for stride-1 convolutions whose taps form complete x-LINES: taps [(dz, dy)] x [dx = -R .. R] (the 3x3x3 second convolutions of the
// heads, R = 1; conv_rgb's 5x5, R = 2). conv_igemm_n16_kernel stages the 256 tile rows once per TAP: at 16 output columns a K-step is 8 MFMAs per wave
// against a 16 KB global -> LDS round trip, so every one of its 54 (heads) / 50 (conv_rgb) K-steps is load-latency bound. The dx taps of one
// (dz, dy) line read the SAME rows shifted by dx voxels = dx rows of the channels-last tensor, so here a K-step stages the slab of 256 + 2R rows ONCE
// (LDS-DMA, common.h) and runs all 2R + 1 taps on shifted views of it: (2R + 1) x fewer global loads, barriers and K-steps, (2R + 1) x the MFMA
// work per barrier. A shifted row that crosses the end of its x-line is another line's voxel: masked per (row, dx) at use. Rows whose (z + dz, y + dy)
// leave the grid are staged as zeros. Same sums as conv_igemm_n16_kernel in a different order (taps of a line innermost).
template <int R>
__global__ __launch_bounds__(NTHREADS) void conv_igemm_n16_lines_kernel(const ConvArgs a) {
    constexpr int NT = 2 * R + 1, SR = BM16 + 2 * R;               // taps per line, slab rows
    constexpr int A_FLOATS = ((SR * BK16 + 255) / 256) * 256, B_FLOATS = NT * 16 * BK16;     // A padded to whole 1 KB wave blocks
    constexpr int APASS = (SR * 4 + NTHREADS - 1) / NTHREADS;      // 16-byte chunks of the slab per thread (3)
    __shared__ __attribute__((aligned(16))) float smem[2 * (A_FLOATS + B_FLOATS)];

    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const long long M = (long long)a.n * a.D * a.H * a.W;
    const unsigned bid = xcd_remap(blockIdx.x, gridDim.x);
    const long long m0 = (long long)bid * BM16;
    const int Cin = a.C1 + a.C2;
    const int kchunks = Cin / BK16;
    const int nlines = a.ntaps / NT;
    const int nsteps = nlines * kchunks;
    const forge_v4i32 w1 = make_rsrc_words(a.in1, a.span1), w2 = make_rsrc_words(a.in2 ? a.in2 : a.in1, a.in2 ? a.span2 : 0),
                      ww = make_rsrc_words(a.wp, (long long)a.ntaps * a.Cout * Cin * 4);
    const unsigned lds_wave = lds_addr(smem) + (unsigned)__builtin_amdgcn_readfirstlane(wave) * 1024u;

    // slab chunk q = tid + 512 j: slab row q >> 2 = the voxel m0 - R + (q >> 2), physical 16-byte chunk q & 3 (LDS byte 16 q: lane-linear)
    const int cp = tid & 3;
    int sz[APASS], sy[APASS], sn[APASS], sx[APASS], asrc[APASS];
    bool sval[APASS];
#pragma unroll
    for (int j = 0; j < APASS; ++j) {
        const int srow = (tid + NTHREADS * j) >> 2;
        const long long vl = m0 - R + srow;
        sval[j] = srow < SR && vl >= 0 && vl < M;
        unsigned v = sval[j] ? (unsigned)vl : 0u;
        unsigned q = v / (unsigned)a.W;
        sx[j] = (int)(v - q * (unsigned)a.W); v = q;
        q = v / (unsigned)a.H;
        sy[j] = (int)(v - q * (unsigned)a.H); v = q;
        q = v / (unsigned)a.D;
        sz[j] = (int)(v - q * (unsigned)a.D);
        sn[j] = (int)q;
        asrc[j] = (cp ^ ((4 - ((srow >> 2) & 3)) & 3)) << 2;
    }
    // weights of a line: NT x 16 rows (tap-in-line, cout) x 4 chunks, staged by the first NT waves; LDS byte 16 tid of the B image
    const int brow = tid >> 2;
    const unsigned boff = (wave < NT && (brow & 15) < a.Cout)
                              ? (unsigned)((((brow >> 4) * a.Cout + (brow & 15)) * Cin + ((cp ^ ((4 - ((brow >> 2) & 3)) & 3)) << 2)) * 4) : OOB;
    int line = 0, kc = 0;
    auto issue_step = [&](int buf) {
        const int dz = a.tap[line * NT][0], dy = a.tap[line * NT][1];
        const int c0 = kc * BK16;
        const unsigned stage = lds_wave + (unsigned)buf * (unsigned)((A_FLOATS + B_FLOATS) * 4);
#pragma unroll
        for (int j = 0; j < APASS; ++j) {
            if ((tid + NTHREADS * j) < A_FLOATS / 4) {                 // whole waves in or out except in the last pass (exec-masked lanes do not write)
                const int zi = sz[j] + dz, yi = sy[j] + dy;
                const bool ok = sval[j] && (unsigned)zi < (unsigned)a.D && (unsigned)yi < (unsigned)a.H;
                const int sp = (zi * a.H + yi) * a.W + sx[j];
                unsigned off = OOB;
                if (c0 < a.C1) { if (ok) off = (unsigned)(((sn[j] * (int)a.bs1r + sp) * a.ld1 + c0 + asrc[j]) * 4); lds_dma16(w1, off, stage + (unsigned)(j * 8 * 1024)); }
                else { if (ok) off = (unsigned)(((sn[j] * (int)a.bs2r + sp) * a.ld2 + (c0 - a.C1) + asrc[j]) * 4); lds_dma16(w2, off, stage + (unsigned)(j * 8 * 1024)); }
            }
        }
        if (wave < NT) lds_dma16(ww, boff == OOB ? OOB : boff + (unsigned)((line * NT * a.Cout * Cin + c0) * 4), stage + (unsigned)(A_FLOATS * 4));
    };
    auto advance = [&]() { if (++kc == kchunks) { kc = 0; ++line; } };

    f32x4 acc[2];
#pragma unroll
    for (int i = 0; i < 2; ++i)
#pragma unroll
        for (int r = 0; r < 4; ++r) acc[i][r] = 0.f;
    const int kq = lane >> 4, l15 = lane & 15;
    int xr[2];                                                     // x coordinate of this lane's two MFMA rows: shifted reads beyond the line end are masked
#pragma unroll
    for (int i = 0; i < 2; ++i) {
        const long long m = m0 + wave * 32 + i * 16 + l15;
        xr[i] = (int)((unsigned)(m < M ? m : 0) % (unsigned)a.W);
    }
    issue_step(0);
    lds_dma_wait();
    __syncthreads();
    for (int s = 0; s < nsteps; ++s) {
        const int buf = s & 1;
        if (s + 1 < nsteps) { advance(); issue_step(buf ^ 1); }
        const float* sa = smem + buf * (A_FLOATS + B_FLOATS);
        const float* sb = sa + A_FLOATS;
#pragma unroll
        for (int d = 0; d < NT; ++d) {
            const float4 fb = *reinterpret_cast<const float4*>(sb + lds_off16(d * 16 + l15, kq));
            float4 fa[2];
#pragma unroll
            for (int i = 0; i < 2; ++i) {
                fa[i] = *reinterpret_cast<const float4*>(sa + lds_off16(wave * 32 + i * 16 + l15 + d, kq));     // slab row = tile row + R + dx, d = dx + R
                if ((unsigned)(xr[i] + d - R) >= (unsigned)a.W) fa[i] = make_float4(0.f, 0.f, 0.f, 0.f);
            }
#pragma unroll
            for (int i = 0; i < 2; ++i) acc[i] = __builtin_amdgcn_mfma_f32_16x16x4f32(fa[i].x, fb.x, acc[i], 0, 0, 0);
#pragma unroll
            for (int i = 0; i < 2; ++i) acc[i] = __builtin_amdgcn_mfma_f32_16x16x4f32(fa[i].y, fb.y, acc[i], 0, 0, 0);
#pragma unroll
            for (int i = 0; i < 2; ++i) acc[i] = __builtin_amdgcn_mfma_f32_16x16x4f32(fa[i].z, fb.z, acc[i], 0, 0, 0);
#pragma unroll
            for (int i = 0; i < 2; ++i) acc[i] = __builtin_amdgcn_mfma_f32_16x16x4f32(fa[i].w, fb.w, acc[i], 0, 0, 0);
        }
        lds_dma_wait();
        __syncthreads();
    }
    // ---- epilogue (bias / folded-BN + LeakyReLU + residual): identity row mapping (stride 1, same grid). 16x16 C layout: col = lane & 15, row = (lane >> 4) * 4 + r
    const int col = l15;
    if (col < a.Cout) {
        const float bias = a.bias ? a.bias[col] : 0.f;
        float sc = 1.f, sh = 0.f;
        if (a.epi == EPI_AFFINE_ACT) { sc = a.scale[col]; sh = a.shift[col]; }
#pragma unroll
        for (int i = 0; i < 2; ++i)
#pragma unroll
            for (int r = 0; r < 4; ++r) {
                const long long orow = m0 + wave * 32 + i * 16 + kq * 4 + r;
                if (orow < M) {
                    float v = acc[i][r] + bias;
                    if (a.epi == EPI_AFFINE_ACT) {
                        v = fmaf(v, sc, sh);
                        if (a.residual) v += a.residual[orow * a.ldr + col];
                        v = v > 0.f ? v : v * a.slope;
                    }
                    a.out[orow * a.ldo + col] = v;
                }
            }
    }
}

// ------------------------------------------------------------------------------------------------
// Launch plan: tile shape and split-K factor from a makespan model of the 256-CU chip (times in us).
//   tiles      A 128x128 / 8 waves   B 64x128 / 8   C 128x64 / 8   D 64x64 / 4   E 128x32 / 4 (Cout <= 32)
//   tile_us    = BM x BN x K-steps / (8000 x eff): CU-time of one tile with the CU's resident workgroups saturating the MFMA
//                pipes; eff = measured rate of each tile on M = 131072 problems relative to tile A (130 TF)
//   makespan   = full waves of 256 x occupancy workgroups, then the remainder with r = ceil(rem / 256) workgroups per CU running at
//                g(r)/g(occ) of the saturated rate (a lone workgroup cannot overlap its own barriers: g(1) = 0.8); floored by the
//                operand/result traffic at 4 TB/s (short-K 1x1 convs are traffic-bound; narrow N tiles re-read A); plus a
//                prologue/epilogue term per round, larger for tile A whose 2 resident workgroups overlap it worst
//   split-K    slices K across workgroups when the chip is under-filled (M = 5120: ResNet, one scene); costs the reduction kernel
// Constants fitted (tools/fit_plan_model.py) on tools/conv_plan_sweep.py (54 shapes x 30 plans, 1 and 4 scenes; re-fitted in round 3 for the
// LDS-DMA loop, which moved the 64x128 tile from 0.98 to 1.06 of tile A): chosen plans are within 0.5 % / 0.1 % of the per-shape best in total. Callers may pass an explicit (tile, ksplit) to forge_conv_igemm instead (tools/conv_plan_sweep.py, tests).
struct ConvPlan { char tile; int ksplit; };

static ConvPlan plan_conv(long long M, int Cout, int Cin, int ntaps, bool can_split, long long ws_bytes) {
    struct Tile { char id; int bm, bn, occ; double eff, ov; };
    static const Tile tiles[5] = {{'A', 128, 128, 2, 1.00, 1.0}, {'B', 64, 128, 3, 1.06, 0.5}, {'C', 128, 64, 3, 0.93, 0.5},
                                  {'D', 64, 64, 5, 0.90, 1.0}, {'E', 128, 32, 4, 0.90, 0.5}};
    static const int splits[6] = {1, 2, 3, 4, 6, 8};
    auto g = [](long long o) { return o <= 1 ? 0.8 : o == 2 ? 0.85 : o == 3 ? 0.96 : 1.0; };
    const int nsteps = ntaps * (Cin / BK);
    const double K = (double)ntaps * Cin;
    ConvPlan best{'D', 1};
    double best_us = 1e300;
    for (const Tile& t : tiles) {
        if (t.id == 'E' && Cout > 32) continue;
        const long long ntn = (Cout + t.bn - 1) / t.bn;
        const long long nb = ((M + t.bm - 1) / t.bm) * ntn;
        for (int k : splits) {
            if (k > 1 && (!can_split || nsteps / k < 8 || (long long)k * M * Cout * 4 > ws_bytes)) continue;
            const long long wgs = nb * k, slots = 256LL * t.occ;
            const long long full = wgs / slots, rem = wgs - full * slots;
            const double tile_us = (double)t.bm * t.bn * ((nsteps + k - 1) / k) / (8000.0 * t.eff);
            double us = (double)full * t.occ * tile_us;
            long long rounds = full;
            if (rem > 0) {
                const long long r = (rem + 255) / 256;
                us += (double)r * tile_us * g(t.occ) / g(r);
                ++rounds;
            }
            const double traffic_us = ((double)M * K * 4.0 * ntn + (double)Cout * K * 4.0 + (double)M * Cout * 4.0) / 4.0e6;
            if (us < traffic_us) us = traffic_us;
            us += (double)(rounds + 1) * t.ov * sqrt((double)t.bm * t.bn / 16384.0);
            if (k > 1) us += 3.0 + (double)(k + 1) * M * Cout * 4.0 / 2.0e6;
            if (us < best_us) { best_us = us; best = ConvPlan{t.id, k}; }
        }
    }
    return best;
}

// Launch conv_igemm_kernel with the planned tile: ceil(M / BM) x ceil(Cout / BN) workgroups per (K slice, phase, batched problem).
static int launch_conv_tile(const ConvArgs& a, char tile, hipStream_t st, bool row_stage = false) {
    const long long M = (long long)a.n * a.D * a.H * a.W;
    auto nblk = [&](int bm, int bn) { return ((M + bm - 1) / bm) * ((a.Cout + bn - 1) / bn); };
    if (row_stage) {                                                 // forge_wino_gemm_half: a.nbat = 4 point columns, four points per workgroup
        const long long grid = nblk(64, 128) * a.nbat;
        FORGE_REQUIRE(grid < (1ll << 31), FORGE_ESHAPE, "forge_wino_gemm_half: grid too large");
        const size_t lds = 2 * (64 * BK + 128 * BK) * sizeof(float);
        FORGE_SET_MAX_LDS_ONCE((conv_igemm_kernel<64, 128, 8, 1, 1>), lds);
        hipLaunchKernelGGL((conv_igemm_kernel<64, 128, 8, 1, 1>), dim3((unsigned)grid), dim3(8 * 64), lds, st, a);
        return 0;
    }
#define FORGE_LAUNCH_CONV(BMv, BNv, NWv)                                                                                   \
    do {                                                                                                                   \
        const long long grid = nblk(BMv, BNv) * a.ksplit * a.nphase * a.nbat;                                              \
        FORGE_REQUIRE(grid < (1ll << 31), FORGE_ESHAPE, "forge_conv_igemm: grid too large");                                \
        const size_t lds = 2 * (BMv * BK + BNv * BK) * sizeof(float);                                                      \
        FORGE_SET_MAX_LDS_ONCE((conv_igemm_kernel<BMv, BNv, NWv>), lds);                                                   \
        hipLaunchKernelGGL((conv_igemm_kernel<BMv, BNv, NWv>), dim3((unsigned)grid), dim3(NWv * 64), lds, st, a);          \
    } while (0)
    switch (tile) {
        case 'A': FORGE_LAUNCH_CONV(128, 128, 8); break;
        case 'B': FORGE_LAUNCH_CONV(64, 128, 8); break;
        case 'C': FORGE_LAUNCH_CONV(128, 64, 8); break;
        case 'E': FORGE_LAUNCH_CONV(128, 32, 4); break;
        default: FORGE_LAUNCH_CONV(64, 64, 4); break;
    }
#undef FORGE_LAUNCH_CONV
    return 0;
}

}  // namespace forge

using namespace forge;

extern "C" int forge_conv_igemm_plan(long long M, int Cout, int Cin, int ntaps, int nphase, int epilogue, int ldo, long long splitk_ws_bytes,
                                     int* tile, int* ksplit) {
    FORGE_REQUIRE(tile && ksplit && M > 0 && Cout > 0 && Cin > 0 && ntaps > 0 && (nphase == 1 || nphase == 4 || nphase == 8) && ntaps % nphase == 0,
                  FORGE_EINVAL, "forge_conv_igemm_plan: bad argument");
    if (Cout <= 16) { *tile = 'N'; *ksplit = 1; return 0; }                         // conv_igemm_n16_kernel
    const ConvPlan pl = plan_conv(M * nphase, Cout, Cin, ntaps / nphase,
                                  nphase == 1 && splitk_ws_bytes > 0 && (epilogue == EPI_BIAS || epilogue == EPI_AFFINE_ACT) && Cout % 4 == 0 && ldo % 4 == 0,
                                  splitk_ws_bytes);
    *tile = pl.tile; *ksplit = pl.ksplit;
    return 0;
}

// Generic entry; see include/forge_hip.h for the contract.
extern "C" int forge_conv_igemm(const float* in1, int C1, int ld1, long long bs1, const float* in2, int C2, int ld2, long long bs2,
                                const float* wp,
                                const float* bias, const float* scale, const float* shift, float slope, const float* residual,
                                const float* aux_h, const float* aux_z, float* out, float* out2, float* out3,
                                int n, int D, int H, int W, int is, int Di, int Hi, int Wi, int Cout, int ldo,
                                const int* taps, int ntaps, int os, int pz, int py, int px, int Do, int Ho, int Wo,
                                int epilogue, int lift, int tile, int ksplit, float* splitk_ws, long long splitk_ws_bytes, double* stats,
                                forge_stream_t stream) {
    FORGE_REQUIRE(in1 && wp && out && taps, FORGE_EINVAL, "forge_conv_igemm: null pointer argument");
    FORGE_REQUIRE(stats == nullptr || (epilogue == EPI_BIAS && ksplit <= 1 && Cout > 16 && !(pz < 0) && lift == 0), FORGE_EINVAL,
                  "forge_conv_igemm: output statistics need the plain bias epilogue of the wide kernel without split-K, phases or lift");
    FORGE_REQUIRE(n > 0 && D > 0 && H > 0 && W > 0 && Cout > 0 && ntaps > 0 && ntaps <= MAX_TAPS, FORGE_EINVAL,
                  "forge_conv_igemm: bad dims n=%d D=%d H=%d W=%d Cout=%d ntaps=%d", n, D, H, W, Cout, ntaps);
    const int kstep = Cout <= 16 ? 16 : 32;
    FORGE_REQUIRE(C1 > 0 && C1 % kstep == 0 && C2 >= 0 && C2 % kstep == 0, FORGE_ESHAPE,
                  "forge_conv_igemm: C1=%d / C2=%d must be multiples of the K-step %d (16 for Cout <= 16, else 32)", C1, C2, kstep);
    FORGE_REQUIRE((C2 == 0) == (in2 == nullptr), FORGE_EINVAL, "forge_conv_igemm: in2/C2 mismatch");
    FORGE_REQUIRE(epilogue >= 0 && epilogue <= 3, FORGE_EINVAL, "forge_conv_igemm: unknown epilogue %d", epilogue);
    FORGE_REQUIRE(epilogue != EPI_AFFINE_ACT || (scale && shift), FORGE_EINVAL, "forge_conv_igemm: affine epilogue needs scale/shift");
    FORGE_REQUIRE(epilogue != EPI_GRU_GATES || (aux_h && out2 && Cout % 2 == 0), FORGE_EINVAL, "forge_conv_igemm: GRU gate epilogue needs aux_h, out2");
    FORGE_REQUIRE(epilogue != EPI_GRU_OUT || (aux_h && aux_z), FORGE_EINVAL, "forge_conv_igemm: GRU out epilogue needs aux_h, aux_z");
    FORGE_REQUIRE(os >= 1 && Do > 0 && Ho > 0 && Wo > 0, FORGE_EINVAL, "forge_conv_igemm: bad output mapping");
    FORGE_REQUIRE(is >= 1 && Di > 0 && Hi > 0 && Wi > 0, FORGE_EINVAL, "forge_conv_igemm: bad input mapping");
    FORGE_REQUIRE(lift == 0 || (lift > 0 && epilogue == EPI_AFFINE_ACT && Cout % lift == 0 && D == 1 && os == 1 && Cout > 64), FORGE_EINVAL,
                  "forge_conv_igemm: lift needs the affine epilogue on a 2-D (D=1) conv with Cout %% lift == 0 and Cout > 64");
    FORGE_REQUIRE(ld1 >= C1 && ld1 % 4 == 0 && (C2 == 0 || (ld2 >= C2 && ld2 % 4 == 0)) && ldo >= 1, FORGE_EINVAL,
                  "forge_conv_igemm: row strides must cover the channels and keep 16-byte alignment");
    ConvArgs a;
    a.in1 = in1; a.in2 = in2; a.C1 = C1; a.C2 = C2; a.ld1 = ld1; a.ld2 = ld2; a.bs1r = bs1 > 0 ? bs1 : (long long)Di * Hi * Wi; a.bs2r = bs2 > 0 ? bs2 : (long long)Di * Hi * Wi;
    a.span1 = ((long long)(n - 1) * a.bs1r + (long long)Di * Hi * Wi) * ld1 * 4;
    a.span2 = in2 ? ((long long)(n - 1) * a.bs2r + (long long)Di * Hi * Wi) * ld2 * 4 : 0;
    FORGE_REQUIRE(a.span1 < (1ll << 31) && a.span2 < (1ll << 31) && (long long)ntaps * Cout * (C1 + C2) * 4 < (1ll << 31), FORGE_ESHAPE,
                  "forge_conv_igemm: an operand spans >= 2 GiB (32-bit buffer offsets); split the batch"); a.is = is; a.Di = Di; a.Hi = Hi; a.Wi = Wi; a.residual = residual; a.ldr = (lift > 0 || epilogue == EPI_GRU_GATES || epilogue == EPI_GRU_OUT) ? Cout : ldo; a.lift = lift; a.ws = nullptr; a.ksplit = 1; a.stats = stats; a.wp = wp; a.bias = bias; a.scale = scale; a.shift = shift; a.slope = slope;
    a.aux_h = aux_h; a.aux_z = aux_z; a.out = out; a.out2 = out2; a.out3 = out3; a.n = n; a.D = D; a.H = H; a.W = W; a.Cout = Cout; a.ldo = ldo;
    a.ntaps = ntaps; a.os = os; a.pz = pz; a.py = py; a.px = px; a.Do = Do; a.Ho = Ho; a.Wo = Wo; a.epi = epilogue;
    a.nphase = 1; a.tpp = ntaps; a.nbat = 1; a.pt1 = a.pt2 = a.ptw = a.pto = 0;
    if (pz < 0) {   // all output phases of a stride-2 transposed convolution in one launch
        FORGE_REQUIRE(os == 2 && py < 0 && px < 0 && Ho == 2 * H && Wo == 2 * W && (Do == 2 * D || Do == D), FORGE_EINVAL,
                      "forge_conv_igemm: merged phases (pz = py = px = -1) need os = 2 and a doubled output grid");
        a.nphase = Do == 2 * D ? 8 : 4;
        FORGE_REQUIRE(ntaps % a.nphase == 0, FORGE_EINVAL, "forge_conv_igemm: merged phases need ntaps %% %d == 0", a.nphase);
        a.tpp = ntaps / a.nphase;
        a.pz = a.py = a.px = 0;
    }
    for (int t = 0; t < MAX_TAPS; ++t) {
        for (int k = 0; k < 3; ++k) a.tap[t][k] = (signed char)(t < ntaps ? taps[t * 3 + k] : 0);
        a.tap[t][3] = 0;
    }
    const long long M = (long long)n * D * H * W;
    FORGE_REQUIRE(M < (1ll << 31), FORGE_ESHAPE, "forge_conv_igemm: more than 2^31 GEMM rows; split the batch");
    FORGE_REQUIRE(out3 == nullptr || epilogue == EPI_GRU_GATES || epilogue == EPI_GRU_OUT, FORGE_EINVAL, "forge_conv_igemm: out3 is a GRU-epilogue output");
    hipStream_t st = (hipStream_t)stream;
    if (Cout <= 16) {
        FORGE_REQUIRE(epilogue == EPI_BIAS || epilogue == EPI_AFFINE_ACT, FORGE_EINVAL, "forge_conv_igemm: GRU epilogues need Cout > 16");
        const long long grid = (M + BM16 - 1) / BM16 * a.nphase;
        FORGE_REQUIRE(grid < (1ll << 31), FORGE_ESHAPE, "forge_conv_igemm: grid too large");
        // stride-1 convolution on its own grid whose taps are complete x-lines [(dz, dy)] x [dx = -R .. R], dx fastest (R = 1, 2): the lines kernel
        int lines_r = 0;
        if (a.nphase == 1 && is == 1 && os == 1 && lift == 0 && Do == D && Ho == H && Wo == W && Di == D && Hi == H && Wi == W && pz == 0 && py == 0 && px == 0) {
            for (int r = 1; r <= 2 && !lines_r; ++r) {
                const int nt = 2 * r + 1;
                bool ok = ntaps % nt == 0 && W > r;
                for (int t = 0; ok && t < ntaps; ++t)
                    ok = taps[t * 3 + 2] == (t % nt) - r && taps[t * 3] == taps[(t - t % nt) * 3] && taps[t * 3 + 1] == taps[(t - t % nt) * 3 + 1];
                if (ok) lines_r = r;
            }
        }
        if (lines_r == 1) hipLaunchKernelGGL(conv_igemm_n16_lines_kernel<1>, dim3((unsigned)grid), dim3(NTHREADS), 0, st, a);
        else if (lines_r == 2) hipLaunchKernelGGL(conv_igemm_n16_lines_kernel<2>, dim3((unsigned)grid), dim3(NTHREADS), 0, st, a);
        else hipLaunchKernelGGL(conv_igemm_n16_kernel, dim3((unsigned)grid), dim3(NTHREADS), 0, st, a);
    } else {
        const bool can_split = a.nphase == 1 && splitk_ws && (epilogue == EPI_BIAS || epilogue == EPI_AFFINE_ACT) && Cout % 4 == 0 && ldo % 4 == 0;
        ConvPlan pl;
        if (tile == 0) {
            pl = plan_conv(M * a.nphase, Cout, C1 + C2, a.tpp, can_split, splitk_ws_bytes);
        } else {                                                       // the caller's plan (forge_conv_igemm_plan's answer, or a sweep / test override)
            pl = ConvPlan{(char)tile, ksplit > 1 ? ksplit : 1};
            FORGE_REQUIRE(tile >= 'A' && tile <= 'E', FORGE_EINVAL, "forge_conv_igemm: tile '%c' is not one of A..E (0 = planned here)", (char)tile);
            FORGE_REQUIRE(pl.ksplit == 1 || (can_split && pl.ksplit <= 8 && (long long)pl.ksplit * M * Cout * 4 <= splitk_ws_bytes &&
                                             pl.ksplit <= a.tpp * ((C1 + C2) / BK)), FORGE_EINVAL,
                          "forge_conv_igemm: ksplit=%d needs epilogue 0/1 without phases, Cout, ldo %% 4 == 0 and a workspace of ksplit M Cout floats", pl.ksplit);
        }
        // the statistics by-product is written by the GEMM epilogue in 32-row blocks of the PLANNED tile: a caller that asks for it must know the
        // tile (to size `stats` and to tell forge_bn_train_fwd how many blocks to read) and must not land on a split-K plan, whose launches skip
        // the epilogue - so it has to hand the plan over (forge_conv_igemm_plan) instead of leaving it to this call
        FORGE_REQUIRE(stats == nullptr || (tile != 0 && pl.ksplit == 1), FORGE_EINVAL,
                      "forge_conv_igemm: output statistics need an explicit tile ('A'..'E' from forge_conv_igemm_plan) and a plan without split-K");
        if (pl.ksplit > 1) { a.ksplit = pl.ksplit; a.ws = splitk_ws; }
        if (int rc = launch_conv_tile(a, pl.tile, st)) return rc;
        if (a.ksplit > 1) {
            const long long total = M * (Cout / 4);
            hipLaunchKernelGGL(conv_splitk_epilogue_kernel, dim3((unsigned)((total + 255) / 256)), dim3(256), 0, st, a);
        }
    }
    FORGE_LAUNCH_CHECK("forge_conv_igemm");
    return 0;
}

// Workgroup tile of forge_wino_gemm for R tile rows per point and K = kd Cin. The makespan model (plan_conv) was fitted on single-problem
// launches; for the 16 short-K problems of one launch the measured optimum (tools/wino_gemm_sweep.py at 1 and 4 scenes, round 3, LDS-DMA loop)
// is the 64x128 tile on every shape of the step (gates 129 TF against 127 / 122 for 128x128 / 64x64; state 128 / 125 / 121; conv1 102 / 98 / 93);
// the 2-D trunk's tiny launches want many small workgroups. forge_wino_gemm's `tile` argument overrides it (that tool).
extern "C" int forge_wino_gemm_tile(long long R, int Cout, int Cin) {
    (void)Cin;
    if (Cout <= 64) return R < 2048 ? 'D' : 'C';               // a 64-wide output (conv1's data gradient, 128 -> 64) would leave half of a 64x128 tile's columns empty
    return R < 2048 ? 'D' : 'B';                                // R < 2048: the 2-D trunk's layer3/4 (R = 320 / 80)
}

// The 16 point-GEMMs of a Winograd F(2x2, 3x3) x 3-depth-tap convolution (winograd.hip) in ONE launch: problem p = (i, j) multiplies
// the transformed inputs V[p] (rows = (n, z, tile row, tile col), channels-last, the channel concatenation of V1 and V2) with the
// transformed weights U[p] [kd depth taps][Cout][C1 + C2] into Mm[p] [rows][Cout] - a kd-tap implicit GEMM over the tile grid, K = kd (C1 + C2);
// kd = 3 for the 3x3x3 convolutions, kd = 1 for the 3x3 convolutions of a 2-D network (D = 1 or D = images: planes do not mix).
static int wino_gemm_impl(const float* V1, int C1, int ld1, long long bs1, long long pt1, const float* V2, int C2, int ld2, long long bs2,
                          long long pt2, const float* U, float* Mm, int n, int D, int Ht, int Wt, int Cout, int kd, int tile, bool half, forge_stream_t stream) {
    FORGE_REQUIRE(tile == 0 || (tile >= 'A' && tile <= 'E'), FORGE_EINVAL, "forge_wino_gemm: tile must be 0 (default rule) or 'A'..'E'");
    FORGE_REQUIRE(V1 && U && Mm && (kd == 1 || kd == 3), FORGE_EINVAL, "forge_wino_gemm: null pointer argument / kd not 1 or 3");
    FORGE_REQUIRE(n > 0 && D > 0 && Ht > 0 && Wt > 0 && Cout > 16, FORGE_EINVAL, "forge_wino_gemm: bad dims n=%d D=%d Ht=%d Wt=%d Cout=%d (Cout > 16)", n,
                  D, Ht, Wt, Cout);
    FORGE_REQUIRE(C1 > 0 && C1 % BK == 0 && C2 >= 0 && C2 % BK == 0 && (C2 == 0) == (V2 == nullptr), FORGE_ESHAPE,
                  "forge_wino_gemm: C1=%d / C2=%d must be multiples of %d, V2 given iff C2 > 0", C1, C2, BK);
    FORGE_REQUIRE(ld1 >= C1 && ld1 % 4 == 0 && (C2 == 0 || (ld2 >= C2 && ld2 % 4 == 0)), FORGE_EINVAL, "forge_wino_gemm: bad row strides");
    ConvArgs a;
    memset(&a, 0, sizeof(a));
    const long long vol = (long long)D * Ht * Wt, R = (long long)n * vol;
    a.in1 = V1; a.in2 = V2; a.C1 = C1; a.C2 = C2; a.ld1 = ld1; a.ld2 = ld2; a.bs1r = bs1 > 0 ? bs1 : vol; a.bs2r = bs2 > 0 ? bs2 : vol;
    a.span1 = ((long long)(n - 1) * a.bs1r + vol) * ld1 * 4;
    a.span2 = V2 ? ((long long)(n - 1) * a.bs2r + vol) * ld2 * 4 : 0;
    FORGE_REQUIRE(a.span1 < (1ll << 31) && a.span2 < (1ll << 31) && R < (1ll << 31), FORGE_ESHAPE,
                  "forge_wino_gemm: an operand spans >= 2 GiB per Winograd point (32-bit buffer offsets); split the batch");
    a.wp = U; a.slope = 1.f; a.out = Mm; a.n = n; a.D = D; a.H = Ht; a.W = Wt; a.is = 1; a.Di = D; a.Hi = Ht; a.Wi = Wt;
    a.Cout = Cout; a.ldo = Cout; a.ldr = Cout; a.ntaps = kd; a.os = 1; a.Do = D; a.Ho = Ht; a.Wo = Wt; a.nphase = 1; a.tpp = kd; a.epi = EPI_BIAS;
    a.ksplit = 1; a.nbat = half ? 4 : 16; a.pt1 = pt1; a.pt2 = pt2; a.ptw = (long long)kd * Cout * (C1 + C2); a.pto = R * Cout;
    if (kd == 3) { a.tap[0][0] = -1; a.tap[2][0] = 1; }                 // depth taps (-1,0,0), (0,0,0), (1,0,0); kd = 1: the 2-D convolution's single tap
    if (int rc = launch_conv_tile(a, (char)(tile ? tile : forge_wino_gemm_tile(R, Cout, C1 + C2)), (hipStream_t)stream, half)) return rc;
    FORGE_LAUNCH_CHECK("forge_wino_gemm");
    return 0;
}

extern "C" int forge_wino_gemm(const float* V1, int C1, int ld1, long long bs1, long long pt1, const float* V2, int C2, int ld2, long long bs2,
                               long long pt2, const float* U, float* Mm, int n, int D, int Ht, int Wt, int Cout, int kd, int tile, forge_stream_t stream) {
    return wino_gemm_impl(V1, C1, ld1, bs1, pt1, V2, C2, ld2, bs2, pt2, U, Mm, n, D, Ht, Wt, Cout, kd, tile, false, stream);
}

// forge_wino_gemm with the ROW stage of the inverse transform applied in the GEMM's epilogue (conv_igemm_kernel<..., RS = 1>): Mm8 [2][4][R][Cout],
// Mm8[i'][j] = sum over the four points (i, j) of A^T[i'][i] Mm[i][j] = (m0 + m1) + m2 | (m1 - m2) - m3 - bitwise what forge_wino_output computes
// first. forge_wino_output_half finishes the transform. 64 x 128 tile only (the launches forge_wino_gemm_tile gives 'B').
extern "C" int forge_wino_gemm_half(const float* V1, int C1, int ld1, long long bs1, long long pt1, const float* V2, int C2, int ld2, long long bs2,
                                    long long pt2, const float* U, float* Mm8, int n, int D, int Ht, int Wt, int Cout, int kd, forge_stream_t stream) {
    return wino_gemm_impl(V1, C1, ld1, bs1, pt1, V2, C2, ld2, bs2, pt2, U, Mm8, n, D, Ht, Wt, Cout, kd, 'B', true, stream);
}
