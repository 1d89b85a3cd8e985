// conv_wgrad.hip — weight gradient of the implicit-GEMM convolution on the fp32 matrix cores.
//
//   dW[t][co][ci] = sum_m dY[m][co] * X[row(m) + tap_t][ci]          (zero outside the grid)
//
// i.e. for every tap a GEMM  dW_t = dY^T (Cout x M)  @  X_t (M x Cin)  whose reduction dimension is the voxel index m.
// Both operands are row-major [m][channels] in HBM = "k-major" for this GEMM, which is exactly what the 32x32x2 fp32 MFMA
// wants from LDS without any transpose: lane l supplies A[i = l&31][k = l>>5], so a half-wave reads 32 CONSECUTIVE floats
// of one LDS row (conflict-free ds_read_b32). Workgroup = 4 waves (2 x 2, wave tile 64 x 64 with even/odd channel interleave: one 8-byte LDS read feeds two MFMAs), tile 128(co) x 128(ci) for one
// tap over one M-chunk, K-step = 16 voxels (32 KiB of LDS, 112 VGPRs: 4 workgroups per CU; with 32-voxel steps and 2 workgroups per
// CU the same kernel ran 7-16 % slower), LDS-DMA staged operands (`buffer_load ... lds`, out-of-range rows / taps -> 0; common.h), double-buffered LDS, partial sums added to dW with hardware fp32 atomics (split-K over M-chunks so that the chip is filled:
// taps x tiles alone is only ~100 workgroups). dW must be zero-filled by the caller.
//
// Replaces torch's conv3d weight-gradient (cuDNN/MIOpen) for the ConvGRU / fusion_conv / conv1 convolutions
// (models/fusion.py:29-35,61-68; models/encoder.py:36-40) in training (scripts/kubric_trainer.py:56).
#include "common.h"

namespace forge {

typedef __attribute__((ext_vector_type(16))) float f32x16w;
typedef __attribute__((__vector_size__(4 * sizeof(unsigned)))) unsigned u32x4w;
constexpr unsigned OOBW = 0x80000000u;

__device__ __forceinline__ float4 buf_load16w(__amdgpu_buffer_rsrc_t r, unsigned off) {
    u32x4w v = __builtin_amdgcn_raw_buffer_load_b128(r, off, 0, 0);
    float4 f;
    __builtin_memcpy(&f, &v, 16);
    return f;
}

struct WgradArgs {
    const float* dy; int ldy;                 // [M][ldy], Cout channels used
    const float* x1; const float* x2;         // channel-concatenated inputs [rows][ld1], [rows][ld2]; x2 nullable
    int C1, C2, ld1, ld2;
    long long bs1r, bs2r, span1, span2, spany;
    float* dw;                                // [ntaps][Cout][C1+C2], zero-filled, atomically accumulated
    int n, D, H, W;                           // GEMM-row grid of dY (M = n D H W)
    int is, Di, Hi, Wi;                       // input voxel = (z is + dz, ...)
    int Cout, ntaps, mchunk;                  // rows per M-chunk (multiple of 32)
    int tpp; long long pty, pt1, pt2;         // tpp > 0: ntaps / tpp independent problems in one launch (forge_wino_wgrad: the 16 Winograd points);
                                              // taps [p tpp, (p+1) tpp) read dy + p pty, x1 + p pt1, x2 + p pt2 (floats); dw[t] stays [ntaps][Cout][Cin]
    signed char tap[64][4];
};

constexpr int WT = 128, WK = 16, WJ = WK / 8;  // tile 128 x 128, K-step 16 voxels (32 KiB of LDS: 4 workgroups per CU)

// CIW = width of the Cin tile: 128 (4 waves as 2 x 2, wave tile 64 co x 64 ci, even/odd interleave on both operands), or for
// narrow inputs 64 / 32 (4 waves as 4 x 1, wave tile 32 co x CIW ci) so that a Cin <= 64 problem (conv1: 64, the transpose conv
// of the heads: 32) does not execute a mostly empty 128-wide tile.
// TG > 1 (single input, Cin <= CIW): the 128 columns of the B image are TG TAPS x CIW channels - column c belongs to tap t TG + c / CIW,
// channel c % CIW - so a narrow-input problem with many taps (the heads' ConvTranspose3d(128, 32, 4, s2): 64 taps x 32 channels; conv1: 27 x 64)
// runs the full 128 x 128 tile with its 32 MFMAs per wave and K-step instead of 8 / 16 (4 / 2 x the matrix work per barrier and per staged
// dY row): a thread's staged chunk simply gathers at ITS tap's offset.
template <int CIW, int TG = 1>
__global__ __launch_bounds__(256) void conv_wgrad_kernel(const WgradArgs a) {
    constexpr int NWID = CIW * TG;                                  // used columns of the B image
    static_assert(NWID <= 128 && (TG == 1 || NWID == 128), "tap groups fill the 128-column tile");
    constexpr int P = NWID == 128 ? 2 : 1, Q = NWID == 32 ? 1 : 2;  // accumulators per wave: P co-parities x Q ci-parities
    extern __shared__ __attribute__((aligned(16))) float smem[];   // [2][2][WK][WT]: A (dY) and B (X) images, row = voxel, col = channel
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int wm = NWID == 128 ? wave >> 1 : wave, wn = NWID == 128 ? wave & 1 : 0;
    const int l31 = lane & 31, half = lane >> 5;
    const long long M = (long long)a.n * a.D * a.H * a.W;
    const int Cin = a.C1 + a.C2;
    const int cot = (a.Cout + WT - 1) / WT, cit = TG > 1 ? 1 : (Cin + CIW - 1) / CIW;
    // workgroup order: taps fastest, then the Cin / Cout tiles, the voxel chunk slowest, and each XCD a contiguous run of it - every workgroup
    // that consumes one chunk's dY / X rows (all taps: the same dY rows, X rows a few voxels apart; all tiles) is dispatched back to back on
    // ONE XCD, whose L2 then serves them (with the chunk fastest every tap re-read the whole operands from the fabric: conv1's weight gradient
    // 7 GB per launch): weight gradients of the 4-scene training step 35.9 -> 35.5 ms, conv1's 2.64 -> 2.52 ms)
    const int ntg = (a.ntaps + TG - 1) / TG;
    unsigned bid = xcd_remap(blockIdx.x, gridDim.x);
    const int t = bid % ntg; bid /= ntg;                             // tap (TG = 1) / group of TG taps
    const int ci_t = bid % cit; bid /= cit;
    const int co_t = bid % cot; bid /= cot;
    const int chunk = bid;
    const long long mbeg = (long long)chunk * a.mchunk;
    const long long mend = mbeg + a.mchunk < M ? mbeg + a.mchunk : M;
    const int nsteps = (int)((mend - mbeg + WK - 1) / WK);
    const int co0 = co_t * WT, ci0 = ci_t * CIW;

    const long long pb = a.tpp > 0 ? t / a.tpp : 0;                  // batched problems: this tap's operands
    const forge_v4i32 ry = make_rsrc_words(a.dy + pb * a.pty, a.spany);
    const bool second = ci0 >= a.C1;                                 // this ci tile lives in x2
    const forge_v4i32 rx = make_rsrc_words(second ? a.x2 + pb * a.pt2 : a.x1 + pb * a.pt1, second ? a.span2 : a.span1);
    const int ldx = second ? a.ld2 : a.ld1, cx0 = second ? ci0 - a.C1 : ci0, Cx = second ? a.C2 : a.C1;
    const long long bsx = second ? a.bs2r : a.bs1r;
    // staging: tile rows = 32 voxels, 32 chunks of 16 B per row; thread -> (row = (tid >> 5) + 8 j, chunk = tid & 31)
    const int sc4 = (tid & 31) << 2;
    const int tg = TG > 1 ? sc4 / CIW : 0, sc4c = TG > 1 ? sc4 % CIW : sc4;      // this thread's chunk: tap t TG + tg, channel sc4c
    int dz = 0, dy_ = 0, dx = 0;
    bool tap_ok = false;
#pragma unroll
    for (int g = 0; g < TG; ++g) {                                   // uniform indices into the kernarg table, selected per thread
        const int tt = t * TG + g;
        if (tg == g && tt < a.ntaps) { dz = a.tap[tt][0]; dy_ = a.tap[tt][1]; dx = a.tap[tt][2]; tap_ok = true; }
    }
    const bool ycol_ok = co0 + sc4 < a.Cout, xcol_ok = tap_ok && (TG > 1 || sc4 < CIW) && cx0 + sc4c < Cx;
    // voxel coordinates of the staged rows, advanced by WK rows per K-step (no per-step divisions)
    int rx_[WJ], ry_[WJ], rz_[WJ], rn_[WJ];
#pragma unroll
    for (int j = 0; j < WJ; ++j) {
        unsigned v = (unsigned)(mbeg + (tid >> 5) + 8 * j);
        rx_[j] = (int)(v % (unsigned)a.W); v /= (unsigned)a.W;
        ry_[j] = (int)(v % (unsigned)a.H); v /= (unsigned)a.H;
        rz_[j] = (int)(v % (unsigned)a.D); v /= (unsigned)a.D;
        rn_[j] = (int)v;
    }
    // loop-invariant scalars pulled out of the kernarg struct once (re-loading them inside the K-loop costs a scalar-memory
    // round trip + an lgkmcnt(0) wait — which also drains the LDS queue — per use)
    const unsigned ldy_u = (unsigned)a.ldy, ldx_u = (unsigned)ldx, bsx_u = (unsigned)bsx, mbeg_u = (unsigned)mbeg, mend_u = (unsigned)mend;
    const unsigned ycol_off = (unsigned)(co0 + sc4), xcol_off = (unsigned)(cx0 + sc4c);
    const int is_ = a.is, Wg = a.W, Hg = a.H, Dg = a.D, Wi_ = a.Wi, Hi_ = a.Hi, Di_ = a.Di;
    const int trow = tid >> 5;
    // LDS-DMA staging (common.h: lds_dma16): chunk j of this thread is LDS bytes 16 tid + 4096 j of the A / B image = per wave a lane-linear
    // 1 KB block; rows / taps / channels out of range read offset OOBW and land as zeros.
    const unsigned lds_wave = lds_addr(smem) + (unsigned)__builtin_amdgcn_readfirstlane(wave) * 1024u;
    auto issue_step = [&](int s, int buf) {
        const unsigned stage = lds_wave + (unsigned)buf * (unsigned)(2 * WK * WT * 4);
#pragma unroll
        for (int j = 0; j < WJ; ++j) {
            // 32-bit offsets: every operand span is < 2 GiB (checked on the host side)
            const unsigned m = mbeg_u + (unsigned)(s * WK + trow + 8 * j);
            const bool mok = m < mend_u;
            lds_dma16(ry, (mok && ycol_ok) ? (m * ldy_u + ycol_off) * 4u : OOBW, stage + (unsigned)(j * 4096));
            const int x = rx_[j] * is_ + dx, y = ry_[j] * is_ + dy_, z = rz_[j] * is_ + dz;
            const bool ok = mok && xcol_ok && (unsigned)z < (unsigned)Di_ && (unsigned)y < (unsigned)Hi_ && (unsigned)x < (unsigned)Wi_;
            const unsigned e = (unsigned)rn_[j] * bsx_u + (unsigned)((z * Hi_ + y) * Wi_ + x);
            lds_dma16(rx, ok ? (e * ldx_u + xcol_off) * 4u : OOBW, stage + (unsigned)(WK * WT * 4 + j * 4096));
            rx_[j] += WK;                                         // advance to the row of the next K-step
            while (rx_[j] >= Wg) {
                rx_[j] -= Wg;
                if (++ry_[j] == Hg) { ry_[j] = 0; if (++rz_[j] == Dg) { rz_[j] = 0; ++rn_[j]; } }
            }
        }
    };

    // accumulators: [co parity][ci parity]; MFMA row i <-> co = co0 + wm*64 + 2 i + pco, col j <-> ci = ci0 + wn*64 + 2 j + pci:
    // one 8-byte LDS read per lane then feeds TWO MFMA operands (even / odd channel), halving the LDS instruction count.
    f32x16w acc[P][Q];
#pragma unroll
    for (int p = 0; p < P; ++p)
#pragma unroll
        for (int q = 0; q < Q; ++q)
#pragma unroll
            for (int r = 0; r < 16; ++r) acc[p][q][r] = 0.f;

    if (nsteps > 0) issue_step(0, 0);
    lds_dma_wait();
    __syncthreads();
    for (int s = 0; s < nsteps; ++s) {
        const int buf = s & 1;
        if (s + 1 < nsteps) issue_step(s + 1, buf ^ 1);           // the other stage was last read in step s - 1; in flight under this step's MFMAs
        const float* sa = smem + buf * (2 * WK * WT) + (P == 2 ? wm * 64 + 2 * l31 : wm * 32 + l31);
        const float* sb = smem + buf * (2 * WK * WT) + WK * WT + (Q == 2 ? wn * 64 + 2 * l31 : l31);
#pragma unroll
        for (int kk = 0; kk < WK / 2; ++kk) {
            const int row = 2 * kk + half;
            float fa[2], fb[2];
            if constexpr (P == 2) { const float2 v = *reinterpret_cast<const float2*>(sa + row * WT); fa[0] = v.x; fa[1] = v.y; }
            else fa[0] = fa[1] = sa[row * WT];
            if constexpr (Q == 2) { const float2 v = *reinterpret_cast<const float2*>(sb + row * WT); fb[0] = v.x; fb[1] = v.y; }
            else fb[0] = fb[1] = sb[row * WT];
#pragma unroll
            for (int p = 0; p < P; ++p)
#pragma unroll
                for (int q = 0; q < Q; ++q) acc[p][q] = __builtin_amdgcn_mfma_f32_32x32x2f32(fa[p], fb[q], acc[p][q], 0, 0, 0);
        }
        lds_dma_wait();
        __syncthreads();
    }

    // D[i][j]: col j = lane & 31, row i = (r & 3) + 8 (r >> 2) + 4 (lane >> 5)
#pragma unroll
    for (int q = 0; q < Q; ++q) {
        const int col = Q == 2 ? wn * 64 + 2 * l31 + q : l31;       // column of the tile
        const int te = TG > 1 ? t * TG + col / CIW : t;              // its tap
        const int ci = TG > 1 ? col % CIW : ci0 + col;
        if (te >= a.ntaps || ci >= Cin || (second ? ci - a.C1 >= a.C2 : ci >= a.C1)) continue;
#pragma unroll
        for (int p = 0; p < P; ++p)
#pragma unroll
            for (int r = 0; r < 16; ++r) {
                const int i = (r & 3) + 8 * (r >> 2) + 4 * half;
                const int co = co0 + (P == 2 ? wm * 64 + 2 * i + p : wm * 32 + i);
                if (co < a.Cout) atomic_add_f32(a.dw + ((long long)te * a.Cout + co) * Cin + ci, acc[p][q][r]);
            }
    }
}

// ------------------------------------------------------------------------------------------------
// Narrow variant for Cout <= 32 and Cin <= 32 (the heads' 32->16 / 32->8 / 8->1 convolutions, channel-padded to 32 by the host
// code): a 128x128 tile would execute 16x the useful matrix work. Here every WAVE is independent — no LDS, no barriers: it owns
// a group of up to 4 taps and a chunk of voxels, and feeds the 32x32x2 MFMA straight from global memory (lane l loads
// dY[row][l & 31] and X[row + tap][l & 31], i.e. each half-wave reads one coalesced 128-byte row per operand per MFMA; the dY
// element is shared by the wave's taps). Partial 32x32 tiles are added to dW with fp32 atomics.
__global__ __launch_bounds__(256) void conv_wgrad_small_kernel(const WgradArgs a, int tap_groups, int rows_per_wave) {
    const int lane = threadIdx.x & 63, l31 = lane & 31, half = lane >> 5;
    const long long wave_id = (long long)blockIdx.x * 4 + (threadIdx.x >> 6);
    const long long M = (long long)a.n * a.D * a.H * a.W;
    const long long nchunk = (M + rows_per_wave - 1) / rows_per_wave;
    if (wave_id >= nchunk * tap_groups) return;
    const int tg = (int)(wave_id % tap_groups);
    const long long mbeg = (wave_id / tap_groups) * rows_per_wave;
    const long long mend = mbeg + rows_per_wave < M ? mbeg + rows_per_wave : M;
    const int t0 = tg * 4;
    const int Cin = a.C1;                                          // single input, Cin <= 32
    const __amdgpu_buffer_rsrc_t ry = __builtin_amdgcn_make_buffer_rsrc((void*)a.dy, 0, (int)a.spany, 0x00020000);
    const __amdgpu_buffer_rsrc_t rx = __builtin_amdgcn_make_buffer_rsrc((void*)a.x1, 0, (int)a.span1, 0x00020000);
    const bool yc = l31 < a.Cout, xc = l31 < Cin;

    int tdz[4], tdy[4], tdx[4]; bool tok[4];
#pragma unroll
    for (int q = 0; q < 4; ++q) {
        tok[q] = t0 + q < a.ntaps;
        const int t = tok[q] ? t0 + q : 0;
        tdz[q] = a.tap[t][0]; tdy[q] = a.tap[t][1]; tdx[q] = a.tap[t][2];
    }
    // this lane's row walks mbeg + half, +2, +4, ...
    unsigned v = (unsigned)(mbeg + half);
    int cx = (int)(v % (unsigned)a.W); v /= (unsigned)a.W;
    int cy = (int)(v % (unsigned)a.H); v /= (unsigned)a.H;
    int cz = (int)(v % (unsigned)a.D); v /= (unsigned)a.D;
    int cn = (int)v;

    f32x16w acc[4];
#pragma unroll
    for (int q = 0; q < 4; ++q)
#pragma unroll
        for (int r = 0; r < 16; ++r) acc[q][r] = 0.f;

    for (long long m = mbeg + half; m < mend + half; m += 2) {      // all 64 lanes iterate together (m - half is wave-uniform)
        const bool mok = m < mend;
        const float fa = __builtin_bit_cast(float, __builtin_amdgcn_raw_buffer_load_b32(ry, (mok && yc) ? (unsigned)((m * a.ldy + l31) * 4) : OOBW, 0, 0));
#pragma unroll
        for (int q = 0; q < 4; ++q) {
            const int x = cx * a.is + tdx[q], y = cy * a.is + tdy[q], z = cz * a.is + tdz[q];
            const bool ok = mok && xc && tok[q] && (unsigned)z < (unsigned)a.Di && (unsigned)y < (unsigned)a.Hi && (unsigned)x < (unsigned)a.Wi;
            const long long e = cn * a.bs1r + ((long long)z * a.Hi + y) * a.Wi + x;
            const float fb = __builtin_bit_cast(float, __builtin_amdgcn_raw_buffer_load_b32(rx, ok ? (unsigned)((e * a.ld1 + l31) * 4) : OOBW, 0, 0));
            acc[q] = __builtin_amdgcn_mfma_f32_32x32x2f32(fa, fb, acc[q], 0, 0, 0);
        }
        cx += 2;
        while (cx >= a.W) {
            cx -= a.W;
            if (++cy == a.H) { cy = 0; if (++cz == a.D) { cz = 0; ++cn; } }
        }
    }
#pragma unroll
    for (int q = 0; q < 4; ++q) {
        if (!tok[q] || l31 >= Cin) continue;
#pragma unroll
        for (int r = 0; r < 16; ++r) {
            const int co = (r & 3) + 8 * (r >> 2) + 4 * half;
            if (co < a.Cout) atomic_add_f32(a.dw + ((long long)(t0 + q) * a.Cout + co) * Cin + l31, acc[q][r]);
        }
    }
}


// ------------------------------------------------------------------------------------------------
// Narrow variant with operand reuse across taps (stride-1 "same" convolutions with Cout, Cin <= 32 whose taps span <= 9 distinct
// (dz, dy) lines and |dx| <= 2: the heads' 3x3x3 convolutions, conv_rgb's 5x5). conv_wgrad_small_kernel above loads one X element per
// lane per MFMA (12.8 FLOP per byte of L1/L2 traffic -> 35 TF); but the 27 taps of a voxel segment read the same dY rows and almost
// the same X rows. Here a workgroup stages, for a segment of 32 consecutive x-voxels of one (n, z, y) line, the dY rows and the X
// halo - every needed (dz, dy) line, 32 + 2 rx voxels long - in LDS once (43 KB for 3x3x3 at 32 channels) and feeds all taps from
// it: each wave owns every 4th tap (<= 7 accumulator tiles), one ds_read_b32 per operand per MFMA, conflict-free (a half-wave reads
// 32 consecutive floats of one LDS row). Workgroups walk segments grid-stride with the next segment's global loads in flight under the
// current segment's MFMAs, and add their partial 32x32 tiles to dW with fp32 atomics once at the end.
constexpr int LSEG = 32, LMAXL = 9, LMAXR = 3, LROWS = LSEG + 2 * LMAXR, LTAPS = 7;

struct LineTable { signed char dz[LMAXL], dy[LMAXL]; int nlines, rx; };

__global__ __launch_bounds__(256) void conv_wgrad_lines_kernel(const WgradArgs a, const LineTable lt) {
    extern __shared__ __attribute__((aligned(16))) float smem[];   // dYs [LSEG][32] | Xs [nlines][LSEG + 2 rx][32]
    float* dYs = smem;
    float* Xs = smem + LSEG * 32;
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6, l31 = lane & 31, half = lane >> 5;
    const int Cin = a.C1, xrows = LSEG + 2 * lt.rx;
    const int nsx = (a.W + LSEG - 1) / LSEG;
    const long long nseg = (long long)a.n * a.D * a.H * nsx;
    const __amdgpu_buffer_rsrc_t ry = __builtin_amdgcn_make_buffer_rsrc((void*)a.dy, 0, (int)a.spany, 0x00020000);
    const __amdgpu_buffer_rsrc_t rx_ = __builtin_amdgcn_make_buffer_rsrc((void*)a.x1, 0, (int)a.span1, 0x00020000);
    // staging geometry: one float4 (4 channels) per thread per pass; 8 threads per 32-channel row
    const int c4 = (tid & 7) << 2, srow = tid >> 3;                 // 32 rows per pass
    const int xchunks = lt.nlines * xrows;                           // X rows to stage
    constexpr int XP = (LMAXL * LROWS + 31) / 32;                    // passes (upper bound 11)
    float4 ry4, rx4[XP];
    auto load_seg = [&](long long sg) {
        long long q = sg;
        const int sx = (int)(q % nsx); q /= nsx;
        const int y = (int)(q % a.H); q /= a.H;
        const int z = (int)(q % a.D); q /= a.D;
        const int nn = (int)q, x0 = sx * LSEG;
        {   // dY row srow of the segment
            const int x = x0 + srow;
            const long long m = (((long long)nn * a.D + z) * a.H + y) * a.W + x;
            ry4 = buf_load16w(ry, (x < a.W && c4 < a.Cout) ? (unsigned)((m * a.ldy + c4) * 4) : OOBW);
        }
#pragma unroll
        for (int p = 0; p < XP; ++p) {
            const int r = srow + 32 * p;                             // staged X row: (line, xr)
            unsigned off = OOBW;
            if (r < xchunks && c4 < Cin) {
                const int ln = r / xrows, xr = r - ln * xrows;
                const int zi = z + lt.dz[ln], yi = y + lt.dy[ln], xi = x0 - lt.rx + xr;
                if ((unsigned)zi < (unsigned)a.D && (unsigned)yi < (unsigned)a.H && (unsigned)xi < (unsigned)a.W)
                    off = (unsigned)((((long long)nn * a.bs1r + ((long long)zi * a.H + yi) * a.W + xi) * a.ld1 + c4) * 4);
            }
            rx4[p] = buf_load16w(rx_, off);
        }
    };
    auto store_seg = [&]() {
        *reinterpret_cast<float4*>(dYs + srow * 32 + c4) = ry4;
#pragma unroll
        for (int p = 0; p < XP; ++p) {
            const int r = srow + 32 * p;
            if (r < xchunks) *reinterpret_cast<float4*>(Xs + r * 32 + c4) = rx4[p];
        }
    };
    // this wave's taps: t = wave, wave + 4, ... (<= LTAPS); LDS row offset of each tap's first X row
    int tb[LTAPS];
    bool tok[LTAPS];
#pragma unroll
    for (int j = 0; j < LTAPS; ++j) {
        const int t = wave + 4 * j;
        tok[j] = t < a.ntaps;
        const int tt = tok[j] ? t : 0;
        tb[j] = (a.tap[tt][3] * xrows + a.tap[tt][2] + lt.rx) * 32;  // (line, dx + rx)
    }
    f32x16w acc[LTAPS];
#pragma unroll
    for (int j = 0; j < LTAPS; ++j)
#pragma unroll
        for (int r = 0; r < 16; ++r) acc[j][r] = 0.f;

    long long sg = blockIdx.x;
    if (sg < nseg) load_seg(sg);
    for (; sg < nseg; sg += gridDim.x) {
        __syncthreads();                                             // previous segment's MFMAs have read the LDS images
        store_seg();
        __syncthreads();
        if (sg + gridDim.x < nseg) load_seg(sg + gridDim.x);         // in flight under this segment's MFMAs
#pragma unroll 4
        for (int k = 0; k < LSEG / 2; ++k) {
            const int row = 2 * k + half;
            const float fa = dYs[row * 32 + l31];
#pragma unroll
            for (int j = 0; j < LTAPS; ++j) {
                if (!tok[j]) continue;                               // wave-uniform
                const float fb = Xs[tb[j] + row * 32 + l31];
                acc[j] = __builtin_amdgcn_mfma_f32_32x32x2f32(fa, fb, acc[j], 0, 0, 0);
            }
        }
    }
#pragma unroll
    for (int j = 0; j < LTAPS; ++j) {
        if (!tok[j] || l31 >= Cin) continue;
        const int t = wave + 4 * j;
#pragma unroll
        for (int r = 0; r < 16; ++r) {
            const int co = (r & 3) + 8 * (r >> 2) + 4 * half;
            if (co < a.Cout && acc[j][r] != 0.f) atomic_add_f32(a.dw + ((long long)t * a.Cout + co) * Cin + l31, acc[j][r]);
        }
    }
}

// The same for Cout <= 16 on the 16x16x4 fp32 MFMA (A[i = l & 15][k = l >> 4], B[k = l >> 4][j = l & 15], D[row = 4 (l >> 4) + r][col = l & 15]):
// the 32-row tile of the kernel above executes 2x (Cout = 16: the features head's 32 -> 16) to 4x (Cout = 8: the density head's 32 -> 8,
// conv_rgb's 16 -> 8) the useful matrix work. CIT = 32: the Cin tile is two 16-column MFMAs fed by ONE 8-byte LDS read per lane (even / odd
// input channel); CIT = 16: one MFMA. dYs rows are 16 floats, so the four voxel rows a wave-instruction reads fall into disjoint banks.
typedef __attribute__((ext_vector_type(4))) float f32x4w;

// IS = 2 (CIT = 16, LT = 9 taps per wave): the gathered operand lives on the 2x finer grid (voxel m reads row 2 m + tap) - the weight gradient of
// a stride-2 transposed convolution (conv_rgb's ConvTranspose2d(16, 16, 6, stride 2): 36 taps on 6 lines, |dx| <= 3), whose staged X segment is
// 2 LSEG - 1 + 2 rx rows long and is read with a row stride of 2.
// NWV = waves per workgroup: 4, or 8 for CIT = 32 (the heads' 32 -> 16 / 32 -> 8 layers): with 4 waves that instantiation needs 189 VGPRs
// (7 taps x 2 accumulators per wave + 11 staging passes), i.e. spills at 3 workgroups per CU and runs 1.5x slower at 2 (1124 vs 732 us);
// 8 waves share one staged segment with 4 taps and 6 staging passes each.
template <int CIT, int IS = 1, int LT = LTAPS, int NWV = 4>
__global__ __launch_bounds__(64 * NWV, NWV == 8 ? 2 : ((CIT == 16 && IS == 1) ? 4 : 3)) void conv_wgrad_lines16_kernel(const WgradArgs a, const LineTable lt) {
    constexpr int NT = CIT / 16, NTHR = 64 * NWV;
    extern __shared__ __attribute__((aligned(16))) float smem[];   // dYs [LSEG][16] | Xs [nlines][LSEG + 2 rx][CIT]
    float* dYs = smem;
    float* Xs = smem + LSEG * 16;
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6, l15 = lane & 15, kq = lane >> 4;
    const int Cin = a.C1, xrows = (LSEG - 1) * IS + 1 + 2 * lt.rx;
    const int nsx = (a.W + LSEG - 1) / LSEG;
    const long long nseg = (long long)a.n * a.D * a.H * nsx;
    const __amdgpu_buffer_rsrc_t ry = __builtin_amdgcn_make_buffer_rsrc((void*)a.dy, 0, (int)a.spany, 0x00020000);
    const __amdgpu_buffer_rsrc_t rx_ = __builtin_amdgcn_make_buffer_rsrc((void*)a.x1, 0, (int)a.span1, 0x00020000);
    // staging: one float4 per thread per pass; CIT / 4 threads per X row (4 per dY row: threads 0..127 stage the 32 dY rows)
    constexpr int TPR = CIT / 4, RPP = NTHR / TPR;                   // threads per row, rows per pass
    const int c4 = (tid % TPR) << 2, srow = tid / TPR;
    const int yc4 = (tid & 3) << 2, yrow = tid >> 2;
    const int xchunks = lt.nlines * xrows;
    constexpr int XP = ((IS == 2 ? 6 : LMAXL) * ((LSEG - 1) * IS + 1 + 2 * LMAXR) + RPP - 1) / RPP;   // IS = 2: <= 6 lines (host)
    float4 ry4 = make_float4(0.f, 0.f, 0.f, 0.f), rx4[XP];
    auto load_seg = [&](long long sg) {
        long long q = sg;
        const int sx = (int)(q % nsx); q /= nsx;
        const int y = (int)(q % a.H); q /= a.H;
        const int z = (int)(q % a.D); q /= a.D;
        const int nn = (int)q, x0 = sx * LSEG;
        if (yrow < LSEG) {
            const int x = x0 + yrow;
            const long long m = (((long long)nn * a.D + z) * a.H + y) * a.W + x;
            ry4 = buf_load16w(ry, (x < a.W && yc4 < a.Cout) ? (unsigned)((m * a.ldy + yc4) * 4) : OOBW);
        }
#pragma unroll
        for (int p = 0; p < XP; ++p) {
            const int r = srow + RPP * p;
            unsigned off = OOBW;
            if (r < xchunks && c4 < Cin) {
                const int ln = r / xrows, xr = r - ln * xrows;
                const int zi = z * IS + lt.dz[ln], yi = y * IS + lt.dy[ln], xi = x0 * IS - lt.rx + xr;
                if ((unsigned)zi < (unsigned)a.Di && (unsigned)yi < (unsigned)a.Hi && (unsigned)xi < (unsigned)a.Wi)
                    off = (unsigned)((((long long)nn * a.bs1r + ((long long)zi * a.Hi + yi) * a.Wi + xi) * a.ld1 + c4) * 4);
            }
            rx4[p] = buf_load16w(rx_, off);
        }
    };
    auto store_seg = [&]() {
        if (yrow < LSEG) *reinterpret_cast<float4*>(dYs + yrow * 16 + yc4) = ry4;
#pragma unroll
        for (int p = 0; p < XP; ++p) {
            const int r = srow + RPP * p;
            if (r < xchunks) *reinterpret_cast<float4*>(Xs + r * CIT + c4) = rx4[p];
        }
    };
    int tb[LT];
    bool tok[LT];
#pragma unroll
    for (int j = 0; j < LT; ++j) {
        const int t = wave + NWV * j;
        tok[j] = t < a.ntaps;
        const int tt = tok[j] ? t : 0;
        tb[j] = (a.tap[tt][3] * xrows + a.tap[tt][2] + lt.rx) * CIT;  // (line, dx + rx)
    }
    f32x4w acc[LT][NT];
#pragma unroll
    for (int j = 0; j < LT; ++j)
#pragma unroll
        for (int n = 0; n < NT; ++n)
#pragma unroll
            for (int r = 0; r < 4; ++r) acc[j][n][r] = 0.f;

    long long sg = blockIdx.x;
    if (sg < nseg) load_seg(sg);
    for (; sg < nseg; sg += gridDim.x) {
        __syncthreads();
        store_seg();
        __syncthreads();
        if (sg + gridDim.x < nseg) load_seg(sg + gridDim.x);
#pragma unroll 4
        for (int k = 0; k < LSEG / 4; ++k) {
            const int row = 4 * k + kq;
            const float fa = dYs[row * 16 + l15];
#pragma unroll
            for (int j = 0; j < LT; ++j) {
                if (!tok[j]) continue;                               // wave-uniform
                if constexpr (NT == 2) {
                    const float2 fb = *reinterpret_cast<const float2*>(Xs + tb[j] + row * (CIT * IS) + 2 * l15);     // input channels 2 l15, 2 l15 + 1
                    acc[j][0] = __builtin_amdgcn_mfma_f32_16x16x4f32(fa, fb.x, acc[j][0], 0, 0, 0);
                    acc[j][1] = __builtin_amdgcn_mfma_f32_16x16x4f32(fa, fb.y, acc[j][1], 0, 0, 0);
                } else {
                    const float fb = Xs[tb[j] + row * (CIT * IS) + l15];
                    acc[j][0] = __builtin_amdgcn_mfma_f32_16x16x4f32(fa, fb, acc[j][0], 0, 0, 0);
                }
            }
        }
    }
#pragma unroll
    for (int j = 0; j < LT; ++j) {
        if (!tok[j]) continue;
        const int t = wave + NWV * j;
#pragma unroll
        for (int n = 0; n < NT; ++n) {
            const int ci = NT == 2 ? 2 * l15 + n : l15;
            if (ci >= Cin) continue;
#pragma unroll
            for (int r = 0; r < 4; ++r) {
                const int co = 4 * kq + r;
                if (co < a.Cout && acc[j][n][r] != 0.f) atomic_add_f32(a.dw + ((long long)t * a.Cout + co) * Cin + ci, acc[j][n][r]);
            }
        }
    }
}

}  // namespace forge

using namespace forge;

// conv_wgrad_kernel<ciw> over (taps x Cout tiles x Cin tiles x voxel chunks) workgroups.
static int launch_wgrad_tiles(WgradArgs& a, int ciw, hipStream_t stream) {
    const long long M = (long long)a.n * a.D * a.H * a.W;
    const int Cin = a.C1 + a.C2, Cout = a.Cout, ntaps = a.ntaps;
    // narrow single inputs with several taps: TG taps share one 128-column tile (kernel header)
    // (large problems only: on the trunk's 64 -> 64 3x3 layers, M = 81920, the coarser tiles leave the chip under-filled: 125 -> 166 us)
    const bool grp = a.x2 == nullptr && a.tpp == 0 && M >= 131072;
    const int tg = (grp && ciw == 32 && Cin <= 32 && ntaps >= 4) ? 4 : (grp && ciw == 64 && Cin <= 64 && ntaps >= 2) ? 2 : 1;
    const long long tiles = (long long)((ntaps + tg - 1) / tg) * ((Cout + WT - 1) / WT) * (tg > 1 ? 1 : (Cin + ciw - 1) / ciw);
    // split the voxel (reduction) axis so that ~4096 workgroups exist, but keep >= 32 K-steps (1024 voxels) per workgroup: every
    // workgroup ends with up to 16 K fp32 atomics for its tile, which must stay small next to its MFMA work. When that leaves the
    // chip under-filled (ResNet at one scene: M = 5120, a handful of tiles) the floor drops to 8 K-steps: those launches are
    // latency-bound and more, shorter workgroups are what shortens them.
    long long target = 4096;      // many short workgroups: 512 are resident at a time, a coarse split leaves a mostly empty last round
    long long nchunk = (target + tiles - 1) / tiles;
    if (nchunk > M / 1024) nchunk = M / 1024;
    if (nchunk < 1) nchunk = 1;
    if (tiles * nchunk < 512) {
        nchunk = (512 + tiles - 1) / tiles;
        if (nchunk > M / 256) nchunk = M / 256;
        if (nchunk < 1) nchunk = 1;
    }
    long long mchunk = ((M + nchunk - 1) / nchunk + WK - 1) / WK * WK;
    a.mchunk = (int)mchunk;
    nchunk = (M + mchunk - 1) / mchunk;
    const long long grid = tiles * nchunk;
    FORGE_REQUIRE(grid < (1ll << 31), FORGE_ESHAPE, "forge_conv_wgrad: grid too large");
    const size_t lds = 2 * 2 * WK * WT * sizeof(float);     // 32 KiB
#define FORGE_LAUNCH_WGRAD(CIWv, TGv)                                                                                                \
    do {                                                                                                                             \
        FORGE_SET_MAX_LDS_ONCE((conv_wgrad_kernel<CIWv, TGv>), lds);                                                                 \
        hipLaunchKernelGGL((conv_wgrad_kernel<CIWv, TGv>), dim3((unsigned)grid), dim3(256), lds, stream, a);           \
    } while (0)
    if (ciw == 32 && tg == 4) FORGE_LAUNCH_WGRAD(32, 4);
    else if (ciw == 64 && tg == 2) FORGE_LAUNCH_WGRAD(64, 2);
    else if (ciw == 32) FORGE_LAUNCH_WGRAD(32, 1);
    else if (ciw == 64) FORGE_LAUNCH_WGRAD(64, 1);
    else FORGE_LAUNCH_WGRAD(128, 1);
#undef FORGE_LAUNCH_WGRAD
    FORGE_LAUNCH_CHECK("forge_conv_wgrad");
    return 0;
}

extern "C" int forge_conv_wgrad(const float* dy, int ldy, const float* x1, int C1, int ld1, long long bs1, const float* x2, int C2, int ld2,
                                long long bs2, float* dw, int n, int D, int H, int W, int is, int Di, int Hi, int Wi, int Cout,
                                const int* taps, int ntaps, forge_stream_t stream) {
    FORGE_REQUIRE(dy && x1 && dw && taps, FORGE_EINVAL, "forge_conv_wgrad: null pointer argument");
    FORGE_REQUIRE(n > 0 && D > 0 && H > 0 && W > 0 && Cout > 0 && ntaps > 0 && ntaps <= 64 && is >= 1 && Di > 0 && Hi > 0 && Wi > 0, FORGE_EINVAL,
                  "forge_conv_wgrad: bad dims");
    FORGE_REQUIRE(C1 > 0 && C1 % 4 == 0 && C2 >= 0 && C2 % 4 == 0 && Cout % 4 == 0 && ldy >= Cout && ldy % 4 == 0 && ld1 >= C1 && ld1 % 4 == 0 &&
                  (C2 == 0 || (ld2 >= C2 && ld2 % 4 == 0)), FORGE_ESHAPE, "forge_conv_wgrad: channel counts / row strides must be multiples of 4");
    FORGE_REQUIRE((C2 == 0) == (x2 == nullptr), FORGE_EINVAL, "forge_conv_wgrad: x2/C2 mismatch");
    FORGE_REQUIRE(C2 == 0 || C1 % WT == 0, FORGE_ESHAPE, "forge_conv_wgrad: with two inputs C1 must be a multiple of %d", WT);
    WgradArgs a;
    a.dy = dy; a.ldy = ldy; a.x1 = x1; a.x2 = x2; a.C1 = C1; a.C2 = C2; a.ld1 = ld1; a.ld2 = ld2; a.dw = dw;
    a.n = n; a.D = D; a.H = H; a.W = W; a.is = is; a.Di = Di; a.Hi = Hi; a.Wi = Wi; a.Cout = Cout; a.ntaps = ntaps;
    a.tpp = 0; a.pty = a.pt1 = a.pt2 = 0;
    a.bs1r = bs1 > 0 ? bs1 : (long long)Di * Hi * Wi; a.bs2r = bs2 > 0 ? bs2 : (long long)Di * Hi * Wi;
    const long long M = (long long)n * D * H * W;
    a.spany = M * ldy * 4;
    a.span1 = ((long long)(n - 1) * a.bs1r + (long long)Di * Hi * Wi) * ld1 * 4;
    a.span2 = x2 ? ((long long)(n - 1) * a.bs2r + (long long)Di * Hi * Wi) * ld2 * 4 : 0;
    FORGE_REQUIRE(a.spany < (1ll << 31) && a.span1 < (1ll << 31) && a.span2 < (1ll << 31), FORGE_ESHAPE,
                  "forge_conv_wgrad: an operand spans >= 2 GiB (32-bit buffer offsets); split the batch");
    for (int t = 0; t < 64; ++t) {
        for (int k = 0; k < 3; ++k) a.tap[t][k] = (signed char)(t < ntaps ? taps[t * 3 + k] : 0);
        a.tap[t][3] = 0;
    }
    const int Cin = C1 + C2;
    const bool lines_s2 = is == 2 && Cout <= 16 && Cin <= 16 && ntaps <= 4 * 9;     // stride-2 gathered operand: the 16x16 kernel only
    if (Cout <= 32 && Cin <= 32 && x2 == nullptr && ((is == 1 && Di == D && Hi == H && Wi == W && ntaps <= 4 * LTAPS) || lines_s2)) {
        // taps grouped by (dz, dy) line; usable when <= 9 lines and |dx| <= 3
        LineTable lt;
        lt.nlines = 0; lt.rx = 0;
        bool ok = true;
        for (int t = 0; t < ntaps && ok; ++t) {
            const int dz = taps[t * 3], dy_ = taps[t * 3 + 1], dx = taps[t * 3 + 2];
            if (dx > LMAXR || dx < -LMAXR) { ok = false; break; }
            if (dx > lt.rx) lt.rx = dx;
            if (-dx > lt.rx) lt.rx = -dx;
            int ln = -1;
            for (int i = 0; i < lt.nlines; ++i)
                if (lt.dz[i] == dz && lt.dy[i] == dy_) ln = i;
            if (ln < 0) {
                if (lt.nlines == (lines_s2 ? 6 : LMAXL)) { ok = false; break; }
                ln = lt.nlines++;
                lt.dz[ln] = (signed char)dz; lt.dy[ln] = (signed char)dy_;
            }
            a.tap[t][3] = (signed char)ln;
        }
        if (ok) {
            const long long nseg = (long long)n * D * H * ((W + LSEG - 1) / LSEG);
            const size_t lds = (size_t)(LSEG * 32 + lt.nlines * (LSEG + 2 * lt.rx) * 32) * sizeof(float);
            FORGE_SET_MAX_LDS_ONCE(conv_wgrad_lines_kernel, (LSEG * 32 + LMAXL * LROWS * 32) * sizeof(float));
            a.mchunk = 0;
            if (lines_s2) {
                const size_t lds16 = (size_t)(LSEG * 16 + lt.nlines * (2 * LSEG - 1 + 2 * lt.rx) * 16) * sizeof(float);
                const long long grid16 = nseg < 768 ? nseg : 768;                   // 3 resident workgroups per CU
                FORGE_SET_MAX_LDS_ONCE((conv_wgrad_lines16_kernel<16, 2, 9>), (LSEG * 16 + 6 * (2 * LSEG - 1 + 2 * LMAXR) * 16) * sizeof(float));
                hipLaunchKernelGGL((conv_wgrad_lines16_kernel<16, 2, 9>), dim3((unsigned)grid16), dim3(256), lds16, (hipStream_t)stream, a, lt);
                FORGE_LAUNCH_CHECK("forge_conv_wgrad");
                return 0;
            }
            if (Cout <= 16) {
                // narrow outputs: the 16x16x4 MFMA tile (no 32-row padding); 4 workgroups per CU walk the segments
                const int cit = Cin <= 16 ? 16 : 32;
                const size_t lds16 = (size_t)(LSEG * 16 + lt.nlines * (LSEG + 2 * lt.rx) * cit) * sizeof(float);
                const long long grid16 = nseg < (cit == 16 ? 1024 : 768) ? nseg : (cit == 16 ? 1024 : 768);    // 4 / 3 resident workgroups per CU (108 / ~150 VGPRs)
                if (cit == 16) {
                    FORGE_SET_MAX_LDS_ONCE(conv_wgrad_lines16_kernel<16>, (LSEG * 16 + LMAXL * LROWS * 16) * sizeof(float));
                    hipLaunchKernelGGL(conv_wgrad_lines16_kernel<16>, dim3((unsigned)grid16), dim3(256), lds16, (hipStream_t)stream, a, lt);
                } else {
                    const long long grid32 = nseg < 512 ? nseg : 512;      // 8-wave workgroups, 2 per CU
                    FORGE_SET_MAX_LDS_ONCE((conv_wgrad_lines16_kernel<32, 1, 4, 8>), (LSEG * 16 + LMAXL * LROWS * 32) * sizeof(float));
                    hipLaunchKernelGGL((conv_wgrad_lines16_kernel<32, 1, 4, 8>), dim3((unsigned)grid32), dim3(512), lds16, (hipStream_t)stream, a, lt);
                }
                FORGE_LAUNCH_CHECK("forge_conv_wgrad");
                return 0;
            }
            const long long grid = nseg < 512 ? nseg : 512;          // 2 workgroups per CU (244 VGPRs), each walking its share of the segments
            hipLaunchKernelGGL(conv_wgrad_lines_kernel, dim3((unsigned)grid), dim3(256), lds, (hipStream_t)stream, a, lt);
            FORGE_LAUNCH_CHECK("forge_conv_wgrad");
            return 0;
        }
        for (int t = 0; t < ntaps; ++t) a.tap[t][3] = 0;
    }
    if (Cout <= 32 && Cin <= 32 && x2 == nullptr && W % 2 == 0) {
        // narrow channels: independent waves, 4 taps each, fed straight from global memory
        const int tap_groups = (ntaps + 3) / 4;
        long long rows = (M * tap_groups + 4095) / 4096;            // ~4096 waves
        rows = (rows + 1) / 2 * 2;
        if (rows < 256) rows = 256;
        a.mchunk = (int)rows;
        const long long waves = ((M + rows - 1) / rows) * tap_groups;
        hipLaunchKernelGGL(conv_wgrad_small_kernel, dim3((unsigned)((waves + 3) / 4)), dim3(256), 0, (hipStream_t)stream, a, tap_groups, (int)rows);
        FORGE_LAUNCH_CHECK("forge_conv_wgrad");
        return 0;
    }
    return launch_wgrad_tiles(a, (x2 == nullptr && Cin <= 32) ? 32 : (x2 == nullptr && Cin <= 64) ? 64 : WT, (hipStream_t)stream);
}

// Weight gradient in the Winograd domain (csrc/winograd.hip): dU[p][kd][co][ci] = sum_r dMm[p][r][co] (V1 | V2)[p][r + kd plane][ci] for the 16
// points in ONE launch of conv_wgrad_kernel - 16 kd "taps" whose operands advance by one point every kd taps. dU [16][kd][Cout][C1+C2] must
// be zero-filled (fp32 atomics over voxel chunks). 2.25x fewer FLOPs than forge_conv_wgrad on the same convolution.
extern "C" int forge_wino_wgrad(const float* dMm, const float* V1, int C1, long long bs1, long long pt1, const float* V2, int C2, long long bs2,
                                long long pt2, float* dU, int n, int D, int Ht, int Wt, int Cout, int kd, forge_stream_t stream) {
    FORGE_REQUIRE(dMm && V1 && dU && (kd == 1 || kd == 3), FORGE_EINVAL, "forge_wino_wgrad: null pointer argument / kd not 1 or 3");
    FORGE_REQUIRE(n > 0 && D > 0 && Ht > 0 && Wt > 0 && Cout > 0 && Cout % 4 == 0 && C1 > 0 && C1 % 4 == 0 && C2 >= 0 && C2 % 4 == 0 &&
                  (C2 == 0) == (V2 == nullptr) && (C2 == 0 || C1 % WT == 0), FORGE_ESHAPE,
                  "forge_wino_wgrad: bad dims (channel counts multiples of 4; with two inputs C1 a multiple of %d)", WT);
    WgradArgs a;
    memset(&a, 0, sizeof(a));
    const long long vol = (long long)D * Ht * Wt, R = (long long)n * vol;
    a.dy = dMm; a.ldy = Cout; a.x1 = V1; a.x2 = V2; a.C1 = C1; a.C2 = C2; a.ld1 = C1; a.ld2 = C2; a.dw = dU;
    a.n = n; a.D = D; a.H = Ht; a.W = Wt; a.is = 1; a.Di = D; a.Hi = Ht; a.Wi = Wt; a.Cout = Cout; a.ntaps = 16 * kd;
    a.bs1r = bs1 > 0 ? bs1 : vol; a.bs2r = bs2 > 0 ? bs2 : vol;
    a.spany = R * Cout * 4;
    a.span1 = ((long long)(n - 1) * a.bs1r + vol) * C1 * 4;
    a.span2 = V2 ? ((long long)(n - 1) * a.bs2r + vol) * C2 * 4 : 0;
    FORGE_REQUIRE(a.spany < (1ll << 31) && a.span1 < (1ll << 31) && a.span2 < (1ll << 31), FORGE_ESHAPE,
                  "forge_wino_wgrad: an operand spans >= 2 GiB per Winograd point (32-bit buffer offsets); split the batch");
    a.tpp = kd; a.pty = R * Cout; a.pt1 = pt1 > 0 ? pt1 : R * C1; a.pt2 = V2 ? (pt2 > 0 ? pt2 : R * C2) : 0;
    for (int t = 0; t < 16 * kd; ++t) a.tap[t][0] = (signed char)(kd == 3 ? t % 3 - 1 : 0);
    const int Cin = C1 + C2;
    return launch_wgrad_tiles(a, (V2 == nullptr && Cin <= 32) ? 32 : (V2 == nullptr && Cin <= 64) ? 64 : WT, (hipStream_t)stream);
}
