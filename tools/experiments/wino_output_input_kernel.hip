// Rejected round-3 experiment, kept out of libforge_hip.so: inverse Winograd transform + tail fused with the NEXT
// convolution's input transform through LDS. It compiled inside forge_amd/csrc/winograd.hip (WinoOutArgs, f4_add / f4_sub, W_* epilogues) and was
// bit-identical to forge_wino_output + forge_wino_input (12 shape x epilogue cases). Measured on the b = 1 fusion (tools/stage_replay.py, hipGraph
// replay): two launches 4.27 ms -> fused 4.41 ms (16-channel chunks, 256 threads, 256 workgroups) / 4.90 ms (8-channel chunks, 512 threads);
// whole step 8.19 -> 8.33 / 8.86 ms, 4 steps in flight 6.95 -> 7.19 / 7.87 ms. Why it loses: the next transform's 4x4 patches need a halo of
// the produced tensor, so a workgroup must own whole image planes (or pay 1.25-2x halo re-reads of Mm, which is 4x the tensor), and with a
// plane per workgroup the 160 KB LDS leaves room for 16 channels only - 64-byte instead of 512-byte contiguous reads of the 134 MB Mm operand
// and 256 workgroups for 256 CUs. The separate kernels run at 4.6-4.8 TB/s of algorithmic traffic with full-row coalescing; what the fusion
// saves (10 launches, 0.25 GB of h / h*r round trips per step) is less than what it loses.

// ---- inverse transform + tail of one convolution FUSED with the input transform of the next: in the ConvGRU chain every
// convolution's output (fusion_conv activations, h * r, the new state h) is immediately the next convolution's input, so
//     Mm -> y = tail(A^T Mm A) -> HBM -> wino_input -> V_next      becomes      Mm -> y (LDS) -> V_next,
// one launch instead of two per convolution and no HBM round trip of y when nothing else needs it (the activation between the two fusion_conv
// layers, h * r). Workgroup = one (n, z) plane x band of TB tile rows x 16 channels of y: phase 1 (thread = 2x2 output tile x 4 channels, the
// same arithmetic as wino_output_kernel, element for element) computes the band plus one halo tile row above and below into LDS (the next
// transform's 4x4 patches reach one output row into the neighbouring tiles; at H = 32 the band is the whole plane: no halo) and writes the
// band's own rows of out / out2 / out3; phase 2 (thread = tile x 4 channels) reads its 4x4 patch from LDS (zero outside the plane) and stores
// V_next = B^T y B - bit-identical to wino_input_kernel on the stored y.
//   y = out (W_AFFINE_ACT), h * r (W_GRU_GATES: the z half is produced alongside, no transform), the new state hn (W_GRU_OUT).
constexpr int WOI_CK = 8, WOI_THREADS = 512;
struct WinoOutInArgs {
    WinoOutArgs o;
    float* Vn; int ldvn; long long ptvn;          // V_next[p] = Vn + p ptvn, rows [n][D][H/2][W/2] x ldvn floats
    int CY;                                        // channels of y (Cout, or Cout / 2 for the gate epilogue)
    int TB;                                        // tile rows per band (divides H/2)
};

template <int EPI>
__global__ __launch_bounds__(WOI_THREADS) void wino_output_input_kernel(const WinoOutInArgs aa) {
    const WinoOutArgs& a = aa.o;
    extern __shared__ __attribute__((aligned(16))) float ylds[];    // [2 TB + 2][W][WOI_CK]
    const int Ht = a.H >> 1, Wt = a.W >> 1, TB = aa.TB;
    const int nband = Ht / TB, nchunk = aa.CY / WOI_CK;
    unsigned bid = blockIdx.x;
    const int chunk = (int)(bid % (unsigned)nchunk); bid /= (unsigned)nchunk;
    const int band = (int)(bid % (unsigned)nband); bid /= (unsigned)nband;
    const long long plane = bid;                                    // (n, z) plane index
    const int c0 = chunk * WOI_CK, th0 = band * TB;                 // first y channel / first own tile row
    const int ybase = 2 * th0 - 1, nrows = 2 * TB + 2;              // LDS row 0 = image row ybase
    const int Ch = a.Cout >> 1;
    const long long R = (long long)a.n * a.D * Ht * Wt;
    const long long prow = plane * a.H * a.W;                       // first output row of the plane
    const unsigned R1 = (unsigned)(a.D * Ht * Wt);

    // ---- phase 1
    const int t_lo = th0 > 0 ? th0 - 1 : 0, t_hi = (th0 + TB < Ht) ? th0 + TB : Ht - 1;       // tile rows to evaluate (halo included)
    const int ntask1 = (t_hi - t_lo + 1) * Wt * (WOI_CK / 4);
    for (int task = threadIdx.x; task < ntask1; task += WOI_THREADS) {
        const int q = task % (WOI_CK / 4), tt = task / (WOI_CK / 4);
        const int tw = tt % Wt, th = t_lo + tt / Wt;
        const bool own = th >= th0 && th < th0 + TB;
        const int cy = c0 + 4 * q;                                   // y channel; Mm column of y: cy (+ Ch for the gate epilogue's reset half)
        const long long r = plane * Ht * Wt + (long long)th * Wt + tw;
        const int ncol = (EPI == W_GRU_GATES && own) ? 2 : 1;        // gate epilogue: the band's own tiles also produce z (columns cy)
        for (int part = ncol - 1; part >= 0; --part) {               // part 1 = z half (own tiles only), part 0 = the y half
            const int c = (EPI == W_GRU_GATES) ? (part == 1 ? cy : Ch + cy) : cy;
            const float* mp = a.Mm + r * a.Cout + c;
            float4 m[4][4];
#pragma unroll
            for (int i = 0; i < 4; ++i)
#pragma unroll
                for (int j = 0; j < 4; ++j) m[i][j] = *reinterpret_cast<const float4*>(mp + (4 * i + j) * a.ptm);
            if (a.Mm2) {
                const unsigned nn = (unsigned)r / R1;
                const float* mp2 = a.Mm2 + ((long long)nn * a.bs2 + ((unsigned)r - nn * R1)) * a.Cout + c;
#pragma unroll
                for (int i = 0; i < 4; ++i)
#pragma unroll
                    for (int j = 0; j < 4; ++j) m[i][j] = f4_add(m[i][j], *reinterpret_cast<const float4*>(mp2 + (4 * i + j) * a.ptm2));
            }
            float4 s[2][4];
#pragma unroll
            for (int j = 0; j < 4; ++j) {
                s[0][j] = f4_add(f4_add(m[0][j], m[1][j]), m[2][j]);
                s[1][j] = f4_sub(f4_sub(m[1][j], m[2][j]), m[3][j]);
            }
            float4 y[2][2];
#pragma unroll
            for (int i = 0; i < 2; ++i) {
                y[i][0] = f4_add(f4_add(s[i][0], s[i][1]), s[i][2]);
                y[i][1] = f4_sub(f4_sub(s[i][1], s[i][2]), s[i][3]);
            }
            float4 bias = make_float4(0.f, 0.f, 0.f, 0.f), sc = make_float4(1.f, 1.f, 1.f, 1.f), sh = bias;
            if (a.bias) bias = *reinterpret_cast<const float4*>(a.bias + c);
            if ((EPI == W_AFFINE_ACT || (EPI == W_GRU_OUT && a.out2)) && a.scale) {
                sc = *reinterpret_cast<const float4*>(a.scale + c);
                sh = *reinterpret_cast<const float4*>(a.shift + c);
            }
            const float scv[4] = {sc.x, sc.y, sc.z, sc.w}, shv[4] = {sh.x, sh.y, sh.z, sh.w};
#pragma unroll
            for (int i = 0; i < 2; ++i)
#pragma unroll
                for (int j = 0; j < 2; ++j) {
                    const int yy = 2 * th + i, xx = 2 * tw + j;
                    const long long orow = prow + (long long)yy * a.W + xx;
                    float v[4] = {y[i][j].x + bias.x, y[i][j].y + bias.y, y[i][j].z + bias.z, y[i][j].w + bias.w};
                    if (a.residual) {
                        const float4 rr = *reinterpret_cast<const float4*>(a.residual + orow * a.Cout + c);
                        v[0] += rr.x; v[1] += rr.y; v[2] += rr.z; v[3] += rr.w;
                    }
                    float4 yv;                                       // what the next convolution reads
                    if constexpr (EPI == W_AFFINE_ACT) {
#pragma unroll
                        for (int k = 0; k < 4; ++k) {
                            const float u = fmaf(v[k], scv[k], shv[k]);
                            v[k] = u > 0.f ? u : u * a.slope;
                        }
                        yv = make_float4(v[0], v[1], v[2], v[3]);
                        if (own && a.out) *reinterpret_cast<float4*>(a.out + orow * a.ldo + c) = yv;
                    } else if constexpr (EPI == W_GRU_GATES) {
                        float g[4];
#pragma unroll
                        for (int k = 0; k < 4; ++k) g[k] = 1.f / (1.f + expf(-v[k]));
                        if (part == 1) {                             // update gate z (own tiles only)
                            *reinterpret_cast<float4*>(a.out + orow * Ch + cy) = make_float4(g[0], g[1], g[2], g[3]);
                            continue;
                        }
                        const float4 h = *reinterpret_cast<const float4*>(a.aux_h + orow * Ch + cy);
                        yv = make_float4(h.x * g[0], h.y * g[1], h.z * g[2], h.w * g[3]);
                        if (own) {
                            if (a.out2) *reinterpret_cast<float4*>(a.out2 + orow * Ch + cy) = yv;
                            if (a.out3) *reinterpret_cast<float4*>(a.out3 + orow * Ch + cy) = make_float4(g[0], g[1], g[2], g[3]);
                        }
                    } else {                                         // W_GRU_OUT
                        const float4 z4 = *reinterpret_cast<const float4*>(a.aux_z + orow * a.Cout + c);
                        const float4 h4 = *reinterpret_cast<const float4*>(a.aux_h + orow * a.Cout + c);
                        const float zv[4] = {z4.x, z4.y, z4.z, z4.w}, hv[4] = {h4.x, h4.y, h4.z, h4.w};
                        float cand[4], hn[4];
#pragma unroll
                        for (int k = 0; k < 4; ++k) {
                            cand[k] = tanhf(v[k]);
                            hn[k] = hv[k] * (1.f - zv[k]) + cand[k] * zv[k];
                        }
                        yv = make_float4(hn[0], hn[1], hn[2], hn[3]);
                        if (own) {
                            *reinterpret_cast<float4*>(a.out + orow * a.ldo + c) = yv;
                            if (a.out2)
                                *reinterpret_cast<float4*>(a.out2 + orow * a.ldo + c) =
                                    make_float4(fmaf(hn[0], scv[0], shv[0]), fmaf(hn[1], scv[1], shv[1]), fmaf(hn[2], scv[2], shv[2]), fmaf(hn[3], scv[3], shv[3]));
                            if (a.out3) *reinterpret_cast<float4*>(a.out3 + orow * a.ldo + c) = make_float4(cand[0], cand[1], cand[2], cand[3]);
                        }
                    }
                    const int lr = yy - ybase;                       // halo tiles contribute only their row that the band's patches reach
                    if (lr >= 0 && lr < nrows) *reinterpret_cast<float4*>(ylds + ((long long)lr * a.W + xx) * WOI_CK + 4 * q) = yv;
                }
        }
    }
    __syncthreads();
    // ---- phase 2: V_next tiles of the band
    const int ntask2 = TB * Wt * (WOI_CK / 4);
    for (int task = threadIdx.x; task < ntask2; task += WOI_THREADS) {
        const int q = task % (WOI_CK / 4), tt = task / (WOI_CK / 4);
        const int tw = tt % Wt, th = th0 + tt / Wt;
        float4 d[4][4];
#pragma unroll
        for (int i = 0; i < 4; ++i) {
            const int yy = 2 * th - 1 + i;
#pragma unroll
            for (int j = 0; j < 4; ++j) {
                const int xx = 2 * tw - 1 + j;
                const bool ok = (unsigned)yy < (unsigned)a.H && (unsigned)xx < (unsigned)a.W;
                d[i][j] = ok ? *reinterpret_cast<const float4*>(ylds + ((long long)(yy - ybase) * a.W + xx) * WOI_CK + 4 * q) : make_float4(0.f, 0.f, 0.f, 0.f);
            }
        }
        float4 w[4][4];
#pragma unroll
        for (int j = 0; j < 4; ++j) {
            w[0][j] = f4_sub(d[0][j], d[2][j]);
            w[1][j] = f4_add(d[1][j], d[2][j]);
            w[2][j] = f4_sub(d[2][j], d[1][j]);
            w[3][j] = f4_sub(d[1][j], d[3][j]);
        }
        const long long r = plane * Ht * Wt + (long long)th * Wt + tw;
        float* vp = aa.Vn + r * aa.ldvn + c0 + 4 * q;
#pragma unroll
        for (int i = 0; i < 4; ++i) {
            *reinterpret_cast<float4*>(vp + (4 * i + 0) * aa.ptvn) = f4_sub(w[i][0], w[i][2]);
            *reinterpret_cast<float4*>(vp + (4 * i + 1) * aa.ptvn) = f4_add(w[i][1], w[i][2]);
            *reinterpret_cast<float4*>(vp + (4 * i + 2) * aa.ptvn) = f4_sub(w[i][2], w[i][1]);
            *reinterpret_cast<float4*>(vp + (4 * i + 3) * aa.ptvn) = f4_sub(w[i][1], w[i][3]);
        }
    }
    (void)R;
}


// forge_wino_output + forge_wino_input of the result in ONE launch (wino_output_input_kernel): the arguments of forge_wino_output, then the next
// convolution's transformed input Vn [16][R][ldvn] (ptvn = 0: dense R x ldvn). y - the tensor that is transformed - is `out` (epilogue 1),
// out2 = h * r (epilogue 2) or `out` = the new state (epilogue 3); outputs nothing else reads may be NULL: out (epilogue 1), out2 (epilogue 2).
extern "C" int forge_wino_output_input(const float* Mm, const float* Mm2, long long bs2, long long pt2, const float* bias, const float* scale, const float* shift,
                                       float slope, const float* residual, const float* aux_h, const float* aux_z, float* out, float* out2, float* out3,
                                       int n, int D, int H, int W, int Cout, int ldo, int epilogue, float* Vn, int ldvn, long long ptvn, forge_stream_t stream) {
    FORGE_REQUIRE(Mm && Vn, FORGE_EINVAL, "forge_wino_output_input: null pointer argument");
    FORGE_REQUIRE(epilogue >= 1 && epilogue <= 3, FORGE_EINVAL, "forge_wino_output_input: epilogue must be 1 (affine + activation), 2 (GRU gates) or 3 (GRU state)");
    const int CY = epilogue == W_GRU_GATES ? Cout / 2 : Cout;
    FORGE_REQUIRE(n > 0 && D > 0 && H > 0 && W > 0 && H % 2 == 0 && W % 2 == 0 && Cout > 0 && CY % WOI_CK == 0 && ldo % 4 == 0 && ldvn >= CY && ldvn % 4 == 0,
                  FORGE_ESHAPE, "forge_wino_output_input: n=%d D=%d H=%d W=%d Cout=%d ldo=%d ldvn=%d (H, W even; transformed channels a multiple of %d)", n, D, H, W,
                  Cout, ldo, ldvn, WOI_CK);
    FORGE_REQUIRE(epilogue != W_AFFINE_ACT || (scale && shift), FORGE_EINVAL, "forge_wino_output_input: affine epilogue needs scale/shift");
    FORGE_REQUIRE(epilogue != W_GRU_GATES || (aux_h && out), FORGE_EINVAL, "forge_wino_output_input: GRU gate epilogue needs aux_h and out (z)");
    FORGE_REQUIRE(epilogue != W_GRU_OUT || (aux_h && aux_z && out && (!out2 || (scale && shift))), FORGE_EINVAL,
                  "forge_wino_output_input: GRU state epilogue needs aux_h, aux_z, out (and scale/shift with out2)");
    FORGE_REQUIRE(out3 == nullptr || epilogue != W_AFFINE_ACT, FORGE_EINVAL, "forge_wino_output_input: out3 is a GRU-epilogue output");
    WinoOutInArgs aa;
    WinoOutArgs& a = aa.o;
    const int Ht = H / 2, Wt = W / 2;
    const long long R = (long long)n * D * Ht * Wt;
    a.Mm = Mm; a.ptm = R * Cout; a.Mm2 = Mm2; a.bs2 = bs2 > 0 ? bs2 : (long long)D * Ht * Wt; a.ptm2 = pt2 > 0 ? pt2 : R * Cout; a.bias = bias; a.scale = scale;
    a.shift = shift; a.slope = slope; a.residual = residual; a.aux_h = aux_h; a.aux_z = aux_z; a.out = out; a.out2 = out2; a.out3 = out3; a.ldo = ldo;
    a.n = n; a.D = D; a.H = H; a.W = W; a.Cout = Cout; a.epi = epilogue;
    aa.Vn = Vn; aa.ldvn = ldvn; aa.ptvn = ptvn > 0 ? ptvn : R * ldvn; aa.CY = CY;
    FORGE_REQUIRE(R < (1ll << 31), FORGE_ESHAPE, "forge_wino_output_input: more than 2^31 tiles; split the batch");
    // band height: the largest divisor of Ht whose (2 TB + 2) image rows x W pixels x 16 channels fit 38 KB of LDS (four workgroups per CU; H = W = 64: 8 tile rows)
    int TB = Ht;
    while (TB > 1 && ((size_t)(2 * TB + 2) * W * WOI_CK * sizeof(float) > 38 * 1024 || Ht % TB)) --TB;
    const size_t lds = (size_t)(2 * TB + 2) * W * WOI_CK * sizeof(float);
    FORGE_REQUIRE(lds <= 150 * 1024, FORGE_ESHAPE, "forge_wino_output_input: W=%d too wide for the LDS band", W);
    aa.TB = TB;
    const long long grid = (long long)n * D * (Ht / TB) * (CY / WOI_CK);
    FORGE_REQUIRE(grid < (1ll << 31), FORGE_ESHAPE, "forge_wino_output_input: grid too large");
    const dim3 g((unsigned)grid), b(WOI_THREADS);
    hipStream_t st = (hipStream_t)stream;
    switch (epilogue) {
        case W_AFFINE_ACT:
            FORGE_SET_MAX_LDS_ONCE((wino_output_input_kernel<W_AFFINE_ACT>), 150 * 1024);
            hipLaunchKernelGGL(wino_output_input_kernel<W_AFFINE_ACT>, g, b, lds, st, aa); break;
        case W_GRU_GATES:
            FORGE_SET_MAX_LDS_ONCE((wino_output_input_kernel<W_GRU_GATES>), 150 * 1024);
            hipLaunchKernelGGL(wino_output_input_kernel<W_GRU_GATES>, g, b, lds, st, aa); break;
        default:
            FORGE_SET_MAX_LDS_ONCE((wino_output_input_kernel<W_GRU_OUT>), 150 * 1024);
            hipLaunchKernelGGL(wino_output_input_kernel<W_GRU_OUT>, g, b, lds, st, aa); break;
    }
    FORGE_LAUNCH_CHECK("forge_wino_output_input");
    return 0;
}
