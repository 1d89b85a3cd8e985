/*
 * forge_hip.h — C-ABI of libforge_hip.so, the MI355X (gfx950) kernels behind the FORGE
 * reconstruction hot path (UT-Austin-RPL/FORGE: models/rotate.py, models/volume_render.py,
 * models/fusion.py, models/encoder.py).
 *
 * The reference is pure Python/PyTorch (no FFI of its own); the drop-in boundary for callers is
 * the Python nn.Module surface in forge_amd/ (same class/method/state_dict names as
 * models/model.py etc.).  This header is the boundary ONE level lower: what forge_amd's host
 * code (or any other host language) binds.  Each entry point cites the
 * reference call site it replaces.
 *
 * Conventions
 *   - plain C types only: device pointers (const float*), ints, floats, an opaque stream handle
 *     (hipStream_t passed as void*; NULL = the default stream).  No torch types.
 *   - every function returns 0 on success, a positive hipError_t when a HIP call failed, or a
 *     negative FORGE_E* code for a rejected argument; forge_last_error() gives the message
 *     (thread-local).
 *   - functions are re-entrant, allocate nothing and keep no global state: all buffers
 *     (including workspaces) are owned by the caller.
 *   - launches are asynchronous on `stream`; nothing synchronises.
 *   - volumes are CHANNELS-LAST fp32: vol[n][D][H][W][C] (torch: an NCDHW tensor in
 *     torch.channels_last_3d memory format); grid axes (x,y,z) <-> tensor axes (W,H,D)
 *     exactly as in F.grid_sample / pytorch3d Volumes.
 */
#ifndef FORGE_HIP_H
#define FORGE_HIP_H

#ifdef __cplusplus
extern "C" {
#endif

typedef void* forge_stream_t; /* hipStream_t */

#define FORGE_EINVAL (-1)   /* bad argument value (null pointer, non-positive dim, ...) */
#define FORGE_ESHAPE (-2)   /* unsupported shape (e.g. C not a multiple of 4)            */

/* Library version: major*10000 + minor*100 + patch. */
int forge_version(void);
/* Message for the last non-zero return on this thread ("" if none). */
const char* forge_last_error(void);

/* ---------------------------------------------------------------------------------------------
 * a2  voxel-grid pose warp — replaces models/rotate.py:127-141 (affine grid + F.grid_sample
 * trilinear, zeros padding, align_corners=FALSE, then cat with view 0).
 *
 *   vox  [n][D][H][W][C]   input volumes (all views of all scenes, n = B*t)
 *   xf   [n][12]           row-major 3x4 affine in NORMALISED grid coords: for the output voxel
 *                          with g = (gx,gy,gz), g_a = 2 i_a/(N_a-1) - 1, the sample coordinate is
 *                          s = A g + b  (A = R_T, b = t_T / grid_coord_max; rotate.py:132-135)
 *   mode [n]               0: copy the volume unchanged (view 0, rotate.py:141); 1: warp
 *   out  [n][D][H][W][C]
 * pixel coord p_a = ((s_a+1) N_a - 1)/2 (align_corners=False), 8 taps each zeroed when out of range.
 * Requires C % 4 == 0.
 */
int forge_rotate_fwd(const float* vox, const float* xf, const int* mode, float* out,
                     int n, int C, int D, int H, int W, forge_stream_t stream);

/* forge_rotate_fwd with the view ordering of models/model.py:127-128 (`chose_selected(features_transformed, idxs)`) fused into the
 * store: input volume i is written to output volume dst_slot[i] (a permutation of 0..n-1 on the device) instead of i - saves the
 * gather copy of all warped volumes (84 MB read + written per 5-view scene). Inference only (no backward counterpart). */
int forge_rotate_fwd_slots(const float* vox, const float* xf, const int* mode, const int* dst_slot, float* out,
                           int n, int C, int D, int H, int W, forge_stream_t stream);

/* models/rotate.py:64-89,132-135 on the device: poses [B][t][4][4] (row-major camera poses, view 0 = reference) ->
 * xf [B*t][12] = [R_T | t_T / half_extent] with T = P_0 P_i^-1 (general 4x4 inverse), mode [B*t] = (0,1,1,...).
 * Feeds forge_rotate_fwd without any host round trip. (Gradients w.r.t. poses go through the host-side torch
 * algebra instead, see forge_amd/rotate.py.)
 * slot (nullable) [B*t]: the view order of models/model.py:152-158 (`sequence_from_distance`: views sorted by squared distance of their
 * camera position to view 0's, ties by view index = a stable sort) as the dst_slot array of forge_rotate_fwd_slots:
 * slot[b t + i] = b t + rank of view i. dist (nullable) [B*t]: the sort keys, when the caller wants its own distance values ranked
 * (the torch host passes sequence_from_distance's: symmetric camera rigs produce near-ties whose order must not depend on who
 * rounded the sum of squares); NULL: computed here as ((dx^2 + dy^2) + dz^2) without fused multiply-adds.
 */
int forge_rotate_xf_from_poses(const float* poses, float* xf, int* mode, int* slot, const float* dist, int B, int t, float half_extent,
                               forge_stream_t stream);

/* Pose chain of the refinement loop (kubric_eval.py:412-530 `do_refinement`, demo.py:115-188): from the optimised relative poses of the
 * non-reference views - rot [b (t-1)][4] raw quaternions (w, x, y, z; normalised here as F.normalize + utils/geo_utils.py:122-137 do),
 * trans [b (t-1)][3] - to both kernel operands in one launch:
 *   poses [b t][16]  canonical_pose @ [R(q) | trans] (view 0: canonical_pose itself)            utils/geo_utils.py:268-287
 *   xf [b t][12], mode [b t]   the warp's affine [R_T | t_T / half_extent], T = P_0 P_i^-1       models/rotate.py:64-89,132-135
 *   slot [b t]       (nullable) the view order of models/model.py:152-158 as forge_rotate_fwd_slots' dst_slot (see forge_rotate_xf_from_poses)
 *   cam16 [b t][16]  the ray-marcher's cameras [R | T | fx fy cx cy] from the extrinsics P_i^-1 and K / 2   models/volume_render.py:40-51
 *   origin [b t][2]  (nullable) projection of the world origin (models/volume_render.py:77-79)
 *   jac [b (t-1)][24][7] (nullable) d(xf[0..11], cam16[0..11]) / d(quaternion, translation), by forward-mode differentiation in the kernel
 * can_pose / can_extr: row-major 4x4 pose / extrinsics of the reference view; K [b t][9] full-resolution intrinsics.
 * forge_pose_chain_bwd: drot [b (t-1)][4], dtrans [b (t-1)][3] = jac^T (dxf row | dcam row[0..11]) (dxf [b t][12] from forge_rotate_bwd,
 * dcam [b t][16] from forge_render_bwd; either nullable). Replaces ~250 torch launches of 3-8 us per refinement iteration. */
int forge_pose_chain_fwd(const float* rot, const float* trans, const float* can_pose, const float* can_extr, const float* K, float half_extent,
                         int b, int t, float* xf, int* mode, int* slot, float* cam16, float* poses, float* origin, float* jac, forge_stream_t stream);
int forge_pose_chain_bwd(const float* jac, const float* dxf, const float* dcam, float* drot, float* dtrans, int b, int t, forge_stream_t stream);

/* Backward of forge_rotate_fwd w.r.t. the volumes (and optionally the affine).
 *   dout [n][D][H][W][C]   upstream gradient
 *   dvox [n][D][H][W][C]   nullable (frozen volumes: pose refinement wants dxf only); written (not accumulated): the exact transpose of
 *                          the warp, computed as a gather per source voxel (deterministic, no atomics); mode 0 volumes: plain copy of dout
 *   dxf  [n][12]           nullable; MUST be zero-filled; d loss / d xf (pose refinement,
 *                          kubric_eval.py:469-503). Needs vox (nullable when dxf is NULL).
 */
int forge_rotate_bwd(const float* dout, const float* vox, const float* xf, const int* mode,
                     float* dvox, float* dxf, int n, int C, int D, int H, int W,
                     forge_stream_t stream);
/* The same for a forward that stored view i at volume slot[i] (forge_rotate_fwd_slots): view i's upstream gradient is read at dout[slot[i]]. */
int forge_rotate_bwd_slots(const float* dout, const float* vox, const float* xf, const int* mode, const int* src_slot,
                           float* dvox, float* dxf, int n, int C, int D, int H, int W, forge_stream_t stream);

/* ---------------------------------------------------------------------------------------------
 * a6  fused ray sampler + volume sampler + emission-absorption ray-marcher — replaces
 * models/volume_render.py:53-63 (cameras_from_opencv_projection, NDCGridRaysampler,
 * VolumeRenderer/VolumeSampler with align_corners=TRUE trilinear zero-pad sampling,
 * EmissionAbsorptionRaymarcher incl. the README.md:26-33 depth patch).
 *
 *   feat     [nvol][D][H][W][C]   render features, channels-last, C % 4 == 0, C <= 32
 *   dens     [nvol][D][H][W]      densities (>= 0, NOT clamped <= 1: models/model.py:140-141)
 *   cam      [V][16]              per view: R[9] row-major (OpenCV world->cam), T[3], fx, fy, cx, cy
 *                                 of the HALF-resolution intrinsics (volume_render.py:50-51)
 *   view2vol [V]                  volume index rendered by each view (replaces the V-fold
 *                                 `repeat` of the volumes, models/model.py:138-139)
 *   out_feat [V][Hr][Wr][C]       sum_s w_s f_s           (channels-last, what the conv_rgb GEMM consumes)
 *   out_opac [V][Hr][Wr]          1 - prod_s (1 - d_s)
 *   out_depth[V][Hr][Wr]          sum_s w_s z_s, nullable (render_depth=False)
 * Ray (h,w): d_cam = ((w+.5-cx)/fx, (h+.5-cy)/fy, 1); p_s = -R^T T + R^T d_cam z_s,
 * z_s = zmin + s (zmax-zmin)/(S-1); local = p / (hx,hy,hz); pix = (local+1)/2 (N-1).
 */
int forge_render_fwd(const float* feat, const float* dens, const float* cam, const int* view2vol,
                     float* out_feat, float* out_opac, float* out_depth,
                     int V, int nvol, int C, int D, int H, int W, int Hr, int Wr, int S,
                     float zmin, float zmax, float hx, float hy, float hz,
                     forge_stream_t stream);

/* Camera dict of models/volume_render.py:40-51 ({'R' [V,3,3], 'T' [V,3], 'K' [V,3,3]}, element strides given: the tensors are usually
 * slices of [V,4,4] extrinsics) -> cam [V][16] as forge_render_fwd takes it, with the reference's half-resolution intrinsics
 * (K / 2, K[2][2] = 1, on a copy), and (origin nullable) the screen projection of the world origin of :77-79,
 * (fx tx / tz + cx, fy ty / tz + cy). One launch instead of a dozen tensor ops per forward; inference only (the differentiable form
 * for camera gradients is host-side torch algebra, forge_amd/volume_render.py). */
int forge_pack_cameras(const float* R, long long r0, long long r1, long long r2, const float* T, long long t0, long long t1,
                       const float* K, long long k0, long long k1, long long k2, float* cam16, float* origin, int V, forge_stream_t stream);

/* Backward of forge_render_fwd w.r.t. volumes (and optionally cameras) - what autograd through PyTorch3D's VolumeSampler /
 * EmissionAbsorptionRaymarcher computes for models/volume_render.py:63. Deterministic (no atomics): a ray-parallel pass leaves the two
 * per-sample scalars (dL/dd_s, T_s d_s) in the workspace, a voxel-parallel gather sums every voxel's tap contributions in a fixed order.
 *   g_feat [V][Hr][Wr][C], g_opac [V][Hr][Wr], g_depth [V][Hr][Wr] (nullable)
 *   dfeat  [nvol][D][H][W][C], ddens [nvol][D][H][W]   WRITTEN (every element; no zero-fill needed, nothing accumulated)
 *   dcam   [V][16] nullable, WRITTEN: d loss / d (R, T, fx, fy, cx, cy)
 *   ws     caller-owned scratch of at least forge_render_bwd_ws_bytes(V, C, Hr, Wr, S, dcam != NULL) bytes, 16-byte aligned
 *          (8 V Hr Wr S bytes; when dcam is requested + 64 bytes per 8 x (64 / (C/4)) pixel tile and view + 24 V Hr Wr S bytes of per-sample
 *          position-gradient scalars that the ray pass's forward march leaves for its own backward sweep); no allocation inside.
 */
long long forge_render_bwd_ws_bytes(int V, int C, int Hr, int Wr, int S, int want_cam);     /* < 0: unsupported arguments */
int forge_render_bwd(const float* feat, const float* dens, const float* cam, const int* view2vol,
                     const float* g_feat, const float* g_opac, const float* g_depth,
                     float* dfeat, float* ddens, float* dcam,
                     int V, int nvol, int C, int D, int H, int W, int Hr, int Wr, int S,
                     float zmin, float zmax, float hx, float hy, float hz,
                     void* ws, long long ws_bytes, forge_stream_t stream);

/* a7  mask / depth up-sampling (models/volume_render.py:69,74: F.upsample(size=img_size, mode='bilinear') = align_corners False): P planes
 * [Hi][Wi] -> [Ho][Wo] with ATen's source-index / weight arithmetic; _bwd is the adjoint (a deterministic gather per input pixel, written). */
int forge_resize_bilinear_fwd(const float* in, float* out, int P, int Hi, int Wi, int Ho, int Wo, forge_stream_t stream);
int forge_resize_bilinear_bwd(const float* g, float* din, int P, int Hi, int Wi, int Ho, int Wo, forge_stream_t stream);

/* ---------------------------------------------------------------------------------------------
 * a1/a4/a5  implicit-GEMM convolution on the fp32 matrix cores (v_mfma_f32_32x32x2_f32, exact fp32)
 * with fused element-wise tails — replaces the cuDNN convolutions + separate element-wise ops of
 * models/fusion.py:29-35 (ConvGRU cell: cat, conv, split, sigmoid, mul, cat, conv, tanh, lerp),
 * models/fusion.py:61-68 (fusion_conv), models/encoder.py:36-40 (conv1) and :16-34 (heads; the
 * ConvTranspose3d k4 s2 p1 runs as 8 output-phase GEMMs of 8 taps each).
 *
 *   rows                        the GEMM rows enumerate an (n,D,H,W) grid, M = n D H W
 *   in1 / in2                   channels-last inputs on an (n,Di,Hi,Wi) grid; row of GEMM-grid voxel
 *                               (z,y,x) and tap (dz,dy,dx) is input voxel (z is+dz, y is+dy, x is+dx),
 *                               zero outside the grid. The conv sees C1 channels of in1 followed by C2 of
 *                               in2 (in2 NULL <=> C2 = 0); ld1/ld2 = row strides in floats (>= C1/C2, so a
 *                               channel slice of a wider tensor can be fed). C1, C2 % 32 == 0. bs1/bs2 = batch
 *                               strides in rows (0 = dense Di Hi Wi): lets view t of a [b][t] stack be fed in place.
 *   wp  [ntaps][Cout][C1+C2]    packed weights: wp[t][co][ci] multiplies in[voxel + taps[t]][ci]
 *   taps [ntaps][3]             HOST array of (dz,dy,dx) input offsets, ntaps <= 27
 *   out rows                    output voxel of GEMM-grid voxel (z,y,x) is (z os+pz, y os+py, x os+px)
 *                               in an (n,Do,Ho,Wo) grid; row stride ldo floats.
 *                               plain conv: is=os=1, Di=Do=D..; strided conv: is=2; ConvTranspose phase: os=2.
 *                               pz = py = px = -1 with os = 2: ALL output phases of a stride-2 transposed convolution in one
 *                               launch - taps / wp hold the phases back to back (8 phases for Do = 2D, 4 for a 2-D grid with
 *                               Do = D; phase p = pz*4 + py*2 + px uses taps [p*ntaps/P, (p+1)*ntaps/P)); no split-K.
 *   epilogue 0  out = acc + bias
 *            1  out = lrelu((acc + bias) * scale + shift + residual, slope)   eval-BN folded; residual
 *               [rows][ldo] nullable (ResNet shortcut); slope 1 = identity, 0 = ReLU
 *            2  GRU gates (Cout = 2 Ch): out[m][c] = sigmoid(v_c), c < Ch (update);
 *                                        out2[m][c] = aux_h[m][c] * sigmoid(v_{Ch+c})  (h * reset)
 *            3  GRU state: hn = aux_h (1 - aux_z) + tanh(v) aux_z; out = hn;
 *                          out2 (nullable) = hn * scale + shift          (fusion_norm on the last step)
 *            epilogues 2 / 3 add `residual` [rows][Cout] (nullable) to the pre-activation: the input half conv(x, W_x) of
 *            conv([x, h], W) = conv(x, W_x) + conv(h, W_h), computed once per view when several fusions share views.
 *   out3 (nullable, epilogues 2 / 3 only): what a hand-written backward of the fused cell needs besides out / out2 -
 *            epilogue 2: the reset gate sigmoid(v_{Ch+c}) [M][Ch]; epilogue 3: the candidate tanh(v) [M][Cout]
 *            (pose refinement with frozen weights runs the fused epilogues in forward, forge_amd/fusion.py).
 * bias/scale/shift [Cout]; bias nullable.
 *   lift > 0 (epilogue 1, 2-D conv i.e. D = 1): fuses the 2D->3D feature lift of models/encoder.py:49 into the store —
 *            GEMM column j = z*(Cout/lift) + c of row (n, h, w) is written to out[n][z][h][w][c] (a channels-last
 *            (n, lift, H, W) volume with Cout/lift channels). The caller orders the weight rows accordingly.
 *   tile / ksplit: the launch plan. tile = 0: planned inside the call (forge_conv_igemm_plan's model; split-K only if splitk_ws is given).
 *            tile = 'A'..'E' with ksplit >= 1: the caller's plan, taken verbatim (normally forge_conv_igemm_plan's answer, so that the
 *            caller can size splitk_ws to ksplit M Cout floats; also how tools / tests pin a tile). Ignored for Cout <= 16.
 *   splitk_ws (nullable, splitk_ws_bytes): scratch for split-K. When the M x Cout tile grid alone cannot fill the chip
 *            (< 512 workgroups, e.g. ResNet layers at M = 5120) the tap x channel reduction is sliced over up to 8 workgroups
 *            per tile; raw partial tiles go to splitk_ws[slice][M][Cout] and a second kernel sums them in a fixed order and
 *            applies the epilogue (epilogues 0 and 1 only). NULL disables it. Results are deterministic either way.
 *   stats (nullable; epilogue 0 of the wide kernel, plain output mapping, no split-K): the column sums and sums of squares of the OUTPUT per
 *            32-row block of M, float64, stats[block][2][Cout] for ceil(M / BM) BM / 32 blocks, BM = the tile's rows (128 for tiles A, C, E; 64 for B, D;
 *            every block is written, blocks past M with partial / zero sums) - the
 *            batch statistics of the BatchNorm behind the convolution as a by-product of the epilogue (fixed order, no atomics):
 *            forge_bn_train_fwd / forge_bn_sync_stats take them as their partial sums (nblk_pre) instead of re-reading the activation.
 *            The block count depends on the tile, and a split-K plan skips the epilogue: a call with stats != NULL must pass an explicit
 *            `tile` ('A'..'E', forge_conv_igemm_plan's answer) and ksplit <= 1 - tile = 0 (planned inside) is refused with FORGE_EINVAL.
 *   M = n D H W must be < 2^31.
 */
int forge_conv_igemm(const float* in1, int C1, int ld1, long long bs1, const float* in2, int C2, int ld2, long long bs2,
                     const float* wp,
                     const float* bias, const float* scale, const float* shift, float slope, const float* residual,
                     const float* aux_h, const float* aux_z, float* out, float* out2, float* out3,
                     int n, int D, int H, int W, int is, int Di, int Hi, int Wi, int Cout, int ldo,
                     const int* taps, int ntaps, int os, int pz, int py, int px, int Do, int Ho, int Wo,
                     int epilogue, int lift, int tile, int ksplit, float* splitk_ws, long long splitk_ws_bytes, double* stats, forge_stream_t stream);

/* The launch plan forge_conv_igemm will use for a problem (M = n*D*H*W GEMM rows, Cout, Cin = C1 + C2, ntaps): *tile gets the
 * workgroup tile ('A' 128x128, 'B' 64x128, 'C' 128x64, 'D' 64x64, 'E' 128x32 output rows x channels; 'N' = the Cout <= 16 kernel),
 * *ksplit the number of K-slices (1 = no split-K); nphase = 1, or 4 / 8 for a merged-phase transposed-conv launch (M rows and ntaps / nphase
 * taps per phase). Pure host arithmetic (a makespan model of the 256-CU chip), no launch; lets a
 * caller size the split-K workspace (ksplit * M * Cout floats) and lets profilers attribute launches to kernel instantiations. */
int forge_conv_igemm_plan(long long M, int Cout, int Cin, int ntaps, int nphase, int epilogue, int ldo, long long splitk_ws_bytes,
                          int* tile, int* ksplit);

/* ---------------------------------------------------------------------------------------------
 * a4 (inference)  the stride-1 3x3x3 convolutions of the ConvGRU fusion (models/fusion.py:29-35, 61-68, 88-95) as Winograd
 * F(2x2, 3x3) over (H, W) with the three depth taps summed directly: 12 instead of 27 multiplies per output and channel pair.
 *   y[z, 2th+i, 2tw+j] = sum_kd A^T [ U[kd] (.) (B^T d[z+kd-1] B) ] A,   U[kd] = G w[kd] G^T,   d = the 4x4 patch at (2th-1.., 2tw-1..)
 * Three launches per convolution, all on channels-last rows, tile rows r = ((n D + z) H/2 + th) W/2 + tw, point p = 4 i + j:
 *   forge_wino_input   V[p][r][c] = (B^T d B)[p] for the C channels of in (rows [n][D][H][W] x ld floats, batch stride bs rows,
 *                      0 = dense); V[p] starts at V + p ptv floats (0 = dense R x ldv), rows of ldv floats. H, W even; C % 4 == 0.
 *   forge_wino_gemm    Mm[p][r][co] = sum_kd sum_ci U[p][kd][co][ci] (V1 | V2)[p][r + kd plane][ci]  for the 16 points in ONE launch of the
 *                      fp32-MFMA implicit-GEMM kernel (forge_conv_igemm's: 3 depth taps over the (n, D, Ht, Wt) tile grid, K = 3 (C1+C2));
 *                      V1 / V2: channel-concatenated operands (V2 nullable with C2 = 0) with row strides ld, batch strides bs rows
 *                      (0 = dense) and point strides pt floats; U [16][3][Cout][C1+C2]; Mm [16][R][Cout] dense. C1, C2 % 32 == 0.
 *   forge_wino_output  y = A^T (Mm + Mm2) A per tile, then forge_conv_igemm's epilogue 0..3 with the same operands (bias, scale/shift/slope,
 *                      residual [rows][Cout] added to the pre-activation, aux_h / aux_z, out / out2 / out3; out rows of ldo floats).
 *                      Mm2 (nullable): a second set of point products with batch stride bs2 rows and point stride pt2 floats (0 = as Mm) -
 *                      the input half conv(x, W_x) of conv([x, h], W), computed once per view when several fusions share views.
 *   forge_wino_weights U [16][kd][Cout][Cin] = G w[kd] G^T from forge_conv_igemm's packed weights wp [9 kd][Cout][Cin] (float64 inside, rounded
 *                      once); transpose != 0: the weights of the DATA GRADIENT of that convolution instead, U [16][kd][Cin][Cout]
 *                      (dx = the Winograd convolution of dy with them) - a few microseconds, so training re-derives U every step.
 * kd = 3: 3x3x3 kernels (three depth taps summed inside the point GEMMs); kd = 1: the 3x3 kernels of a 2-D convolution (ResNet
 * bottlenecks: the planes of the (n, D) grid do not mix, so images can sit on either axis).
 * B^T and A^T hold 0 / +-1 only (exact additions); U is rounded once from a float64 product. Not bit-identical to
 * forge_conv_igemm (different order of the fp32 additions); error vs a float64 convolution is ~1.4x the direct fp32 kernel's. */
int forge_wino_weights(const float* wp, float* U, int Cout, int Cin, int kd, int transpose, forge_stream_t stream);
/* Weight gradient of the same convolution in the Winograd domain (training): dMm = A dy A^T (forge_wino_dy, the adjoint of the inverse
 * transform; dM [16][R][Cout]), dU[p][kd][co][ci] = sum_r dMm[p][r][co] (V1 | V2)[p][r + kd plane][ci] (forge_wino_wgrad: the weight-gradient
 * GEMM kernel of forge_conv_wgrad on 16 batched problems, dU [16][kd][Cout][C1+C2] ZERO-FILLED by the caller, fp32 atomics over voxel chunks),
 * dw[9 kd][Cout][Cin] += G^T dU G (forge_wino_dw: accumulates, like forge_conv_wgrad). V1 / V2 as in forge_wino_gemm (row strides = channel counts). */
int forge_wino_dy(const float* dy, int ldy, float* dM, int n, int D, int H, int W, int Cout, forge_stream_t stream);
int forge_wino_wgrad(const float* dMm, const float* V1, int C1, long long bs1, long long pt1, const float* V2, int C2, long long bs2,
                     long long pt2, float* dU, int n, int D, int Ht, int Wt, int Cout, int kd, forge_stream_t stream);
int forge_wino_dw(const float* dU, float* dw, int Cout, int Cin, int kd, forge_stream_t stream);
int forge_wino_input(const float* in, int ld, long long bs, float* V, int ldv, long long ptv, int n, int D, int H, int W, int C,
                     int nsum /* >1: transform the MEAN of nsum tensors sum_stride rows apart (models/encoder.py:62 view mean) */, long long sum_stride,
                     forge_stream_t stream);
/* An upstream gradient dy [n D H W][ld] (C channels) in BOTH transformed forms with one pass over it: V = B^T dy B (as forge_wino_input,
 * for the data-gradient point GEMMs) and dM = A dy A^T (as forge_wino_dy, for forge_wino_wgrad); both [16][n D H/2 W/2][C], dense.
 * (Backward of the reference's 3x3x3 convolutions under autograd: models/fusion.py:29-35, 61-68 via scripts/kubric_trainer.py:56.) */
int forge_wino_input_dy(const float* dy, int ld, float* V, float* dM, int n, int D, int H, int W, int C, forge_stream_t stream);
int forge_wino_gemm(const float* V1, int C1, int ld1, long long bs1, long long pt1, const float* V2, int C2, int ld2, long long bs2,
                    long long pt2, const float* U, float* Mm, int n, int D, int Ht, int Wt, int Cout, int kd, int tile /* 0 = forge_wino_gemm_tile's rule */,
                    forge_stream_t stream);
/* The same pair with the inverse transform's ROW stage moved into the GEMM's epilogue (inference of models/fusion.py:29-35, 61-68 and models/encoder.py:36-40):
 * one workgroup runs the four points (i = 0..3, j) of a point column on its 64 x 128 tile and stores s0 = (m0 + m1) + m2, s1 = (m1 - m2) - m3 - the
 * operations forge_wino_output performs first, in its order - as Mm8 [2][4][R][Cout]: the point products cross HBM as 2x instead of 4x the output
 * tensor. forge_wino_output_half runs the column stage + the same fused tails on Mm8. Without a second addend the results are bitwise those of
 * forge_wino_gemm + forge_wino_output; a second addend (the shared input halves) arrives in the same 8-plane form and is added after the row stage. For the launches forge_wino_gemm_tile answers with 'B' (R >= 2048 tile rows, Cout > 64). */
int forge_wino_gemm_half(const float* V1, int C1, int ld1, long long bs1, long long pt1, const float* V2, int C2, int ld2, long long bs2,
                         long long pt2, const float* U, float* Mm8, int n, int D, int Ht, int Wt, int Cout, int kd, forge_stream_t stream);
int forge_wino_output_half(const float* Mm8, const float* Mm2_8 /* nullable second addend, 8 planes too */, long long bs2, long long pt2, const float* bias, const float* scale,
                           const float* shift, float slope, const float* residual, const float* aux_h, const float* aux_z, float* out, float* out2, float* out3,
                           int n, int D, int H, int W, int Cout, int ldo, int epilogue, forge_stream_t stream);
int forge_wino_gemm_tile(long long R, int Cout, int Cin);   /* the workgroup tile letter ('A'..'E', see forge_conv_igemm_plan) forge_wino_gemm uses for R tile rows per point, Cin = C1 + C2 */
int forge_wino_output(const float* Mm, const float* Mm2, long long bs2, long long pt2, const float* bias, const float* scale, const float* shift, float slope, const float* residual,
                      const float* aux_h, const float* aux_z, float* out, float* out2, float* out3, int n, int D, int H, int W, int Cout,
                      int ldo, int epilogue, forge_stream_t stream);

/* Weight gradient of forge_conv_igemm's convolution (training, scripts/kubric_trainer.py:56 -> torch conv backward):
 *   dw[t][co][ci] += sum_m dy[m][co] * x[voxel(m) + taps[t]][ci]      (x = channel concat of x1 | x2, zero outside the grid)
 * dy [M][ldy] is the upstream gradient of the conv output on the (n,D,H,W) row grid; x1/x2, is, Di.. as in forge_conv_igemm.
 * dw [ntaps][Cout][C1+C2] MUST be zero-filled: partial sums over voxel chunks are accumulated with fp32 atomics (the order of
 * the additions, hence the last bits, is not deterministic). With two inputs C1 must be a multiple of 128.
 * The data gradient needs no extra entry point: it is forge_conv_igemm on dy with negated taps and wp[t][ci][co] (transposed).
 */
int forge_conv_wgrad(const float* dy, int ldy, const float* x1, int C1, int ld1, long long bs1, const float* x2, int C2, int ld2,
                     long long bs2, float* dw, int n, int D, int H, int W, int is, int Di, int Hi, int Wi, int Cout,
                     const int* taps, int ntaps, forge_stream_t stream);

/* ---------------------------------------------------------------------------------------------
 * a5 / a7  direct convolution for TINY channel counts (Cin in {4, 8, 16}, Cout in 1..4): the last convolutions of the density head
 * (Conv3d(8, 1, 3), models/encoder.py:31) and of conv_rgb (Conv2d(8, 3, 5), models/volume_render.py:36). A matrix-core tile pads such
 * a layer to 16 or 32 channels on both sides (up to 128x the useful FLOPs); these are streaming problems and run on the vector ALUs.
 * Stride-1 "same" geometry on channels-last rows: in [M][ld_in], w [ntaps][Cout][Cin], taps (dz,dy,dx) inside the (n,D,H,W) grid.
 *   fwd    out[m][co] = act(bias[co] + sum_t sum_ci w[t][co][ci] in[m + tap_t][ci])     (bias nullable; act = LeakyReLU(slope):
 *          slope 1 = none, 0 = ReLU - the inference path fuses the layer's trailing ReLU)
 *   dgrad  dx[m][ci]  = sum_t sum_co w[t][co][ci] dy[m - tap_t][co]
 *   wgrad  dw[t][co][ci] += sum_m dy[m][co] x[m + tap_t][ci]                              (dw zero-filled by the caller; fp32 atomics)
 */
int forge_conv_direct_fwd(const float* in, int ld_in, const float* w, const float* bias, float slope, float* out, int ld_out,
                          int n, int D, int H, int W, int Cin, int Cout, const int* taps, int ntaps, forge_stream_t stream);
int forge_conv_direct_dgrad(const float* dy, int ld_dy, const float* w, float* dx, int ld_dx,
                            int n, int D, int H, int W, int Cin, int Cout, const int* taps, int ntaps, forge_stream_t stream);
int forge_conv_direct_wgrad(const float* dy, int ld_dy, const float* x, int ld_x, float* dw,
                            int n, int D, int H, int W, int Cin, int Cout, const int* taps, int ntaps, forge_stream_t stream);

/* ---------------------------------------------------------------------------------------------
 * a4 (training)  element-wise halves of the ConvGRU cell, models/fusion.py:29-35 under autograd. Inference fuses them into
 * forge_conv_igemm's GRU epilogues; with an autograd graph the two convolutions run with the bias epilogue and each half of the
 * cell is one kernel per direction. All arrays are channels-last rows [M][C] fp32, C % 4 == 0, g / dg are [M][2C] (update | reset).
 *   gates fwd: z = sigmoid(g[:, :C]), r = sigmoid(g[:, C:]), hr = h * r
 *   state fwd: cand = tanh(c) (written over c), hn = h (1 - z) + cand z
 *   state bwd: dh = dhn (1 - z), dz = dhn (cand - h), dc = dhn z (1 - cand^2)        (dhn rows may be strided: ld_dhn floats)
 *   gates bwd: dg = (dz z (1 - z) | dhr h r (1 - r)), dh_out = dh + dhr r  (dhr rows may be strided: ld_dhr floats; dh_out NULL = in
 *              place over dh, else rows of stride ld_dh_out - it may alias dhr, which lets the sum land beside the x-gradient half)
 */
int forge_gru_gates_fwd(const float* g, const float* h, float* z, float* r, float* hr, long long M, int C, forge_stream_t stream);
int forge_gru_state_fwd(float* c_cand, const float* h, const float* z, float* hn, long long M, int C, forge_stream_t stream);
/* dc_acc / dg_acc (acc_mode 0: unused, 1: written, 2: added to): a second copy of dc / dg for the gradient of a convolution INPUT HALF that several
 * fusions share (model_single_pose_estimator.py:108-120 fuses views (0,1,2), (3,4), (0..4) of the same features): rows of batch element n land at
 * acc + (n acc_bs + row within the volume of `vol` rows) x (C | 2C) floats, i.e. view ti of a [b][t] stack is acc = base + ti vol ld, acc_bs = t vol. */
int forge_gru_state_bwd(const float* dhn, int ld_dhn, const float* h, const float* z, const float* cand, float* dh, float* dz, float* dc,
                        long long M, int C, float* dc_acc, long long acc_bs, long long vol, int acc_mode, forge_stream_t stream);
int forge_gru_gates_bwd(const float* dz, const float* dhr, int ld_dhr, const float* h, const float* z, const float* r,
                        float* dg, float* dh, float* dh_out, int ld_dh_out, long long M, int C, float* dg_acc, long long acc_bs, long long vol,
                        int acc_mode, forge_stream_t stream);

/* Train-mode BatchNorm (+ LeakyReLU / ReLU) on channels-last rows (training path; torch.nn.BatchNorm{2,3}d(train) + activation of
 * models/fusion.py:49-58, models/encoder.py:16-40, models/volume_render.py:29-37 and the torchvision bottlenecks):
 *   fwd   batch statistics over the M rows (float64 sums), y = lrelu((x - mean) invstd gamma + beta, slope); mean / invstd [C] are stored for the
 *         backward; running_mean / running_var (nullable) are updated in place with `momentum` (unbiased variance), as nn.BatchNorm does
 *   bwd   g = dy * (pre-activation > 0 ? 1 : slope); dx = gamma invstd (g - mean(g) - xhat mean(g xhat)); dgamma = sum g xhat, dbeta = sum g
 * x / y / dy / dx rows [M][ld] fp32, C % 4 == 0; gamma / beta nullable (1 / 0); slope 1 = no activation, 0 = ReLU; ws = forge_bn_ws_doubles(C)
 * doubles of scratch (one float64 partial per reduction block, summed by a second kernel in a fixed order: deterministic, no atomics).
 * Residual form (torchvision Bottleneck tail, out = relu(bn3(conv3) + identity)): res [M][ldres] (nullable) is added before the activation
 * in the same pass; the backward then takes the forward's OUTPUT y (mask = y > 0, valid for slope >= 0) and also writes d res = g to dres
 * (nullable). num_batches_tracked (nullable, int64 on the device) is incremented by the forward's finalize step - no separate launch. */
int forge_bn_ws_doubles(int C);
/* Column sums out[c] = sum_m x[m][c] of a row-major [M][C] matrix with row stride ldx (floats): the bias gradient of a convolution (torch:
 * dy.sum over every dimension but the channels). float64 partial sums in a fixed order (deterministic); ws = forge_bn_ws_doubles(C) doubles. */
int forge_colsum(const float* x, int ldx, float* out, double* ws, long long M, int C, forge_stream_t stream);
int forge_bn_train_fwd(const float* x, int ldx, const float* gamma, const float* beta, float eps, float slope, float* y, int ldy,
                       float* mean, float* invstd, float* running_mean, float* running_var, float momentum, double* ws,
                       long long M, int C, const float* res, int ldres, long long* num_batches_tracked,
                       int nblk_pre /* > 0: ws already holds that many partial rows [2][C] - forge_conv_igemm's `stats` by-product - and the statistics pass is skipped */,
                       forge_stream_t stream);
int forge_bn_train_bwd(const float* dy, int lddy, const float* x, int ldx, const float* gamma, const float* beta, const float* mean,
                       const float* invstd, float slope, float* dx, int lddx, float* dgamma, float* dbeta, double* ws, long long M, int C,
                       const float* y, int ldy, float* dres, int lddres, forge_stream_t stream);
/* SyncBatchNorm (torch.nn.SyncBatchNorm.convert_sync_batchnorm, kubric_train_pose_3D.py:119, kubric_train_joint.py:136): the same kernels
 * with ONE all-reduce (SUM) of 2 C doubles (+ the row count) between the reduction and the apply step, issued by the caller (RCCL):
 *   forge_bn_sync_stats       ws[0 .. 2C) = this rank's (sum x, sum x^2) over its M rows          -> all-reduce -> totals, M_total
 *   forge_bn_sync_fwd_apply   mean / invstd (written) and running statistics (updated, unbiased over M_total) from the totals; y as above
 *   forge_bn_sync_bwd_reduce  ws[0 .. 2C) = this rank's (sum g, sum g xhat); dgamma / dbeta = those LOCAL sums (torch's SyncBatchNorm
 *                             does the same: DDP averages parameter gradients afterwards)           -> all-reduce -> totals
 *   forge_bn_sync_bwd_apply   dx = gamma invstd (g - totals_g / M_total - xhat totals_gx / M_total)
 * ws as for forge_bn_train_fwd (forge_bn_ws_doubles(C) doubles); totals may alias ws. M_total = 0: the all-rank row count is read from
 * device memory at totals[2 C] (a double, all-reduced with the sums: no host round trip per layer). With M_total = M and no all-reduce the four calls
 * reproduce forge_bn_train_fwd / _bwd bit for bit. */
int forge_bn_sync_stats(const float* x, int ldx, double* ws, long long M, int C, int nblk_pre /* as forge_bn_train_fwd */, forge_stream_t stream);
int forge_bn_sync_fwd_apply(const float* x, int ldx, const float* gamma, const float* beta, float eps, float slope, float* y, int ldy,
                            float* mean, float* invstd, float* running_mean, float* running_var, float momentum, const double* totals,
                            long long M_total, long long M, int C, const float* res, int ldres, long long* num_batches_tracked, forge_stream_t stream);
/* Eval-mode BatchNorm (+ residual) + LeakyReLU(slope) on channels-last rows UNDER AUTOGRAD with trainable gamma / beta (a fine-tune that keeps
 * BatchNorm layers in eval mode: torch's F.batch_norm(training=False), which the reference reaches through nn.BatchNorm*.forward):
 * mean = running_mean, invstd = 1 / sqrt(running_var + eps) are written for the backward, the running statistics are only read.
 * Backward = forge_bn_sync_bwd_reduce (dgamma, dbeta) + forge_bn_sync_bwd_apply with all-zero totals (dx = gamma invstd g: the statistics
 * do not depend on x). */
int forge_bn_eval_fwd(const float* x, int ldx, const float* gamma, const float* beta, const float* running_mean, const float* running_var,
                      float eps, float slope, float* y, int ldy, float* mean, float* invstd, long long M, int C, const float* res, int ldres,
                      forge_stream_t stream);
int forge_bn_sync_bwd_reduce(const float* dy, int lddy, const float* x, int ldx, const float* gamma, const float* beta, const float* mean,
                             const float* invstd, float slope, float* dgamma, float* dbeta, double* ws, long long M, int C,
                             const float* y, int ldy, forge_stream_t stream);
int forge_bn_sync_bwd_apply(const float* dy, int lddy, const float* x, int ldx, const float* gamma, const float* beta, const float* mean,
                            const float* invstd, float slope, float* dx, int lddx, const double* totals, long long M_total, long long M, int C,
                            const float* y, int ldy, float* dres, int lddres, forge_stream_t stream);

/* Backward of forge_conv_igemm's epilogue 1 (folded eval-BatchNorm + LeakyReLU / ReLU) for frozen-weight optimisation loops (pose
 * refinement, kubric_eval.py:412-530): dx[m][c] = dy[m][c] * scale[c] * (y[m][c] > 0 ? 1 : slope), y = the forward OUTPUT (its sign
 * is the pre-activation's for slope >= 0). Rows may be strided (ld_* floats); scale nullable (= 1). The data gradient of the
 * convolution itself is forge_conv_igemm on dx with negated taps and transposed weights. */
int forge_affine_act_bwd(const float* dy, int ld_dy, const float* y, int ld_y, const float* scale, float slope, float* dx, int ld_dx,
                         long long M, int C, forge_stream_t stream);

/* Single-head dot-product attention out = softmax(q k^T) v, fp32 on the matrix cores with the N x N matrix kept in registers (online softmax over
 * 32-key tiles). Replaces `Attention.forward` / `Attention.get_attn` + the matmul behind it (models/model_utils.py:207-229, called by
 * models/pose_estimator_3d.py:116-144 with N = 4096 tokens per volume pair) in predicted-pose INFERENCE; torch computes the same sums with the matrix
 * materialised three times. Unscaled logits, as the reference. q [B][Nq][d], k [B][Nk][d], v [.][Nk][d] with a batch stride of v_batch_rows rows
 * (0 = one v shared by all batch elements: the positional table of the cross attention), out [B][Nq][d]; d = 64, Nq and Nk multiples of 64. */
int forge_attention_fwd(const float* q, const float* k, const float* v, long long v_batch_rows, float* out, int B, int Nq, int Nk, int d,
                        forge_stream_t stream);

/* ---------------------------------------------------------------------------------------------
 * a1  ResNet stem helpers (torchvision conv1/bn1/relu/maxpool behind models/encoder.py:71-73).
 * forge_im2col_nchw: img [N][C][H][W] -> patch rows out [N*Ho*Wo][Kpad], k = (ky*kw + kx)*C + c, zeros outside the image and
 *   for k >= kh*kw*C; Ho = (H + 2 pad - kh)/stride + 1. The 7x7/s2 stem conv then runs as forge_conv_igemm with one tap and
 *   C1 = Kpad (BN + ReLU in its epilogue).
 * forge_maxpool2d_nhwc: in [N][H][W][C] -> out [N][Ho][Wo][C], window k, -inf padding (nn.MaxPool2d semantics), C % 4 == 0.
 */
int forge_im2col_nchw(const float* img, float* out, int N, int C, int H, int W, int kh, int kw, int stride, int pad,
                      int Kpad, forge_stream_t stream);
int forge_maxpool2d_nhwc(const float* in, float* out, int N, int H, int W, int C, int k, int stride, int pad,
                         forge_stream_t stream);

/* ---------------------------------------------------------------------------------------------
 * f1  squared-error sums of the reconstruction losses (scripts/kubric_compute_loss.py:26-29, 136-139: four F.mse_loss terms on
 * reshaped / repeated tensors) in one pass over the rendered maps, and their backward.
 *   pred    [B][Vp][C][H][W] addressed with element strides (sn per view, sc, sh, sw): the rgb maps are channels-last memory behind
 *           an NCHW view (conv_rgb's output), the masks plain NCHW
 *   target  [B][Vt][C][H][W] contiguous; rendered view v of scene b is compared with target view v % Vt and counted in group
 *           v / gsize (GT-pose model: Vp = 2t, Vt = gsize = t; joint model: Vp = Vt = 2t, gsize = t); at most 4 groups
 *   fwd     partial [forge_sse_groups_blocks()][G]: per-workgroup sums of (pred - target)^2 (fixed reduction tree; the caller adds
 *           the rows in a fixed order: deterministic, no atomics)
 *   bwd     dpred (same layout as pred) = coef[group] * (pred - target)
 */
int forge_sse_groups_blocks(void);
int forge_sse_groups_fwd(const float* pred, long long sn, long long sc, long long sh, long long sw, const float* target,
                         float* partial, int B, int Vp, int Vt, int gsize, int C, int H, int W, forge_stream_t stream);
int forge_sse_groups_bwd(const float* pred, long long sn, long long sc, long long sh, long long sw, const float* target,
                         const float* coef, float* dpred, int B, int Vp, int Vt, int gsize, int C, int H, int W, forge_stream_t stream);

/* f2  torch.optim.Adam (betas, eps; no weight decay / amsgrad) on one SMALL tensor in one launch, the step count `step` [1] (float) kept and
 * advanced on the device: the pose-refinement loop (kubric_eval.py:440-449) optimises b (t-1) x 7 numbers, for which the capturable torch
 * optimiser issues ~30 launches inside the captured iteration. param / exp_avg / exp_avg_sq [n] updated in place. */
int forge_adam_small(float* param, const float* grad, float* exp_avg, float* exp_avg_sq, float* step, int n, float lr, float beta1, float beta2,
                     float eps, forge_stream_t stream);

/* ---------------------------------------------------------------------------------------------
 * layout helpers: NCDHW <-> channels-last for callers that hold plain-contiguous volumes.
 *   src [n][C][P] -> dst [n][P][C]   (P = D*H*W)   and back.
 */
int forge_ncdhw_to_ndhwc(const float* src, float* dst, int n, int C, long long P, forge_stream_t stream);
int forge_ndhwc_to_ncdhw(const float* src, float* dst, int n, int C, long long P, forge_stream_t stream);

#ifdef __cplusplus
}
#endif
#endif /* FORGE_HIP_H */
