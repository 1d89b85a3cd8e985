"""Host-side plumbing for the fp32-MFMA implicit-GEMM convolution kernel (forge_conv_igemm).

Weight packing ([Cout,Cin,k..] -> [tap][Cout][Cin]), eval-mode BatchNorm folding, tap lists,
ConvTranspose phase decomposition and the launch wrapper. No arithmetic of the hot path happens
here apart from the one-off weight repacks (cached per parameter version).
"""
import ctypes

import torch

from . import _lib

EPI_BIAS, EPI_AFFINE_ACT, EPI_GRU_GATES, EPI_GRU_OUT = 0, 1, 2, 3


_TAPS_CACHE = {}


def _taps_array(taps):
    key = tuple(tuple(t) for t in taps)
    arr = _TAPS_CACHE.get(key)
    if arr is None:
        flat = [int(v) for t in key for v in t]
        arr = _TAPS_CACHE[key] = (ctypes.c_int * len(flat))(*flat)
    return arr


TAPS_3x3x3 = [(kz - 1, ky - 1, kx - 1) for kz in range(3) for ky in range(3) for kx in range(3)]
TAPS_3x3 = [(0, ky - 1, kx - 1) for ky in range(3) for kx in range(3)]        # a 2-D 3x3 convolution on a D = 1 grid (ResNet bottleneck conv2)
WINO2D_MIN_C = 256                  # 2-D Winograd (one "depth" tap, K = Cin per point) from this channel count: ResNet layer3 / layer4 (inference sweep, TUNING_LOG)


def pack_conv3d_weight(w):
    """nn.Conv3d weight [Cout,Cin,3,3,3] (cross-correlation, pad 1) -> [27][Cout][Cin]; tap t = (kz*3+ky)*3+kx
    multiplies in[z+kz-1, y+ky-1, x+kx-1]."""
    co, ci = w.shape[:2]
    return w.detach().reshape(co, ci, -1).permute(2, 0, 1).contiguous()


def pad_cin(wp, cin_pad):
    """Zero-pad packed weights [T][Cout][Cin] along Cin (inputs narrower than the 32-channel K-step)."""
    t, co, ci = wp.shape
    if ci == cin_pad:
        return wp
    out = torch.zeros(t, co, cin_pad, dtype=wp.dtype, device=wp.device)
    out[:, :, :ci] = wp
    return out


def convT_phases(w, pad, nd):
    """nn.ConvTranspose{2,3}d(stride=2, padding=pad) weight [Cin,Cout,k(,k),k] -> 2^nd output-phase GEMMs.
    Per axis, output o = 2 z + ph receives in[z + d] * W[k] for every k with k = ph + pad (mod 2), d = (ph + pad - k) / 2
    (from o = 2 i - pad + k). Returns [(phase (pz,py,px), taps [(dz,dy,dx)], wp [ntaps][Cout][Cin])]; 2-D uses pz = dz = 0."""
    k = w.shape[-1]
    per_axis = {ph: [((ph + pad - kk) // 2, kk) for kk in range(k) if (kk - ph - pad) % 2 == 0] for ph in (0, 1)}
    wd = w.detach()
    out = []
    for pz in ((0, 1) if nd == 3 else (0,)):
        for py in (0, 1):
            for px in (0, 1):
                taps, mats = [], []
                for dz, kz in (per_axis[pz] if nd == 3 else [(0, None)]):
                    for dy, ky in per_axis[py]:
                        for dx, kx in per_axis[px]:
                            taps.append((dz, dy, dx))
                            mats.append((wd[:, :, kz, ky, kx] if nd == 3 else wd[:, :, ky, kx]).t())     # [Cout][Cin]
                out.append(((pz, py, px), taps, torch.stack(mats).contiguous()))
    return out


def convT_phases_merged(w, pad, nd):
    """All output phases of convT_phases back to back for ONE forge_conv_igemm launch (phase = (-1,-1,-1)): (taps, wp)."""
    ph = convT_phases(w, pad, nd)
    taps = [t for _, tp, _ in ph for t in tp]
    return taps, torch.cat([wp for _, _, wp in ph], dim=0).contiguous()


def convT3d_k4s2p1_phases(w):
    """nn.ConvTranspose3d(k=4, s=2, p=1): 8 phases x 8 taps (per axis p=0: (d=0,k=1), (d=-1,k=3); p=1: (d=+1,k=0), (d=0,k=2))."""
    return convT_phases(w, 1, 3)


def pack_conv2d_weight(w):
    """nn.Conv2d weight [Cout,Cin,kh,kw] (pad k//2) -> ([kh*kw][Cout][Cin], taps (0,dy,dx)) for a D=1 grid."""
    co, ci, kh, kw = w.shape
    taps = [(0, ky - kh // 2, kx - kw // 2) for ky in range(kh) for kx in range(kw)]
    return w.detach().reshape(co, ci, kh * kw).permute(2, 0, 1).contiguous(), taps


def bn_affine(bn):
    """Eval-mode BatchNorm as y = x*scale + shift."""
    scale = bn.weight.detach() / torch.sqrt(bn.running_var.detach() + bn.eps)
    shift = bn.bias.detach() - bn.running_mean.detach() * scale
    return scale.contiguous(), shift.contiguous()


class PackCache:
    """Derived tensors (packed weights, folded BatchNorm) of one fused launch group, rebuilt when a source parameter / buffer changed.

    The key is (data_ptr, version counter, device) of every source: optimizer steps, `load_state_dict`, `.to()` and in-place tensor ops
    all bump one of them. In-place edits made through `.data` (`p.data.mul_()`, EMA swaps of legacy loaders) bump NONE of them, so the
    owning modules (PackedModule) also drop their caches on `train()` / `eval()`, `load_state_dict` and `_apply` (`.to()`, `.float()`),
    and `forge_amd.invalidate_packed(model)` is the explicit call after any other `.data` surgery."""

    def __init__(self):
        self._key = None
        self.val = None

    def clear(self):
        self._key = None
        self.val = None

    def key_of(self, sources):
        return tuple((t.data_ptr(), t._version, str(t.device)) for t in sources)

    def get(self, sources, build):
        key = self.key_of(sources)
        if key != self._key:
            self.val = build()
            self._key = key
        return self.val


class PackedModule(torch.nn.Module):
    """nn.Module whose fused HIP inference path keeps PackCache attributes: they are dropped whenever the module's mode or storage can
    have changed behind the version counters (see PackCache)."""

    def __init__(self):
        super().__init__()
        self.register_load_state_dict_post_hook(lambda module, incompatible_keys: module.invalidate_packed())

    def invalidate_packed(self):
        for v in self.__dict__.values():
            if isinstance(v, PackCache):
                v.clear()

    def train(self, mode=True):
        self.invalidate_packed()
        return super().train(mode)

    def _apply(self, fn, *args, **kwargs):
        self.invalidate_packed()
        return super()._apply(fn, *args, **kwargs)


def invalidate_packed(module):
    """Drop every packed-weight cache below `module` (call after editing parameters through `.data`)."""
    for m in module.modules():
        if isinstance(m, PackedModule):
            m.invalidate_packed()


class _ZeroArena:
    """Zero-filled scratch for the weight gradients of ONE backward pass from one allocation and ONE fill per HIP stream: every weight-gradient
    launch accumulates into a zero-filled dW (fp32 atomics over voxel chunks), i.e. ~100 `torch.zeros` launches of 4-5 us per training step. The
    first request a backward pass makes on a stream allocates the bytes the PREVIOUS pass used on that stream (+ slack) and zero-fills them once,
    on that stream; requests are carved out of it in order (256-byte aligned); when it runs out - or on the very first pass - a request falls
    back to its own `torch.zeros`. One pool per stream because nothing orders a side stream's launches (the weight gradients of the grouped
    ConvGRU fusion, FORGE's 2-D pose estimator: autograd replays their nodes on the stream their forward ran on) behind a fill queued on another
    stream. The pass ends with an autograd-engine callback (queued on the first request). The carved tensors are views of the pool: they stay
    valid for as long as anything (a parameter's .grad) references them; the next pass gets fresh pools."""

    def __init__(self):
        self.task, self.device = -1, None
        self.pools = {}               # stream handle -> [buffer | None, next free byte, bytes requested]  (this pass)
        self.want = {}                # stream handle -> bytes the previous pass requested on that stream

    def _end(self):
        self.want = {k: p[2] for k, p in self.pools.items()}
        self.pools, self.task = {}, -1

    # the arena needs two private autograd hooks (the id of the running backward pass, an end-of-pass callback); a torch build without them
    # gets plain torch.zeros for every request (feature-detected, not assumed)
    HAVE_HOOKS = hasattr(torch._C, "_current_graph_task_id") and hasattr(getattr(torch.autograd.Variable, "_execution_engine", None), "queue_callback")

    def zeros(self, shape, device):
        n = 1
        for d in shape:
            n *= int(d)
        nbytes = (n * 4 + 255) // 256 * 256
        if not self.HAVE_HOOKS:
            return torch.zeros(shape, dtype=torch.float32, device=device)
        task = torch._C._current_graph_task_id()                  # -1 outside a backward pass
        if task < 0:
            return torch.zeros(shape, dtype=torch.float32, device=device)
        if task != self.task:
            # a new pass - also when the previous one never reached its callback (an exception inside backward) or a nested backward runs inside
            # this one: the old pools are dropped, never carved again (their slices may be live gradients), and this pass gets fresh ones
            if self.task >= 0:
                self._end()
            self.task, self.device = task, device
            torch.autograd.Variable._execution_engine.queue_callback(self._end_of(task))
        if device != self.device:
            return torch.zeros(shape, dtype=torch.float32, device=device)
        key = torch.cuda.current_stream(device).cuda_stream if torch.device(device).type == "cuda" else 0
        pool = self.pools.get(key)
        if pool is None:                                          # allocated AND filled on the stream whose launches will accumulate into it
            want = self.want.get(key, 0)
            pool = self.pools[key] = [torch.zeros(want // 4 + 64, dtype=torch.float32, device=device) if want else None, 0, 0]
        pool[2] += nbytes
        if pool[0] is None or pool[1] + nbytes > pool[0].numel() * 4:
            return torch.zeros(shape, dtype=torch.float32, device=device)
        out = pool[0][pool[1] // 4:pool[1] // 4 + n].view(shape)
        pool[1] += nbytes
        return out

    def _end_of(self, task):
        def cb():
            if self.task == task:                                 # (a nested pass may have replaced it meanwhile)
                self._end()
        return cb


GRAD_ZEROS = _ZeroArena()


def grad_zeros(shape, device, scratch=False):
    """A zero-filled float32 tensor for a weight gradient (see _ZeroArena): inside a backward pass a slice of the pass's arena, else torch.zeros.
    scratch = True: a transient the caller does NOT hand to autograd as a gradient (Winograd-domain dU accumulators) - its own allocation, so that
    parameter gradients (which AccumulateGrad keeps as views of the arena) do not pin transients for as long as they live."""
    if scratch:
        return torch.zeros(tuple(shape), dtype=torch.float32, device=device)
    return GRAD_ZEROS.zeros(tuple(shape), device)


SPLITK_WS_BYTES = 128 << 20          # cap the plan model may assume for split-K partial tiles (ksplit x M x Cout floats)

TILE_NAMES = {"A": "128, 128, 8, 1", "B": "64, 128, 8, 1", "C": "128, 64, 8, 1", "D": "64, 64, 4, 1", "E": "128, 32, 4, 1"}

_PLAN_CACHE = {}


class _State:
    """Process-wide measurement switches (defaults = the product). Set through force_plan() / winograd(), or monkeypatched by tests."""
    plan_override = None              # (tile letter or None, ksplit or None): set by force_plan() only (tools/conv_plan_sweep.py, tests)
    winograd = True                   # False: the direct implicit-GEMM kernel for every 3x3(x3) convolution


STATE = _State()


class force_plan:
    """Context manager for tools / tests: pin the workgroup tile ('A'..'E') and / or the split-K factor of every forge_conv_igemm and
    forge_wino_gemm launch made inside it, instead of the library's plan model. The library itself reads no environment variables; the
    override travels as the explicit (tile, ksplit) arguments of the C-ABI calls."""

    def __init__(self, tile=None, ksplit=None):
        self.val = (tile, ksplit)

    def __enter__(self):
        self.prev = STATE.plan_override
        STATE.plan_override = self.val
        return self

    def __exit__(self, *exc):
        STATE.plan_override = self.prev
        return False


def conv_plan(M, Cout, Cin, ntaps, epilogue, ldo, nphase=1):
    """(tile letter, ksplit) of this problem: forge_conv_igemm_plan's makespan model (host arithmetic only, cached per shape), or the
    force_plan() override. nphase = 4 / 8 for a merged-phase transposed-conv launch (M rows per phase, ntaps over all phases). The
    launcher passes the answer to forge_conv_igemm explicitly and sizes the split-K scratch from it."""
    key = (int(M), int(Cout), int(Cin), int(ntaps), int(nphase), int(epilogue), int(ldo) % 4 == 0)
    pl = _PLAN_CACHE.get(key)
    if pl is None:
        tile, ks = ctypes.c_int(0), ctypes.c_int(0)
        _lib.check(_lib.lib().forge_conv_igemm_plan(int(M), int(Cout), int(Cin), int(ntaps), int(nphase), int(epilogue), int(ldo), SPLITK_WS_BYTES,
                                                    ctypes.byref(tile), ctypes.byref(ks)), "forge_conv_igemm_plan")
        pl = _PLAN_CACHE[key] = (chr(tile.value), ks.value)
    ov = STATE.plan_override
    if ov is not None and pl[0] != "N":
        tile = ov[0] or pl[0]
        ks = ov[1] if ov[1] is not None else (pl[1] if ov[0] is None else 1)
        can_split = nphase == 1 and epilogue in (EPI_BIAS, EPI_AFFINE_ACT) and Cout % 4 == 0 and ldo % 4 == 0
        if ks > 1 and not (can_split and ks * M * Cout * 4 <= SPLITK_WS_BYTES and ks <= (ntaps // nphase) * (Cin // 32)):
            ks = 1
        return tile, ks
    return pl


MAX_OPERAND_BYTES = (1 << 31) - 1       # 32-bit buffer offsets of the kernel's gathered operands (tests lower it to exercise the chunking)


@_lib.on_tensor_device
def conv_igemm(in1, C1, ld1, in2, C2, ld2, wp, bias, scale, shift, slope, residual, aux_h, aux_z, out, out2,
               grid, in_grid, Cout, ldo, taps, out_grid=None, istride=1, ostride=1, phase=(0, 0, 0), epilogue=EPI_BIAS,
               bs1=0, bs2=0, lift=0, out3=None, stats=None):
    """Thin launcher. grid = (n,D,H,W) GEMM-row grid; in_grid = (Di,Hi,Wi); out_grid = (Do,Ho,Wo) (default = grid).
    The kernel addresses its gathered operands through 32-bit buffer offsets (< 2 GiB per operand); batches whose inputs span more
    (e.g. 32 scenes of 64^3 x 64-channel head activations) are launched in batch chunks here.
    stats (float64 [stats_blocks(M, tile)][2][Cout], EPI_BIAS only): receives the output's per-block column sums / sums of squares (the batch
    statistics of a following BatchNorm); only for un-chunked, un-split launches - use conv_stats_buffer(), which returns None otherwise."""
    n, D, H, W = grid
    Di, Hi, Wi = in_grid
    Do, Ho, Wo = out_grid if out_grid is not None else (D, H, W)
    if wp.shape != (len(taps), Cout, C1 + C2):
        raise ValueError("packed weight %s does not match taps=%d Cout=%d Cin=%d" % (tuple(wp.shape), len(taps), Cout, C1 + C2))
    in_rows, out_rows = Di * Hi * Wi, Do * Ho * Wo
    b1, b2 = (int(bs1) or in_rows), (int(bs2) or in_rows)
    span = lambda k, br, ld: ((k - 1) * br + in_rows) * ld * 4
    limit = MAX_OPERAND_BYTES
    nc = n
    while nc > 1 and (span(nc, b1, ld1) > limit or (in2 is not None and span(nc, b2, ld2) > limit)):
        nc = (nc + 1) // 2
    L, st, arr = _lib.lib(), _lib.current_stream(), _taps_array(taps)
    nphase = 1
    if tuple(phase) == (-1, -1, -1):
        nphase = 8 if Do == 2 * D else 4

    def off(t, floats):                       # device pointer of tensor t advanced by `floats` elements (None -> NULL)
        return None if t is None else ctypes.c_void_p(t.data_ptr() + 4 * floats)
    gate_w = Cout // 2 if epilogue == EPI_GRU_GATES else Cout           # row width of the GRU side tensors
    for s0 in range(0, n, nc):
        k = min(nc, n - s0)
        orow = s0 * out_rows                  # first output row of the chunk (lift: rows of the un-lifted GEMM grid, same product)
        # the plan is made here and handed over explicitly: a split-K launch gets scratch of exactly ksplit x M x Cout floats from
        # torch's stream-ordered caching allocator (per-stream pools; inside a hipGraph capture it comes from the graph's private
        # pool and is reused by the graph's later launches) - no process-global workspace
        tile, ksplit = conv_plan(k * D * H * W, Cout, C1 + C2, len(taps), epilogue, ldo, nphase)
        if stats is not None and (ksplit > 1 or nc < n):
            raise RuntimeError("forge_amd: conv_igemm output statistics need an un-chunked launch without split-K (use conv_stats_buffer)")
        ws = torch.empty(ksplit * k * D * H * W * Cout, dtype=torch.float32, device=out.device) if ksplit > 1 else None
        o_ld = gate_w if epilogue == EPI_GRU_GATES else ldo
        _lib.check(L.forge_conv_igemm(
            off(in1, s0 * b1 * ld1), C1, ld1, int(bs1), off(in2, s0 * b2 * ld2), C2, ld2, int(bs2), _lib.ptr(wp), _lib.ptr(bias), _lib.ptr(scale),
            _lib.ptr(shift), float(slope), off(residual, orow * (Cout if (lift or epilogue in (EPI_GRU_GATES, EPI_GRU_OUT)) else ldo)),
            off(aux_h, orow * gate_w), off(aux_z, orow * Cout),
            off(out, orow * (Cout if lift else o_ld)), off(out2, orow * o_ld), off(out3, orow * o_ld), k, D, H, W, istride, Di, Hi, Wi, Cout, ldo,
            arr, len(taps), ostride, phase[0], phase[1], phase[2], Do, Ho, Wo, epilogue, int(lift), 0 if tile == "N" else ord(tile), ksplit,
            _lib.ptr(ws), 0 if ws is None else ws.numel() * 4, _lib.ptr(stats), st),
            "forge_conv_igemm")
    return out


TILE_BM = {"A": 128, "B": 64, "C": 128, "D": 64, "E": 128}          # GEMM rows per workgroup tile ('A'..'E' of forge_conv_igemm_plan)


def stats_blocks(M, tile):
    """32-row blocks forge_conv_igemm's `stats` by-product writes for M GEMM rows on workgroup tile `tile`: whole tiles, every block written."""
    bm = TILE_BM[tile]
    return ((int(M) + bm - 1) // bm) * (bm // 32)


def conv_stats_buffer(x1, in2, grid, in_grid, Cout, Cin, ntaps, ldo, device):
    """float64 [stats_blocks(M, tile)][2][Cout] for the BatchNorm statistics by-product of a plain EPI_BIAS conv_igemm launch, or None when this
    problem would not take one un-split launch of the wide kernel (split-K plan, Cout <= 16, batch chunking)."""
    n, D, H, W = grid
    M = n * D * H * W
    if Cout <= 16 or Cout % 4:
        return None
    tile, ksplit = conv_plan(M, Cout, Cin, ntaps, EPI_BIAS, ldo, 1)
    if ksplit > 1 or tile == "N":
        return None
    Di, Hi, Wi = in_grid
    if ((n - 1) * (Di * Hi * Wi) + Di * Hi * Wi) * max(x1.shape[-1], 0 if in2 is None else in2.shape[-1]) * 4 > MAX_OPERAND_BYTES:
        return None
    return torch.empty(stats_blocks(M, tile), 2, Cout, dtype=torch.float64, device=device)


# ------------------------------------------------------------------------------------------------------------------
# Winograd F(2x2, 3x3) x 3 depth taps for the stride-1 3x3x3 convolutions of the inference path (csrc/winograd.hip)
# ------------------------------------------------------------------------------------------------------------------
@_lib.on_tensor_device
def wino_pack_packed(wp, transpose=False):
    """Packed weights [27][Cout][Cin] (pack_conv3d_weight) -> U [16][3][Cout][Cin] = G w[kd] G^T (forge_wino_weights: float64 inside,
    rounded once). transpose: the weights of the DATA GRADIENT instead - the correlation of dy with the flipped kernel and swapped
    channel roles, U [16][3][Cin][Cout]. One small kernel: the training path calls it per convolution and step."""
    T, co_, ci_ = wp.shape
    assert T in (27, 9)                     # 3x3x3 kernels, or the 3x3 kernels of a 2-D convolution (pack_conv2d_weight): kd = 3 / 1 depth taps
    kd = T // 9
    wp = wp.detach()
    wp = wp if wp.is_contiguous() else wp.contiguous()
    U = torch.empty((16, kd, ci_, co_) if transpose else (16, kd, co_, ci_), dtype=torch.float32, device=wp.device)
    _lib.check(_lib.lib().forge_wino_weights(_lib.ptr(wp), _lib.ptr(U), co_, ci_, kd, 1 if transpose else 0, _lib.current_stream()), "forge_wino_weights")
    return U


def wino_pack_weight(w):
    """nn.Conv3d weight [Cout,Cin,3,3,3] -> U [16][3][Cout][Cin]: U[4i+j][kd] = (G w[kd] G^T)[i][j]."""
    return wino_pack_packed(pack_conv3d_weight(w))


def wino_conv_rows(x1, x2, U, bias, out, residual=None, V1=None):
    """out [n,D,H,W,Cout] = conv3x3x3(cat(x1, x2)) + bias (+ residual) on channels-last rows through the three Winograd launches.
    x1 may have a batch stride (a view of a [b,t,...] stack). V1: x1's input transform, if the caller already has it."""
    n, D, H, W, C1 = x1.shape
    return _wino_conv(x1, C1, _batch_stride_rows(x1), x2, 0 if x2 is None else x2.shape[-1], 0 if x2 is None else _batch_stride_rows(x2), U, bias, out,
                      residual, (n, D, H, W), V1=V1)


def _wino_conv(x1, C1, bs1, x2, C2, bs2, U, bias, out, residual, grid, V1=None):
    """The three launches of one Winograd convolution with the bias (+ residual) tail: transforms of x1 / x2, point GEMMs, inverse."""
    n, D, H, W = grid
    Cout = U.shape[2]
    if V1 is None:
        V1 = wino_input(x1, C1, C1, n, D, H, W, bs=bs1)
    V2 = None if x2 is None else wino_input(x2, C2, C2, n, D, H, W, bs=bs2)
    R = n * D * (H // 2) * (W // 2)
    Mm = torch.empty(8 if wino_half_applies(R, Cout, C1 + C2) else 16, R, Cout, dtype=torch.float32, device=out.device)
    wino_gemm(V1, C1, V2, C2, U, Mm, n, D, H // 2, W // 2, Cout)
    return wino_output(Mm, bias, None, None, 1.0, residual, None, None, out, None, None, n, D, H, W, Cout, Cout, EPI_BIAS)


_ONE_ZERO = {}


def _one_zero(device, C):
    key = (str(device), C)
    v = _ONE_ZERO.get(key)
    if v is None:
        v = _ONE_ZERO[key] = (torch.ones(C, device=device), torch.zeros(C, device=device))
    return v


def conv3_launch(x1, C1, x2, C2, wp, bias, out, grid, Cout, bs1=0, residual=None, dgrad=False, U=None, wT=None):
    """out [rows][Cout] = conv3x3x3(cat(x1, x2)) + bias + residual on dense channels-last rows (x1 may have the batch stride bs1 rows),
    from the layer's packed FORWARD weights wp [27][Co][Ci]. dgrad: the data gradient of that layer instead (x1 = dy with C1 = Co
    channels, out = dx with Cout = Ci). Winograd launches when wino_applies, else the direct implicit-GEMM kernel. U / wT: the Winograd
    weights / the transposed packed weights [27][Ci][Co] of the direct data gradient, if the caller caches them (frozen weights)."""
    n, D, H, W = grid
    if wino_applies(TAPS_3x3x3, 1, n, D, H, W, C1, C2, Cout):
        return _wino_conv(x1, C1, bs1, x2, C2, 0, U if U is not None else wino_pack_packed(wp, transpose=dgrad), bias, out, residual, grid)
    w = (wT if wT is not None else wp.transpose(1, 2).contiguous()) if dgrad else wp
    taps = [(-a, -b, -c) for a, b, c in TAPS_3x3x3] if dgrad else TAPS_3x3x3
    if residual is None:
        return conv_igemm(x1, C1, C1, x2, C2, C2, w, bias, None, None, 1.0, None, None, None, out, None, grid, (D, H, W), Cout, Cout, taps,
                          epilogue=EPI_BIAS, bs1=bs1)
    one, zero = _one_zero(out.device, Cout)
    return conv_igemm(x1, C1, C1, x2, C2, C2, w, bias, one, zero, 1.0, residual, None, None, out, None, grid, (D, H, W), Cout, Cout, taps,
                      epilogue=EPI_AFFINE_ACT, bs1=bs1)


def wino_wgrad_applies(n, D, H, W, C1, C2, Cout, taps=None):
    """The Winograd weight gradient runs on the 128-wide tiles of the wgrad kernel: wide layers only (with two inputs its Cin tiles must
    not straddle them: C1 a multiple of 128)."""
    return (wino_applies(TAPS_3x3x3 if taps is None else taps, 1, n, D, H, W, C1, C2, Cout) and C1 + C2 >= 128 and Cout >= 64 and (C2 == 0 or C1 % 128 == 0)
            and n * D * (H // 2) * (W // 2) * Cout * 4 <= MAX_OPERAND_BYTES)


@_lib.on_tensor_device
def conv3_wgrad(dy, x1, C1, x2, C2, dwp, grid, Cout, bs1=0, taps=None, dM=None):
    """dwp [27 | 9][Cout][C1+C2] (zero-filled by the caller) += the weight gradient of conv3x3x3 / conv3x3 (cat(x1, x2)) for the upstream gradient
    dy [rows][Cout]. Wide layers take the Winograd form - dMm = A dy A^T, dU[p] = dMm[p]^T (x) V[p] as 16 batched problems of the wgrad
    GEMM kernel (2.25x fewer FLOPs), dw = G^T dU G - the others the direct kernel (conv_wgrad)."""
    n, D, H, W = grid
    taps = TAPS_3x3x3 if taps is None else list(taps)
    if not wino_wgrad_applies(n, D, H, W, C1, C2, Cout, taps):
        return conv_wgrad(dy, x1, C1, x2, C2, dwp, grid, (D, H, W), Cout, taps, bs1=bs1)
    kd = len(taps) // 9
    L, st = _lib.lib(), _lib.current_stream()
    Ht, Wt = H // 2, W // 2
    R = n * D * Ht * Wt
    dev = dy.device
    V1 = wino_input(x1, C1, C1, n, D, H, W, bs=bs1)
    V2 = None if x2 is None else wino_input(x2, C2, C2, n, D, H, W)
    if dM is None:                                             # (the caller may hold A dy A^T already: wino_input_dy)
        dM = torch.empty(16, R, Cout, dtype=torch.float32, device=dev)
        _lib.check(L.forge_wino_dy(_lib.ptr(dy), dy.shape[-1], _lib.ptr(dM), n, D, H, W, Cout, st), "forge_wino_dy")
    dU = grad_zeros((16, kd, Cout, C1 + C2), dev, scratch=True)
    _lib.check(L.forge_wino_wgrad(_lib.ptr(dM), _lib.ptr(V1), C1, 0, 0, _lib.ptr(V2), C2, 0, 0, _lib.ptr(dU), n, D, Ht, Wt, Cout, kd, st), "forge_wino_wgrad")
    _lib.check(L.forge_wino_dw(_lib.ptr(dU), _lib.ptr(dwp), Cout, C1 + C2, kd, st), "forge_wino_dw")
    return dwp


def wino_applies(taps, istride, n, D, H, W, C1, C2, Cout):
    """Stride-1 3x3x3 problems with GEMM-sized channel counts take the Winograd launches: K = 3 Cin per point must amortise the GEMM
    prologue (measured, tools/wino_gemm_sweep.py: conv1's Cin = 64 -> 128 still runs at 95 TF of MFMA work = 214 TF direct-equivalent
    against 120 TF on the direct kernel); narrower layers (the heads' 32 -> 16 / 8) stay on the direct kernels. 2-D 3x3 convolutions on a
    D = 1 grid (one depth tap, K = Cin per point) from WINO2D_MIN_C channels: the bottleneck conv2 of ResNet layer3 / layer4."""
    tt = tuple(tuple(t) for t in taps)
    if not (wino_enabled() and istride == 1 and C1 % 32 == 0 and C2 % 32 == 0 and Cout % 8 == 0 and wino_fits(n, D, H, W, max(C1, C2, 1))):
        return False
    if len(tt) == 27 and tt == tuple(TAPS_3x3x3):
        return C1 + C2 >= 64 and Cout >= 32
    if len(tt) == 9 and tt == tuple(TAPS_3x3) and D == 1:
        return C2 == 0 and C1 >= WINO2D_MIN_C and Cout >= WINO2D_MIN_C
    return False


def wino_enabled():
    """The stride-1 3x3(x3) convolutions take the Winograd launches unless a tool / test asked for the direct implicit-GEMM kernel
    (`with convops.winograd(False): ...`; A/B: tools/wino_ab.py, bit-equality tests of launcher mechanics)."""
    return STATE.winograd


class winograd:
    """Context manager: run the enclosed launches with (True) / without (False) the Winograd path."""

    def __init__(self, on):
        self.on = bool(on)

    def __enter__(self):
        self.prev = STATE.winograd
        STATE.winograd = self.on
        return self

    def __exit__(self, *exc):
        STATE.winograd = self.prev
        return False


def wino_scene_chunk(b, D, H, W, C, views=1):
    """Largest number of scenes (<= b) whose transformed operands stay within the kernel's buffer range (0: not even one, or odd H / W)."""
    nb = b
    while nb >= 1 and not wino_fits(nb, D, H, W, C, views=views):
        nb = (nb + 1) // 2 if nb > 1 else 0
    return nb


def wino_fits(n, D, H, W, C, views=1):
    """H, W even, and every transformed operand ([n views D H/2 W/2][C] floats per Winograd point) within the kernel's 32-bit buffer offsets."""
    return H % 2 == 0 and W % 2 == 0 and n * views * D * (H // 2) * (W // 2) * C * 4 <= MAX_OPERAND_BYTES


@_lib.on_tensor_device
def wino_input_dy(dy, C, n, D, H, W):
    """(V, dM) = (B^T dy B, A dy A^T), both [16][n D H/2 W/2][C], of an upstream gradient dy (dense rows [n D H W][C]) in ONE pass over it:
    the operands of the data-gradient point GEMMs and of the Winograd weight gradient (forge_wino_input_dy)."""
    R = n * D * (H // 2) * (W // 2)
    V = torch.empty(16, R, C, dtype=torch.float32, device=dy.device)
    dM = torch.empty(16, R, C, dtype=torch.float32, device=dy.device)
    _lib.check(_lib.lib().forge_wino_input_dy(_lib.ptr(dy), C, _lib.ptr(V), _lib.ptr(dM), n, D, H, W, C, _lib.current_stream()), "forge_wino_input_dy")
    return V, dM


@_lib.on_tensor_device
def wino_input(x, C, ld, n, D, H, W, bs=0, out=None, nsum=1, sum_stride=0):
    """V[16][n D H/2 W/2][C] = B^T d B of the channels-last rows x ([n][D][H][W] x ld floats, batch stride bs rows). nsum > 1: d is the
    mean of nsum such tensors sum_stride rows apart (the view mean feeding fusion_conv, models/encoder.py:62)."""
    R = n * D * (H // 2) * (W // 2)
    V = out if out is not None else torch.empty(16, R, C, dtype=torch.float32, device=x.device)
    _lib.check(_lib.lib().forge_wino_input(_lib.ptr(x), ld, int(bs), _lib.ptr(V), C, 0, n, D, H, W, C, int(nsum), int(sum_stride), _lib.current_stream()),
               "forge_wino_input")
    return V


def wino_half_applies(R, Cout, Cin):
    """The point-GEMM launches whose inverse transform's row stage runs in the GEMM epilogue (forge_wino_gemm_half / forge_wino_output_half: half the
    point-product bytes through HBM; bitwise the same result without a second addend): those forge_wino_gemm would give its 64 x 128 tile - a rule
    of (R, Cout) only - and no forced plan."""
    return STATE.plan_override is None and wino_gemm_tile(R, Cout, Cin) == "B"


@_lib.on_tensor_device
def wino_gemm(V1, C1, V2, C2, U, Mm, n, D, Ht, Wt, Cout, view=0, views=1, half=None):
    """Mm[16][n D Ht Wt][Cout] = the 16 point GEMMs. V1 [16][n views D Ht Wt][C1] may hold `views` views per batch element (the
    transformed inputs of every view of a scene, made by ONE wino_input launch): this call reads view `view`. V2 [16][R][C2] or None.
    half: Mm receives the 8 planes [2][4][R][Cout] of forge_wino_gemm_half (the first half of a 16-plane buffer) for wino_output(half=True).
    None = wino_half_applies' rule, which depends on (R, Cout) only - wino_output applies the same rule to ITS row count, so a GEMM / inverse pair over
    the same rows always agrees; a caller whose products are consumed as the second addend (Mm2) of launches over FEWER rows passes their decision."""
    vol = D * Ht * Wt
    if half is None:
        half = wino_half_applies(n * vol, Cout, C1 + C2)
    kd = U.shape[1]
    if U.shape != (16, kd, Cout, C1 + C2) or kd not in (1, 3):
        raise ValueError("transformed weight %s does not match Cout=%d Cin=%d" % (tuple(U.shape), Cout, C1 + C2))
    p1 = ctypes.c_void_p(V1.data_ptr() + 4 * view * vol * C1)
    if half:
        _lib.check(_lib.lib().forge_wino_gemm_half(p1, C1, C1, views * vol if views > 1 else 0, V1.shape[1] * C1, _lib.ptr(V2), C2, C2, 0,
                                                   0 if V2 is None else V2.shape[1] * C2, _lib.ptr(U), _lib.ptr(Mm), n, D, Ht, Wt, Cout, kd,
                                                   _lib.current_stream()), "forge_wino_gemm_half")
        return Mm
    _lib.check(_lib.lib().forge_wino_gemm(p1, C1, C1, views * vol if views > 1 else 0, V1.shape[1] * C1, _lib.ptr(V2), C2, C2, 0,
                                          0 if V2 is None else V2.shape[1] * C2, _lib.ptr(U), _lib.ptr(Mm), n, D, Ht, Wt, Cout, kd,
                                          ord(wino_gemm_tile(n * vol, Cout, C1 + C2)), _lib.current_stream()),
               "forge_wino_gemm")
    return Mm


def wino_gemm_tile(R, Cout, Cin):
    """Tile letter forge_wino_gemm uses for R tile rows per point, Cin input channels (names the kernel instantiation for profilers, bench.py)."""
    ov = STATE.plan_override
    if ov is not None and ov[0]:
        return ov[0]
    return chr(_lib.lib().forge_wino_gemm_tile(int(R), int(Cout), int(Cin)))


@_lib.on_tensor_device
def wino_output(Mm, bias, scale, shift, slope, residual, aux_h, aux_z, out, out2, out3, n, D, H, W, Cout, ldo, epilogue, Mm2=None, view=0, views=1, half=None):
    """out = epilogue(A^T (Mm + Mm2) A): the element-wise tails of conv_igemm (EPI_*) on the inverse-transformed tiles. Mm2 (optional)
    [16][n views D H/2 W/2][Cout]: point products of the input half for `views` views per batch element; this call adds view `view`.
    half: Mm (and Mm2) hold wino_gemm(half=True)'s 8 planes (column stage only); None = the rule wino_gemm applied to make them."""
    vol = D * (H // 2) * (W // 2)
    if half is None:
        half = wino_half_applies(n * vol, Cout, 0)
    if half:                                                            # Mm2 (if any) is in the 8-plane form too
        p2 = None if Mm2 is None else ctypes.c_void_p(Mm2.data_ptr() + 4 * view * vol * Cout)
        _lib.check(_lib.lib().forge_wino_output_half(_lib.ptr(Mm), p2, views * vol, 0 if Mm2 is None else Mm2.shape[1] * Cout, _lib.ptr(bias), _lib.ptr(scale), _lib.ptr(shift), float(slope), _lib.ptr(residual),
                                                     _lib.ptr(aux_h), _lib.ptr(aux_z), _lib.ptr(out), _lib.ptr(out2), _lib.ptr(out3), n, D, H, W, Cout, ldo,
                                                     epilogue, _lib.current_stream()), "forge_wino_output_half")
        return out
    p2 = None if Mm2 is None else ctypes.c_void_p(Mm2.data_ptr() + 4 * view * vol * Cout)
    _lib.check(_lib.lib().forge_wino_output(_lib.ptr(Mm), p2, views * vol, 0 if Mm2 is None else Mm2.shape[1] * Cout, _lib.ptr(bias), _lib.ptr(scale), _lib.ptr(shift), float(slope), _lib.ptr(residual),
                                            _lib.ptr(aux_h), _lib.ptr(aux_z), _lib.ptr(out), _lib.ptr(out2), _lib.ptr(out3), n, D, H, W, Cout, ldo,
                                            epilogue, _lib.current_stream()), "forge_wino_output")
    return out


# ------------------------------------------------------------------------------------------------------------------
# autograd: stride-1 3x3x3 convolution on channels-last rows, forward / data-gradient on forge_conv_igemm, weight-gradient
# on forge_conv_wgrad. Used by the training / pose-refinement paths of the ConvGRU fusion and conv1.
# ------------------------------------------------------------------------------------------------------------------
@_lib.on_tensor_device
def conv_wgrad(dy, x1, C1, x2, C2, dwp, grid, in_grid, Cout, taps, istride=1, bs1=0, bs2=0):
    """dwp += weight gradient (forge_conv_wgrad; dwp zero-filled by the caller, accumulated with atomics). Batches whose operands span
    2 GiB or more (32-bit buffer offsets in the kernel) are accumulated in batch chunks, as conv_igemm launches them."""
    n, D, H, W = grid
    Di, Hi, Wi = in_grid
    in_rows, out_rows = Di * Hi * Wi, D * H * W
    ldy, ld1, ld2 = dy.shape[-1], x1.shape[-1], (0 if x2 is None else x2.shape[-1])
    b1, b2 = (int(bs1) or in_rows), (int(bs2) or in_rows)
    span = lambda k, br, ld, rows: ((k - 1) * br + rows) * ld * 4
    nc = n
    while nc > 1 and (span(nc, out_rows, ldy, out_rows) > MAX_OPERAND_BYTES or span(nc, b1, ld1, in_rows) > MAX_OPERAND_BYTES
                      or (x2 is not None and span(nc, b2, ld2, in_rows) > MAX_OPERAND_BYTES)):
        nc = (nc + 1) // 2
    off = lambda t, floats: None if t is None else ctypes.c_void_p(t.data_ptr() + 4 * floats)
    for s0 in range(0, n, nc):
        k = min(nc, n - s0)
        _lib.check(_lib.lib().forge_conv_wgrad(off(dy, s0 * out_rows * ldy), ldy, off(x1, s0 * b1 * ld1), C1, ld1, int(bs1), off(x2, s0 * b2 * ld2), C2, ld2,
                                               int(bs2), _lib.ptr(dwp), k, D, H, W, istride, Di, Hi, Wi, Cout, _taps_array(taps), len(taps),
                                               _lib.current_stream()), "forge_conv_wgrad")
    return dwp


def _batch_stride_rows(x):
    """rows tensor [n,D,H,W,C] whose only non-dense stride may be the batch one -> batch stride in rows (0 = dense)."""
    n, D, H, W, C = x.shape
    if x.stride()[1:] != (H * W * C, W * C, C, 1):
        raise ValueError("conv_rows needs channels-last rows [n,D,H,W,C] dense within a batch element; got strides %s" % (x.stride(),))
    return 0 if (n == 1 or x.stride(0) == D * H * W * C) else x.stride(0) // C


@_lib.on_tensor_device
def colsum(x):
    """Sum over the rows of a channels-last tensor [..., C] -> [C] - the bias gradient of a convolution - on forge_colsum (float64 partial sums in a
    fixed order: deterministic, and more accurate than a fp32 tree). Channel counts that are not multiples of 4 (the 1- and 3-channel outputs of the
    density head / conv_rgb) take the kernel's flat walk over the dense matrix."""
    C = x.shape[-1]
    rows = x.reshape(-1, C)
    if not (rows.is_cuda and rows.dtype == torch.float32):
        raise RuntimeError("forge_amd: colsum runs only on the MI355X HIP kernels (float32 cuda tensor); got %s on %s" % (rows.dtype, rows.device))
    if C % 4:
        if C > 32:
            raise RuntimeError("forge_amd: colsum needs C %% 4 == 0 or C <= 32 (got %d)" % C)
        rows = rows if rows.is_contiguous() else rows.contiguous()
    else:
        rows = rows if (rows.stride(1) == 1 and rows.stride(0) >= C and rows.stride(0) % 4 == 0) else rows.contiguous()
    out = torch.empty(C, dtype=torch.float32, device=rows.device)
    ws = torch.empty(_lib.lib().forge_bn_ws_doubles(C), dtype=torch.float64, device=rows.device)
    _lib.check(_lib.lib().forge_colsum(_lib.ptr(rows), rows.stride(0), _lib.ptr(out), _lib.ptr(ws), rows.shape[0], C, _lib.current_stream()), "forge_colsum")
    return out


class _ConvTapsRows(torch.autograd.Function):
    """y[m] = bias + sum_t wp[t] @ cat(x1, x2)[voxel(m) * istride + taps[t]] on channels-last rows — the one autograd node behind
    every convolution of the training path. `wp` [T][Cout][Cin] is the differentiable weight: callers build it from the module
    parameter with differentiable torch ops (permute / reshape / pad), so autograd maps its gradient back to the parameter layout.
      forward          forge_conv_igemm
      d/dx  stride 1   forge_conv_igemm on dy with negated taps and transposed weights
            stride 2   (2-D, D = 1) a transposed convolution: one phase GEMM per input-pixel parity over the taps of that parity
      d/dwp            forge_conv_wgrad
    """

    @staticmethod
    def forward(ctx, x1, x2, wp, bias, taps, istride, out_spatial, want_stats=False):
        n, Di, Hi, Wi, C1 = x1.shape
        C2 = 0 if x2 is None else x2.shape[-1]
        T, Cout, Cin = wp.shape
        D, H, W = out_spatial
        wpc = wp.detach().contiguous()
        out = torch.empty(n, D, H, W, Cout, dtype=torch.float32, device=x1.device)
        bs1 = _batch_stride_rows(x1)
        bs2 = 0 if x2 is None else _batch_stride_rows(x2)
        stats = None
        if (D, H, W) == (Di, Hi, Wi) and wino_applies(taps, istride, n, D, H, W, C1, C2, Cout):
            wino_conv_rows(x1, x2, wino_pack_packed(wpc), bias, out)
        else:
            if want_stats and bs1 == 0 and bs2 == 0:
                # the BatchNorm behind this convolution gets its batch statistics from the GEMM epilogue (float64 column sums per 32-row block)
                stats = conv_stats_buffer(x1, x2, (n, D, H, W), (Di, Hi, Wi), Cout, Cin, T, Cout, x1.device)
            conv_igemm(x1, C1, C1, x2, C2, C2, wpc, bias, None, None, 1.0, None, None, None, out, None,
                       (n, D, H, W), (Di, Hi, Wi), Cout, Cout, taps, istride=istride, epilogue=EPI_BIAS, bs1=bs1, bs2=bs2, stats=stats)
        ctx.save_for_backward(x1, x2, wpc)
        ctx.meta = (tuple(taps), istride, (D, H, W), bias is not None)
        if not want_stats:
            return out
        if stats is None:
            stats = torch.empty(0, dtype=torch.float64, device=x1.device)      # "none": the BatchNorm runs its own statistics pass
        ctx.mark_non_differentiable(stats)
        return out, stats

    @staticmethod
    def backward(ctx, dy, _dstats=None):
        x1, x2, wp = ctx.saved_tensors
        taps, istride, (D, H, W), has_bias = ctx.meta
        n, Di, Hi, Wi, C1 = x1.shape
        C2 = 0 if x2 is None else x2.shape[-1]
        T, Cout, Cin = wp.shape
        dy = dy.contiguous()
        dx1 = dx2 = dwp = db = None
        need_dx = ctx.needs_input_grad[0] or ctx.needs_input_grad[1]
        same = (D, H, W) == (Di, Hi, Wi)
        wino_dx = need_dx and istride == 1 and Cout > 16 and same and wino_applies(taps, istride, n, D, H, W, Cout, 0, Cin)
        wino_dw = (ctx.needs_input_grad[2] and same and wino_applies(taps, istride, n, D, H, W, C1, C2, Cout)
                   and (x2 is None or _batch_stride_rows(x2) == 0) and wino_wgrad_applies(n, D, H, W, C1, C2, Cout, taps))
        Vdy = dMdy = None
        if wino_dx and wino_dw:                                                      # both Winograd forms of dy from one pass over it
            Vdy, dMdy = wino_input_dy(dy.reshape(-1, Cout), Cout, n, D, H, W)
        if need_dx:
            wd = wp.transpose(1, 2).contiguous()                                     # [T][Cin][Cout]
            if istride == 1 and Cout <= 16:
                # narrow output (forward ran on the Cout <= 16 kernel): the data gradient has K = Cout <= 16 and N = Cin; run it on the
                # same kernel in 16-column blocks of Cin (K padded to its 16-voxel step) instead of padding both sides to 32
                assert (D, H, W) == (Di, Hi, Wi)
                if Cout < 16:
                    dyk = torch.nn.functional.pad(dy, (0, 16 - Cout))
                    wd = torch.nn.functional.pad(wd, (0, 16 - Cout))
                else:
                    dyk = dy
                dx = torch.empty(n, Di, Hi, Wi, Cin, dtype=torch.float32, device=dy.device)
                ntaps_ = [(-a, -b, -c) for a, b, c in taps]
                for j in range(0, Cin, 16):
                    nb = min(16, Cin - j)
                    conv_igemm(dyk, 16, 16, None, 0, 0, wd[:, j:j + nb].contiguous(), None, None, None, 1.0, None, None, None, dx[..., j:], None,
                               (n, D, H, W), (D, H, W), nb, Cin, ntaps_, epilogue=EPI_BIAS)
            elif istride == 1 and (D, H, W) == (Di, Hi, Wi) and wino_applies(taps, istride, n, D, H, W, Cout, 0, Cin):
                dx = torch.empty(n, Di, Hi, Wi, Cin, dtype=torch.float32, device=dy.device)
                wino_conv_rows(dy, None, wino_pack_packed(wp, transpose=True), None, dx, V1=Vdy)
            elif istride == 1:
                assert (D, H, W) == (Di, Hi, Wi)
                dx = torch.empty(n, Di, Hi, Wi, Cin, dtype=torch.float32, device=dy.device)
                conv_igemm(dy, Cout, Cout, None, 0, 0, wd, None, None, None, 1.0, None, None, None, dx, None, (n, D, H, W), (D, H, W), Cin, Cin,
                           [(-a, -b, -c) for a, b, c in taps], epilogue=EPI_BIAS)
            else:
                # stride 2: a transposed convolution - one phase GEMM per input-voxel parity over the taps of that parity (2-D: D = Di = 1, four
                # phases; 3-D: eight). Parities no tap reaches keep a zero gradient.
                flat = D == 1 and Di == 1 and all(t[0] == 0 for t in taps)
                assert istride == 2 and Hi == 2 * H and Wi == 2 * W and (flat or Di == 2 * D), (istride, (D, H, W), (Di, Hi, Wi))
                dx = torch.zeros(n, Di, Hi, Wi, Cin, dtype=torch.float32, device=dy.device)
                for ph_z in ((0,) if flat else (0, 1)):
                    for ph_y in (0, 1):
                        for ph_x in (0, 1):
                            sel = [i for i, (tz, ty, tx) in enumerate(taps) if (flat or (tz - ph_z) % 2 == 0) and (ty - ph_y) % 2 == 0 and (tx - ph_x) % 2 == 0]
                            if not sel:
                                continue
                            ptaps = [(0 if flat else (ph_z - taps[i][0]) // 2, (ph_y - taps[i][1]) // 2, (ph_x - taps[i][2]) // 2) for i in sel]
                            conv_igemm(dy, Cout, Cout, None, 0, 0, torch.stack([wd[i] for i in sel]), None, None, None, 1.0, None, None, None, dx, None,
                                       (n, D, H, W), (D, H, W), Cin, Cin, ptaps, out_grid=(Di, Hi, Wi), ostride=2, phase=(ph_z, ph_y, ph_x),
                                       epilogue=EPI_BIAS)
            dx1 = dx[..., :C1] if ctx.needs_input_grad[0] else None
            dx2 = dx[..., C1:] if (x2 is not None and ctx.needs_input_grad[1]) else None
        if ctx.needs_input_grad[2]:
            dwp = grad_zeros(wp.shape, wp.device)
            if (D, H, W) == (Di, Hi, Wi) and wino_applies(taps, istride, n, D, H, W, C1, C2, Cout) and (x2 is None or _batch_stride_rows(x2) == 0):
                conv3_wgrad(dy, x1, C1, x2, C2, dwp, (n, D, H, W), Cout, bs1=_batch_stride_rows(x1), taps=taps, dM=dMdy)
            else:
                conv_wgrad(dy, x1, C1, x2, C2, dwp, (n, D, H, W), (Di, Hi, Wi), Cout, list(taps), istride=istride, bs1=_batch_stride_rows(x1),
                           bs2=0 if x2 is None else _batch_stride_rows(x2))
        if has_bias and ctx.needs_input_grad[3]:
            db = colsum(dy.reshape(-1, Cout))
        return dx1, dx2, dwp, db, None, None, None, None


class _Conv1x1RowsSkip(torch.autograd.Function):
    """y = x @ w^T (a 1x1 stride-1 convolution without bias on NHWC rows) that ALSO hands its input through as a second output: the first
    convolution of a ResNet bottleneck, whose input feeds the identity path (or the downsample convolution) as well. Autograd then delivers
    the gradient of that second use to THIS node, and the data-gradient GEMM adds it in its epilogue (residual operand) - instead of a
    separate element-wise accumulation of two activation-sized gradients per block (torchvision Bottleneck.forward: out += identity)."""

    @staticmethod
    def forward(ctx, x, w):
        N, H, W, Cin = x.shape
        Cout = w.shape[0]
        wp = w.detach().reshape(1, Cout, Cin).contiguous()
        out = torch.empty(N, H, W, Cout, dtype=torch.float32, device=x.device)
        stats = conv_stats_buffer(x, None, (N, 1, H, W), (1, H, W), Cout, Cin, 1, Cout, x.device)      # the batch statistics of bn1, from the epilogue
        conv_igemm(x, Cin, Cin, None, 0, 0, wp, None, None, None, 1.0, None, None, None, out, None, (N, 1, H, W), (1, H, W), Cout, Cout, [(0, 0, 0)],
                   epilogue=EPI_BIAS, stats=stats)
        ctx.save_for_backward(x, wp)
        if stats is None:
            stats = torch.empty(0, dtype=torch.float64, device=x.device)
        ctx.mark_non_differentiable(stats)
        return out, x.view_as(x), stats

    @staticmethod
    def backward(ctx, dy, dskip, _dstats=None):
        x, wp = ctx.saved_tensors
        N, H, W, Cin = x.shape
        Cout = wp.shape[1]
        dy = dy.contiguous()
        dx = dw = None
        grid, ig, T1 = (N, 1, H, W), (1, H, W), [(0, 0, 0)]
        if ctx.needs_input_grad[0]:
            wd = wp.transpose(1, 2).contiguous()                               # [1][Cin][Cout]
            dx = torch.empty(N, H, W, Cin, dtype=torch.float32, device=dy.device)
            if dskip is None:
                conv_igemm(dy, Cout, Cout, None, 0, 0, wd, None, None, None, 1.0, None, None, None, dx, None, grid, ig, Cin, Cin, T1, epilogue=EPI_BIAS)
            else:
                one, zero = _one_zero(dy.device, Cin)
                conv_igemm(dy, Cout, Cout, None, 0, 0, wd, None, one, zero, 1.0, dskip.contiguous(), None, None, dx, None, grid, ig, Cin, Cin, T1,
                           epilogue=EPI_AFFINE_ACT)
        if ctx.needs_input_grad[1]:
            dwp = grad_zeros(wp.shape, wp.device)
            conv_wgrad(dy.reshape(N, 1, H, W, Cout), x.reshape(N, 1, H, W, Cin), Cin, None, 0, dwp, grid, ig, Cout, T1)
            dw = dwp.reshape(Cout, Cin, 1, 1)
        return dx, dw


def conv1x1_rows_skip(x, weight):
    """(conv1x1(x, weight), x, stats) on NHWC rows [N,H,W,Cin] with autograd; weight [Cout,Cin,1,1], Cin and Cout multiples of 32. Use the
    returned alias of x for every OTHER consumer of x: its gradient is then added inside this convolution's data-gradient GEMM. stats: the
    output's batch statistics from the GEMM epilogue (conv_taps_rows), for the BatchNorm behind the convolution."""
    return _Conv1x1RowsSkip.apply(x, weight)


def conv_taps_rows(x1, x2, wp, bias, taps, istride=1, out_spatial=None, want_stats=False):
    """want_stats: returns (out, stats) - stats = the output's float64 per-block column sums / sums of squares from the GEMM epilogue
    ([blocks][2][Cout]; an EMPTY tensor when this launch could not produce them), for bn_act_rows(..., stats=)."""
    if out_spatial is None:
        out_spatial = tuple(x1.shape[1:4])
    return _ConvTapsRows.apply(x1, x2, wp, bias, tuple(taps), int(istride), tuple(out_spatial), bool(want_stats))


def _pack3d(weight):        # differentiable pack_conv3d_weight
    co_, ci_ = weight.shape[:2]
    return weight.reshape(co_, ci_, -1).permute(2, 0, 1)


def conv3x3x3_rows(x1, x2, weight, bias):
    """Conv3d(k=3, padding=1, stride=1) of the channel concat (x1 | x2) on channels-last rows [n,D,H,W,C] with autograd.
    C1, C2 and Cout must be multiples of 32 (the GEMM K-step; the data gradient swaps the roles of Cin and Cout)."""
    return conv_taps_rows(x1, x2, _pack3d(weight), bias, TAPS_3x3x3)


def conv2d_rows(x, weight, bias, stride=1, want_stats=False):
    """Conv2d(k, stride, padding=k//2) on NHWC rows [N,H,W,C] with autograd (ResNet bottleneck convolutions in training).
    want_stats: (y, stats) as conv_taps_rows."""
    co_, ci_, kh, kw = weight.shape
    N, H, W, C = x.shape
    taps = [(0, ky - kh // 2, kx - kw // 2) for ky in range(kh) for kx in range(kw)]
    Ho, Wo = (H + 2 * (kh // 2) - kh) // stride + 1, (W + 2 * (kw // 2) - kw) // stride + 1
    y = conv_taps_rows(x.reshape(N, 1, H, W, C), None, weight.reshape(co_, ci_, kh * kw).permute(2, 0, 1), bias, taps, stride, (1, Ho, Wo), want_stats)
    if want_stats:
        return y[0].reshape(N, Ho, Wo, co_), y[1]
    return y.reshape(N, Ho, Wo, co_)


def conv3d_rows(x, weight, bias, stride=1, want_stats=False):
    """Conv3d(k = 3, padding = 1, stride 1 or 2) on channels-last rows [n,D,H,W,C] with autograd (the 3-D pose estimator's convolutions,
    models/pose_estimator_3d.py:24-60); Cin and Cout multiples of 32, even input extents for stride 2. Stride 1 is conv3x3x3_rows (Winograd
    where it applies); stride 2 gathers rows 2 o + t through the same GEMM, its data gradient runs as eight parity-phase GEMMs."""
    co_, ci_ = weight.shape[:2]
    n, D, H, W, C = x.shape
    if tuple(weight.shape[2:]) != (3, 3, 3) or C != ci_ or stride not in (1, 2):
        raise ValueError("conv3d_rows: weight %s / input channels %d / stride %d" % (tuple(weight.shape), C, stride))
    if stride == 1:
        return conv_taps_rows(x, None, _pack3d(weight), bias, TAPS_3x3x3, want_stats=want_stats)
    if D % 2 or H % 2 or W % 2:
        raise ValueError("conv3d_rows: stride 2 needs even input extents, got %s" % ((D, H, W),))
    return conv_taps_rows(x, None, _pack3d(weight), bias, TAPS_3x3x3, 2, (D // 2, H // 2, W // 2), want_stats)


class _ConvDirectRows(torch.autograd.Function):
    """Stride-1 'same' convolution with tiny channel counts (Cin in {4, 8, 16}, Cout <= 4) on channels-last rows, all three directions
    on the direct vector-ALU kernels (csrc/conv_direct.hip). wp [T][Cout][Cin] is the differentiable packed weight."""

    @staticmethod
    @_lib.on_tensor_device
    def forward(ctx, x, wp, bias, taps):
        n, D, H, W, Cin = x.shape
        T, Cout, _ = wp.shape
        x = x.contiguous()
        wpc = wp.detach().contiguous()
        out = torch.empty(n, D, H, W, Cout, dtype=torch.float32, device=x.device)
        _lib.check(_lib.lib().forge_conv_direct_fwd(_lib.ptr(x), Cin, _lib.ptr(wpc), _lib.ptr(bias), 1.0, _lib.ptr(out), Cout, n, D, H, W, Cin, Cout,
                                                    _taps_array(taps), T, _lib.current_stream()), "forge_conv_direct_fwd")
        ctx.save_for_backward(x, wpc)
        ctx.meta = (tuple(taps), bias is not None)
        return out

    @staticmethod
    @_lib.on_tensor_device
    def backward(ctx, dy):
        x, wp = ctx.saved_tensors
        taps, has_bias = ctx.meta
        n, D, H, W, Cin = x.shape
        T, Cout, _ = wp.shape
        dy = dy.contiguous()
        dx = dwp = db = None
        if ctx.needs_input_grad[0]:
            dx = torch.empty_like(x)
            _lib.check(_lib.lib().forge_conv_direct_dgrad(_lib.ptr(dy), Cout, _lib.ptr(wp), _lib.ptr(dx), Cin, n, D, H, W, Cin, Cout,
                                                          _taps_array(taps), T, _lib.current_stream()), "forge_conv_direct_dgrad")
        if ctx.needs_input_grad[1]:
            dwp = grad_zeros(wp.shape, wp.device)
            _lib.check(_lib.lib().forge_conv_direct_wgrad(_lib.ptr(dy), Cout, _lib.ptr(x), Cin, _lib.ptr(dwp), n, D, H, W, Cin, Cout,
                                                          _taps_array(taps), T, _lib.current_stream()), "forge_conv_direct_wgrad")
        if has_bias and ctx.needs_input_grad[2]:
            db = colsum(dy.reshape(-1, Cout))
        return dx, dwp, db, None


@_lib.on_tensor_device
def conv_direct(x, ld_in, wp, bias, slope, out, grid, Cin, Cout, taps):
    """Inference launcher of the direct kernel: out [M][Cout] = LeakyReLU(conv(x [M][ld_in], wp [T][Cout][Cin]) + bias, slope)."""
    n, D, H, W = grid
    _lib.check(_lib.lib().forge_conv_direct_fwd(_lib.ptr(x), ld_in, _lib.ptr(wp), _lib.ptr(bias), float(slope), _lib.ptr(out), Cout, n, D, H, W, Cin, Cout,
                                                _taps_array(taps), len(taps), _lib.current_stream()), "forge_conv_direct_fwd")
    return out


def conv_direct_rows(x, wp, bias, taps):
    return _ConvDirectRows.apply(x, wp, bias, tuple(taps))


def conv3x3x3_rows_any(x, weight, bias):
    """conv3x3x3_rows for channel counts that are not multiples of 32 (the heads' 32->16, 32->8, 8->1 convolutions). Tiny layers
    (Cin in {4, 8, 16}, Cout <= 4: the density head's 8->1) run on the direct kernels, Cout <= 16 layers on the narrow-N GEMM kernel
    (forward and, in 16-column blocks, data gradient); anything else is zero-padded to the GEMM K-step
    with differentiable torch ops and the extra output channels are sliced away, so the same forward / dgrad / wgrad kernels serve
    them (what matters is that no MIOpen 3-D weight-gradient solver - 65-110 ms each on these shapes - is involved)."""
    Cout, Cin = weight.shape[:2]
    if Cin in (4, 8, 16) and Cout <= 4 and x.shape[-1] == Cin:
        return conv_direct_rows(x, _pack3d(weight), bias, TAPS_3x3x3)
    if Cout <= 16 and Cout % 4 == 0 and Cin % 16 == 0 and x.shape[-1] == Cin:
        return conv_taps_rows(x.contiguous(), None, _pack3d(weight), bias, TAPS_3x3x3)          # Cout <= 16 kernel, no channel padding
    Cop, Cip = -(-Cout // 32) * 32, -(-Cin // 32) * 32
    if Cop != Cout or Cip != Cin:
        weight = torch.nn.functional.pad(weight, (0, 0, 0, 0, 0, 0, 0, Cip - Cin, 0, Cop - Cout))
        if bias is not None:
            bias = torch.nn.functional.pad(bias, (0, Cop - Cout))
    if x.shape[-1] != Cip:
        x = torch.nn.functional.pad(x, (0, Cip - x.shape[-1]))
    y = conv_taps_rows(x, None, _pack3d(weight), bias, TAPS_3x3x3)
    return y[..., :Cout] if Cop != Cout else y


class _ConvTS2Rows(torch.autograd.Function):
    """nn.ConvTranspose{2,3}d(kernel k, stride 2, padding pad) on channels-last rows [n,D,H,W,Cin] (2-D: D = 1).
    forward: 2^nd output-phase GEMMs; data gradient: ONE stride-2 gather GEMM over the k^nd kernel taps
    (dX[z] = sum_k dY[2z - pad + k] W[:, :, k]); weight gradient: the wgrad kernel with the roles swapped (reduction over input voxels,
    'dy' operand = x, gathered operand = dY at 2z - pad + k)."""

    @staticmethod
    def forward(ctx, x, weight, bias, pad, nd):
        n, D, H, W, Cin = x.shape
        Cout, k = weight.shape[1], weight.shape[-1]
        Do = 2 * D if nd == 3 else 1
        out = torch.empty(n, Do, 2 * H, 2 * W, Cout, dtype=torch.float32, device=x.device)
        taps, wp = convT_phases_merged(weight, pad, nd)
        conv_igemm(x, Cin, Cin, None, 0, 0, wp, bias, None, None, 1.0, None, None, None, out, None, (n, D, H, W), (D, H, W), Cout, Cout,
                   taps, out_grid=(Do, 2 * H, 2 * W), ostride=2, phase=(-1, -1, -1), epilogue=EPI_BIAS)
        ctx.save_for_backward(x, weight)
        ctx.meta = (bias is not None, pad, nd, k)
        return out

    @staticmethod
    def backward(ctx, dy):
        x, weight = ctx.saved_tensors
        has_bias, pad, nd, k = ctx.meta
        n, D, H, W, Cin = x.shape
        Cout = weight.shape[1]
        Do = 2 * D if nd == 3 else 1
        kk = k ** nd
        taps = [(kz - pad if nd == 3 else 0, ky - pad, kx - pad) for kz in (range(k) if nd == 3 else (0,)) for ky in range(k) for kx in range(k)]
        dy = dy.contiguous()
        dx = dw = db = None
        if ctx.needs_input_grad[0]:
            wd = weight.detach().reshape(Cin, Cout, kk).permute(2, 0, 1).contiguous()         # [k][Cin][Cout]
            dx = torch.empty_like(x)
            conv_igemm(dy, Cout, Cout, None, 0, 0, wd, None, None, None, 1.0, None, None, None, dx, None, (n, D, H, W), (Do, 2 * H, 2 * W),
                       Cin, Cin, taps, istride=2, epilogue=EPI_BIAS)
        if ctx.needs_input_grad[1]:
            dwp = grad_zeros((kk, Cin, Cout), x.device)                                          # [k][ci][co]
            conv_wgrad(x, dy, Cout, None, 0, dwp, (n, D, H, W), (Do, 2 * H, 2 * W), Cin, taps, istride=2)
            dw = dwp.permute(1, 2, 0).reshape(weight.shape)
        if has_bias and ctx.needs_input_grad[2]:
            db = colsum(dy.reshape(-1, Cout))
        return dx, dw, db, None, None


def convT_s2_rows(x, weight, bias, pad, nd):
    """ConvTranspose{2,3}d(Cin, Cout, k, stride=2, padding=pad) on rows [n,D,H,W,Cin] -> [n,(2)D,2H,2W,Cout] with autograd; Cin and
    Cout multiples of 32, or both <= 16 and multiples of 16 (narrow-N kernel)."""
    return _ConvTS2Rows.apply(x, weight, bias, int(pad), int(nd))


def convT3d_k4s2p1_rows(x, weight, bias):
    """ConvTranspose3d(Cin, Cout, 4, stride=2, padding=1) on rows [n,D,H,W,Cin] -> [n,2D,2H,2W,Cout]; Cin, Cout % 32 == 0."""
    return convT_s2_rows(x, weight, bias, 1, 3)


def conv2d_rows_any(x, weight, bias):
    """Conv2d(k, stride 1, padding k//2) on NHWC rows [N,H,W,Cin] with autograd for the narrow layers of conv_rgb: tiny layers
    (Cin in {4, 8, 16}, Cout <= 4) on the direct kernels, Cout <= 16 with Cin % 16 == 0 on the narrow-N GEMM kernel."""
    co_, ci_, kh, kw = weight.shape
    N, H, W, C = x.shape
    taps = [(0, ky - kh // 2, kx - kw // 2) for ky in range(kh) for kx in range(kw)]
    wp = weight.reshape(co_, ci_, kh * kw).permute(2, 0, 1)
    x5 = x.reshape(N, 1, H, W, C)
    if ci_ in (4, 8, 16) and co_ <= 4:
        y = conv_direct_rows(x5, wp, bias, taps)
    elif co_ <= 16 and co_ % 4 == 0 and ci_ % 16 == 0:
        y = conv_taps_rows(x5.contiguous(), None, wp, bias, taps)
    else:
        raise ValueError("conv2d_rows_any: Cin=%d Cout=%d is not a narrow layer; use conv2d_rows" % (ci_, co_))
    return y.reshape(N, H, W, co_)


# ------------------------------------------------------------------------------------------------------------------
# frozen-weight data gradients (pose refinement): helpers shared by the heads / conv_rgb backward of the fused inference path
# ------------------------------------------------------------------------------------------------------------------
def narrow_dgrad(dy16, wT16, dx, grid, taps):
    """Data gradient of a stride-1 'same' convolution with Cout <= 16 on the narrow-N kernel: dy16 [.., 16] (Cout zero-padded to the
    16-wide K-step), wT16 [T][Cin][16] (transposed, padded packed weights), dx [.., ld] receives Cin columns in 16-column blocks."""
    Cin = wT16.shape[1]
    ntaps = [(-a, -b, -c) for a, b, c in taps]
    n, D, H, W = grid
    for j in range(0, Cin, 16):
        nb = min(16, Cin - j)
        conv_igemm(dy16, 16, 16, None, 0, 0, wT16[:, j:j + nb].contiguous(), None, None, None, 1.0, None, None, None, dx[..., j:], None,
                   (n, D, H, W), (D, H, W), nb, dx.stride(-2), ntaps, epilogue=EPI_BIAS)
    return dx


def pad_last(w, k):
    """zero-pad the last dim of a packed weight to k"""
    return w if w.shape[-1] == k else torch.nn.functional.pad(w, (0, k - w.shape[-1]))


def direct_dgrad(dy, wp, dx, grid, Cin, Cout, taps):
    n, D, H, W = grid
    _lib.check(_lib.lib().forge_conv_direct_dgrad(_lib.ptr(dy), Cout, _lib.ptr(wp), _lib.ptr(dx), Cin, n, D, H, W, Cin, Cout, _taps_array(taps), len(taps),
                                                  _lib.current_stream()), "forge_conv_direct_dgrad")
    return dx
