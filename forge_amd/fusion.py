"""a4 — multi-view voxel aggregation: 3-D ConvGRU over the view sequence.

Mirror of the reference's models/fusion.py (ConvGRUCell_3D :7-35, ConvGRU_3D :39-95): same class
names, constructor arguments, sub-module names and therefore state_dict keys
(`cells.0.conv_gate.weight [256,256,3,3,3]`, `cells.0.out_gate.*`, `fusion_norm.*`,
`fusion_conv.{0,1,3,4}.*`). The dead bookkeeping of the reference forward (layer_output_list,
last_state_list, torch.stack of every h) is not reproduced — only the returned value is.
"""
import torch
import torch.nn as nn

from . import _lib, convops as co


def hip_inference(module, x):
    """The fused HIP convolution path is taken for inference: eval-mode BN (folded into the GEMM epilogue)
    and no autograd graph. With an autograd graph (training, pose refinement) the same GEMM kernel runs with the plain bias epilogue,
    data / weight gradients on the GEMM / wgrad kernels, and BatchNorm stays a torch module (batch statistics, SyncBN)."""
    return x.is_cuda and x.dtype == torch.float32 and not module.training and not torch.is_grad_enabled()


def frozen_eval(module, x):
    """Eval mode with an autograd graph but NO trainable parameter below `module` (test-time pose refinement, kubric_eval.py:412-530:
    gradients flow only to the poses): forward runs the fused inference epilogues, backward is hand-written data-gradient only."""
    return (x.is_cuda and x.dtype == torch.float32 and not module.training and torch.is_grad_enabled()
            and not any(p.requires_grad for p in module.parameters()))


def affine_act_bwd(dy, y, scale, slope, out=None):
    """dx = dy * scale[c] * (y > 0 ? 1 : slope) on rows [..., C] (last-dim stride 1, row stride arbitrary)."""
    C = dy.shape[-1]
    M = dy.numel() // C
    if out is None:
        out = torch.empty(dy.shape, dtype=torch.float32, device=dy.device)
    _lib.check(_lib.lib().forge_affine_act_bwd(_lib.ptr(dy), dy.stride(-2), _lib.ptr(y), y.stride(-2), _lib.ptr(scale), float(slope), _lib.ptr(out),
                                               out.stride(-2), M, C, _lib.current_stream()), "forge_affine_act_bwd")
    return out


def _rows_ok(t, C):
    return t.stride(1) == 1 and t.stride(0) >= C and t.stride(0) % 4 == 0


@_lib.on_tensor_device
def bn_rows_fwd(x, gamma, beta, running_mean, running_var, momentum, eps, slope, residual=None, nbt=None, group=None, stats=None):
    """Train-mode (Sync)BatchNorm (+ residual) + LeakyReLU(slope) on channels-last rows x [M, C] (csrc/bnorm.hip): float64 batch statistics, running
    statistics / num_batches_tracked updated in place by the same launches. group (a process group with > 1 ranks): the statistics are those of
    all ranks - ONE all-reduce of the float64 (sum x, sum x^2, row count) (RCCL over xGMI; torch's SyncBatchNorm all-gathers per-rank mean /
    invstd / count instead). stats (float64 [blocks][2][C], optional): the per-block column sums / sums of squares of x that the producing
    convolution's GEMM epilogue already wrote (convops.conv_taps_rows(want_stats=True)) - the statistics pass over x is then skipped.
    Returns (y, saved) with `saved` what bn_rows_bwd needs."""
    M, C = x.shape
    dev = x.device
    L, p, st = _lib.lib(), _lib.ptr, _lib.current_stream
    y = torch.empty(M, C, dtype=torch.float32, device=dev)
    mean, invstd = torch.empty(C, dtype=torch.float32, device=dev), torch.empty(C, dtype=torch.float32, device=dev)
    res = None if residual is None else (residual if _rows_ok(residual, C) else residual.contiguous())
    ldres = 0 if res is None else res.stride(0)
    count = None
    pre = 0
    if stats is not None and stats.numel():
        if stats.dim() != 3 or stats.shape[1:] != (2, C) or stats.dtype != torch.float64:
            raise ValueError("bn_rows_fwd: stats must be float64 [blocks][2][%d], got %s %s" % (C, stats.dtype, tuple(stats.shape)))
        pre = stats.shape[0]
        # the producing GEMM wrote whole workgroup tiles of 64 or 128 rows in 32-row blocks (convops.stats_blocks): a stats tensor of another
        # convolution's output (other M) cannot have a block count in that window - refuse it instead of normalising with foreign sums
        lo, hi = (M + 63) // 64 * 2, (M + 127) // 128 * 4
        if not (min(lo, hi) <= pre <= max(lo, hi)):
            raise ValueError("bn_rows_fwd: %d statistics blocks do not belong to an output of %d rows (expected %d or %d)" % (pre, M, lo, hi))
    if group is None:
        ws = stats if pre else torch.empty(L.forge_bn_ws_doubles(C), dtype=torch.float64, device=dev)
        _lib.check(L.forge_bn_train_fwd(p(x), x.stride(0), p(gamma), p(beta), float(eps), float(slope), p(y), C, p(mean), p(invstd), p(running_mean),
                                        p(running_var), float(momentum), p(ws), M, C, p(res), ldres, p(nbt), pre, st()), "forge_bn_train_fwd")
    else:
        import torch.distributed as tdist
        if pre:                                                     # the totals land in the first partial row, the row count behind them (pre >= 2 rows)
            ws = stats.reshape(-1)
        else:
            ws = torch.empty(L.forge_bn_ws_doubles(C) + 1, dtype=torch.float64, device=dev)
        _lib.check(L.forge_bn_sync_stats(p(x), x.stride(0), p(ws), M, C, pre, st()), "forge_bn_sync_stats")
        tot = ws[:2 * C + 1]
        tot[2 * C:].fill_(float(M))                                 # the row count rides on the same all-reduce and stays on the device
        tdist.all_reduce(tot, op=tdist.ReduceOp.SUM, group=group)
        _lib.check(L.forge_bn_sync_fwd_apply(p(x), x.stride(0), p(gamma), p(beta), float(eps), float(slope), p(y), C, p(mean), p(invstd),
                                             p(running_mean), p(running_var), float(momentum), p(tot), 0, M, C, p(res), ldres, p(nbt), st()),
                   "forge_bn_sync_fwd_apply")                       # M_total = 0: "read the all-rank row count at totals[2C]"
        count = tot[2 * C:].clone()
    return y, (x, gamma, beta, mean, invstd, y if res is not None else None, count, float(slope), group)


@_lib.on_tensor_device
def bn_rows_bwd(saved, dy, need_dres=False):
    """Backward of bn_rows_fwd: (dx, dgamma, dbeta, dres). Under a process group: one all-reduce of (sum g, sum g xhat); dgamma / dbeta are this
    rank's sums, as torch's SyncBatchNorm (DDP averages them)."""
    x, gamma, beta, mean, invstd, y, count, slope, group = saved
    M, C = x.shape
    dev = x.device
    L, p, st = _lib.lib(), _lib.ptr, _lib.current_stream
    dy = dy if _rows_ok(dy, C) else dy.contiguous()
    dx = torch.empty(M, C, dtype=torch.float32, device=dev)
    dres = torch.empty(M, C, dtype=torch.float32, device=dev) if (y is not None and need_dres) else None
    dg = torch.empty(C, dtype=torch.float32, device=dev) if gamma is not None else None
    db = torch.empty(C, dtype=torch.float32, device=dev) if beta is not None else None
    if group is None:
        ws = torch.empty(L.forge_bn_ws_doubles(C), dtype=torch.float64, device=dev)
        _lib.check(L.forge_bn_train_bwd(p(dy), dy.stride(0), p(x), x.stride(0), p(gamma), p(beta), p(mean), p(invstd), slope, p(dx), C, p(dg), p(db), p(ws),
                                        M, C, p(y), C, p(dres), C, st()), "forge_bn_train_bwd")
    else:
        import torch.distributed as tdist
        ws = torch.empty(L.forge_bn_ws_doubles(C) + 1, dtype=torch.float64, device=dev)
        _lib.check(L.forge_bn_sync_bwd_reduce(p(dy), dy.stride(0), p(x), x.stride(0), p(gamma), p(beta), p(mean), p(invstd), slope, p(dg), p(db), p(ws),
                                              M, C, p(y), C, st()), "forge_bn_sync_bwd_reduce")
        tot = ws[:2 * C]
        tdist.all_reduce(tot, op=tdist.ReduceOp.SUM, group=group)
        ws[2 * C:2 * C + 1].copy_(count)                            # the forward's all-rank row count, read by the kernel at totals[2C]
        _lib.check(L.forge_bn_sync_bwd_apply(p(dy), dy.stride(0), p(x), x.stride(0), p(gamma), p(beta), p(mean), p(invstd), slope, p(dx), C, p(ws),
                                             0, M, C, p(y), C, p(dres), C, st()), "forge_bn_sync_bwd_apply")
    return dx, dg, db, dres


class _BNTrainRows(torch.autograd.Function):
    """Train-mode (Sync)BatchNorm (+ residual) + LeakyReLU(slope) on channels-last rows [M, C] with autograd: bn_rows_fwd / bn_rows_bwd.
    slope 1 = no activation, 0 = ReLU. residual [M, C] (optional): y = act(bn(x) + residual) - the bottleneck tail of the ResNet trunk -
    and the backward returns d residual."""

    @staticmethod
    def forward(ctx, x, gamma, beta, running_mean, running_var, momentum, eps, slope, residual, nbt, group, stats=None):
        y, saved = bn_rows_fwd(x, gamma, beta, running_mean, running_var, momentum, eps, slope, residual, nbt, group, stats)
        tensors = tuple(saved[:7])
        ctx.save_for_backward(*tensors)
        ctx.slope, ctx.group = saved[7], saved[8]
        return y

    @staticmethod
    def backward(ctx, dy):
        dx, dg, db, dres = bn_rows_bwd(tuple(ctx.saved_tensors) + (ctx.slope, ctx.group), dy, need_dres=ctx.needs_input_grad[8])
        return dx, dg, db, None, None, None, None, None, dres, None, None, None


class _BNEvalRows(torch.autograd.Function):
    """Eval-mode BatchNorm (+ residual) + LeakyReLU(slope) on channels-last rows [M, C] under autograd with trainable gamma / beta
    (forge_bn_eval_fwd): running statistics read, never updated. Backward: dgamma / dbeta from forge_bn_sync_bwd_reduce, dx = gamma invstd g from
    forge_bn_sync_bwd_apply with all-zero totals - the statistics do not depend on x."""

    @staticmethod
    @_lib.on_tensor_device
    def forward(ctx, x, gamma, beta, running_mean, running_var, eps, slope, residual):
        M, C = x.shape
        L, p = _lib.lib(), _lib.ptr
        y = torch.empty(M, C, dtype=torch.float32, device=x.device)
        mean, invstd = torch.empty(C, dtype=torch.float32, device=x.device), torch.empty(C, dtype=torch.float32, device=x.device)
        res = None if residual is None else (residual if _rows_ok(residual, C) else residual.contiguous())
        _lib.check(L.forge_bn_eval_fwd(p(x), x.stride(0), p(gamma), p(beta), p(running_mean), p(running_var), float(eps), float(slope), p(y), C, p(mean),
                                       p(invstd), M, C, p(res), 0 if res is None else res.stride(0), _lib.current_stream()), "forge_bn_eval_fwd")
        ctx.save_for_backward(x, gamma, beta, mean, invstd, y if res is not None else None)
        ctx.slope = float(slope)
        return y

    @staticmethod
    @_lib.on_tensor_device
    def backward(ctx, dy):
        x, gamma, beta, mean, invstd, y = ctx.saved_tensors
        M, C = x.shape
        L, p, st = _lib.lib(), _lib.ptr, _lib.current_stream
        dy = dy if _rows_ok(dy, C) else dy.contiguous()
        dx = torch.empty(M, C, dtype=torch.float32, device=x.device)
        dres = torch.empty(M, C, dtype=torch.float32, device=x.device) if (y is not None and ctx.needs_input_grad[7]) else None
        dg = torch.empty(C, dtype=torch.float32, device=x.device) if gamma is not None else None
        db = torch.empty(C, dtype=torch.float32, device=x.device) if beta is not None else None
        if dg is not None or db is not None:
            ws = torch.empty(L.forge_bn_ws_doubles(C), dtype=torch.float64, device=x.device)
            _lib.check(L.forge_bn_sync_bwd_reduce(p(dy), dy.stride(0), p(x), x.stride(0), p(gamma), p(beta), p(mean), p(invstd), ctx.slope, p(dg), p(db), p(ws),
                                                  M, C, p(y), C, st()), "forge_bn_sync_bwd_reduce")
        zero = torch.zeros(2 * C, dtype=torch.float64, device=x.device)
        _lib.check(L.forge_bn_sync_bwd_apply(p(dy), dy.stride(0), p(x), x.stride(0), p(gamma), p(beta), p(mean), p(invstd), ctx.slope, p(dx), C, p(zero),
                                             M, M, C, p(y), C, p(dres), C, st()), "forge_bn_sync_bwd_apply")
        return dx, dg, db, None, None, None, None, dres


def _sync_world(bn):
    """World size of the SyncBatchNorm module's process group (1: not initialised / single process -> plain batch statistics)."""
    import torch.distributed as tdist
    if not (isinstance(bn, nn.SyncBatchNorm) and tdist.is_available() and tdist.is_initialized()):
        return 1
    return tdist.get_world_size(bn.process_group)


def _bn_group(bn):
    """The process group a SyncBatchNorm module's statistics are taken over (torch.distributed's default group when the module has none), or
    None for plain BatchNorm / a single process."""
    if _sync_world(bn) <= 1:
        return None
    import torch.distributed as tdist
    return bn.process_group if bn.process_group is not None else tdist.group.WORLD


def bn_hip_train(bn, rows):
    """True when bn_act_rows runs `bn` on the HIP kernels with batch statistics (train mode, fp32 rows on the MI355X, C % 4 == 0, a momentum)."""
    return (bn.training and rows.is_cuda and rows.dtype == torch.float32 and rows.shape[-1] % 4 == 0
            and (bn.momentum is not None or not bn.track_running_stats))


def bn_module_args(bn):
    """(running_mean, running_var, momentum, eps, num_batches_tracked, group) of a BatchNorm module for bn_rows_fwd."""
    track = bn.track_running_stats and bn.running_mean is not None
    return (bn.running_mean if track else None, bn.running_var if track else None, bn.momentum if bn.momentum is not None else 0.0, bn.eps,
            bn.num_batches_tracked if (track and bn.num_batches_tracked is not None) else None, _bn_group(bn))


def bn_act_rows(bn, rows, slope=1.0, residual=None, stats=None):
    """BatchNorm module `bn` (+ `residual`, same shape as rows) + LeakyReLU(slope) (1 = none, 0 = ReLU) applied to channels-last rows [..., C].
    Train mode runs the HIP kernels of csrc/bnorm.hip - per-process batch statistics for nn.BatchNorm*, statistics over the module's process
    group for nn.SyncBatchNorm (one all-reduce of 2C+1 float64 forward, 2C backward: bn_rows_fwd / bn_rows_bwd), running statistics and
    num_batches_tracked updated by the same launches; eval mode under autograd (a fine-tune with frozen statistics) runs forge_bn_eval_fwd with
    the running statistics (_BNEvalRows). One path: host tensors, C % 4 != 0 and a cumulative-average momentum (momentum=None) raise."""
    C = rows.shape[-1]
    require_hip_input("BatchNorm on channels-last rows", rows)
    if C % 4:
        raise RuntimeError("forge_amd: the HIP BatchNorm kernels need a channel count that is a multiple of 4 (got %d)" % C)
    if bn.training and bn.momentum is None and bn.track_running_stats:
        raise RuntimeError("forge_amd: BatchNorm with momentum=None (cumulative moving average) is not implemented by the HIP kernels "
                           "(the reference's modules use the default momentum 0.1)")
    if not bn.training and bn.track_running_stats and bn.running_mean is not None:
        x = rows.reshape(-1, C)
        x = x if _rows_ok(x, C) else x.contiguous()
        res = None if residual is None else residual.reshape(-1, C)
        return _BNEvalRows.apply(x, bn.weight, bn.bias, bn.running_mean, bn.running_var, bn.eps, slope, res).reshape(rows.shape)
    if bn_hip_train(bn, rows) or not bn.track_running_stats or bn.running_mean is None:
        x = rows.reshape(-1, C)
        x = x if _rows_ok(x, C) else x.contiguous()
        rm, rv, mom, eps, nbt, group = bn_module_args(bn)
        res = None if residual is None else residual.reshape(-1, C)
        args = (x, bn.weight, bn.bias, rm, rv, mom, eps, slope, res, nbt)
        # SyncBatchNorm in a job with > 1 ranks (the reference's training configuration): statistics over all ranks, one all-reduce each way
        y = _BNTrainRows.apply(*args, group, stats if (stats is not None and stats.numel() and x.data_ptr() == rows.data_ptr()) else None)
        return y.reshape(rows.shape)
    raise RuntimeError("forge_amd: unsupported BatchNorm configuration %r" % (bn,))


def require_hip_input(what, x, channels=None):
    """The product has ONE implementation per op - the HIP kernels. Anything they cannot take is an error, never a silent stock-PyTorch
    detour (north_star: no dual code paths)."""
    if not (torch.is_tensor(x) and x.is_cuda and x.dtype == torch.float32):
        raise RuntimeError("forge_amd: %s runs only on the MI355X HIP kernels and needs a float32 tensor on a cuda/HIP device (got %s on %s); "
                           "there is no CPU or stock-PyTorch path" % (what, getattr(x, "dtype", type(x)), getattr(x, "device", "?")))
    if channels is not None and channels % 32:
        raise RuntimeError("forge_amd: %s needs a channel count that is a multiple of the GEMM K-step 32 (got %d)" % (what, channels))


class _GRUCellRows(torch.autograd.Function):
    """One ConvGRU step (models/fusion.py:29-35) on channels-last rows with an autograd graph. Both convolutions run on the MFMA
    implicit-GEMM kernel (bias epilogue), data / weight gradients on the same GEMM / the wgrad kernel, and each element-wise half of
    the cell is one HIP kernel per direction (csrc/gru.hip) instead of ~23 generic tensor ops per step.
      x [b,D,H,W,C] (a view with a batch stride is fine), h [b,D,H,W,C] dense, wg [27][2C][2C], wo [27][C][2C] packed weights."""

    @staticmethod
    @_lib.on_tensor_device
    def forward(ctx, x, h, wg, bg, wo, bo):
        b, D, H, W, C = h.shape
        M = b * D * H * W
        dev = h.device
        h = h.contiguous()
        wgc, woc = wg.detach().contiguous(), wo.detach().contiguous()
        bsx = co._batch_stride_rows(x)
        grid = (b, D, H, W)
        new = lambda c: torch.empty(b, D, H, W, c, dtype=torch.float32, device=dev)
        g = new(2 * C)
        co.conv3_launch(x, C, h, C, wgc, bg, g, grid, 2 * C, bs1=bsx)
        z, r, hr = new(C), new(C), new(C)
        L, p, st = _lib.lib(), _lib.ptr, _lib.current_stream
        _lib.check(L.forge_gru_gates_fwd(p(g), p(h), p(z), p(r), p(hr), M, C, st()), "forge_gru_gates_fwd")
        cand = new(C)
        co.conv3_launch(x, C, hr, C, woc, bo, cand, grid, C, bs1=bsx)
        hn = new(C)
        _lib.check(L.forge_gru_state_fwd(p(cand), p(h), p(z), p(hn), M, C, st()), "forge_gru_state_fwd")     # cand <- tanh(conv)
        ctx.save_for_backward(x, h, z, r, hr, cand, wgc, woc)
        ctx.has_bias = (bg is not None, bo is not None)
        return hn

    @staticmethod
    @_lib.on_tensor_device
    def backward(ctx, dhn):
        x, h, z, r, hr, cand, wg, wo = ctx.saved_tensors
        b, D, H, W, C = h.shape
        M = b * D * H * W
        dev = h.device
        grid = (b, D, H, W)
        bsx = co._batch_stride_rows(x)
        new = lambda c: torch.empty(b, D, H, W, c, dtype=torch.float32, device=dev)
        L, p, st = _lib.lib(), _lib.ptr, _lib.current_stream
        dhn = dhn.contiguous()
        dh, dz, dc = new(C), new(C), new(C)
        _lib.check(L.forge_gru_state_bwd(p(dhn), C, p(h), p(z), p(cand), p(dh), p(dz), p(dc), M, C, None, 0, 0, 0, st()), "forge_gru_state_bwd")
        # candidate conv: c = conv([x | h r], wo)
        dxh = new(2 * C)                                                                   # (dx | d(h r))
        co.conv3_launch(dc, C, None, 0, wo, None, dxh, grid, 2 * C, dgrad=True)
        dwo = dbo = dwg = dbg = None
        if ctx.needs_input_grad[4]:
            dwo = co.grad_zeros(wo.shape, wo.device)
            co.conv3_wgrad(dc, x, C, hr, C, dwo, grid, C, bs1=bsx)
        if ctx.has_bias[1] and ctx.needs_input_grad[5]:
            dbo = co.colsum(dc.reshape(M, C))
        # gates: g = conv([x | h], wg); z = sigmoid(g[:C]), r = sigmoid(g[C:]), hr = h r
        dg = new(2 * C)
        _lib.check(L.forge_gru_gates_bwd(p(dz), _lib.ptr(dxh[..., C:]), 2 * C, p(h), p(z), p(r), p(dg), p(dh), None, 0, M, C, None, 0, 0, 0, st()), "forge_gru_gates_bwd")
        dxh2 = new(2 * C)
        co.conv3_launch(dg, 2 * C, None, 0, wg, None, dxh2, grid, 2 * C, dgrad=True)
        if ctx.needs_input_grad[2]:
            dwg = co.grad_zeros(wg.shape, wg.device)
            co.conv3_wgrad(dg, x, C, h, C, dwg, grid, 2 * C, bs1=bsx)
        if ctx.has_bias[0] and ctx.needs_input_grad[3]:
            dbg = co.colsum(dg.reshape(M, 2 * C))
        dx = (dxh[..., :C] + dxh2[..., :C]) if ctx.needs_input_grad[0] else None
        dh_total = (dh + dxh2[..., C:]) if ctx.needs_input_grad[1] else None
        return dx, dh_total, dwg, dbg, dwo, dbo


class _GRUCellPreRows(torch.autograd.Function):
    """ConvGRU step whose INPUT contributions are precomputed: with W = (W_x | W_h) along Cin, conv([x, h], W) = conv(x, W_x) +
    conv(h, W_h), so  g = gx + conv(h, Wg_h) + bg,  c = cx + conv(h r, Wo_h) + bo  (gx, cx enter through the GEMM's residual
    operand). Used when several fusions share input views (FORGE_poseEstimator3D fuses views (0,1,2), (3,4) and (0..4) of the same
    rotated features, model_single_pose_estimator.py:108-120): the x halves are then computed once per view instead of once per
    (fusion, view) - 25 % fewer GRU FLOPs forward and backward. Same arithmetic up to the order of the fp32 additions.
      gx [b,D,H,W,2C], cx [b,D,H,W,C], h [b,D,H,W,C] dense rows; wgh [27][2C][C], woh [27][C][C] packed hidden-state weights."""

    @staticmethod
    @_lib.on_tensor_device
    def forward(ctx, gx, cx, h, wgh, bg, woh, bo):
        b, D, H, W, C = h.shape
        M = b * D * H * W
        dev = h.device
        h, gx, cx = h.contiguous(), gx.contiguous(), cx.contiguous()
        wg, wo = wgh.detach().contiguous(), woh.detach().contiguous()
        grid = (b, D, H, W)
        new = lambda c: torch.empty(b, D, H, W, c, dtype=torch.float32, device=dev)
        L, p, st = _lib.lib(), _lib.ptr, _lib.current_stream
        g = new(2 * C)
        co.conv3_launch(h, C, None, 0, wg, bg, g, grid, 2 * C, residual=gx)
        z, r, hr = new(C), new(C), new(C)
        _lib.check(L.forge_gru_gates_fwd(p(g), p(h), p(z), p(r), p(hr), M, C, st()), "forge_gru_gates_fwd")
        cand = new(C)
        co.conv3_launch(hr, C, None, 0, wo, bo, cand, grid, C, residual=cx)
        hn = new(C)
        _lib.check(L.forge_gru_state_fwd(p(cand), p(h), p(z), p(hn), M, C, st()), "forge_gru_state_fwd")
        ctx.save_for_backward(h, z, r, hr, cand, wg, wo)
        ctx.has_bias = (bg is not None, bo is not None)
        return hn

    @staticmethod
    @_lib.on_tensor_device
    def backward(ctx, dhn):
        h, z, r, hr, cand, wg, wo = ctx.saved_tensors
        b, D, H, W, C = h.shape
        M = b * D * H * W
        dev = h.device
        grid = (b, D, H, W)
        new = lambda c: torch.empty(b, D, H, W, c, dtype=torch.float32, device=dev)
        L, p, st = _lib.lib(), _lib.ptr, _lib.current_stream
        dhn = dhn.contiguous()
        dh, dz, dc = new(C), new(C), new(C)
        _lib.check(L.forge_gru_state_bwd(p(dhn), C, p(h), p(z), p(cand), p(dh), p(dz), p(dc), M, C, None, 0, 0, 0, st()), "forge_gru_state_bwd")
        dhr = new(C)
        co.conv3_launch(dc, C, None, 0, wo, None, dhr, grid, C, dgrad=True)
        dwo = dbo = dwg = dbg = None
        if ctx.needs_input_grad[5]:
            dwo = co.grad_zeros(wo.shape, wo.device)
            co.conv3_wgrad(dc, hr, C, None, 0, dwo, grid, C)
        if ctx.has_bias[1] and ctx.needs_input_grad[6]:
            dbo = co.colsum(dc.reshape(M, C))
        dg = new(2 * C)
        _lib.check(L.forge_gru_gates_bwd(p(dz), p(dhr), C, p(h), p(z), p(r), p(dg), p(dh), None, 0, M, C, None, 0, 0, 0, st()), "forge_gru_gates_bwd")
        dh_total = new(C)                                          # dh (state + reset paths) + conv^T(dg, Wg_h), added in the GEMM epilogue
        co.conv3_launch(dg, 2 * C, None, 0, wg, None, dh_total, grid, C, residual=dh, dgrad=True)
        if ctx.needs_input_grad[3]:
            dwg = co.grad_zeros(wg.shape, wg.device)
            co.conv3_wgrad(dg, h, C, None, 0, dwg, grid, 2 * C)
        if ctx.has_bias[0] and ctx.needs_input_grad[4]:
            dbg = co.colsum(dg.reshape(M, 2 * C))
        return (dg if ctx.needs_input_grad[0] else None), (dc if ctx.needs_input_grad[1] else None), dh_total, dwg, dbg, dwo, dbo


class _FuseFrozen(torch.autograd.Function):
    """Encoder3D.fuse for frozen weights (eval mode, gradients only w.r.t. the input views): forward = ConvGRU_3D.fuse_hip - the fused
    conv + BN + LeakyReLU launches of fusion_conv and TWO launches per view (gates: cat / conv / sigmoid / h*r; state: cat / conv /
    tanh / lerp, final BatchNorm folded) - with the reset gate and the candidate additionally stored (forge_conv_igemm out3); backward
    = per view: state half (one kernel), data-gradient GEMM of the candidate conv, gate half (one kernel, its dh sum written beside
    the x-gradient half), data-gradient GEMM of the gate conv with that buffer as residual -> (dx_t | dh) in one tensor; no weight
    gradients, no generic element-wise kernels."""

    @staticmethod
    @_lib.on_tensor_device
    def forward(ctx, x, gru, skip_dx0=False, const0=None):
        ctx.skip_dx0 = bool(skip_dx0)
        b, t, C, D, H, W = x.shape
        xr = x.permute(0, 1, 3, 4, 5, 2)
        xr = xr if xr.is_contiguous() else xr.contiguous()
        dev, M, vol = x.device, b * D * H * W, D * H * W
        grid, ig, taps = (b, D, H, W), (D, H, W), co.TAPS_3x3x3
        new = lambda c=C: torch.empty(M, c, dtype=torch.float32, device=dev)
        t0, h = new(), new()
        wino = co.wino_enabled() and co.wino_fits(b, D, H, W, C, views=t)
        steps, out = [], new()
        if wino:                                                             # ConvGRU_3D._fuse_wino with the reset gate / candidate kept
            p = gru._packed_wino()
            Ht, Wt = H // 2, W // 2
            R = b * D * Ht * Wt
            Vx = co.wino_input(xr, C, C, b * t, D, H, W)
            Vh = torch.empty(16, R, C, dtype=torch.float32, device=dev)
            Mm = torch.empty(16, R, 2 * C, dtype=torch.float32, device=dev)
            Mc = Mm.view(-1)[:16 * R * C].view(16, R, C)
            gru._wino_h0(p, xr, grid, Vh, Mc, t0, h, nsum=t, sum_stride=vol, bs=t * vol)
            # const0 (a dict the caller keeps across calls; with skip_dx0): view 0 holds the SAME values in every call (the un-warped reference
            # view of frozen features, kubric_eval.py:456-470) - the point products of its input halves, V_x0 (x) U_x of both GRU convolutions,
            # are made once and added inside the inverse transforms of step 0, whose GEMMs then contract the hidden-state half only (K = 3 C
            # instead of 6 C). Same arithmetic up to the order of the fp32 additions (the halves meet before A^T . A instead of inside the K loop).
            hoist = const0 is not None and skip_dx0
            if hoist:
                ps = gru._packed_wino_halves()
                if "MXg0" not in const0:
                    Vx0 = co.wino_input(xr[:, 0], C, C, b, D, H, W, bs=t * vol)
                    const0["MXg0"] = torch.empty(16, R, 2 * C, dtype=torch.float32, device=dev)
                    const0["MXc0"] = torch.empty(16, R, C, dtype=torch.float32, device=dev)
                    co.wino_gemm(Vx0, C, None, 0, ps["gate_Ux"], const0["MXg0"], b, D, Ht, Wt, 2 * C)
                    co.wino_gemm(Vx0, C, None, 0, ps["out_Ux"], const0["MXc0"], b, D, Ht, Wt, C)
            for ti in range(t):
                z, hr, r, hn, cand = new(), new(), new(), new(), new()
                first = hoist and ti == 0
                co.wino_input(h, C, C, b, D, H, W, out=Vh)
                if first:
                    co.wino_gemm(Vh, C, None, 0, ps["gate_Uh"], Mm, b, D, Ht, Wt, 2 * C)
                else:
                    co.wino_gemm(Vx, C, Vh, C, p["gate_U"], Mm, b, D, Ht, Wt, 2 * C, view=ti, views=t)
                co.wino_output(Mm, p["gate_b"], None, None, 1.0, None, h, None, z, hr, r, *grid, 2 * C, C, co.EPI_GRU_GATES,
                               Mm2=const0["MXg0"] if first else None)
                co.wino_input(hr, C, C, b, D, H, W, out=Vh)
                if first:
                    co.wino_gemm(Vh, C, None, 0, ps["out_Uh"], Mc, b, D, Ht, Wt, C)
                else:
                    co.wino_gemm(Vx, C, Vh, C, p["out_U"], Mc, b, D, Ht, Wt, C, view=ti, views=t)
                co.wino_output(Mc, p["out_b"], p["norm"][0], p["norm"][1], 1.0, None, h, z, hn, out if ti == t - 1 else None, cand, *grid, C, C,
                               co.EPI_GRU_OUT, Mm2=const0["MXc0"] if first else None)
                steps.append((h, z, r, cand))
                h = hn
        else:
            p = gru._packed()
            mean = xr.mean(dim=1).reshape(M, C)
            co.conv_igemm(mean, C, C, None, 0, 0, p["fc0_w"], p["fc0_b"], p["bn1"][0], p["bn1"][1], 0.01, None, None, None, t0, None, grid, ig, C, C, taps,
                          epilogue=co.EPI_AFFINE_ACT)
            co.conv_igemm(t0, C, C, None, 0, 0, p["fc3_w"], p["fc3_b"], p["bn4"][0], p["bn4"][1], 0.01, None, None, None, h, None, grid, ig, C, C, taps,
                          epilogue=co.EPI_AFFINE_ACT)
            for ti in range(t):
                xt = xr[:, ti]
                z, hr, r, hn, cand = new(), new(), new(), new(), new()
                co.conv_igemm(xt, C, C, h, C, C, p["gate_w"], p["gate_b"], None, None, 1.0, None, h, None, z, hr, grid, ig, 2 * C, C, taps,
                              epilogue=co.EPI_GRU_GATES, bs1=t * vol, out3=r)
                co.conv_igemm(xt, C, C, hr, C, C, p["out_w"], p["out_b"], p["norm"][0], p["norm"][1], 1.0, None, h, z, hn, out if ti == t - 1 else None,
                              grid, ig, C, C, taps, epilogue=co.EPI_GRU_OUT, bs1=t * vol, out3=cand)
                steps.append((h, z, r, cand))
                h = hn
        ctx.gru, ctx.shape = gru, (b, t, C, D, H, W)
        ctx.save_for_backward(t0, *[v for st in steps for v in st])          # per step (h, z, r, cand); steps[0][0] is h0
        return out.reshape(b, D, H, W, C).permute(0, 4, 1, 2, 3)

    @staticmethod
    @_lib.on_tensor_device
    def backward(ctx, dout):
        gru = ctx.gru
        b, t, C, D, H, W = ctx.shape
        saved = ctx.saved_tensors
        t0, steps = saved[0], [saved[1 + 4 * i:5 + 4 * i] for i in range(t)]
        p = gru._packed_T()
        dev, M = dout.device, b * D * H * W
        grid = (b, D, H, W)
        new = lambda c=C: torch.empty(M, c, dtype=torch.float32, device=dev)
        L, ptr, st = _lib.lib(), _lib.ptr, _lib.current_stream
        # data gradients of the four convolutions: Winograd launches with the cached transposed-domain weights, else the direct kernel
        dgrad = lambda dy, Cdy, k, dst, Cdst, residual=None: co.conv3_launch(dy, Cdy, None, 0, p[k + "_w"], None, dst, grid, Cdst, residual=residual,
                                                                           dgrad=True, U=p[k + "_UT"], wT=p[k + "_wT"])
        dr = dout.permute(0, 2, 3, 4, 1)
        dr = (dr if dr.is_contiguous() else dr.contiguous()).reshape(M, C)
        dhn = dr * p["norm_scale"]                                           # out = fusion_norm(h_T) = h_T * scale + shift
        ld_dhn = C
        dx = torch.empty(b, t, D, H, W, C, dtype=torch.float32, device=dev)
        for ti in reversed(range(t)):
            h, z, r, cand = steps[ti]
            dh, dz, dc, dg = new(), new(), new(), new(2 * C)
            _lib.check(L.forge_gru_state_bwd(ptr(dhn), ld_dhn, ptr(h), ptr(z), ptr(cand), ptr(dh), ptr(dz), ptr(dc), M, C, None, 0, 0, 0, st()), "forge_gru_state_bwd")
            if ti == 0 and ctx.skip_dx0:
                # the first view of the sequence is the un-warped reference view and the caller needs no gradient for it (pose refinement: frozen
                # features, the reference pose is fixed): both data gradients produce their hidden-state half only (N = C instead of 2C)
                dhr = dgrad(dc, C, "out_h", new(), C)
                dhp = new()
                _lib.check(L.forge_gru_gates_bwd(ptr(dz), ptr(dhr), C, ptr(h), ptr(z), ptr(r), ptr(dg), ptr(dh), ptr(dhp), C, M, C, None, 0, 0, 0, st()),
                           "forge_gru_gates_bwd")
                toth = dgrad(dg, 2 * C, "gate_h", new(), C, residual=dhp)
                dx[:, 0].zero_()
                dhn, ld_dhn = toth, C
                continue
            dxh = new(2 * C)                                                 # (d x_t | d (h r)) of the candidate conv
            dgrad(dc, C, "out", dxh, 2 * C)
            # dg = gate pre-activation gradients; dh + d(hr) r lands in dxh's right half (over d(hr)): dxh = (dx_t part 1 | dh partial)
            _lib.check(L.forge_gru_gates_bwd(ptr(dz), ptr(dxh[:, C:]), 2 * C, ptr(h), ptr(z), ptr(r), ptr(dg), ptr(dh), ptr(dxh[:, C:]), 2 * C, M, C,
                                             None, 0, 0, 0, st()), "forge_gru_gates_bwd")
            tot = new(2 * C)                                                 # conv^T(dg, Wg) + dxh = (d x_t | d h_{t-1})
            dgrad(dg, 2 * C, "gate", tot, 2 * C, residual=dxh)
            dx[:, ti] = tot.view(b, D, H, W, 2 * C)[..., :C]
            dhn, ld_dhn = tot[:, C:], 2 * C
        # h0 = lrelu(bn4(conv(lrelu(bn1(conv(mean_t x))))))
        h0 = steps[0][0]
        g = torch.empty(M, C, dtype=torch.float32, device=dev)
        affine_act_bwd(dhn, h0, p["bn4_scale"], 0.01, out=g)
        g2 = new()
        dgrad(g, C, "fc3", g2, C)
        affine_act_bwd(g2, t0, p["bn1_scale"], 0.01, out=g)
        dgrad(g, C, "fc0", g2, C)
        dx.add_(g2.reshape(b, 1, D, H, W, C), alpha=1.0 / t)
        return dx.permute(0, 1, 5, 2, 3, 4), None, None, None


def _flatten_saved(obj, tensors):
    """Nested lists / tuples / dicts of tensors and constants -> a structure description with the tensors moved to `tensors` (for
    ctx.save_for_backward: version-counter checks on the parameters among them, storage released when the backward pass has run)."""
    if torch.is_tensor(obj):
        tensors.append(obj)
        return ("t", len(tensors) - 1)
    if isinstance(obj, (list, tuple)):
        return ("l" if isinstance(obj, list) else "u", [_flatten_saved(o, tensors) for o in obj])
    if isinstance(obj, dict):
        return ("d", [(k, _flatten_saved(v, tensors)) for k, v in obj.items()])
    return ("c", obj)


def _unflatten_saved(spec, tensors):
    kind, val = spec
    if kind == "t":
        return tensors[val]
    if kind in ("l", "u"):
        items = [_unflatten_saved(v, tensors) for v in val]
        return items if kind == "l" else tuple(items)
    if kind == "d":
        return {k: _unflatten_saved(v, tensors) for k, v in val}
    return val


class _FuseGroupsTrain(torch.autograd.Function):
    """Several ConvGRU fusions over subsets of the SAME views WITH weight gradients (the GT-pose training step, model_single_pose_estimator.py:
    108-120: views (0,1,2), (3,4), (0..4)) as ONE autograd node whose forward and backward are hand-scheduled on the Winograd launches:
      forward   the views are transformed once (V_x) and the point products of the INPUT halves of both GRU convolutions are made once, for
                all views, and stay in the Winograd domain (MX = V_x (x) U_x: never inverse-transformed); every (group, view) step runs the
                hidden-state halves only and the inverse transform adds the view's MX before the fused GRU tails (wino_output EPI_GRU_GATES /
                EPI_GRU_OUT, which also emit r and tanh(c) for the backward); train-mode BatchNorm (batch statistics, SyncBN) through
                bn_rows_fwd. The transforms V_h / V_hr of every step are KEPT: the weight gradient needs exactly them.
      backward  per step (reverse): state half, data + weight gradient of the candidate conv's hidden half, gate half, data + weight gradient
                of the gate conv's hidden half; the two element-wise kernels also SUM their dc / dg into per-view buffers
                (forge_gru_*_bwd acc_mode), so that the shared input halves are differentiated once per view at the end - two data-gradient
                and two weight-gradient launches over all views - instead of once per (group, view); weight gradients accumulate in the
                Winograd domain over all steps (dU) and are brought back (G^T dU G) once per weight.
    Against the autograd-composed form (_GRUCellPreRows per step) this removes, per step, two input transforms, the separate gate / state
    forward kernels, the gx / cx residual round trips and autograd's gradient accumulations, stacks and zero-fills; the arithmetic is the same up
    to the order of fp32 additions (the input halves are added before the inverse transform instead of after it).
    x [b,t,C,D,H,W] (channels-last memory), weights = the module's parameters; returns one fused volume [b,C,D,H,W] per group."""

    @staticmethod
    @_lib.on_tensor_device
    def forward(ctx, x, Wg, bg, Wo, bo, w0, b0, g1, be1, w3, b3, g4, be4, gn, bn_, gru, groups):
        b, t, C, D, H, W = x.shape
        xr = x.detach().permute(0, 1, 3, 4, 5, 2)
        xr = xr if xr.is_contiguous() else xr.contiguous()
        dev, M, vol, Ht, Wt = x.device, b * D * H * W, D * H * W, H // 2, W // 2
        R1 = D * Ht * Wt
        R = b * R1
        geo = (b, D, H, W)
        fc, norm = gru.fusion_conv, gru.fusion_norm
        new = lambda c=C: torch.empty(M, c, dtype=torch.float32, device=dev)
        newV = lambda rows, c: torch.empty(16, rows, c, dtype=torch.float32, device=dev)
        pk = lambda w: co.pack_conv3d_weight(w)                              # [27][Cout][Cin]
        wpg, wpo = pk(Wg), pk(Wo)
        packs = {"gx": wpg[:, :, :C].contiguous(), "gh": wpg[:, :, C:].contiguous(), "ox": wpo[:, :, :C].contiguous(), "oh": wpo[:, :, C:].contiguous(),
                 "f0": pk(w0), "f3": pk(w3)}
        U = {k: co.wino_pack_packed(v) for k, v in packs.items()}
        Vx = co.wino_input(xr, C, C, b * t, D, H, W)                         # [16][b t R1][C]: all views of all scenes
        MXg, MXc = newV(b * t * R1, 2 * C), newV(b * t * R1, C)
        hx = co.wino_half_applies(R, C, C)                                   # the form (8 / 16 planes) of the per-step launches that consume MXg / MXc as second addends
        co.wino_gemm(Vx, C, None, 0, U["gx"], MXg, b * t, D, Ht, Wt, 2 * C, half=hx)
        co.wino_gemm(Vx, C, None, 0, U["ox"], MXc, b * t, D, Ht, Wt, C, half=hx)
        Mm = newV(R, 2 * C)
        Mc = Mm.view(-1)[:16 * R * C].view(16, R, C)                         # the C-column problems reuse the front of the buffer
        bnargs = lambda m: bn_module_args(m)
        outs, saved_groups = [], []
        for grp in groups:
            grp = list(grp)
            run = grp == list(range(grp[0], grp[0] + len(grp)))
            Vm = newV(R, C)
            if run:                                                          # the view mean is taken inside the input transform
                co.wino_input(xr[:, grp[0]:], C, C, b, D, H, W, bs=t * vol, out=Vm, nsum=len(grp), sum_stride=vol)
            else:
                co.wino_input(torch.stack([xr[:, ti] for ti in grp], dim=1).mean(dim=1).reshape(M, C), C, C, b, D, H, W, out=Vm)
            a0 = new()
            co.wino_gemm(Vm, C, None, 0, U["f0"], Mc, b, D, Ht, Wt, C)
            co.wino_output(Mc, b0, None, None, 1.0, None, None, None, a0, None, None, *geo, C, C, co.EPI_BIAS)
            rm, rv, mom, eps, nbt, group = bnargs(fc[1])
            t0, sv1 = bn_rows_fwd(a0, g1, be1, rm, rv, mom, eps, 0.01, None, nbt, group)
            Vt0 = co.wino_input(t0, C, C, b, D, H, W)
            a1 = new()
            co.wino_gemm(Vt0, C, None, 0, U["f3"], Mc, b, D, Ht, Wt, C)
            co.wino_output(Mc, b3, None, None, 1.0, None, None, None, a1, None, None, *geo, C, C, co.EPI_BIAS)
            rm, rv, mom, eps, nbt, group = bnargs(fc[4])
            h, sv4 = bn_rows_fwd(a1, g4, be4, rm, rv, mom, eps, 0.01, None, nbt, group)
            steps = []
            for ti in grp:
                Vh = co.wino_input(h, C, C, b, D, H, W)
                co.wino_gemm(Vh, C, None, 0, U["gh"], Mm, b, D, Ht, Wt, 2 * C)
                z, hr, r = new(), new(), new()
                co.wino_output(Mm, bg, None, None, 1.0, None, h, None, z, hr, r, *geo, 2 * C, C, co.EPI_GRU_GATES, Mm2=MXg, view=ti, views=t)
                Vhr = co.wino_input(hr, C, C, b, D, H, W)
                co.wino_gemm(Vhr, C, None, 0, U["oh"], Mc, b, D, Ht, Wt, C)
                hn, cand = new(), new()
                co.wino_output(Mc, bo, None, None, 1.0, None, h, z, hn, None, cand, *geo, C, C, co.EPI_GRU_OUT, Mm2=MXc, view=ti, views=t)
                steps.append((ti, h, z, r, cand, Vh, Vhr))
                h = hn
            rm, rv, mom, eps, nbt, group = bnargs(norm)
            out, svn = bn_rows_fwd(h, gn, bn_, rm, rv, mom, eps, 1.0, None, nbt, group)
            outs.append(out.reshape(b, D, H, W, C).permute(0, 4, 1, 2, 3))
            saved_groups.append((grp, Vm, sv1, Vt0, sv4, steps, svn))
        tensors = []
        ctx.spec = _flatten_saved((packs, Vx, saved_groups), tensors)       # incl. the BatchNorm parameters inside the bn_rows_fwd states
        ctx.save_for_backward(*tensors)
        ctx.shape = (b, t, C, D, H, W)
        ctx.has_bias = tuple(v is not None for v in (bg, bo, b0, b3))
        return tuple(outs)

    @staticmethod
    @_lib.on_tensor_device
    def backward(ctx, *douts):
        b, t, C, D, H, W = ctx.shape
        packs, Vx, saved_groups = _unflatten_saved(ctx.spec, ctx.saved_tensors)
        dev = Vx.device
        M, vol, Ht, Wt = b * D * H * W, D * H * W, H // 2, W // 2
        R1 = D * Ht * Wt
        R = b * R1
        geo = (b, D, H, W)
        L, p, st = _lib.lib(), _lib.ptr, _lib.current_stream
        new = lambda c=C: torch.empty(M, c, dtype=torch.float32, device=dev)
        newV = lambda rows, c: torch.empty(16, rows, c, dtype=torch.float32, device=dev)
        UT = {k: co.wino_pack_packed(v, transpose=True) for k, v in packs.items()}     # Winograd-domain data-gradient weights [16][3][Cin][Cout]
        dU = {k: co.grad_zeros((16, 3, v.shape[1], v.shape[2]), dev, scratch=True) for k, v in packs.items()}
        Mm = newV(R, 2 * C)
        Mc = Mm.view(-1)[:16 * R * C].view(16, R, C)

        def both(dy, Cdy, Vin, wkey, dkey, dst, residual=None):
            """One pass over dy [M, Cdy] for both of its Winograd forms (forge_wino_input_dy), then
               dU[wkey] += (A dy A^T)^T (x) Vin                                  (weight gradient)
               dst [M, C] = conv^T(dy, weights `dkey`) (+ residual)              (data gradient: point GEMMs on B^T dy B, inverse transform)"""
            V, dM = co.wino_input_dy(dy, Cdy, b, D, H, W)
            _lib.check(L.forge_wino_wgrad(p(dM), p(Vin), C, 0, 0, None, 0, 0, 0, p(dU[wkey]), b, D, Ht, Wt, Cdy, 3, st()), "forge_wino_wgrad")
            del dM
            co.wino_gemm(V, Cdy, None, 0, UT[dkey], Mc, b, D, Ht, Wt, C)
            co.wino_output(Mc, None, None, None, 1.0, residual, None, None, dst, None, None, *geo, C, C, co.EPI_BIAS)
            return dst

        dg_acc = torch.empty(b, t, D, H, W, 2 * C, dtype=torch.float32, device=dev)     # d (gate pre-activations) summed per view over the groups
        dc_acc = torch.empty(b, t, D, H, W, C, dtype=torch.float32, device=dev)
        seen = [False] * t
        grads_bn = {k: None for k in ("g1", "be1", "g4", "be4", "gn", "bn")}
        add = lambda old, g: g if old is None else old + g
        db0 = db3 = None
        dmeans = []
        for gi in reversed(range(len(saved_groups))):
            grp, Vm, sv1, Vt0, sv4, steps, svn = saved_groups[gi]
            dout = douts[gi]
            if dout is None:
                dout = torch.zeros(b, C, D, H, W, dtype=torch.float32, device=dev)
            dr = dout.permute(0, 2, 3, 4, 1)
            dr = (dr if dr.is_contiguous() else dr.contiguous()).reshape(M, C)
            dhn, dgn_, dbn_, _ = bn_rows_bwd(svn, dr)
            grads_bn["gn"], grads_bn["bn"] = add(grads_bn["gn"], dgn_), add(grads_bn["bn"], dbn_)
            for ti, h, z, r, cand, Vh, Vhr in reversed(steps):
                mode = 2 if seen[ti] else 1
                seen[ti] = True
                dh, dz, dc = new(), new(), new()
                _lib.check(L.forge_gru_state_bwd(p(dhn), C, p(h), p(z), p(cand), p(dh), p(dz), p(dc), M, C, p(dc_acc[:, ti]), t * vol, vol, mode, st()),
                           "forge_gru_state_bwd")
                dhr = both(dc, C, Vhr, "oh", "oh", new())
                dg = new(2 * C)
                _lib.check(L.forge_gru_gates_bwd(p(dz), p(dhr), C, p(h), p(z), p(r), p(dg), p(dh), None, 0, M, C, p(dg_acc[:, ti]), t * vol, vol, mode, st()),
                           "forge_gru_gates_bwd")
                dhn = both(dg, 2 * C, Vh, "gh", "gh", new(), residual=dh)  # dh (state + reset paths) + conv^T(dg, Wg_h)
            # h0 = lrelu(bn4(conv(lrelu(bn1(conv(mean))))))
            da1, dg4, dbe4, _ = bn_rows_bwd(sv4, dhn)
            grads_bn["g4"], grads_bn["be4"] = add(grads_bn["g4"], dg4), add(grads_bn["be4"], dbe4)
            dt0 = both(da1, C, Vt0, "f3", "f3", new())
            if ctx.has_bias[3]:
                db3 = add(db3, co.colsum(da1))
            da0, dg1, dbe1, _ = bn_rows_bwd(sv1, dt0)
            grads_bn["g1"], grads_bn["be1"] = add(grads_bn["g1"], dg1), add(grads_bn["be1"], dbe1)
            dmeans.append((grp, both(da0, C, Vm, "f0", "f0", new())))
            if ctx.has_bias[2]:
                db0 = add(db0, co.colsum(da0))
        for ti in range(t):
            if not seen[ti]:                                                 # a view no group uses: no gradient through the GRU input halves
                dg_acc[:, ti].zero_()
                dc_acc[:, ti].zero_()
        # the shared input halves, once for all views: weight gradients against V_x, data gradient = conv^T(dg, Wg_x) + conv^T(dc, Wo_x)
        nbt_, Rall = b * t, b * t * R1
        dx = torch.empty(b, t, D, H, W, C, dtype=torch.float32, device=dev)
        MA, MB = newV(Rall, C), newV(Rall, C)
        for acc, Cacc, key, Mout in ((dg_acc, 2 * C, "gx", MA), (dc_acc, C, "ox", MB)):
            Va, dMa = co.wino_input_dy(acc.reshape(-1, Cacc), Cacc, nbt_, D, H, W)
            _lib.check(L.forge_wino_wgrad(p(dMa), p(Vx), C, 0, 0, None, 0, 0, 0, p(dU[key]), nbt_, D, Ht, Wt, Cacc, 3, st()), "forge_wino_wgrad")
            del dMa
            co.wino_gemm(Va, Cacc, None, 0, UT[key], Mout, nbt_, D, Ht, Wt, C)
            del Va
        co.wino_output(MA, None, None, None, 1.0, None, None, None, dx, None, None, nbt_, D, H, W, C, C, co.EPI_BIAS, Mm2=MB, view=0, views=1)
        for grp, dmean in dmeans:                                            # d mean / d x_ti = 1 / |group|
            dm = dmean.reshape(b, 1, D, H, W, C)
            if grp == list(range(grp[0], grp[0] + len(grp))):
                dx[:, grp[0]:grp[0] + len(grp)].add_(dm, alpha=1.0 / len(grp))
            else:
                for ti in grp:
                    dx[:, ti].add_(dm[:, 0], alpha=1.0 / len(grp))
        dbg = co.colsum(dg_acc.reshape(-1, 2 * C)) if ctx.has_bias[0] else None
        dbo = co.colsum(dc_acc.reshape(-1, C)) if ctx.has_bias[1] else None

        def unpack(*keys):
            """G^T dU G of the listed halves, concatenated along Cin, in the nn.Conv3d layout [Cout, Cin, 3, 3, 3]"""
            parts = []
            for k in keys:
                dwp = co.grad_zeros(packs[k].shape, dev)
                _lib.check(L.forge_wino_dw(p(dU[k]), p(dwp), dwp.shape[1], dwp.shape[2], 3, st()), "forge_wino_dw")
                parts.append(dwp)
            dwp = parts[0] if len(parts) == 1 else torch.cat(parts, dim=2)
            return dwp.permute(1, 2, 0).reshape(dwp.shape[1], dwp.shape[2], 3, 3, 3)
        need = ctx.needs_input_grad
        return (dx.permute(0, 1, 5, 2, 3, 4) if need[0] else None, unpack("gx", "gh") if need[1] else None, dbg, unpack("ox", "oh") if need[3] else None, dbo,
                unpack("f0") if need[5] else None, db0, grads_bn["g1"], grads_bn["be1"], unpack("f3") if need[9] else None, db3,
                grads_bn["g4"], grads_bn["be4"], grads_bn["gn"], grads_bn["bn"], None, None)


def gru_cell_rows(x, h, gate_weight, gate_bias, out_weight, out_bias):
    """ConvGRUCell_3D.forward on rows [b,D,H,W,C] with autograd; weights are the module's Conv3d parameters."""
    return _GRUCellRows.apply(x, h, co._pack3d(gate_weight), gate_bias, co._pack3d(out_weight), out_bias)


class ConvGRUCell_3D(nn.Module):
    """models/fusion.py:7-35. Gate split order is (update, reset) (:30)."""

    def __init__(self, config, input_size, hidden_size):
        super().__init__()
        self.input_size = input_size
        self.hidden_size = hidden_size
        self.conv_gate = nn.Conv3d(input_size + hidden_size, hidden_size * 2, 3, padding=1)
        self.out_gate = nn.Conv3d(input_size + hidden_size, hidden_size, 3, padding=1)

    def forward(self, x, prev_state=None):
        """x [b,C,d,h,w], prev_state [b,Ch,d,h,w] or None -> new state (models/fusion.py:21-35), on the HIP kernels (with autograd)."""
        require_hip_input("ConvGRUCell_3D", x, self.input_size)
        require_hip_input("ConvGRUCell_3D", x, self.hidden_size)
        b, c, d, h, w = x.shape
        if prev_state is None:
            prev_state = torch.zeros([b, self.hidden_size, d, h, w], dtype=x.dtype, device=x.device)
        rows = lambda v: v.permute(0, 2, 3, 4, 1).contiguous()
        hn = gru_cell_rows(rows(x), rows(prev_state), self.conv_gate.weight, self.conv_gate.bias, self.out_gate.weight, self.out_gate.bias)
        return hn.permute(0, 4, 1, 2, 3)


class ConvGRU_3D(co.PackedModule):
    """models/fusion.py:39-95."""

    def __init__(self, config, n_layers=1, input_size=16, hidden_size=16):
        super().__init__()
        self._pack_cache = co.PackCache()
        self.input_size = input_size
        self.hidden_size = hidden_size
        self.n_layers = n_layers
        self.cells = nn.ModuleList([
            ConvGRUCell_3D(config, input_size if i == 0 else hidden_size, hidden_size) for i in range(n_layers)])
        self.fusion_norm = nn.BatchNorm3d(hidden_size)
        self.fusion_conv = nn.Sequential(
            nn.Conv3d(input_size, input_size, 3, padding=1),
            nn.BatchNorm3d(input_size),
            nn.LeakyReLU(inplace=True),
            nn.Conv3d(input_size, input_size, 3, padding=1),
            nn.BatchNorm3d(input_size),
            nn.LeakyReLU(inplace=True),
        )

    # ---------------------------------------------------------------- fused HIP inference path
    def _packed(self):
        cell = self.cells[0]
        fc = self.fusion_conv
        src = [cell.conv_gate.weight, cell.conv_gate.bias, cell.out_gate.weight, cell.out_gate.bias,
               fc[0].weight, fc[0].bias, fc[3].weight, fc[3].bias] + \
              [t for bn in (fc[1], fc[4], self.fusion_norm) for t in (bn.weight, bn.bias, bn.running_mean, bn.running_var)]

        def build():
            return {
                "gate_w": co.pack_conv3d_weight(cell.conv_gate.weight), "gate_b": cell.conv_gate.bias.detach().contiguous(),
                "out_w": co.pack_conv3d_weight(cell.out_gate.weight), "out_b": cell.out_gate.bias.detach().contiguous(),
                "fc0_w": co.pack_conv3d_weight(fc[0].weight), "fc0_b": fc[0].bias.detach().contiguous(),
                "fc3_w": co.pack_conv3d_weight(fc[3].weight), "fc3_b": fc[3].bias.detach().contiguous(),
                "bn1": co.bn_affine(fc[1]), "bn4": co.bn_affine(fc[4]), "norm": co.bn_affine(self.fusion_norm),
            }
        return self._pack_cache.get(src, build)

    def _packed_T(self):
        """Transposed packed weights ([tap][Cin][Cout]) and BatchNorm scales for the hand-written data-gradient path (_FuseFrozen)."""
        p = self._packed()
        if "gate_wT" not in p:
            tr = lambda w: w.transpose(1, 2).contiguous()
            p.update({"gate_wT": tr(p["gate_w"]), "out_wT": tr(p["out_w"]), "fc0_wT": tr(p["fc0_w"]), "fc3_wT": tr(p["fc3_w"]),
                      "norm_scale": p["norm"][0], "bn1_scale": p["bn1"][0], "bn4_scale": p["bn4"][0]})
            p.update({k + "_UT": co.wino_pack_packed(p[k + "_w"], transpose=True) for k in ("gate", "out", "fc0", "fc3")})   # Winograd-domain data-gradient weights
            C = self.hidden_size
            for k in ("gate", "out"):                                   # the hidden-state halves alone (data gradient w.r.t. h only: _FuseFrozen skip_dx0)
                p[k + "_h_w"] = p[k + "_w"][:, :, C:].contiguous()
                p[k + "_h_wT"] = p[k + "_h_w"].transpose(1, 2).contiguous()
                p[k + "_h_UT"] = p[k + "_UT"][:, :, C:, :].contiguous()
        return p

    def fuse_frozen_hip(self, x, skip_dx0=False, const0=None):
        """Encoder3D.fuse with frozen weights under autograd (pose refinement): fused forward, hand-written data-gradient backward.
        skip_dx0: the caller needs no gradient for view 0 of the sequence (the un-warped reference view of a refinement problem): its slice of
        the returned gradient holds only the fusion_conv(mean) share and the last backward step produces the hidden-state halves of its two data gradients only.
        const0 (with skip_dx0; a dict the caller owns, initially empty): the caller PROMISES that view 0 holds the same values in every call made
        with this dict and that the weights do not change meanwhile - the products of view 0's input halves are then computed by the first call
        only (kept in the dict) and step 0 of later calls contracts the hidden-state half alone."""
        assert self.n_layers == 1 and self.input_size == self.hidden_size
        require_hip_input("ConvGRU_3D.fuse_frozen_hip", x, x.shape[2])
        return _FuseFrozen.apply(x, self, bool(skip_dx0), const0)

    def fuse_hip(self, x, h0=None):
        """Encoder3D.fuse on the MI355X: h0 = fusion_conv(mean_t x) as two fused conv+BN+LeakyReLU GEMMs (or the caller's h0
        [b,C,D,H,W]), then per view two implicit-GEMM launches (gates: cat/conv/sigmoid/h*r fused; state: cat/conv/tanh/lerp fused,
        the final fusion_norm folded into the last one). x [b,t,C,D,H,W] -> [b,C,D,H,W] (channels-last memory)."""
        assert self.n_layers == 1 and self.input_size == self.hidden_size
        require_hip_input("ConvGRU_3D.fuse_hip", x, x.shape[2])
        b, t, C, D, H, W = x.shape
        xr = x.permute(0, 1, 3, 4, 5, 2)                          # [b,t,D,H,W,C] rows view
        if not xr.is_contiguous():
            xr = xr.contiguous()
        p = self._packed()
        dev, M, vol = x.device, b * D * H * W, D * H * W
        nb = co.wino_scene_chunk(b, D, H, W, C, views=t) if co.wino_enabled() else 0
        if nb:                                                        # scenes are independent: batches beyond the buffer range run in scene chunks
            if nb >= b:
                return self._fuse_wino(xr, h0)
            return torch.cat([self._fuse_wino(xr[i:i + nb], None if h0 is None else h0[i:i + nb]) for i in range(0, b, nb)], dim=0)
        grid, ig = (b, D, H, W), (D, H, W)
        new = lambda: torch.empty(M, C, dtype=torch.float32, device=dev)
        taps = co.TAPS_3x3x3
        t0, h = new(), new()
        if h0 is None:
            mean = xr.mean(dim=1).reshape(M, C)
            co.conv_igemm(mean, C, C, None, 0, 0, p["fc0_w"], p["fc0_b"], p["bn1"][0], p["bn1"][1], 0.01, None, None, None,
                          t0, None, grid, ig, C, C, taps, epilogue=co.EPI_AFFINE_ACT)
            co.conv_igemm(t0, C, C, None, 0, 0, p["fc3_w"], p["fc3_b"], p["bn4"][0], p["bn4"][1], 0.01, None, None, None,
                          h, None, grid, ig, C, C, taps, epilogue=co.EPI_AFFINE_ACT)
        else:
            h.copy_(h0.permute(0, 2, 3, 4, 1).reshape(M, C))
        z, hr, h2, out = new(), new(), t0, new()
        for ti in range(t):
            xt = xr[:, ti]                                        # base pointer of view ti; batch stride t*vol rows
            co.conv_igemm(xt, C, C, h, C, C, p["gate_w"], p["gate_b"], None, None, 1.0, None, h, None, z, hr,
                          grid, ig, 2 * C, C, taps, epilogue=co.EPI_GRU_GATES, bs1=t * vol)
            last = ti == t - 1
            co.conv_igemm(xt, C, C, hr, C, C, p["out_w"], p["out_b"], p["norm"][0], p["norm"][1], 1.0, None, h, z,
                          h2, out if last else None, grid, ig, C, C, taps, epilogue=co.EPI_GRU_OUT, bs1=t * vol)
            h, h2 = h2, h
        return out.reshape(b, D, H, W, C).permute(0, 4, 1, 2, 3)

    def _packed_wino(self):
        """Winograd-domain weights U = G w G^T [16][3][Cout][Cin] of the four 3x3x3 convolutions (convops.wino_pack_weight)."""
        p = self._packed()
        if "gate_U" not in p:
            cell, fc = self.cells[0], self.fusion_conv
            p.update({"gate_U": co.wino_pack_weight(cell.conv_gate.weight), "out_U": co.wino_pack_weight(cell.out_gate.weight),
                      "fc0_U": co.wino_pack_weight(fc[0].weight), "fc3_U": co.wino_pack_weight(fc[3].weight)})
        return p

    def _packed_wino_halves(self):
        """The input (x) and hidden-state (h) halves of the two GRU convolutions as separate Winograd-domain weights [16][3][Cout][C]
        (the frozen-weight refinement forward with a hoisted reference view, _FuseFrozen)."""
        p = self._packed_wino()
        if "gate_Ux" not in p:
            C = self.hidden_size
            cell = self.cells[0]
            for k, w in (("gate", cell.conv_gate.weight), ("out", cell.out_gate.weight)):
                wp = co.pack_conv3d_weight(w)                                # [27][Cout][2 C]: (x | h) input channels
                p[k + "_Ux"] = co.wino_pack_packed(wp[:, :, :C].contiguous())
                p[k + "_Uh"] = co.wino_pack_packed(wp[:, :, C:].contiguous())
        return p

    @staticmethod
    def _wino_h0(p, src, geo, Vh, Mc, t0, h, nsum=1, sum_stride=0, bs=0):
        """h = fusion_conv(mean of the nsum view tensors starting at `src`, sum_stride rows apart): two Winograd convolutions with the folded
        BatchNorm + LeakyReLU tail (scratch Vh / Mc / t0); the view mean of models/encoder.py:62 is taken inside the first input transform."""
        b, D, H, W = geo
        C = h.shape[-1]
        co.wino_input(src, C, C, b, D, H, W, bs=bs, out=Vh, nsum=nsum, sum_stride=sum_stride)
        co.wino_gemm(Vh, C, None, 0, p["fc0_U"], Mc, b, D, H // 2, W // 2, C)
        co.wino_output(Mc, p["fc0_b"], p["bn1"][0], p["bn1"][1], 0.01, None, None, None, t0, None, None, b, D, H, W, C, C, co.EPI_AFFINE_ACT)
        co.wino_input(t0, C, C, b, D, H, W, out=Vh)
        co.wino_gemm(Vh, C, None, 0, p["fc3_U"], Mc, b, D, H // 2, W // 2, C)
        co.wino_output(Mc, p["fc3_b"], p["bn4"][0], p["bn4"][1], 0.01, None, None, None, h, None, None, b, D, H, W, C, C, co.EPI_AFFINE_ACT)

    def _fuse_wino(self, xr, h0=None):
        """fuse_hip with every 3x3x3 convolution as Winograd F(2x2, 3x3) x 3 depth taps (csrc/winograd.hip): 2.25x fewer MFMA FLOPs.
        The views are transformed once, by one launch; the view mean is taken inside the input transform that feeds fusion_conv; per GRU step:
        transform h, 16 point GEMMs over [V_x | V_h] (K = 3 x 256), inverse transform fused with the gate epilogue; transform h*r, point GEMMs,
        inverse transform fused with the state update. (Fusing each inverse transform with the next input transform through LDS was built and
        measured slower - tools/experiments/wino_output_input_kernel.hip.)"""
        b, t, D, H, W, C = xr.shape
        p = self._packed_wino()
        dev, M, Ht, Wt = xr.device, b * D * H * W, H // 2, W // 2
        R = b * D * Ht * Wt
        new = lambda c=C: torch.empty(M, c, dtype=torch.float32, device=dev)
        geo = (b, D, H, W)
        Vx = co.wino_input(xr, C, C, b * t, D, H, W)                       # [16][b t D Ht Wt][C]: all views of all scenes
        Vh = torch.empty(16, R, C, dtype=torch.float32, device=dev)
        Mm = torch.empty(16, R, 2 * C, dtype=torch.float32, device=dev)
        Mc = Mm.view(-1)[:16 * R * C].view(16, R, C)                        # the C-column problems reuse the front of the buffer
        t0, h = new(), new()
        vol = D * H * W
        if h0 is None:
            self._wino_h0(p, xr, geo, Vh, Mc, t0, h, nsum=t, sum_stride=vol, bs=t * vol)
        else:
            h.copy_(h0.permute(0, 2, 3, 4, 1).reshape(M, C))
        z, hr, h2, out = new(), new(), t0, new()
        for ti in range(t):
            co.wino_input(h, C, C, b, D, H, W, out=Vh)
            co.wino_gemm(Vx, C, Vh, C, p["gate_U"], Mm, b, D, Ht, Wt, 2 * C, view=ti, views=t)
            co.wino_output(Mm, p["gate_b"], None, None, 1.0, None, h, None, z, hr, None, *geo, 2 * C, C, co.EPI_GRU_GATES)
            co.wino_input(hr, C, C, b, D, H, W, out=Vh)
            co.wino_gemm(Vx, C, Vh, C, p["out_U"], Mc, b, D, Ht, Wt, C, view=ti, views=t)
            last = ti == t - 1
            co.wino_output(Mc, p["out_b"], p["norm"][0], p["norm"][1], 1.0, None, h, z, h2, out if last else None, None, *geo, C, C, co.EPI_GRU_OUT)
            h, h2 = h2, h
        return out.reshape(b, D, H, W, C).permute(0, 4, 1, 2, 3)

    def _fuse_groups_wino(self, xr, groups):
        """fuse_groups_hip in the Winograd domain: the views are transformed once and the point products of the INPUT halves of both GRU
        convolutions are computed once, for all views in one launch each (Mm_x = V_x (x) U_x); every (group, view) step then runs the
        hidden-state halves only (K = 3 x 128) and the inverse transform adds the view's Mm_x before the fused GRU tail."""
        b, t, D, H, W, C = xr.shape
        p = self._packed_wino()
        if "gate_Ux" not in p:                                         # U = (U_x | U_h) along Cin
            p.update({"gate_Ux": p["gate_U"][..., :C].contiguous(), "gate_Uh": p["gate_U"][..., C:].contiguous(),
                      "out_Ux": p["out_U"][..., :C].contiguous(), "out_Uh": p["out_U"][..., C:].contiguous()})
        dev, M, Ht, Wt = xr.device, b * D * H * W, H // 2, W // 2
        R1 = D * Ht * Wt
        R = b * R1
        new = lambda c=C: torch.empty(M, c, dtype=torch.float32, device=dev)
        geo = (b, D, H, W)
        Vx = co.wino_input(xr, C, C, b * t, D, H, W)
        # 8-plane form (row stage of the inverse transform in the GEMM epilogue, convops.wino_half_applies) for the launches on the 64 x 128 tile: the
        # shared input-half products and the per-step hidden-half products are then row-combined separately and added in the column-stage kernel
        hx = co.wino_half_applies(R, C, C)                            # decided by the per-step launches (R rows); the all-view launches follow it
        P = 8 if hx else 16
        MXg = torch.empty(P, b * t * R1, 2 * C, dtype=torch.float32, device=dev)
        MXc = torch.empty(P, b * t * R1, C, dtype=torch.float32, device=dev)
        co.wino_gemm(Vx, C, None, 0, p["gate_Ux"], MXg, b * t, D, Ht, Wt, 2 * C, half=hx)
        co.wino_gemm(Vx, C, None, 0, p["out_Ux"], MXc, b * t, D, Ht, Wt, C, half=hx)
        Vh = torch.empty(16, R, C, dtype=torch.float32, device=dev)
        Mm = torch.empty(P, R, 2 * C, dtype=torch.float32, device=dev)
        Mc = Mm.view(-1)[:P * R * C].view(P, R, C)
        outs = []
        for grp in groups:
            grp = list(grp)
            run = grp == list(range(grp[0], grp[0] + len(grp)))       # a run of views: the mean is taken inside the input transform
            mean = None if run else torch.stack([xr[:, ti] for ti in grp], dim=1).mean(dim=1).reshape(M, C)
            t0, h = new(), new()
            if run:
                self._wino_h0(p, xr[:, grp[0]:], geo, Vh, Mc, t0, h, nsum=len(grp), sum_stride=D * H * W, bs=t * D * H * W)
            else:
                self._wino_h0(p, mean, geo, Vh, Mc, t0, h)
            z, hr, h2, out = new(), new(), t0, new()
            for k, ti in enumerate(grp):
                co.wino_input(h, C, C, b, D, H, W, out=Vh)
                co.wino_gemm(Vh, C, None, 0, p["gate_Uh"], Mm, b, D, Ht, Wt, 2 * C)
                co.wino_output(Mm, p["gate_b"], None, None, 1.0, None, h, None, z, hr, None, *geo, 2 * C, C, co.EPI_GRU_GATES, Mm2=MXg, view=ti, views=t)
                co.wino_input(hr, C, C, b, D, H, W, out=Vh)
                co.wino_gemm(Vh, C, None, 0, p["out_Uh"], Mc, b, D, Ht, Wt, C)
                last = k == len(grp) - 1
                co.wino_output(Mc, p["out_b"], p["norm"][0], p["norm"][1], 1.0, None, h, z, h2, out if last else None, None, *geo, C, C,
                               co.EPI_GRU_OUT, Mm2=MXc, view=ti, views=t)
                h, h2 = h2, h
            outs.append(out.reshape(b, D, H, W, C).permute(0, 4, 1, 2, 3))
        return outs

    def fuse_groups_hip(self, x, groups):
        """Several fusions over subsets of the SAME views in inference (FORGE_poseEstimator3D fuses views (0,1,2), (3,4) and (0..4) of
        the same rotated features, models/model_single_pose_estimator.py:108-120): conv([x, h], W) = conv(x, W_x) + conv(h, W_h), so the
        input halves of both GRU convolutions are computed ONCE per view (two launches per view) and every (group, view) step then runs
        the hidden-state halves only, the input half entering the fused GRU epilogues as a residual - 25 % fewer GRU FLOPs for the
        three fusions. Same arithmetic as fuse_hip up to the order of the fp32 additions. Returns one fused volume per group."""
        assert self.n_layers == 1 and self.input_size == self.hidden_size
        require_hip_input("ConvGRU_3D.fuse_groups_hip", x, x.shape[2])
        b, t, C, D, H, W = x.shape
        xr = x.permute(0, 1, 3, 4, 5, 2)
        xr = xr if xr.is_contiguous() else xr.contiguous()
        nb = co.wino_scene_chunk(b, D, H, W, C, views=t) if co.wino_enabled() else 0
        if nb:
            if nb >= b:
                return self._fuse_groups_wino(xr, groups)
            parts = [self._fuse_groups_wino(xr[i:i + nb], groups) for i in range(0, b, nb)]
            return [torch.cat([p_[g] for p_ in parts], dim=0) for g in range(len(groups))]
        p = self._packed()
        if "gate_wx" not in p:                                         # W = (W_x | W_h) along Cin
            p.update({"gate_wx": p["gate_w"][:, :, :C].contiguous(), "gate_wh": p["gate_w"][:, :, C:].contiguous(),
                      "out_wx": p["out_w"][:, :, :C].contiguous(), "out_wh": p["out_w"][:, :, C:].contiguous()})
        dev, M, vol = x.device, b * D * H * W, D * H * W
        grid, ig, taps = (b, D, H, W), (D, H, W), co.TAPS_3x3x3
        new = lambda c=C: torch.empty(M, c, dtype=torch.float32, device=dev)
        used = sorted({ti for g in groups for ti in g})
        gx, cx = {}, {}
        for ti in used:                                                # input halves, once per view (no bias: added with the hidden half)
            gx[ti], cx[ti] = new(2 * C), new()
            co.conv_igemm(xr[:, ti], C, C, None, 0, 0, p["gate_wx"], None, None, None, 1.0, None, None, None, gx[ti], None, grid, ig, 2 * C, 2 * C, taps,
                          epilogue=co.EPI_BIAS, bs1=t * vol)
            co.conv_igemm(xr[:, ti], C, C, None, 0, 0, p["out_wx"], None, None, None, 1.0, None, None, None, cx[ti], None, grid, ig, C, C, taps,
                          epilogue=co.EPI_BIAS, bs1=t * vol)
        outs = []
        for grp in groups:
            grp = list(grp)
            if grp == list(range(grp[0], grp[0] + len(grp))):          # a run of views: a slice (no index tensor: capturable into a hipGraph)
                mean = xr[:, grp[0]:grp[0] + len(grp)].mean(dim=1).reshape(M, C)
            else:
                mean = torch.stack([xr[:, ti] for ti in grp], dim=1).mean(dim=1).reshape(M, C)
            t0, h = new(), new()
            co.conv_igemm(mean, C, C, None, 0, 0, p["fc0_w"], p["fc0_b"], p["bn1"][0], p["bn1"][1], 0.01, None, None, None, t0, None, grid, ig, C, C, taps,
                          epilogue=co.EPI_AFFINE_ACT)
            co.conv_igemm(t0, C, C, None, 0, 0, p["fc3_w"], p["fc3_b"], p["bn4"][0], p["bn4"][1], 0.01, None, None, None, h, None, grid, ig, C, C, taps,
                          epilogue=co.EPI_AFFINE_ACT)
            z, hr, h2, out = new(), new(), t0, new()
            for k, ti in enumerate(grp):
                co.conv_igemm(h, C, C, None, 0, 0, p["gate_wh"], p["gate_b"], None, None, 1.0, gx[ti], h, None, z, hr, grid, ig, 2 * C, C, taps,
                              epilogue=co.EPI_GRU_GATES)
                last = k == len(grp) - 1
                co.conv_igemm(hr, C, C, None, 0, 0, p["out_wh"], p["out_b"], p["norm"][0], p["norm"][1], 1.0, cx[ti], h, z, h2, out if last else None,
                              grid, ig, C, C, taps, epilogue=co.EPI_GRU_OUT)
                h, h2 = h2, h
            outs.append(out.reshape(b, D, H, W, C).permute(0, 4, 1, 2, 3))
        return outs

    # ---------------------------------------------------------------- HIP training / autograd path
    @staticmethod
    def _bn_rows(bn, rows, act=None):
        """nn.BatchNorm3d / SyncBatchNorm module applied to channels-last rows [b,D,H,W,C] (batch statistics in train mode)."""
        return bn_act_rows(bn, rows, 1.0 if act is None else act)

    def fuse_autograd_hip(self, x):
        """Encoder3D.fuse with an autograd graph (model.train(), or eval-mode pose refinement): the six convolutions per GRU step
        and fusion_conv run on the MFMA implicit-GEMM kernel (forward and data gradient) and the wgrad kernel, the cell's sigmoid /
        tanh / lerp halves on the element-wise kernels of csrc/gru.hip (_GRUCellRows); BatchNorm (batch statistics / SyncBN) stays
        a torch module so that SyncBatchNorm conversion keeps working."""
        assert self.n_layers == 1
        require_hip_input("ConvGRU_3D.fuse_autograd_hip", x, x.shape[2])
        b, t, C, D, H, W = x.shape
        xr = x.permute(0, 1, 3, 4, 5, 2)
        xr = xr if xr.is_contiguous() else xr.contiguous()
        cell, fc = self.cells[0], self.fusion_conv
        lrelu = 0.01
        h = self._bn_rows(fc[1], co.conv3x3x3_rows(xr.mean(dim=1), None, fc[0].weight, fc[0].bias), lrelu)
        h = self._bn_rows(fc[4], co.conv3x3x3_rows(h, None, fc[3].weight, fc[3].bias), lrelu)
        for xv in xr.unbind(1):                       # unbind: ONE stack in backward instead of a zero-filled [b,t,...] scatter per view
            h = gru_cell_rows(xv, h, cell.conv_gate.weight, cell.conv_gate.bias, cell.out_gate.weight, cell.out_gate.bias)
        return bn_act_rows(self.fusion_norm, h).permute(0, 4, 1, 2, 3)

    def fuse_groups_autograd_hip(self, x, groups):
        """Several fusions over subsets of the SAME views (groups = lists of view indices into x [b,t,C,D,H,W]) with an autograd graph:
        the input halves of both GRU convolutions are computed once per view for all groups (_GRUCellPreRows), each group then
        runs h0 = fusion_conv(mean of its views) and its own recurrence on the hidden-state halves. Returns one fused volume per group."""
        assert self.n_layers == 1 and self.input_size == self.hidden_size
        require_hip_input("ConvGRU_3D.fuse_groups_autograd_hip", x, x.shape[2])
        b, t, C, D, H, W = x.shape
        cell, fc = self.cells[0], self.fusion_conv
        bns = (fc[1], fc[4], self.fusion_norm)
        rows_probe = x.permute(0, 1, 3, 4, 5, 2)
        if (co.wino_enabled() and co.wino_fits(b, D, H, W, 2 * C, views=t) and co.wino_wgrad_applies(b, D, H, W, C, 0, C)
                and all(bn_hip_train(m, rows_probe) and m.weight is not None for m in bns)
                and all(isinstance(a, nn.LeakyReLU) and a.negative_slope == 0.01 for a in (fc[2], fc[5]))):       # the node's kernels are launched with slope 0.01 for BOTH activations
            # the training step proper (every BatchNorm on batch statistics): one hand-scheduled autograd node for all groups
            outs = _FuseGroupsTrain.apply(x, cell.conv_gate.weight, cell.conv_gate.bias, cell.out_gate.weight, cell.out_gate.bias,
                                          fc[0].weight, fc[0].bias, fc[1].weight, fc[1].bias, fc[3].weight, fc[3].bias, fc[4].weight, fc[4].bias,
                                          self.fusion_norm.weight, self.fusion_norm.bias, self, tuple(tuple(g) for g in groups))
            return list(outs)
        xt = x.permute(1, 0, 3, 4, 5, 2).contiguous()                                   # [t,b,D,H,W,C]: a view's rows are dense
        Wg, Wo = cell.conv_gate.weight, cell.out_gate.weight
        flat = xt.reshape(t * b, D, H, W, C)
        # per-view tensors via unbind: their gradients (one per group that uses the view) are summed view-sized and stacked once
        gx_all = co.conv3x3x3_rows(flat, None, Wg[:, :C], None).reshape(t, b, D, H, W, 2 * C).unbind(0)
        cx_all = co.conv3x3x3_rows(flat, None, Wo[:, :C], None).reshape(t, b, D, H, W, C).unbind(0)
        xviews = xt.unbind(0)
        wgh, woh = co._pack3d(Wg[:, C:]).contiguous(), co._pack3d(Wo[:, C:]).contiguous()      # dense once: every GRU step of every group reads them
        lrelu = 0.01
        outs = []
        for grp in groups:
            grp = list(grp)
            xm = sum(xviews[ti] for ti in grp) / float(len(grp))
            h = self._bn_rows(fc[1], co.conv3x3x3_rows(xm, None, fc[0].weight, fc[0].bias), lrelu)
            h = self._bn_rows(fc[4], co.conv3x3x3_rows(h, None, fc[3].weight, fc[3].bias), lrelu)
            for ti in grp:
                h = _GRUCellPreRows.apply(gx_all[ti], cx_all[ti], h, wgh, cell.conv_gate.bias, woh, cell.out_gate.bias)
            outs.append(bn_act_rows(self.fusion_norm, h).permute(0, 4, 1, 2, 3))
        return outs

    def forward(self, x, hidden=None):
        """x [b,t,c,d,h,w], hidden = [h0 per layer] (models/fusion.py:71-95; Encoder3D.fuse passes [fusion_conv(mean_t x)]) ->
        fusion_norm(h_T) [b,c',d,h,w]. One implementation: the HIP kernels (fused epilogues in inference, the autograd cell otherwise)."""
        require_hip_input("ConvGRU_3D", x, x.shape[2])
        if not hidden:
            hidden = [None] * self.n_layers
        if self.n_layers == 1 and hip_inference(self, x) and self.input_size == self.hidden_size:
            h0 = hidden[0]
            if h0 is None:
                h0 = torch.zeros(x.shape[0], self.hidden_size, *x.shape[3:], dtype=x.dtype, device=x.device)
            return self.fuse_hip(x, h0=h0)
        cur = x
        h = None
        for layer_idx in range(self.n_layers):
            h = hidden[layer_idx]
            outs = []
            for t in range(x.shape[1]):
                h = self.cells[layer_idx](cur[:, t], h)
                if layer_idx + 1 < self.n_layers:
                    outs.append(h)
            if layer_idx + 1 < self.n_layers:
                cur = torch.stack(outs, dim=1)
        return self.fusion_norm(h)
