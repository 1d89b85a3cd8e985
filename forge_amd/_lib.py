"""ctypes binding of libforge_hip.so (include/forge_hip.h). No fallbacks: if the library is
missing or a call fails, a RuntimeError is raised — the product never routes through a CPU path."""
import ctypes
import os

# torch FIRST: PyTorch-ROCm ships its own libamdhip64; libforge_hip.so must bind to the HIP runtime that
# owns torch's streams and allocations. Loading our library before torch would pull in /opt/rocm's copy
# and every launch on a torch stream would fail with "no ROCm-capable device is detected".
import torch  # noqa: F401

_HERE = os.path.dirname(os.path.abspath(__file__))
LIB_PATH = os.environ.get("FORGE_AMD_LIB") or os.path.join(_HERE, "libforge_hip.so")     # override: instrumented debug builds (tools/debug)
_lib = None

_P = ctypes.c_void_p
_I = ctypes.c_int
_F = ctypes.c_float
_LL = ctypes.c_longlong

# name -> argtypes; must list every symbol declared in include/forge_hip.h (tests check this)
SIGNATURES = {
    "forge_version": [],
    "forge_last_error": [],
    "forge_rotate_fwd": [_P, _P, _P, _P, _I, _I, _I, _I, _I, _P],
    "forge_rotate_fwd_slots": [_P, _P, _P, _P, _P, _I, _I, _I, _I, _I, _P],
    "forge_rotate_xf_from_poses": [_P, _P, _P, _P, _P, _I, _I, _F, _P],
    "forge_rotate_bwd": [_P, _P, _P, _P, _P, _P, _I, _I, _I, _I, _I, _P],
    "forge_pose_chain_fwd": [_P, _P, _P, _P, _P, _F, _I, _I, _P, _P, _P, _P, _P, _P, _P, _P],
    "forge_colsum": [_P, _I, _P, _P, _LL, _I, _P],
    "forge_adam_small": [_P, _P, _P, _P, _P, _I, _F, _F, _F, _F, _P],
    "forge_rotate_bwd_slots": [_P, _P, _P, _P, _P, _P, _P, _I, _I, _I, _I, _I, _P],
    "forge_pose_chain_bwd": [_P, _P, _P, _P, _P, _I, _I, _P],
    "forge_pack_cameras": [_P, _LL, _LL, _LL, _P, _LL, _LL, _P, _LL, _LL, _LL, _P, _P, _I, _P],
    "forge_render_fwd": [_P, _P, _P, _P, _P, _P, _P] + [_I] * 9 + [_F] * 5 + [_P],
    "forge_render_bwd_ws_bytes": [_I] * 6,
    "forge_render_bwd": [_P, _P, _P, _P, _P, _P, _P, _P, _P, _P] + [_I] * 9 + [_F] * 5 + [_P, _LL, _P],
    "forge_resize_bilinear_fwd": [_P, _P, _I, _I, _I, _I, _I, _P],
    "forge_resize_bilinear_bwd": [_P, _P, _I, _I, _I, _I, _I, _P],
    "forge_conv_igemm": [_P, _I, _I, _LL, _P, _I, _I, _LL, _P, _P, _P, _P, _F, _P, _P, _P, _P, _P, _P] + [_I] * 10 + [_P] + [_I] * 12 + [_P, _LL, _P, _P],
    "forge_wino_weights": [_P, _P, _I, _I, _I, _I, _P],
    "forge_wino_dy": [_P, _I, _P, _I, _I, _I, _I, _I, _P],
    "forge_wino_wgrad": [_P, _P, _I, _LL, _LL, _P, _I, _LL, _LL, _P, _I, _I, _I, _I, _I, _I, _P],
    "forge_wino_dw": [_P, _P, _I, _I, _I, _P],
    "forge_wino_input": [_P, _I, _LL, _P, _I, _LL, _I, _I, _I, _I, _I, _I, _LL, _P],
    "forge_wino_input_dy": [_P, _I, _P, _P, _I, _I, _I, _I, _I, _P],
    "forge_wino_gemm": [_P, _I, _I, _LL, _LL, _P, _I, _I, _LL, _LL, _P, _P, _I, _I, _I, _I, _I, _I, _I, _P],
    "forge_wino_gemm_half": [_P, _I, _I, _LL, _LL, _P, _I, _I, _LL, _LL, _P, _P, _I, _I, _I, _I, _I, _I, _P],
    "forge_wino_output_half": [_P, _P, _LL, _LL, _P, _P, _P, _F, _P, _P, _P, _P, _P, _P, _I, _I, _I, _I, _I, _I, _I, _P],
    "forge_wino_gemm_tile": [_LL, _I, _I],
    "forge_wino_output": [_P, _P, _LL, _LL, _P, _P, _P, _F, _P, _P, _P, _P, _P, _P, _I, _I, _I, _I, _I, _I, _I, _P],
    "forge_conv_igemm_plan": [_LL, _I, _I, _I, _I, _I, _I, _LL, _P, _P],
    "forge_conv_wgrad": [_P, _I, _P, _I, _I, _LL, _P, _I, _I, _LL, _P] + [_I] * 9 + [_P, _I, _P],
    "forge_conv_direct_fwd": [_P, _I, _P, _P, _F, _P, _I] + [_I] * 6 + [_P, _I, _P],
    "forge_conv_direct_dgrad": [_P, _I, _P, _P, _I] + [_I] * 6 + [_P, _I, _P],
    "forge_conv_direct_wgrad": [_P, _I, _P, _I, _P] + [_I] * 6 + [_P, _I, _P],
    "forge_gru_gates_fwd": [_P, _P, _P, _P, _P, _LL, _I, _P],
    "forge_gru_state_fwd": [_P, _P, _P, _P, _LL, _I, _P],
    "forge_gru_state_bwd": [_P, _I, _P, _P, _P, _P, _P, _P, _LL, _I, _P, _LL, _LL, _I, _P],
    "forge_gru_gates_bwd": [_P, _P, _I, _P, _P, _P, _P, _P, _P, _I, _LL, _I, _P, _LL, _LL, _I, _P],
    "forge_bn_ws_doubles": [_I],
    "forge_bn_train_fwd": [_P, _I, _P, _P, _F, _F, _P, _I, _P, _P, _P, _P, _F, _P, _LL, _I, _P, _I, _P, _I, _P],
    "forge_bn_train_bwd": [_P, _I, _P, _I, _P, _P, _P, _P, _F, _P, _I, _P, _P, _P, _LL, _I, _P, _I, _P, _I, _P],
    "forge_bn_sync_stats": [_P, _I, _P, _LL, _I, _I, _P],
    "forge_bn_eval_fwd": [_P, _I, _P, _P, _P, _P, _F, _F, _P, _I, _P, _P, _LL, _I, _P, _I, _P],
    "forge_bn_sync_fwd_apply": [_P, _I, _P, _P, _F, _F, _P, _I, _P, _P, _P, _P, _F, _P, _LL, _LL, _I, _P, _I, _P, _P],
    "forge_bn_sync_bwd_reduce": [_P, _I, _P, _I, _P, _P, _P, _P, _F, _P, _P, _P, _LL, _I, _P, _I, _P],
    "forge_bn_sync_bwd_apply": [_P, _I, _P, _I, _P, _P, _P, _P, _F, _P, _I, _P, _LL, _LL, _I, _P, _I, _P, _I, _P],
    "forge_affine_act_bwd": [_P, _I, _P, _I, _P, _F, _P, _I, _LL, _I, _P],
    "forge_sse_groups_blocks": [],
    "forge_sse_groups_fwd": [_P, _LL, _LL, _LL, _LL, _P, _P] + [_I] * 7 + [_P],
    "forge_sse_groups_bwd": [_P, _LL, _LL, _LL, _LL, _P, _P, _P] + [_I] * 7 + [_P],
    "forge_attention_fwd": [_P, _P, _P, _LL, _P, _I, _I, _I, _I, _P],
    "forge_im2col_nchw": [_P, _P] + [_I] * 9 + [_P],
    "forge_maxpool2d_nhwc": [_P, _P] + [_I] * 7 + [_P],
    "forge_ncdhw_to_ndhwc": [_P, _P, _I, _I, _LL, _P],
    "forge_ndhwc_to_ncdhw": [_P, _P, _I, _I, _LL, _P],
}


_LL_RESULTS = ("forge_render_bwd_ws_bytes",)       # byte counts: long long results


def lib():
    """Load (once) and return the ctypes handle. Raises if the library has not been built."""
    global _lib
    if _lib is None:
        if not os.path.exists(LIB_PATH):
            raise RuntimeError(
                "forge_amd: %s not found. Build it with `python -m forge_amd.build` (needs hipcc, "
                "targets gfx950). There is no CPU/PyTorch fallback for the HIP kernels." % LIB_PATH)
        h = ctypes.CDLL(LIB_PATH)
        for name, args in SIGNATURES.items():
            fn = getattr(h, name)
            fn.argtypes = args
            fn.restype = ctypes.c_char_p if name == "forge_last_error" else (_LL if name in _LL_RESULTS else _I)
        _lib = h
    return _lib


def check(rc, what):
    if rc != 0:
        msg = lib().forge_last_error().decode("utf-8", "replace")
        raise RuntimeError("forge_amd: %s failed (code %d): %s" % (what, rc, msg))


def ptr(t):
    """Device pointer of a torch tensor (None -> NULL)."""
    return None if t is None else ctypes.c_void_p(t.data_ptr())


def current_stream():
    """torch's current stream of the CURRENT device (launchers run under on_tensor_device, which makes the operands' device current)."""
    return ctypes.c_void_p(torch.cuda.current_stream().cuda_stream)


def on_tensor_device(fn):
    """Decorator for launch wrappers: run `fn` with the device of its first cuda tensor argument (positional or keyword) current, so that
    `current_stream()` is that device's stream and per-device kernel attributes are applied to it (a process may hold models on
    several GPUs while torch's current device is another one)."""
    import functools

    @functools.wraps(fn)
    def wrapped(*args, **kw):
        for a in list(args) + list(kw.values()):       # keyword call sites too (the models call rotate / render with keywords)
            if torch.is_tensor(a) and a.is_cuda:
                if a.device.index != torch.cuda.current_device():
                    with torch.cuda.device(a.device):
                        return fn(*args, **kw)
                break
        return fn(*args, **kw)
    return wrapped
