"""CPU: the C-ABI library loads and exports every symbol include/forge_hip.h declares; argument
validation returns the documented codes without touching a GPU; the Python surface keeps the
reference's state_dict keys."""
import os
import re

import numpy as np
import pytest
import torch

from forge_amd import _lib, synthetic as syn

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def declared_symbols():
    src = open(os.path.join(ROOT, "include", "forge_hip.h")).read()
    src = re.sub(r"/\*.*?\*/", "", src, flags=re.S)
    return sorted(set(re.findall(r"\b(forge_[a-z0-9_]+)\s*\(", src)))


def test_header_symbols_exported(built_lib):
    syms = declared_symbols()
    assert "forge_render_fwd" in syms and "forge_rotate_fwd" in syms and len(syms) >= 8
    import ctypes
    h = ctypes.CDLL(built_lib)
    for s in syms:
        assert hasattr(h, s), "libforge_hip.so does not export %s" % s
    # and the ctypes binding table covers exactly the header
    assert sorted(_lib.SIGNATURES) == syms


def test_c_host_program_compiles_against_the_header():
    """tests/c_host/c_abi_smoke.c (the plain-C driver of the boundary, run on the GPU by tests/test_gpu_c_host.py) must compile against the
    CURRENT include/forge_hip.h: a changed entry-point signature is caught here, on the CPU, not at round end."""
    import shutil
    import subprocess
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    rocm = os.environ.get("ROCM_PATH", "/opt/rocm")
    gcc = shutil.which("gcc")
    if gcc is None or not os.path.exists(os.path.join(rocm, "include", "hip", "hip_runtime_api.h")):
        pytest.skip("needs gcc and the HIP runtime headers")
    r = subprocess.run([gcc, "-std=c11", "-fsyntax-only", "-Werror=implicit-function-declaration", "-I", os.path.join(rocm, "include"),
                        "-I", os.path.join(root, "include"), "-D__HIP_PLATFORM_AMD__", os.path.join(root, "tests", "c_host", "c_abi_smoke.c")],
                       capture_output=True, text=True, timeout=120)
    assert r.returncode == 0, r.stderr


def test_version_and_error_codes(built_lib):
    l = _lib.lib()
    assert l.forge_version() >= 100
    assert l.forge_rotate_fwd(None, None, None, None, 1, 4, 8, 8, 8, None) == -1          # FORGE_EINVAL
    assert b"null pointer" in l.forge_last_error()
    fake = 0x1000   # never dereferenced: argument checks run before any launch
    assert l.forge_rotate_fwd(fake, fake, fake, fake, 1, 6, 8, 8, 8, None) == -2           # FORGE_ESHAPE (C % 4)
    assert l.forge_render_fwd(fake, fake, fake, fake, fake, fake, None, 1, 1, 12, 8, 8, 8, 4, 4, 8,
                              0.5, 2.0, 0.5, 0.5, 0.5, None) == -2                          # C unsupported
    assert l.forge_render_fwd(fake, fake, fake, fake, fake, fake, None, 1, 1, 16, 8, 8, 8, 4, 4, 1,
                              0.5, 2.0, 0.5, 0.5, 0.5, None) == -1                          # S < 2
    with pytest.raises(RuntimeError, match="forge_rotate"):
        _lib.check(l.forge_rotate_fwd(None, None, None, None, 1, 4, 8, 8, 8, None), "forge_rotate_fwd")


def test_ops_refuse_cpu_tensors():
    from forge_amd import ops
    with pytest.raises(RuntimeError, match="no CPU fallback"):
        ops.rotate_warp(torch.zeros(1, 4, 4, 4, 4), torch.zeros(1, 12), torch.zeros(1, dtype=torch.int32))
    q = torch.zeros(1, 64, 64)
    with torch.no_grad():
        assert not ops.attention_applies(q, q, q)                           # a CPU tensor: the caller keeps torch's own ops
        with pytest.raises(RuntimeError, match="on the MI355X"):
            ops.attention(q, q, q)


@pytest.mark.parametrize("which", ["pose3d", "joint"])
def test_state_dict_keys_match_reference(golden, which):
    g = golden("state_dict_keys_" + which)
    ref = dict(zip(g["keys"].tolist(), g["shapes"].tolist()))
    if which == "pose3d":
        from forge_amd.model_single_pose_estimator import FORGE_poseEstimator3D as M
        m = M(syn.kubric_config())
    else:
        from forge_amd.model import FORGE as M
        m = M(syn.kubric_config(use_gt_pose=False, parameter="joint"))
    mine = {k: str(tuple(v.shape)) for k, v in m.state_dict().items()}
    assert set(mine) == set(ref)
    assert all(mine[k] == ref[k] for k in ref)
    # spot-check the hot-path keys SURVEY.md Appendix B names
    assert ref["encoder_3d.fusion_feature.cells.0.conv_gate.weight"] == "(256, 256, 3, 3, 3)"
    assert ref["render.conv_rgb.0.weight"] == "(16, 16, 6, 6)"
    assert "rotate.conv3d_4.weight" in ref


def test_module_surface_names():
    from forge_amd.model import FORGE, chose_selected, sequence_from_distance  # noqa: F401  (demo.py:21)
    from forge_amd.model_single_pose_estimator import FORGE_poseEstimator3D
    m = FORGE_poseEstimator3D(syn.kubric_config())
    for attr in ("get_feat3D", "fuse", "get_density3D", "get_render_features", "fusion_feature", "density_head"):
        assert hasattr(m.encoder_3d, attr)
    assert hasattr(m.encoder_traj, "toSE3") and m.encoder_traj.pose_dim == 7
    assert m.rotate.grid_coord_max == pytest.approx(0.484375)
    import inspect
    assert list(inspect.signature(m.render.forward).parameters)[:5] == [
        "camera_params", "feature_3d", "density_3d", "render_depth", "return_origin_proj"]
    assert list(inspect.signature(m.rotate.forward).parameters)[:3] == ["voxels", "camPoses_cv2", "grid_size"]     # + optional `order` extension
    # the reference's three arguments first; `features_recon` is this package's optional extension (128^3-voxel scenes: feature volumes the encoder cannot produce)
    assert list(inspect.signature(m.forward).parameters) == ["sample", "dataset", "device", "features_recon"]
    assert inspect.signature(m.forward).parameters["features_recon"].default is None


def test_reference_import_aliases():
    import sys
    import forge_amd
    saved = {k: v for k, v in sys.modules.items() if k == "models" or k.startswith("models.")}
    try:
        forge_amd.install_reference_aliases()
        from models.model import FORGE, chose_selected  # noqa: F401
        from models.model_single_pose_estimator import FORGE_poseEstimator3D  # noqa: F401
        assert FORGE.__module__ == "forge_amd.model"
    finally:
        for k in [k for k in sys.modules if k == "models" or k.startswith("models.")]:
            del sys.modules[k]
        sys.modules.update(saved)


def test_view_ordering_helpers_match_oracle():
    import forge_oracle as fo
    from forge_amd.model import chose_selected, sequence_from_distance
    g = torch.Generator().manual_seed(0)
    trans = torch.randn(3, 5, 3, generator=g)
    idx = sequence_from_distance(trans)
    assert torch.equal(idx, fo.sequence_from_distance(trans))
    x = torch.randn(3, 5, 2, 4, generator=g)
    assert torch.equal(chose_selected(x, idx), fo.chose_selected(x, idx))


def test_synthetic_sample_schema():
    s = syn.make_sample(2, 10, 64, 1.5, seed=1)
    assert s["images"].shape == (2, 10, 3, 64, 64) and s["fg_probabilities"].shape == (2, 10, 1, 64, 64)
    assert s["K_cv2"].shape == (2, 10, 3, 3) and s["cam_poses_cv2_canonicalized"].shape == (2, 10, 4, 4)
    E, P = s["cam_extrinsics_cv2_canonicalized"], s["cam_poses_cv2_canonicalized"]
    assert torch.allclose(E @ P, torch.eye(4).expand(2, 10, 4, 4), atol=1e-5)
    can = syn.SyntheticDataset(1.5).get_canonical_extrinsics_cv2()
    assert torch.allclose(E[:, 0], can.expand(2, 4, 4), atol=1e-6)      # dataset/kubric.py:100-104
    # every camera looks at the object centre from distance camera_z
    assert torch.allclose(P[..., :3, 3].norm(dim=-1), torch.full((2, 10), 1.5), atol=1e-4)
    sd = syn.seeded_state_dict({"a.weight": (4, 3, 3, 3), "a.bias": (4,)}, 0)
    sd2 = syn.seeded_state_dict({"a.bias": (4,), "a.weight": (4, 3, 3, 3)}, 0)
    assert torch.equal(sd["a.weight"], sd2["a.weight"])                 # per-key streams: order independent


def test_look_at_view_transform_kat_and_shim():
    """row f3: at azim=180 the PyTorch3D look-at camera IS the canonical OpenCV camera (R = I, T = (0,0,dist)) — the reason
    demo.py:85-87 can feed PyTorch3D (R,T) to render(); and the restatement agrees with the oracle-side shim."""
    import sys
    from forge_amd import nvs
    R, T = nvs.look_at_view_transform(dist=1.5, elev=0.0, azim=180.0)
    assert torch.allclose(R[0], torch.eye(3), atol=1e-6) and torch.allclose(T[0], torch.tensor([0.0, 0.0, 1.5]), atol=1e-6)
    sys.path.insert(0, os.path.join(ROOT, "oracle", "shims"))
    try:
        from pytorch3d.renderer import look_at_view_transform as shim
    finally:
        sys.path.remove(os.path.join(ROOT, "oracle", "shims"))
    elev = torch.tensor([0.0, 10.0, -25.0, 40.0])
    azim = torch.tensor([180.0, 13.0, 250.0, 359.0])
    R1, T1 = nvs.look_at_view_transform(dist=1.5, elev=elev, azim=azim)
    R2, T2 = shim(dist=1.5, elev=elev, azim=azim)
    assert torch.allclose(R1, R2, atol=1e-6) and torch.allclose(T1, T2, atol=1e-6)
    R, T = nvs.nvs_cameras(1.5)
    assert R.shape == (28, 3, 3) and torch.allclose(T.norm(dim=1), torch.full((28,), 1.5), atol=1e-5)
    assert torch.allclose(R @ R.transpose(1, 2), torch.eye(3).expand(28, 3, 3), atol=1e-5)


def test_adjust_lr_schedule():
    """utils/train_utils.py:149-164 + config/kubric/gt_pose.yaml:41,54"""
    from forge_amd import train
    cfg = syn.kubric_config()
    opt = torch.optim.Adam([torch.nn.Parameter(torch.zeros(1))], lr=cfg.train.lr)
    for k, it in enumerate(cfg.train.adjust_iter_num):
        assert train.adjust_lr(cfg, opt, it, cfg.train.adjust_iter_num) == pytest.approx(cfg.train.lr * 0.5 ** (k + 1))
        assert opt.param_groups[0]["lr"] == pytest.approx(cfg.train.lr * 0.5 ** (k + 1))
    assert train.adjust_lr(cfg, opt, 123, cfg.train.adjust_iter_num) is None


def test_conv_plan_is_host_arithmetic_and_sane():
    """forge_conv_igemm_plan runs without a GPU: plans for the step's shapes (csrc/conv_igemm.hip: plan_conv)."""
    from forge_amd import convops as co
    assert co.conv_plan(32768, 256, 256, 27, co.EPI_GRU_GATES, 128) in (("A", 1), ("B", 1))   # ConvGRU gates: a large tile, never split (853 / 854 us measured)
    assert co.conv_plan(262144, 16, 32, 27, co.EPI_AFFINE_ACT, 16)[0] == "N"             # Cout <= 16 kernel
    assert co.conv_plan(786432, 32, 32, 27, co.EPI_BIAS, 32) == ("E", 1)                 # 32-channel tile
    t, k = co.conv_plan(5120, 512, 512, 9, co.EPI_AFFINE_ACT, 512)                       # ResNet layer4 3x3 at one scene: split-K
    assert k > 1 and t in "ABCD"
    assert co.conv_plan(5120, 512, 512, 9, co.EPI_GRU_OUT, 512)[1] == 1                  # GRU epilogues cannot be split
    assert co.conv_plan(32768, 64, 128, 64, co.EPI_AFFINE_ACT, 64, nphase=8)[1] == 1     # merged transposed-conv phases: no split-K
    for M in (1, 63, 5120, 20480, 131072):
        for N in (17, 32, 64, 96, 2048):
            t, k = co.conv_plan(M, N, 64, 1, co.EPI_BIAS, N)
            assert t in "ABCDE" and k == 1                                               # 2 K-steps: never split
    # forge_wino_gemm_tile (host arithmetic too): the 16-problem Winograd launches take the 64x128 tile, the 2-D trunk's tiny ones 64x64
    assert co.wino_gemm_tile(8192, 256, 256) == "B" and co.wino_gemm_tile(40960, 128, 64) == "B" and co.wino_gemm_tile(320, 256, 256) == "D"


def test_plan_model_replica_of_the_fitting_tool_matches_the_library():
    """tools/fit_plan_model.py re-fits plan_conv's constants on sweep data with a Python replica of the model; the replica (and the constants it
    starts from) must stay the library's: same (tile, split-K) on a grid of 405 shapes."""
    from forge_amd import convops as co
    ns = {"__name__": "replica"}
    try:
        exec(open(os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "tools", "fit_plan_model.py")).read(), ns)
    except SystemExit:
        pass
    avail = {"%s%d" % (t, k) for t in "ABCDE" for k in ns["SPLITS"]}
    n = 0
    for M in (1280, 5120, 20480, 32768, 131072, 786432):
        for N in (32, 64, 128, 256, 512, 2048):
            for C in (32, 64, 256, 1024):
                for T in (1, 9, 27):
                    if M * max(N, C) * 4 >= 1 << 31:
                        continue
                    n += 1
                    assert ns["choose"](ns["TILES"], (M, N, C, T), avail) == "%s%d" % co.conv_plan(M, N, C, T, co.EPI_AFFINE_ACT, N), (M, N, C, T)
    assert n == 405


def test_loss_functions_refuse_host_tensors():
    """f1 / north_star "no dual code paths": the loss functions have ONE implementation (csrc/loss.hip through train.grouped_mse); rendered maps on
    the host raise instead of taking a stock F.mse_loss detour. Their values are pinned against the reference's own numbers on the GPU
    (tests/test_gpu_parity.py::test_loss_functions_match_reference_golden)."""
    import types
    from forge_amd import train as tr
    cfg = types.SimpleNamespace(loss=types.SimpleNamespace(recon_rgb=5.0, recon_mask=1.0, perceptual_img=0.0, regu_origin_proj=0.0))
    smp = {"images": torch.zeros(1, 10, 3, 8, 8), "fg_probabilities": torch.zeros(1, 10, 1, 8, 8)}
    pose = {"pred": torch.zeros(4, 7), "gt": torch.zeros(4, 7)}
    s5 = {k: v[:, :5] for k, v in smp.items()}
    for fn, sample, model in ((tr.compute_reconstruction_loss, s5, lambda s, d, dev: (torch.zeros(10, 3, 8, 8), torch.zeros(10, 1, 8, 8))),
                              (tr.compute_all_loss, s5, lambda s, d, dev: (torch.zeros(10, 3, 8, 8), torch.zeros(10, 1, 8, 8), torch.zeros(5, 2), pose)),
                              (tr.compute_all_loss_nvs, smp, lambda s, d, dev: (torch.zeros(10, 3, 8, 8), torch.zeros(10, 1, 8, 8), torch.zeros(5, 2), pose))):
        with pytest.raises(TypeError, match="on the MI355X"):
            fn(cfg, 0, sample, None, model, {}, "cpu", None)


def test_packed_caches_are_dropped_on_mode_load_and_apply():
    """packed-weight caches must not survive train()/eval(), load_state_dict or _apply (.to/.float) — `.data` edits do not
    bump the version counters the cache keys on — and forge_amd.invalidate_packed() is the explicit call."""
    import forge_amd
    from forge_amd import convops as co
    from forge_amd.fusion import ConvGRU_3D
    gru = ConvGRU_3D(syn.kubric_config(), n_layers=1, input_size=32, hidden_size=32)
    w = gru.cells[0].conv_gate.weight

    def arm():
        gru._pack_cache._key, gru._pack_cache.val = gru._pack_cache.key_of([w]), "packed"
    arm()
    assert gru._pack_cache.get([w], lambda: "rebuilt") == "packed"
    w.data.mul_(2.0)                                                   # invisible to the key ...
    assert gru._pack_cache.get([w], lambda: "rebuilt") == "packed"
    forge_amd.invalidate_packed(gru)                                   # ... so the explicit call exists
    assert gru._pack_cache.get([w], lambda: "rebuilt") == "rebuilt"
    for action in (lambda: gru.eval(), lambda: gru.train(), lambda: gru.load_state_dict(gru.state_dict()), lambda: gru.float()):
        arm()
        action()
        assert gru._pack_cache._key is None and gru._pack_cache.val is None
    with torch.no_grad():
        arm()
        w.mul_(0.5)                                                    # a versioned in-place op IS seen by the key
    assert gru._pack_cache.get([w], lambda: "rebuilt") == "rebuilt"
    assert isinstance(gru, co.PackedModule)


def test_inference_schedule_specs_and_pose_estimator_caches():
    """forge_amd/frozen.py host logic: conv / BatchNorm / activation grouping of the pose estimators' blocks (models/pose_estimator_3d.py:24-60,
    models/pose_estimator_2d.py:36-48), the eligibility test (eval BatchNorm, no autograd graph, fp32 on the GPU), and the launch-argument caches of
    the pose estimators dropped with the module's mode like every other packed cache."""
    from forge_amd import convops as co, frozen as fz
    from forge_amd.pose_estimator_2d import FPN, PoseEstimator2D
    from forge_amd.pose_estimator_3d import PoseEstimator3D
    p3 = PoseEstimator3D(syn.kubric_config())
    s1 = fz.chain_specs(p3.conv3d_1)
    assert [(c.in_channels, c.out_channels, c.stride[0], bn is not None, sl) for c, bn, sl in s1] == [(128, 64, 2, True, 0.01), (64, 64, 1, False, 1.0)]
    sh = fz.chain_specs(p3.pose_head_1)
    assert [(c.out_channels, bn is not None, sl) for c, bn, sl in sh] == [(512, True, 0.01), (1024, False, 1.0)]
    p2 = PoseEstimator2D()
    assert [(c.out_channels, c.stride[0], sl) for c, bn, sl in fz.chain_specs(p2.conv)] == [(256, 2, 0.01), (512, 2, 0.01), (512, 2, 0.01), (1024, 2, 0.01)]
    with pytest.raises(TypeError):
        fz.chain_specs(torch.nn.Sequential(torch.nn.BatchNorm3d(8)))
    n_src = len(fz._sources(s1))
    assert n_src == 2 + 4 + 2                                                      # conv w, b + BatchNorm w, b, mean, var + conv w, b
    x = torch.zeros(1, 8)
    with torch.no_grad():
        assert not fz.frozen_ok(x, p3)                                             # a CPU tensor never takes the HIP schedule
    for m in (p3, p2, p2.backbone):
        assert isinstance(m, co.PackedModule)
    caches = [p3._frozen_cache, p2._conv_cache, p2.backbone._res_cache, p2.backbone._head_cache]
    for c in caches:
        c._key, c.val = ("armed",), "packed"
    p3.eval(), p2.eval()
    assert all(c._key is None and c.val is None for c in caches)
    assert "_frozen_cache" not in p3.state_dict() and not any("cache" in k for k in p2.state_dict())


def test_load_imagenet_trunk_key_mapping():
    """torchvision resnet50 keys -> the nn.Sequential trunk (conv1 -> 0, bn1 -> 1, layerN -> N+3), fc dropped, shapes validated."""
    from forge_amd.encoder import get_resnet50, load_imagenet_trunk
    src, dst = get_resnet50(), get_resnet50()
    names = {"0": "conv1", "1": "bn1", "4": "layer1", "5": "layer2", "6": "layer3", "7": "layer4"}
    tv = {}
    for k, v in src.state_dict().items():
        head, rest = k.split(".", 1)
        tv[names[head] + "." + rest] = v.clone() + (0.25 if v.is_floating_point() else 0)
    tv["fc.weight"], tv["fc.bias"] = torch.zeros(1000, 2048), torch.zeros(1000)
    res = load_imagenet_trunk(dst, tv)
    assert not res.missing_keys and not res.unexpected_keys
    for (k, a), (_, b) in zip(src.state_dict().items(), dst.state_dict().items()):
        assert torch.equal(b, a + (0.25 if a.is_floating_point() else 0)), k
    bad = dict(tv)
    bad["layer1.0.conv1.weight"] = torch.zeros(3, 3)
    with pytest.raises(ValueError):
        load_imagenet_trunk(dst, bad)
    with pytest.raises(KeyError):
        load_imagenet_trunk(dst, {"avgpool.weight": torch.zeros(1)})


def test_stage_sample_passthrough_and_cpu():
    """f4 staging helper: tensors already on the target device pass through untouched (same objects); non-tensor entries are kept."""
    from forge_amd.staging import stage_sample
    s = syn.make_sample(1, 5, 32, 1.5, seed=0)
    s["seq_name"] = ["scene0"]
    out = stage_sample(s, "cpu")
    assert out is s
    s64 = dict(s, K_cv2=s["K_cv2"].double())
    out = stage_sample(s64, "cpu")
    assert out["K_cv2"].dtype == torch.float32 and out["images"] is s["images"] and out["seq_name"] == ["scene0"]


def test_modules_refuse_cpu_tensors_instead_of_falling_back():
    """north_star "no dual code paths": the module entry points raise on inputs the HIP kernels cannot take."""
    from forge_amd.encoder import Encoder3D
    enc = Encoder3D(syn.kubric_config()).eval()
    with torch.no_grad():
        for call in (lambda: enc.get_feat3D(torch.zeros(1, 3, 64, 64)), lambda: enc.fuse(torch.zeros(1, 2, 128, 4, 4, 4)),
                     lambda: enc.get_density3D(torch.zeros(1, 128, 4, 4, 4)), lambda: enc.get_render_features(torch.zeros(1, 128, 4, 4, 4)),
                     lambda: enc.heads(torch.zeros(1, 128, 4, 4, 4)),
                     lambda: enc.fusion_feature(torch.zeros(1, 2, 128, 4, 4, 4), None)):
            with pytest.raises(RuntimeError, match="no CPU or stock-PyTorch path"):
                call()


def test_grad_zero_arena_never_hands_out_memory_twice():
    """convops.grad_zeros (one zero-filled arena per backward pass for the atomically accumulated weight gradients): plain torch.zeros outside
    a backward pass; inside one, slices of ONE buffer sized by the previous pass; a pass that dies with an exception, or a nested backward,
    never makes a later request alias memory handed out before (the old arena is dropped, not rewound)."""
    from forge_amd import convops as co
    arena = co._ZeroArena()
    dev = torch.device("cpu")
    assert arena.zeros((3, 4), dev).abs().sum() == 0 and arena.task == -1          # outside backward: no arena

    got = []

    class Node(torch.autograd.Function):
        @staticmethod
        def forward(ctx, x, fail):
            ctx.fail = fail
            return x * 2

        @staticmethod
        def backward(ctx, g):
            a, b = arena.zeros((5, 7), dev), arena.zeros((64,), dev)
            a.add_(1.0)
            b.add_(2.0)                                                         # "gradients" accumulated into the slices
            got.append((a, b))
            if ctx.fail:
                raise RuntimeError("boom")
            return g * 2, None

    x = torch.ones(2, requires_grad=True)
    Node.apply(x, False).sum().backward()                                        # pass 1: nothing known yet -> own allocations
    assert arena.task == -1 and arena.want[0] >= (5 * 7 + 64) * 4               # keyed by stream handle (0 off the GPU)
    Node.apply(x, False).sum().backward()                                        # pass 2: both carved from one arena
    a, b = got[-1]
    assert a.untyped_storage().data_ptr() == b.untyped_storage().data_ptr() and a.data_ptr() != b.data_ptr()
    with pytest.raises(RuntimeError):
        Node.apply(x, True).sum().backward()                                     # pass 3 dies: its callback never runs
    fa, fb = got[-1]
    Node.apply(x, False).sum().backward()                                        # pass 4 must not rewind pass 3's arena
    a4, b4 = got[-1]
    assert a4.untyped_storage().data_ptr() != fa.untyped_storage().data_ptr()
    assert float(fa.sum()) == 35.0 and float(fb.sum()) == 128.0                  # pass 3's slices untouched by pass 4's zero-fill / adds
    assert float(a4.sum()) == 35.0 and float(b4.sum()) == 128.0 and arena.task == -1


def test_rotate_grid_coordinate_formula_is_torch_linspace():
    """csrc/rotate.hip::linspace_pm1 (the normalised voxel-centre coordinate of the warp) restated on the host: step = fp32(2 / (D - 1)), the lower half
    fma(step, i, -1), the upper half fma(-step, D - 1 - i, 1) - bit for bit torch.linspace(-1, 1, D), the grid models/rotate.py:50-51 gets from PyTorch3D."""
    for D in (7, 16, 32, 33, 48, 64, 128):
        step = np.float32(2.0) / np.float32(D - 1)
        fma = lambda a, b, c: np.float32(np.float64(a) * np.float64(b) + np.float64(c))          # exact product and sum in double, ONE rounding: fmaf
        got = np.array([fma(step, np.float32(i), np.float32(-1.0)) if i < D // 2 else fma(-step, np.float32(D - 1 - i), np.float32(1.0)) for i in range(D)], dtype=np.float32)
        assert np.array_equal(got, torch.linspace(-1.0, 1.0, D).numpy()), D
