"""`extra_configs`: the other BASELINE configurations on one GPU, bounded, measured after the headline region (N = 1)."""
import json
import os

import torch

from forge_amd import synthetic as syn
from benchkit.common import FP32_MFMA_PEAK_TF, ROOT, T_IN, TRAIN_CAVEATS, V_OUT, _timed, floor_of


def extra_configs(dev, steps=5):
    """The other BASELINE configurations on this GPU, bounded (<= `steps` timed steps each), AFTER the headline timed region (N = 1):
    configs[2] (8 scenes), the 128^3-voxel grid (n1 / configs[3]-[4] grid), FORGE_poseEstimator3D inference, one GT-pose training step
    (configs[3] per-GPU step at the reference-native grid) and one pose-refinement iteration (row f2). Each entry: workload, ms_per_step,
    views_per_s and `roofline` = the FLOPs the step's matrix-core launches execute (FlopMeter around one eager pass) as a time floor on the
    157.3 TF fp32 MFMA pipe. A configuration that fails reports its error and the others still run."""
    from forge_amd import geo_utils, refine
    from forge_amd.flopmeter import FlopMeter
    from forge_amd.graph import GraphedCall, GraphedForward, PipelinedForward
    from forge_amd.model import FORGE
    from forge_amd.model_single_pose_estimator import FORGE_poseEstimator3D
    from forge_amd import train
    from forge_amd.train import grouped_mse
    out = []
    holder = {}
    ds = syn.SyntheticDataset(1.5)
    cfg = syn.kubric_config()

    def build(cls, train=False):
        m = cls(cfg)
        m.load_state_dict(syn.seeded_state_dict(m.state_dict(), 0))
        m = m.to(dev)
        return m.train() if train else m.eval()

    def entry(name, workload, views, fn_eager, fn_timed, n=steps, make_pipe=None):
        try:
            with FlopMeter() as fm:
                fn_eager()
            torch.cuda.synchronize()
            ms = _timed(fn_timed, n)
            training = name.startswith(("train_step", "joint_step"))
            e = {"name": name, "workload": workload + (TRAIN_CAVEATS if training else ""), "steps": n, "ms_per_step": ms, "views_per_s": views / ms * 1e3,
                 "roofline": dict(floor_of(fm.gflop, ms), bound="mfma", peak=FP32_MFMA_PEAK_TF, unit="TFLOP/s", achieved=fm.gflop / ms, launches=fm.launches)}
            if training:
                e.update(perceptual_term="excluded", deterministic=False)
            if make_pipe is not None:                          # the same step with several replays in flight (PipelinedForward), as the headline runs it
                holder.clear()
                torch.cuda.empty_cache()
                pipe, depth = make_pipe()
                msp = _timed(pipe, 2 * n, warm=depth)
                e["pipelined"] = dict(floor_of(fm.gflop, msp), depth=depth, ms_per_step=msp, views_per_s=views / msp * 1e3)
                del pipe
            out.append(e)
        except Exception as e:
            out.append({"name": name, "workload": workload, "error": repr(e)[:300]})
        torch.cuda.empty_cache()

    model = build(FORGE)
    # --- configs[2]: 8 scenes per GPU
    s8 = {k: v.to(dev) for k, v in syn.make_sample(8, T_IN, 256, 1.5, seed=1000).items()}

    def eager8():
        with torch.no_grad():
            model(s8, ds, dev)

    def timed8():
        if "g" not in holder:
            holder["g"] = GraphedForward(model, s8, ds, dev)
        holder["g"](s8)
    def pipe8():
        p = PipelinedForward(model, s8, ds, dev, depth=2, warmup=1)
        return (lambda: p(s8)), 2
    entry("configs[2]", "BASELINE configs[2]: FORGE hot path, 8 scenes/GPU x 5 views -> 40 rendered views per step (hipGraph replay)", 40, eager8, timed8,
          make_pipe=pipe8)
    holder.clear()
    del s8
    # --- 128^3-voxel scenes (synthetic 64^3 feature volumes through reconstruct)
    s1 = {k: v.to(dev) for k, v in syn.make_sample(1, T_IN, 256, 1.5, seed=1000).items()}
    gen = torch.Generator(device=dev).manual_seed(77)
    f64 = torch.randn(1, T_IN, 128, 64, 64, 64, device=dev, generator=gen).mul_(0.5).permute(0, 1, 3, 4, 5, 2).contiguous().permute(0, 1, 5, 2, 3, 4)
    p64 = s1["cam_poses_cv2_canonicalized"][:, :T_IN].contiguous()
    c64 = geo_utils.camera_dict(s1["cam_extrinsics_cv2_canonicalized"][:, :V_OUT], s1["K_cv2"][:, :V_OUT])

    def eager64():
        with torch.no_grad():
            return model.reconstruct(f64, p64, c64)[:2]

    def timed64():
        if "g" not in holder:
            holder["g"] = GraphedCall(eager64, dev)
        holder["g"]()
    entry("grid64", "128^3-voxel scenes (configs[3]/[4] grid): 1 scene x 5 synthetic [128,64^3] feature volumes -> rotate(D=64) -> fusion at "
          "M=262144 -> heads -> 128^3 x 17 volume -> 5 views (hipGraph replay)", 5, eager64, timed64)
    holder.clear()
    del f64
    # --- pose refinement iteration (row f2): t = 5 views, 4 free poses, hipGraph replay inside refine_poses
    try:
        with torch.no_grad():
            feats = model.encoder_3d.get_feat3D(s1["images"][0, :T_IN]).reshape(1, T_IN, 128, 32, 32, 32)
            gt7 = geo_utils.mat2quat(s1["cam_poses_rel_cv2"][0, 1:T_IN])
            tgt_i, tgt_m, _, _, _ = refine._render_views(model, cfg, ds, feats, gt7, s1["K_cv2"][:, :T_IN], dev)
        init = gt7.clone()
        init[:, 4:] += 0.02
        with FlopMeter() as fm:                                    # eager iterations only (one here): forward + data-gradient backward
            refine.refine_poses(model, cfg, ds, feats, init, tgt_i, tgt_m, s1["K_cv2"][:, :T_IN], dev, iter_num=0, use_graph=False)
        _, _, dt = refine.refine_poses(model, cfg, ds, feats, init, tgt_i, tgt_m, s1["K_cv2"][:, :T_IN], dev, iter_num=2 * steps + 3, use_graph=True)
        ms = dt * 1e3
        out.append({"name": "refinement", "workload": "pose-refinement iteration (kubric_eval.py:412-530): 1 scene, 5 views, 4 free 7-D poses; rotate -> fuse "
                    "-> heads -> ray-march -> conv_rgb forward + data-gradient backward + Adam, hipGraph replay", "steps": 2 * steps,
                    "ms_per_step": ms, "views_per_s": T_IN / ms * 1e3,
                    "roofline": dict(floor_of(fm.gflop, ms), bound="mfma", peak=FP32_MFMA_PEAK_TF, unit="TFLOP/s", achieved=fm.gflop / ms,
                            launches=fm.launches)})
        try:                                                       # two refinement problems in flight (refine_poses_many): per iteration AND instance
            probs = [(feats, init, tgt_i, tgt_m, s1["K_cv2"][:, :T_IN]), (feats, init.clone(), tgt_i, tgt_m, s1["K_cv2"][:, :T_IN])]
            _, dt2 = refine.refine_poses_many(model, cfg, ds, probs, dev, iter_num=2 * steps, depth=2)
            out[-1]["pipelined"] = dict(floor_of(fm.gflop, dt2 * 1e3), depth=2, ms_per_step=dt2 * 1e3, views_per_s=T_IN / dt2)
        except Exception as e:
            out[-1]["pipelined"] = {"error": repr(e)[:200]}
    except Exception as e:
        out.append({"name": "refinement", "error": repr(e)[:300]})
    del model
    torch.cuda.empty_cache()
    # --- FORGE_poseEstimator3D inference: three fusions, 10 rendered views per scene
    m3 = build(FORGE_poseEstimator3D)

    def eager3():
        with torch.no_grad():
            m3(s1, ds, dev)

    def timed3():
        if "g" not in holder:
            holder["g"] = GraphedForward(m3, s1, ds, dev)
        holder["g"](s1)
    def pipe3():
        p = PipelinedForward(m3, s1, ds, dev, depth=4, warmup=1)
        return (lambda: p(s1)), 4
    entry("pose3d_inference", "FORGE_poseEstimator3D inference (GT poses): 1 scene x 5 views -> 3 fusions (shared input halves) -> 10 rendered views "
          "(hipGraph replay)", 10, eager3, timed3, make_pipe=pipe3)
    holder.clear()
    # --- FORGE with PREDICTED poses in inference (kubric_eval.py:371-410 predict_initial / demo.py): both pose estimators + pose head -> cameras ->
    # reconstruction -> 10 views
    try:
        mj = FORGE(syn.kubric_config(use_gt_pose=False, parameter="joint"))
        mj.load_state_dict(syn.seeded_state_dict(mj.state_dict(), 0))
        mj = mj.to(dev).eval()
        s10 = {k: v.to(dev) for k, v in syn.make_sample(1, 10, 256, 1.5, seed=12).items()}

        def eagerj():
            with torch.no_grad():
                mj(s10, ds, dev)

        def timedj():
            if "g" not in holder:
                holder["g"] = GraphedForward(mj, s10, ds, dev)
            holder["g"](s10)
        entry("joint_inference", "FORGE inference with PREDICTED poses (2-D + 3-D pose estimators + pose head -> cameras): 1 scene x 5 input views -> 10 "
                "rendered views "
              "(5 predicted + 5 given novel cameras); the 2-D estimator on a side HIP stream beside the encoder (hipGraph replay)", 10, eagerj, timedj)
        holder.clear()
        del mj, s10
    except Exception as e:
        out.append({"name": "joint_inference", "error": repr(e)[:300]})
    torch.cuda.empty_cache()
    # --- one GT-pose training step (configs[3] per-GPU step at the reference-native 32^3 / 64^3 grids): forward + backward + clip + Adam, eager
    m3 = m3.train()
    opt = torch.optim.Adam([p for p in m3.parameters() if p.requires_grad], lr=1e-4, fused=True)     # torch's multi-tensor Adam: same update, one launch chain

    def train_step():
        imgs, masks = m3(s1, ds, dev)
        mi = grouped_mse(imgs.reshape(1, 10, 3, 256, 256), s1["images"][:, :T_IN], T_IN)
        mm = grouped_mse(masks.reshape(1, 10, 1, 256, 256), s1["fg_probabilities"][:, :T_IN], T_IN)
        loss = 5.0 * (mi[0] + mi[1]) + mm[0] + mm[1]
        opt.zero_grad(set_to_none=True)
        loss.backward()
        train.clip_grad_norm_(m3.parameters(), 10.0)
        opt.step()
    entry("train_step", "GT-pose training step (kubric_train_pose_3D.py; scripts/kubric_trainer.py:47-59): FORGE_poseEstimator3D, 1 scene x 5 views, "
          "3 fusions, 10 rendered views, fused MSE, backward, clip 10, Adam; train-mode BatchNorm on the HIP kernels; eager launch", 10, train_step, train_step)
    try:
        # how far apart are two backward passes of the SAME state (atomics-ordered weight gradients)? max |g1 - g2| / max |g1| over the parameters
        def grads():
            imgs, masks = m3(s1, ds, dev)
            mi = grouped_mse(imgs.reshape(1, 10, 3, 256, 256), s1["images"][:, :T_IN], T_IN)
            mm = grouped_mse(masks.reshape(1, 10, 1, 256, 256), s1["fg_probabilities"][:, :T_IN], T_IN)
            ps = [p for p in m3.parameters() if p.requires_grad]
            gs = torch.autograd.grad(5.0 * (mi[0] + mi[1]) + mm[0] + mm[1], ps, allow_unused=True)
            return [g for g in gs if g is not None]
        bn_state = {k: v.clone() for k, v in m3.state_dict().items() if "running_" in k or "num_batches" in k}
        g1 = grads()
        m3.load_state_dict(bn_state, strict=False)              # the same running statistics for the second pass
        g2 = grads()
        m3.load_state_dict(bn_state, strict=False)
        rel = max(float((a - b).abs().max() / a.abs().max().clamp_min(1e-30)) for a, b in zip(g1, g2))
        for e_ in out:
            if e_.get("name") == "train_step":
                e_["gradient_run_to_run_rel_max"] = rel
        del g1, g2
    except Exception as e:
        out.append({"name": "train_step_determinism_probe", "error": repr(e)[:200]})
    # the per-GPU shape of BASELINE configs[3]: 4 scenes per GPU (bounded: 3 timed steps of ~175 ms)
    try:
        s4 = {k: v.to(dev) for k, v in syn.make_sample(4, T_IN, 256, 1.5, seed=1001).items()}

        def train_step4():
            imgs, masks = m3(s4, ds, dev)
            mi = grouped_mse(imgs.reshape(4, 10, 3, 256, 256), s4["images"][:, :T_IN], T_IN)
            mm = grouped_mse(masks.reshape(4, 10, 1, 256, 256), s4["fg_probabilities"][:, :T_IN], T_IN)
            loss = 5.0 * (mi[0] + mi[1]) + mm[0] + mm[1]
            opt.zero_grad(set_to_none=True)
            loss.backward()
            train.clip_grad_norm_(m3.parameters(), 10.0)
            opt.step()
        entry("train_step_4_scenes", "the same training step at configs[3]'s per-GPU batch: 4 scenes x 5 views -> 40 rendered views per step; eager launch",
              40, train_step4, train_step4, n=steps)
        del s4
    except Exception as e:
        out.append({"name": "train_step_4_scenes", "error": repr(e)[:300]})
    # BASELINE configs[3] at its REAL per-GPU shape: 4 scenes x 128^3-voxel render grid = 64^3 feature grid (models/rotate.py:115-117; the encoder cannot
    # produce 64^3 features from 256^2 images, models/encoder.py:49, so synthetic [4,5,128,64^3] feature volumes enter at rotate): rotate(D=64), three
    # fusions at M = 4 x 262144, heads to 128^3, 40 ray-marched views, loss, backward (data + weight gradients), clip, Adam
    try:
        s4 = {k: v.to(dev) for k, v in syn.make_sample(4, T_IN, 256, 1.5, seed=1001).items()}
        gen4 = torch.Generator(device=dev).manual_seed(78)
        f4 = torch.randn(4, T_IN, 128, 64, 64, 64, device=dev, generator=gen4).mul_(0.5).permute(0, 1, 3, 4, 5, 2).contiguous().permute(0, 1, 5, 2, 3, 4)
        c4 = geo_utils.camera_dict(s4["cam_extrinsics_cv2_canonicalized"][:, :T_IN].repeat(1, 2, 1, 1), s4["K_cv2"][:, :T_IN].repeat(1, 2, 1, 1))
        p4 = s4["cam_poses_cv2_canonicalized"][:, :T_IN].contiguous()

        def train_step4g():
            imgs, masks = m3.reconstruct(f4, p4, c4)[:2]
            mi = grouped_mse(imgs.reshape(4, 10, 3, 256, 256), s4["images"][:, :T_IN], T_IN)
            mm = grouped_mse(masks.reshape(4, 10, 1, 256, 256), s4["fg_probabilities"][:, :T_IN], T_IN)
            loss = 5.0 * (mi[0] + mi[1]) + mm[0] + mm[1]
            opt.zero_grad(set_to_none=True)
            loss.backward()
            train.clip_grad_norm_(m3.parameters(), 10.0)
            opt.step()
        entry("train_step_4_scenes_grid64", "BASELINE configs[3] per-GPU shape: GT-pose training step, 4 scenes x 5 synthetic [128,64^3] feature volumes "
              "(128^3-voxel render grid; the ENCODER is not run - its forward, backward and gradient all-reduce are not in this number) -> rotate(D=64) -> "
                      "3 fusions -> heads -> 128^3 x 17 volumes -> 40 rendered views, backward, clip 10, Adam; eager launch",
              40, train_step4g, train_step4g, n=steps)
        del s4, f4
    except Exception as e:
        out.append({"name": "train_step_4_scenes_grid64", "error": repr(e)[:300]})
    torch.cuda.empty_cache()
    out.extend(joint_configs(dev, steps=max(3, steps // 2)))
    # the same step captured into ONE hipGraph (forge_amd.graph.GraphedStep: forward, loss, backward, clip, capturable Adam) - single-process
    # training is host-bound at one scene (~1000 launches per step); reported beside the eager number, which is what a DDP wrapper runs
    try:
        from forge_amd.graph import GraphedStep
        opt_g = torch.optim.Adam([p for p in m3.parameters() if p.requires_grad], lr=1e-4, capturable=True)

        def graph_fn():
            imgs, masks = m3(s1, ds, dev)
            mi = grouped_mse(imgs.reshape(1, 10, 3, 256, 256), s1["images"][:, :T_IN], T_IN)
            mm = grouped_mse(masks.reshape(1, 10, 1, 256, 256), s1["fg_probabilities"][:, :T_IN], T_IN)
            loss = 5.0 * (mi[0] + mi[1]) + mm[0] + mm[1]
            loss.backward()
            train.clip_grad_norm_(m3.parameters(), 10.0)
            opt_g.step()
            return loss.detach()
        gs = GraphedStep(graph_fn, opt_g, warmup=2)
        msg = _timed(gs, steps)
        ts = [e for e in out if e.get("name") == "train_step" and "ms_per_step" in e]
        if ts:
            ts[-1]["hipgraph_replay"] = dict(floor_of(ts[-1]["roofline"]["executed_gflop"], msg), ms_per_step=msg, views_per_s=10 / msg * 1e3)
        del gs
    except Exception as e:
        ts = [x for x in out if x.get("name") == "train_step"]
        if ts:
            ts[-1]["hipgraph_replay"] = {"error": repr(e)[:200]}
    return out


def joint_stock_share():
    """Share of the joint step's kernel time spent in stock-torch (MIOpen / rocBLAS / ATen) kernels, from the committed rocprofv3 kernel trace
    of tools/joint_step_probe.py (profiles/r05_joint_*_kernel_share.json, written by tools/joint_kernel_share.py): a per-name attribution the
    process cannot make about itself. None when no profile is committed."""
    import glob
    res = {}
    for f in sorted(glob.glob(os.path.join(ROOT, "profiles", "r*_joint_*kernel_share.json"))):
        try:
            d = json.load(open(f))
            res[d.get("workload", os.path.basename(f))] = {"stock_torch_share_of_kernel_time": d["stock_share"], "forge_share_of_kernel_time": d["forge_share"],
                                                           "kernel_ms_per_step": d.get("kernel_ms_per_step"), "source": os.path.basename(f)}
        except Exception:
            continue
    return res or None


def joint_configs(dev, steps=5):
    """BASELINE configs[4] on one GPU: the joint 2D3D fine-tune iteration of kubric_train_joint.py:111-141 - FORGE with
    PREDICTED poses (attention blocks of the 2-D / 3-D pose estimators and the pose head on stock torch kernels; encoder / rotate / fusion / heads / ray-march /
    conv_rgb and, since round 5, every convolution + BatchNorm of the two pose estimators on the HIP kernels), 5 input + 5 novel views, compute_all_loss_nvs
    (scripts/kubric_compute_loss.py:121-172), backward through the
    pose chain (rotate's d pose, the ray-marcher's d(R, T)), clip 10, Adam over the parameter list of kubric_train_joint.py:111-116.
      joint_step          reference-native grids (32^3 features, 64^3 render volume)
      joint_step_grid64   the configuration's 128^3-voxel scenes: synthetic [1,5,128,64^3] feature volumes enter the reconstruction
                          (FORGE.forward(features_recon=...)), the pose networks keep their native inputs
    Each entry: ms_per_step, views_per_s, `roofline` = the FLOPs libforge's matrix-core launches execute as a time floor (stock-torch FLOPs are
    counted separately by torch's FlopCounterMode and NOT part of that floor), and `stock_torch` = the pose networks' own forward + backward
    timed alone on the same inputs (live) beside the per-kernel-name share of a committed rocprofv3 trace."""
    from forge_amd import train
    from forge_amd.flopmeter import FlopMeter
    from forge_amd.model import FORGE
    out = []
    cfg = syn.kubric_config(use_gt_pose=False, parameter="joint")
    cfg.loss.regu_origin_proj = 1.0                                   # config/kubric/joint_pose_2d3d.yaml:34-38 (perceptual term: no VGG weights offline)
    ds = syn.SyntheticDataset(1.5)
    try:
        model = FORGE(cfg)
        model.load_state_dict(syn.seeded_state_dict(model.state_dict(), 0))
        model = model.to(dev).train()
        params = [p for m in (model.encoder_traj, model.pose_head, model.encoder_3d.fusion_feature, model.encoder_3d.density_head, model.render)
                  for p in m.parameters()]                            # kubric_train_joint.py:111-116
        opt = torch.optim.Adam(params, lr=1e-4, fused=True)
        sample = {k: v.to(dev) for k, v in syn.make_sample(1, 10, 256, 1.5, seed=12).items()}
        gen = torch.Generator(device=dev).manual_seed(79)
        f64 = torch.randn(1, T_IN, 128, 64, 64, 64, device=dev, generator=gen).mul_(0.5).permute(0, 1, 3, 4, 5, 2).contiguous().permute(0, 1, 5, 2, 3, 4)
    except Exception as e:
        return [{"name": "joint_step", "error": repr(e)[:300]}]

    def make_step(feats, smp=None):
        call = model if feats is None else (lambda s, d, dv: model(s, d, dv, features_recon=feats))
        smp = sample if smp is None else smp

        def step():
            loss, _, _, _ = train.compute_all_loss_nvs(cfg, 0, smp, ds, call, {}, dev)
            opt.zero_grad(set_to_none=True)
            loss.backward()
            train.clip_grad_norm_(model.parameters(), 10.0)
            opt.step()
            return loss
        return step

    def pose_nets_only():
        """both pose estimators + pose head alone: forward and backward on the step's own (detached) inputs"""
        with torch.no_grad():
            clips = sample["images"][:, :T_IN]
            feats = model.encoder_3d.get_feat3D(clips.reshape(T_IN, 3, 256, 256)).reshape(1, T_IN, 128, 32, 32, 32)
        feats = feats.detach().requires_grad_(True)

        def run():
            _, _, pose = model.predict_poses(feats, clips, sample, ds, dev)
            (pose["pred"].square().sum() + pose["conf"].sum()).backward()
            for p in model.parameters():
                p.grad = None
            feats.grad = None
        return run

    share = joint_stock_share()
    for name, feats, workload in (
            ("joint_step", None, "BASELINE configs[4] step at the reference-native grids: FORGE joint 2D3D fine-tune (predicted poses), 1 scene x 5 input "
                    "+ 5 novel "
             "views 256^2 -> 10 rendered views, compute_all_loss_nvs, backward incl. the pose chain, clip 10, Adam; train-mode BatchNorm / Dropout; "
                     "eager launch"),
            ("joint_step_grid64", f64, "BASELINE configs[4] at its 128^3-voxel grid: the same step with 5 synthetic [128,64^3] feature volumes entering "
                    "rotate(D=64) -> "
             "fusion at M=262144 -> heads -> 128^3 x 17 volume -> 10 rendered views; pose networks on their native inputs; eager launch")):
        try:
            step = make_step(feats)
            step()                                                    # allocator / MIOpen solver warm-up outside the meters
            torch.cuda.synchronize()
            from torch.utils.flop_counter import FlopCounterMode
            with FlopMeter() as fm, FlopCounterMode(display=False) as fc:
                step()
            torch.cuda.synchronize()
            ms = _timed(step, steps, warm=1)
            ms_pose = _timed(pose_nets_only(), steps, warm=1)
            e = {"name": name, "workload": workload + TRAIN_CAVEATS, "perceptual_term": "excluded", "deterministic": False,
                 "steps": steps, "ms_per_step": ms, "views_per_s": 10 / ms * 1e3,
                 "roofline": dict(floor_of(fm.gflop, ms), bound="mfma", peak=FP32_MFMA_PEAK_TF, unit="TFLOP/s", achieved=fm.gflop / ms, launches=fm.launches,
                                  note="executed_gflop = libforge matrix-core launches (the pose estimators' convolutions included since round 5); the "
                                          "attention "
                                       "blocks' rocBLAS GEMMs are in stock_torch.gflop"),
                 "pose_networks": {"what": "2-D + 3-D pose estimators and pose head alone, forward + backward on the step's inputs (convolutions + "
                         "BatchNorm on libforge, "
                                           "attention blocks on rocBLAS / ATen)", "fwd_bwd_ms": ms_pose, "share_of_step": ms_pose / ms},
                 "stock_torch": {"what": "kernels that are not libforge's (rocBLAS attention GEMMs, ATen element-wise / softmax / LayerNorm / "
                         "optimizer): FLOPs "
                                         "counted by torch's FlopCounterMode; share of kernel time by kernel NAME from the committed rocprofv3 trace",
                                 "gflop": fc.get_total_flops() / 1e9, "rocprofv3": (share or {}).get(name)}}
            # the same step as ONE hipGraph (forge_amd.graph.GraphedStep: forward, loss, backward, clip, capturable Adam): the eager step is host-bound
            # (~3000 launches from Python); reported beside the eager number, which is what a DDP wrapper runs
            try:
                from forge_amd.graph import GraphedStep
                opt_g = torch.optim.Adam(params, lr=1e-4, capturable=True)
                call_g = model if feats is None else (lambda s, d, dv: model(s, d, dv, features_recon=feats))

                def graph_fn():
                    loss, _, _, _ = train.compute_all_loss_nvs(cfg, 0, sample, ds, call_g, {}, dev)
                    loss.backward()
                    train.clip_grad_norm_(model.parameters(), 10.0)
                    opt_g.step()
                    return loss.detach()
                gs = GraphedStep(graph_fn, opt_g, warmup=2)
                msg = _timed(gs, steps, warm=1)
                e["hipgraph_replay"] = dict(floor_of(fm.gflop, msg), ms_per_step=msg, views_per_s=10 / msg * 1e3)
                del gs, opt_g
            except Exception as ex:
                e["hipgraph_replay"] = {"error": repr(ex)[:300]}
            for p_ in model.parameters():
                p_.grad = None
            torch.cuda.empty_cache()
            out.append(e)
        except Exception as e:
            out.append({"name": name, "workload": workload, "error": repr(e)[:300]})
        torch.cuda.empty_cache()
    # the reference's joint configuration trains 4 scenes per GPU (config/kubric/joint_pose_2d3d.yaml: batch_size 4): the GPU-bound regime of the same step
    try:
        s4 = {k: v.to(dev) for k, v in syn.make_sample(4, 10, 256, 1.5, seed=13).items()}
        step4 = make_step(None, s4)
        step4()
        torch.cuda.synchronize()
        with FlopMeter() as fm4:
            step4()
        torch.cuda.synchronize()
        ms4 = _timed(step4, max(2, steps // 2), warm=1)
        out.append({"name": "joint_step_4_scenes", "workload": "the joint step at the reference configuration's per-GPU batch (4 scenes x (5 + 5) views -> "
                "40 rendered views per step); eager launch" + TRAIN_CAVEATS,
                    "perceptual_term": "excluded", "deterministic": False,
                    "steps": max(2, steps // 2), "ms_per_step": ms4, "views_per_s": 40 / ms4 * 1e3,
                            "stock_torch": {"rocprofv3": (share or {}).get("joint_step_4_scenes")},
                    "roofline": dict(floor_of(fm4.gflop, ms4), bound="mfma", peak=FP32_MFMA_PEAK_TF, unit="TFLOP/s", achieved=fm4.gflop / ms4,
                            launches=fm4.launches)})
        del s4
    except Exception as e:
        out.append({"name": "joint_step_4_scenes", "error": repr(e)[:300]})
    for p_ in model.parameters():
        p_.grad = None
    torch.cuda.empty_cache()
    return out
