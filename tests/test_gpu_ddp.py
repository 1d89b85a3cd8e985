"""GPU (-m gpu): the data-parallel TRAINING path (kubric_train_pose_3D.py:119-130): the model wrapped in torch
DistributedDataParallel, one scene per rank, gradients all-reduced in buckets. Two ranks share the single GPU of the test box, so
the process group is gloo (RCCL refuses two ranks on one device; the bucketing / hook logic of DDP is backend-independent); the
averaged gradients must equal those of ONE process that sees both scenes as a batch of 2."""
import os
import socket
import sys

import pytest
import torch
import torch.multiprocessing as mp

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
KEYS = ["encoder_3d.fusion_feature.cells.0.conv_gate.weight", "encoder_3d.conv1.0.weight", "encoder_3d.density_head.6.weight",
        "encoder_3d.features_head.0.weight", "encoder_3d.feature_extraction.0.weight", "render.conv_rgb.6.weight",
        "encoder_3d.fusion_feature.fusion_norm.bias"]


def _free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    p = s.getsockname()[1]
    s.close()
    return p


def _build(dev, batch_stats=False):
    from forge_amd import synthetic as syn
    from forge_amd.model_single_pose_estimator import FORGE_poseEstimator3D
    cfg = syn.kubric_config()
    model = FORGE_poseEstimator3D(cfg)
    model.load_state_dict(syn.seeded_state_dict(model.state_dict(), 0))
    model = model.to(dev).train()
    if not batch_stats:
        for m in model.modules():                   # running statistics: per-rank batch statistics would differ from the batch-of-2 run
            if isinstance(m, torch.nn.modules.batchnorm._BatchNorm):
                m.eval()
    return cfg, model


def _loss(model, sample, dev):
    from forge_amd import synthetic as syn
    imgs, masks = model(sample, syn.SyntheticDataset(1.5), dev)
    b = sample["images"].shape[0]
    tgt_i = sample["images"].repeat(1, 2, 1, 1, 1).reshape(b * 10, 3, 256, 256)
    tgt_m = sample["fg_probabilities"].repeat(1, 2, 1, 1, 1).reshape(b * 10, 1, 256, 256)
    return 5.0 * torch.nn.functional.mse_loss(imgs, tgt_i) + torch.nn.functional.mse_loss(masks, tgt_m)


def _sample(seeds, dev):
    from forge_amd import synthetic as syn
    parts = [syn.make_sample(1, 5, 256, 1.5, seed=s) for s in seeds]
    return {k: torch.cat([p[k] for p in parts], dim=0).to(dev) for k in parts[0]}


def _worker(rank, world, port, q, sync_bn=False):
    sys.path.insert(0, ROOT)
    os.environ.update(RANK=str(rank), LOCAL_RANK=str(rank), WORLD_SIZE=str(world), MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port))
    import torch.distributed as dist
    from forge_amd import dist as fd
    fd.init()                                       # 2 ranks on 1 GPU -> gloo, both on cuda:0
    dev = torch.device("cuda", torch.cuda.current_device())
    _, model = _build(dev, batch_stats=sync_bn)
    if sync_bn:                                     # the reference's configuration (kubric_train_pose_3D.py:119-124): statistics over all ranks
        model = torch.nn.SyncBatchNorm.convert_sync_batchnorm(model)
    ddp = torch.nn.parallel.DistributedDataParallel(model, device_ids=[dev.index], find_unused_parameters=True)
    loss = _loss(ddp, _sample([100 + rank], dev), dev)
    loss.backward()
    torch.cuda.synchronize()
    named = dict(model.named_parameters())
    out = {k: named[k].grad.detach().cpu().numpy() for k in KEYS}      # numpy: pickled by value (tensors travel as fds of a process that may be gone)
    q.put((rank, float(loss.detach()), out))
    dist.barrier()
    dist.destroy_process_group()


@pytest.mark.parametrize("sync_bn", [False, True])
def test_ddp_two_ranks_equal_one_process_batch_of_two(sync_bn):
    """sync_bn = False: BatchNorm on running statistics (eval) in every module; sync_bn = True: SyncBatchNorm.convert_sync_batchnorm + train
    mode, i.e. batch statistics over both ranks - which must equal ordinary train-mode BatchNorm over the batch of 2 in one process."""
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = _free_port()
    procs = [ctx.Process(target=_worker, args=(r, 2, port, q, sync_bn)) for r in range(2)]
    for p in procs:
        p.start()
    import queue
    import time
    res, deadline = [], time.time() + 420
    while len(res) < len(procs):                     # fail fast when a rank dies instead of waiting out the queue timeout
        try:
            res.append(q.get(timeout=2))
        except queue.Empty:
            dead = [p.exitcode for p in procs if p.exitcode not in (None, 0)]
            if dead or time.time() > deadline:
                for p in procs:
                    if p.is_alive():
                        p.terminate()
                pytest.fail("a DDP rank exited with %s before reporting (or the run timed out)" % (dead or "timeout",))
    res.sort(key=lambda t: t[0])
    for p in procs:
        p.join(timeout=120)
        assert p.exitcode == 0
    # every rank holds the same (averaged) gradients
    for k in KEYS:
        assert (res[0][2][k] == res[1][2][k]).all(), k
    # one process, both scenes as a batch of 2: mean loss over 2 scenes -> the same averaged gradients
    dev = torch.device("cuda:0")
    _, model = _build(dev, batch_stats=sync_bn)
    loss = _loss(model, _sample([100, 101], dev), dev)
    loss.backward()
    assert abs(float(loss.detach()) - 0.5 * (res[0][1] + res[1][1])) < 1e-5 * max(1.0, abs(float(loss.detach())))
    named = dict(model.named_parameters())
    for k in KEYS:
        ref = named[k].grad.detach().cpu()
        err = (torch.from_numpy(res[0][2][k]) - ref).abs().max().item()
        gscale = max(named[kk].grad.abs().max().item() for kk in KEYS)
        assert err < (1e-2 if sync_bn else 2e-4) * max(ref.abs().max().item(), 1e-3 * gscale) + 1e-9, (k, err, ref.abs().max().item())


def _ray_worker(rank, world, port, q):
    sys.path.insert(0, ROOT)
    os.environ.update(RANK=str(rank), LOCAL_RANK=str(rank), WORLD_SIZE=str(world), MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port))
    import torch.distributed as dist
    from forge_amd import dist as fd, ops, synthetic as syn
    fd.init()
    dev = torch.device("cuda", torch.cuda.current_device())
    D, C, V, Hr, S = 64, 16, 3, 128, 64
    feat, dens = syn.blob_volumes(1, D, C, seed=5)
    feat, dens = feat.to(dev), dens.to(dev)
    _, extr, _ = syn.orbit_cameras(10, 1.5, 15.0)
    E = extr[:V]
    K = syn.intrinsics(256) / 2.0
    cam = torch.cat([E[:, :3, :3].reshape(V, 9), E[:, :3, 3], K[0, 0].expand(V, 1), K[1, 1].expand(V, 1), K[0, 2].expand(V, 1),
                     K[1, 2].expand(V, 1)], dim=1).contiguous().to(dev)
    v2v = torch.zeros(V, dtype=torch.int32, device=dev)
    h = [0.5 * (D - 1) / D] * 3
    with torch.no_grad():
        got = fd.render_rays_sharded(feat, dens, cam, v2v, Hr, Hr, S, 0.5, 2.0, h, True)       # each rank marches its row band, one all_gather
        ref = ops.render_rays(feat, dens, cam, v2v, Hr, Hr, S, 0.5, 2.0, h, True)
    torch.cuda.synchronize()
    q.put((rank, [bool(torch.equal(a, b)) for a, b in zip(got, ref)], [tuple(a.shape) for a in got]))
    dist.barrier()
    dist.destroy_process_group()


def test_ray_sharded_render_two_ranks_on_the_gpu():
    """BASELINE configs[4] "per-ray sharding": two ranks (sharing the test box's GPU, gloo) each ray-march a band of image rows of every
    view with the HIP kernel and all_gather the bands; the assembled images equal the single-launch render bit for bit."""
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = _free_port()
    procs = [ctx.Process(target=_ray_worker, args=(r, 2, port, q)) for r in range(2)]
    for p in procs:
        p.start()
    import queue
    import time
    res, deadline = [], time.time() + 240
    while len(res) < len(procs):
        try:
            res.append(q.get(timeout=2))
        except queue.Empty:
            dead = [p.exitcode for p in procs if p.exitcode not in (None, 0)]
            if dead or time.time() > deadline:
                for p in procs:
                    if p.is_alive():
                        p.terminate()
                pytest.fail("a rank exited with %s before reporting (or the run timed out)" % (dead or "timeout",))
    for p in procs:
        p.join(timeout=60)
        assert p.exitcode == 0
    for rank, equal, shapes in res:
        assert all(equal), (rank, equal)
        assert shapes == [(3, 16, 128, 128), (3, 1, 128, 128), (3, 1, 128, 128)]


def test_bench_entry_two_ranks_on_the_shared_gpu():
    """`python bench.py --gpus 2` (no torchrun environment) on the one-GPU test box: the entry starts its own two ranks, each captures its
    hipGraph BEFORE the process group exists, the ranks rendezvous (gloo here: two ranks on one device; RCCL on a real node), time the step
    between barriers, all-reduce time / SSE / view counts, and rank 0 prints one line with n_gpus = 2 and twice the per-rank views. Without
    FORGE_BENCH_ALLOW_SHARED_GPUS the same command must refuse (a scaling number from shared devices would be meaningless)."""
    import json
    import subprocess
    import sys
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    env = {k: v for k, v in os.environ.items() if k not in ("RANK", "LOCAL_RANK", "WORLD_SIZE", "MASTER_ADDR", "MASTER_PORT")}
    cmd = [sys.executable, os.path.join(root, "bench.py"), "--gpus", "2", "--steps", "2", "--warmup", "1", "--no-microbench"]
    bad = subprocess.run(cmd, capture_output=True, text=True, timeout=600, env=env, cwd=root)
    assert bad.returncode != 0 and "GPU(s) visible" in (bad.stderr + bad.stdout)
    ok = subprocess.run(cmd, capture_output=True, text=True, timeout=900, env=dict(env, FORGE_BENCH_ALLOW_SHARED_GPUS="1"), cwd=root)
    assert ok.returncode == 0, ok.stderr[-2000:]
    lines = [l for l in ok.stdout.splitlines() if l.startswith("{")]
    assert len(lines) == 1
    d = json.loads(lines[0])
    assert d["n_gpus"] == 2 and d["steps"] == 2 and d["value"] > 0 and "cpu_baseline" not in d
    assert abs(d["value"] - 2 * 5 * 2 / (d["ms_per_step"] * 2 / 1e3)) < 1e-6 * d["value"]          # whole-job views / max-over-ranks time
