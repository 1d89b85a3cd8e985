"""CPU, world_size 2, gloo: the N>1 path of the bench/eval harness — scene sharding without a data-path
collective + scalar metric all-reduce — gives rank-count-invariant results."""
import os
import socket
import sys

import pytest
import torch
import torch.multiprocessing as mp

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    p = s.getsockname()[1]
    s.close()
    return p


def _worker(rank, world, port, q):
    sys.path.insert(0, ROOT)
    os.environ.update(RANK=str(rank), LOCAL_RANK=str(rank), WORLD_SIZE=str(world),
                      MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port))
    from forge_amd import dist as fd
    r, lr, w = fd.init(backend="gloo")
    assert (r, w) == (rank, world)
    n_scenes = 7
    mine = fd.shard_indices(n_scenes, r, w)
    # per-scene "SSE" and pixel count that only depend on the scene index
    sse = sum(float(i + 1) * 0.5 for i in mine)
    cnt = sum(100.0 + i for i in mine)
    tot = fd.all_reduce_scalars([sse, cnt, float(len(mine))], "cpu", "sum")
    tmax = fd.all_reduce_scalars([float(rank + 1)], "cpu", "max")
    fd.barrier()
    q.put((rank, mine, tot, tmax))
    torch.distributed.destroy_process_group()


def test_scene_sharding_and_metric_allreduce_world2():
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = _free_port()
    procs = [ctx.Process(target=_worker, args=(r, 2, port, q)) for r in range(2)]
    for p in procs:
        p.start()
    res = sorted(q.get(timeout=120) for _ in procs)
    for p in procs:
        p.join(timeout=60)
        assert p.exitcode == 0
    shards = [r[1] for r in res]
    assert sorted(shards[0] + shards[1]) == list(range(7)) and not set(shards[0]) & set(shards[1])
    exp = [sum((i + 1) * 0.5 for i in range(7)), sum(100.0 + i for i in range(7)), 7.0]
    for r in res:
        assert r[2] == pytest.approx(exp) and r[3] == [2.0]


def test_world1_is_a_noop():
    from forge_amd import dist as fd
    for k in ("RANK", "LOCAL_RANK", "WORLD_SIZE"):
        os.environ.pop(k, None)
    assert fd.init() == (0, 0, 1)
    assert fd.all_reduce_scalars([1.5, 2.0], "cpu") == [1.5, 2.0]
    assert fd.shard_indices(5, 0, 1) == [0, 1, 2, 3, 4]
    assert fd.psnr_from_sse(1.0, 100.0) == pytest.approx(20.0)


def _ray_worker(rank, world, port, q):
    sys.path.insert(0, ROOT)
    sys.path.insert(0, os.path.join(ROOT, "oracle"))
    os.environ.update(RANK=str(rank), LOCAL_RANK=str(rank), WORLD_SIZE=str(world), MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port))
    import forge_oracle as fo
    from forge_amd import dist as fd, synthetic as syn
    fd.init(backend="gloo")
    feat, dens = syn.blob_volumes(1, 12, 4, seed=2)
    _, extr, _ = syn.orbit_cameras(3, 1.5, 10.0)
    Kh = fo.halve_intrinsics(syn.intrinsics(32)[None].repeat(3, 1, 1))
    cam = torch.cat([extr[:, :3, :3].reshape(3, 9), extr[:, :3, 3], Kh[:, 0, 0:1], Kh[:, 1, 1:2], Kh[:, 0, 2:3], Kh[:, 1, 2:3]], dim=1)
    v2v = torch.zeros(3, dtype=torch.int32)
    h = fo.grid_half_extent(12, 1.0)

    def oracle_fn(f, d, c, v2v_, Hr, Wr, S, zmin, zmax, half, want_depth):      # CPU stand-in with the ops.render_rays signature
        K = torch.zeros(c.shape[0], 3, 3)
        K[:, 0, 0], K[:, 1, 1], K[:, 0, 2], K[:, 1, 2], K[:, 2, 2] = c[:, 12], c[:, 13], c[:, 14], c[:, 15], 1.0
        r = fo.render_rays(f[v2v_.long()], d[v2v_.long()], c[:, :9].reshape(-1, 3, 3), c[:, 9:12], K, Hr, Wr, S, zmin, zmax, 1.0, want_depth)
        C = f.shape[1]
        outs = [r[..., :C].permute(0, 3, 1, 2), r[..., C:C + 1].permute(0, 3, 1, 2)]
        if want_depth:
            outs.append(r[..., C + 1:].permute(0, 3, 1, 2))
        return tuple(outs)
    got = fd.render_rays_sharded(feat, dens, cam, v2v, 16, 16, 24, 0.5, 2.0, (h, h, h), True, render_fn=oracle_fn)
    ref = oracle_fn(feat, dens, cam, v2v, 16, 16, 24, 0.5, 2.0, (h, h, h), True)
    q.put((rank, [float((a - b).abs().max()) for a, b in zip(got, ref)], [tuple(a.shape) for a in got]))
    torch.distributed.destroy_process_group()


def test_ray_sharded_render_world2():
    """config-5 style per-ray sharding: two ranks each march half of the image rows (principal point shifted by the band origin),
    all_gather the bands, and both end up with exactly the full-image result."""
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = _free_port()
    procs = [ctx.Process(target=_ray_worker, args=(r, 2, port, q)) for r in range(2)]
    for p in procs:
        p.start()
    res = sorted(q.get(timeout=180) for _ in procs)
    for p in procs:
        p.join(timeout=60)
        assert p.exitcode == 0
    for rank, errs, shapes in res:
        assert shapes == [(3, 4, 16, 16), (3, 1, 16, 16), (3, 1, 16, 16)]
        assert max(errs) < 1e-6


def test_bench_entry_launches_its_own_ranks_dry_run(tmp_path):
    """`python bench.py --gpus 8` (no torchrun environment) must start 8 ranks itself and report n_gpus: 8 - here as the
    CPU/gloo rehearsal of exactly that entry (`--dry-run`: launch, rendezvous on 127.0.0.1, barrier, MAX / SUM reductions, one JSON
    line from rank 0) - and must FAIL when the launch environment's world size contradicts --gpus."""
    import json
    import subprocess
    env = {k: v for k, v in os.environ.items() if k not in ("RANK", "LOCAL_RANK", "WORLD_SIZE", "MASTER_ADDR", "MASTER_PORT")}
    full = str(tmp_path / "full.json")

    def strict_line(stdout):
        """The driver's view: the LAST stdout line, under 4 KB, strict JSON (no NaN / Infinity tokens)."""
        out_lines = stdout.splitlines()
        assert [l for l in out_lines if l.startswith("{")] == [out_lines[-1]], stdout
        assert len(out_lines[-1]) < 4096

        def refuse(tok):
            raise AssertionError("non-strict JSON token %s" % tok)
        return json.loads(out_lines[-1], parse_constant=refuse)
    r = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "8", "--steps", "3", "--scenes", "2", "--dry-run", "--full-record", full],
                       capture_output=True, text=True, timeout=600, env=env, cwd=ROOT)
    assert r.returncode == 0, r.stderr[-2000:]
    c = strict_line(r.stdout)
    assert c["n_gpus"] == 8 and c["dry_run"] is True and c["ranks_ok"] == 8 and c["full_record"] == full
    assert set(c["multi_rank"]) == {"ddp_train", "failing_record", "ray_sharded_joint"} and c["multi_rank"]["ddp_train"][3] == 8
    assert c["multi_rank"]["failing_record"][3] == 7
    d = json.load(open(full))                                            # every detail lives in the full record
    assert d["n_gpus"] == 8 and d["dry_run"] is True and d["views_counted"] == 8 * 2 * 5 * 3 and d["steps"] == 3
    assert d["ranks_ok"] == 8 and d["error"] is None
    # the sub-record skeleton of the real multi-rank line (benchkit.multirank.multi_rank_records), rehearsed with token workloads: a DDP step
    # with and without no_sync(), a record that fails on ONE rank (reported, the line and the other records survive), the ray-sharded render op
    m = d["multi_rank"]
    assert m["ddp_train"]["ranks_ok"] == 8 and m["ddp_train"]["ms_per_step"] > 0 and m["ddp_train"]["ms_per_step_no_sync"] > 0
    assert m["failing_record"]["ranks_ok"] == 7 and len(m["failing_record"]["errors"]) == 1 and "rank 7" in m["failing_record"]["errors"][0]
    assert m["ray_sharded_joint"]["ranks_ok"] == 8 and m["ray_sharded_joint"]["rows"] == 16 and m["ray_sharded_joint"]["d_feat"] == m["ray_sharded_joint"]["expected_d_feat"]
    # a sub-record that never comes back: the watchdog prints the MAIN line and every rank leaves with exit code 0
    w = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "2", "--steps", "1", "--dry-run", "--rehearse-hang", "--full-record", full],
                       capture_output=True, text=True, timeout=300, env=dict(env, FORGE_BENCH_SUBRECORD_DEADLINE_S="3"), cwd=ROOT)
    assert w.returncode == 0, w.stderr[-2000:]
    dw = strict_line(w.stdout)
    assert dw["n_gpus"] == 2 and "did not finish" in dw["multi_rank"]["error"]
    # --train rehearsal: the same entry wraps a model in DistributedDataParallel over the 8 ranks (gloo here, RCCL on the node), steps it, and
    # the replicas end up identical on every rank
    t = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "8", "--steps", "3", "--scenes", "4", "--dry-run", "--train", "--full-record", full],
                       capture_output=True, text=True, timeout=600, env=env, cwd=ROOT)
    assert t.returncode == 0, t.stderr[-2000:]
    dt_ = strict_line(t.stdout)
    assert dt_["n_gpus"] == 8 and dt_["train"] is True and dt_["ranks_ok"] == 8 and dt_["replicas_identical"] is True
    assert dt_["views_counted"] == 8 * 4 * 10 * 3
    bad = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "4", "--dry-run"], capture_output=True, text=True,
                         timeout=120, env=dict(env, WORLD_SIZE="2", RANK="0", LOCAL_RANK="0"), cwd=ROOT)
    assert bad.returncode != 0 and "WORLD_SIZE=2" in (bad.stderr + bad.stdout)


def _loss_worker(rank, world, port, q):
    sys.path.insert(0, ROOT)
    os.environ.update(RANK=str(rank), LOCAL_RANK=str(rank), WORLD_SIZE=str(world), MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port))
    from forge_amd import dist as fd, train
    fd.init(backend="gloo")
    terms = {"recon_img": torch.tensor(1.0 + rank), "recon_mask": torch.tensor(10.0 * (rank + 1))}
    q.put((rank, train._publish({}, terms)))
    fd.barrier()
    torch.distributed.destroy_process_group()


def test_logged_losses_are_all_reduced_world2():
    """train._publish averages the loss terms over ranks with one all-reduce (north_star: all-reduce of losses)."""
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = _free_port()
    procs = [ctx.Process(target=_loss_worker, args=(r, 2, port, q)) for r in range(2)]
    for p in procs:
        p.start()
    res = dict(q.get(timeout=120) for _ in procs)
    for p in procs:
        p.join(timeout=60)
        assert p.exitcode == 0
    for r in (0, 1):
        assert res[r] == {"recon_img": pytest.approx(1.5), "recon_mask": pytest.approx(15.0)}


def _oracle_render_fn():
    import forge_oracle as fo

    def oracle_fn(f, d, c, v2v_, Hr, Wr, S, zmin, zmax, half, want_depth):      # CPU stand-in with the ops.render_rays signature, differentiable
        z, o = torch.zeros_like(c[:, 0]), torch.ones_like(c[:, 0])
        K = torch.stack([c[:, 12], z, c[:, 14], z, c[:, 13], c[:, 15], z, z, o], dim=1).reshape(-1, 3, 3)
        r = fo.render_rays(f[v2v_.long()], d[v2v_.long()], c[:, :9].reshape(-1, 3, 3), c[:, 9:12], K, Hr, Wr, S, zmin, zmax, 1.0, want_depth)
        C = f.shape[1]
        outs = [r[..., :C].permute(0, 3, 1, 2), r[..., C:C + 1].permute(0, 3, 1, 2)]
        if want_depth:
            outs.append(r[..., C + 1:].permute(0, 3, 1, 2))
        return tuple(outs)
    return oracle_fn


def _ray_case():
    import forge_oracle as fo
    from forge_amd import synthetic as syn
    feat, dens = syn.blob_volumes(1, 12, 4, seed=2)
    _, extr, _ = syn.orbit_cameras(3, 1.5, 10.0)
    Kh = fo.halve_intrinsics(syn.intrinsics(32)[None].repeat(3, 1, 1))
    cam = torch.cat([extr[:, :3, :3].reshape(3, 9), extr[:, :3, 3], Kh[:, 0, 0:1], Kh[:, 1, 1:2], Kh[:, 0, 2:3], Kh[:, 1, 2:3]], dim=1)
    g = torch.Generator().manual_seed(3)
    tg = [torch.randn(3, 4, 16, 16, generator=g), torch.randn(3, 1, 16, 16, generator=g), torch.randn(3, 1, 16, 16, generator=g)]
    return feat, dens, cam, torch.zeros(3, dtype=torch.int32), fo.grid_half_extent(12, 1.0), tg


def _ray_bwd_worker(rank, world, port, q, mode):
    sys.path.insert(0, ROOT)
    sys.path.insert(0, os.path.join(ROOT, "oracle"))
    os.environ.update(RANK=str(rank), LOCAL_RANK=str(rank), WORLD_SIZE=str(world), MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port))
    from forge_amd import dist as fd
    fd.init(backend="gloo")
    feat, dens, cam, v2v, h, tg = _ray_case()
    fn = _oracle_render_fn()
    cam = cam.clone().requires_grad_(True)
    if mode == "all":                        # replicated volume: every rank ends up with the single-process gradients
        feat, dens = feat.clone().requires_grad_(True), dens.clone().requires_grad_(True)
        fin, din = feat, dens
    else:                                    # owner mode: rank 0 owns the (differentiable) volume, the others hold placeholders
        p = torch.full((1,), 0.7, requires_grad=True)
        if rank == 0:
            fin0, din0 = feat * p, dens * (2.0 * p)
        else:
            fin0, din0 = torch.zeros_like(feat) * p, torch.zeros_like(dens) * p
        fin, din = fd.broadcast_from_owner((fin0, din0), src=0)
    outs = fd.render_rays_sharded(fin, din, cam, v2v, 16, 16, 24, 0.5, 2.0, (h, h, h), True, render_fn=fn, reduce=mode)
    loss = sum((o * t).sum() for o, t in zip(outs, tg))
    loss.backward()
    if mode == "all":
        res = (float(loss), feat.grad.numpy(), dens.grad.numpy(), cam.grad.numpy())
    else:
        res = (float(loss), p.grad.numpy(), cam.grad.numpy())
    q.put((rank, res))
    fd.barrier()
    torch.distributed.destroy_process_group()


@pytest.mark.parametrize("mode", ["all", "none"])
def test_ray_sharded_render_backward_world2(mode):
    """BASELINE configs[4] is a FINE-TUNE: the ray-sharded render is differentiable. Two ranks (gloo, the CPU oracle injected as the band
    renderer): loss and gradients of the volume, the density and the 16 camera parameters equal single-process autograd through the
    full render - reduce="all": on EVERY rank (partials all-reduced); reduce="none" + broadcast_from_owner: the volume's owner (rank 0)
    receives the full gradient of its upstream parameter, the other rank zero; camera partials sum to the full camera gradient."""
    import forge_oracle  # noqa: F401  (conftest put oracle/ on the path)
    feat, dens, cam, v2v, h, tg = _ray_case()
    fn = _oracle_render_fn()
    f1, d1, c1 = feat.clone().requires_grad_(True), dens.clone().requires_grad_(True), cam.clone().requires_grad_(True)
    p1 = torch.full((1,), 0.7, requires_grad=True)
    fin, din = (f1, d1) if mode == "all" else (feat * p1, dens * (2.0 * p1))
    ref = sum((o * t).sum() for o, t in zip(fn(fin, din, c1, v2v, 16, 16, 24, 0.5, 2.0, (h, h, h), True), tg))
    ref.backward()
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = _free_port()
    procs = [ctx.Process(target=_ray_bwd_worker, args=(r, 2, port, q, mode)) for r in range(2)]
    for p in procs:
        p.start()
    res = dict(q.get(timeout=240) for _ in procs)
    for p in procs:
        p.join(timeout=60)
        assert p.exitcode == 0
    close = lambda a, b: float(abs(torch.from_numpy(a) - b).max()) <= 2e-5 * max(1.0, float(b.abs().max()))
    for r in (0, 1):
        assert abs(res[r][0] - float(ref.detach())) < 1e-4 * max(1.0, abs(float(ref.detach())))
    if mode == "all":
        for r in (0, 1):
            assert close(res[r][1], f1.grad) and close(res[r][2], d1.grad) and close(res[r][3], c1.grad), r
    else:
        assert close(res[0][1], p1.grad) and float(abs(res[1][1]).max()) == 0.0
        assert close(res[0][2] + res[1][2], c1.grad)
