"""`--dry-run`: the launch / rendezvous / reduction skeleton of bench.py on CPU over gloo (tests/test_dist_cpu.py)."""
import time

import torch

from forge_amd import dist as fdist
from benchkit.common import V_OUT, _bracketed
from benchkit.emit import emit
from benchkit.multirank import multi_rank_records


def dry_run(args, rank, world):
    """`--dry-run`: the launch / rendezvous / timing-reduction skeleton of this entry point on CPU over gloo, with a token CPU workload
    instead of the HIP step (tests/test_dist_cpu.py runs `python bench.py --gpus 8 --dry-run` here, where there is no GPU). With --train the
    token workload is a DistributedDataParallel step (bucketed gradient all-reduce over gloo), as the real --train mode wraps the model."""
    fdist.init(backend="gloo")
    fdist.barrier()
    ddp = opt = None
    if args.train:
        torch.manual_seed(0)
        net = torch.nn.Sequential(torch.nn.Linear(64, 64), torch.nn.ReLU(), torch.nn.Linear(64, 8))
        ddp = torch.nn.parallel.DistributedDataParallel(net) if world > 1 else net
        opt = torch.optim.Adam(net.parameters(), lr=1e-3)
    ok, err = 1.0, None
    t0 = time.perf_counter()
    acc = 0.0
    try:
        for _ in range(args.steps):
            if ddp is not None:
                opt.zero_grad()
                loss = ddp(torch.full((4, 64), 1.0 + rank)).square().mean()
                loss.backward()
                opt.step()
                acc += float(loss)
            else:
                acc += float(torch.ones(64, 64).sum())
    except Exception as e:                                        # a failing rank still joins the reductions below
        ok, err = 0.0, repr(e)
    fdist.barrier()
    dt = fdist.all_reduce_scalars([time.perf_counter() - t0], "cpu", "max")[0]
    units, ranks_ok = fdist.all_reduce_scalars([float(args.scenes * (10 if args.train else V_OUT) * args.steps), ok], "cpu", "sum")
    same = None
    if ddp is not None and world > 1:                             # DDP keeps the replicas identical: the parameter checksum agrees on all ranks
        chk = float(sum(p.detach().double().sum() for p in ddp.parameters()))
        lo, hi = fdist.all_reduce_scalars([chk], "cpu", "min")[0], fdist.all_reduce_scalars([chk], "cpu", "max")[0]
        same = abs(hi - lo) < 1e-9 * max(1.0, abs(hi))
    multi = None
    if world > 1 and not args.train:
        # the sub-record skeleton of the real multi-rank line (multi_rank_records: watchdog, per-record try block, error gathering) with token
        # workloads: a DDP step timed with and without no_sync(), a record that FAILS on the last rank (reported, the others carry on), and the
        # differentiable ray-sharded render (all_gather forward, all-reduce backward) on a toy render function
        def token_ddp():
            torch.manual_seed(0)
            net = torch.nn.Sequential(torch.nn.Linear(64, 64), torch.nn.ReLU(), torch.nn.Linear(64, 8))
            dd = torch.nn.parallel.DistributedDataParallel(net)
            o = torch.optim.Adam(net.parameters(), lr=1e-3)

            def st():
                o.zero_grad()
                dd(torch.full((4, 64), 1.0 + rank)).square().mean().backward()
                o.step()

            def st_ns():
                with dd.no_sync():
                    st()
            a, b_ = _bracketed(st, 2, 1, "cpu"), _bracketed(st_ns, 2, 1, "cpu")
            return {"ms_per_step": a * 1e3, "ms_per_step_no_sync": b_ * 1e3,
                    "gradient_bytes_all_reduced_per_step": sum(p.numel() for p in net.parameters()) * 4}

        def token_fail():
            if rank == world - 1:
                raise RuntimeError("rehearsed failure on rank %d" % rank)
            return {"ms_per_step": 0.0}

        def token_rays():
            Hr = 2 * world
            feat = torch.ones(1, 2, 2, 2, 2, requires_grad=True)
            dens = torch.ones(1, 1, 2, 2, 2, requires_grad=True)
            cam = torch.zeros(3, 16)
            # noqa: E731
            toy = lambda f, d, c, v2v, hr, wr, *a: (f.sum() * torch.ones(3, 2, hr, wr) + c[:, 15].reshape(3, 1, 1, 1), d.sum() * torch.ones(3, 1, hr, wr))
            o = fdist.render_rays_sharded(feat, dens, cam, None, Hr, 4, 8, 0.5, 2.0, (1.0, 1.0, 1.0), render_fn=toy)
            (o[0].sum() + o[1].sum()).backward()
            return {"rows": int(o[0].shape[2]), "d_feat": float(feat.grad.sum()), "expected_d_feat": float(3 * 2 * Hr * 4 * 16)}
        def token_hang():
            if rank == world - 1:
                time.sleep(3600)                                         # a rank stuck for ever; the others block in the next collective
            return {}
        recs = (("ddp_train", token_ddp), ("failing_record", token_fail), ("ray_sharded_joint", token_rays))
        multi = multi_rank_records(args, rank, world, "cpu", {"metric": "dry run", "value": None, "n_gpus": world, "dry_run": True},
                                   records=(("hang", token_hang),) if args.rehearse_hang else recs)
    if rank == 0:
        emit({"metric": "rendered views/sec (5 views, 128^2 px, 64^3 voxel)", "value": None, "unit": "views/s", "n_gpus": world, "multi_rank": multi,
              "steps": args.steps, "warmup": args.warmup, "dry_run": True, "views_counted": units, "ms_per_step": dt / args.steps * 1e3,
              "scaling": "weak", "ranks_ok": int(ranks_ok), "process_group": fdist.group_info(), "train": bool(args.train), "replicas_identical": same,
                      "error": err,
              "config": {"workload": "dry run: no HIP work, launch + rendezvous + reductions only"}}, args.full_record)
    fdist.barrier()
    fdist.shutdown()
