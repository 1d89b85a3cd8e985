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
