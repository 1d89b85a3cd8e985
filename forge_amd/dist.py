"""Single-node data parallelism: one process per MI355X, torch.distributed over RCCL (backend
"nccl" IS RCCL on PyTorch-ROCm) — the only parallelism the reference has (SURVEY.md §2.2:
DDP + DistributedSampler, kubric_train_pose_3D.py:74,124,130; eval sharding by batch index,
kubric_eval.py:56). Scenes are independent, so the data path needs NO collective: scenes are
sharded across ranks and only scalar metrics (losses / SSE / counts / time) are all-reduced.
Gradient all-reduce in training is torch DDP's bucketed RCCL all-reduce, unchanged.
"""
import os

import torch
import torch.distributed as dist


def env_world():
    return int(os.environ.get("RANK", 0)), int(os.environ.get("LOCAL_RANK", 0)), int(os.environ.get("WORLD_SIZE", 1))


def init(backend=None):
    """Initialise the default process group from the torchrun environment (no-op for world size 1).
    Returns (rank, local_rank, world_size)."""
    rank, local_rank, world = env_world()
    if world > 1 and not dist.is_initialized():
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        os.environ.setdefault("MASTER_PORT", "29500")
        if backend is None:
            backend = "nccl" if torch.cuda.is_available() else "gloo"
        if backend == "nccl":
            torch.cuda.set_device(local_rank)
        dist.init_process_group(backend=backend, init_method="env://", rank=rank, world_size=world)
    return rank, local_rank, world


def shard_indices(n_items, rank, world):
    """Round-robin shard of scene indices (kubric_eval.py:56: `batch_idx % split_num == exp_id`)."""
    return list(range(rank, n_items, world))


def all_reduce_scalars(values, device, op="sum"):
    """All-reduce a small list of python floats (SSE, counts, seconds...). float64 on the wire."""
    t = torch.tensor(values, dtype=torch.float64, device=device)
    if dist.is_available() and dist.is_initialized() and dist.get_world_size() > 1:
        dist.all_reduce(t, op={"sum": dist.ReduceOp.SUM, "max": dist.ReduceOp.MAX, "min": dist.ReduceOp.MIN}[op])
    return t.tolist()


def barrier():
    if dist.is_available() and dist.is_initialized() and dist.get_world_size() > 1:
        dist.barrier()


def psnr_from_sse(sse, count):
    """10 log10(1 / MSE), data_range = 1 (utils/eval_utils.py:8-12)."""
    import math
    mse = sse / max(count, 1.0)
    return float("inf") if mse <= 0 else 10.0 * math.log10(1.0 / mse)
