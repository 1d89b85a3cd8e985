"""Rank launch: `python bench.py --gpus N` starts its own N ranks (one process per GPU), rank environment, NUMA pinning."""
import os
import sys

import torch

from benchkit.common import BENCH_PY


def self_launch(args):
    """`python bench.py --gpus N` without a torchrun environment: re-exec under torch.distributed.run with N ranks on this node
    (one process per GPU; rendezvous on 127.0.0.1). Returns only in the children / for N = 1."""
    if args.gpus <= 1 or "WORLD_SIZE" in os.environ:
        return
    import socket
    import subprocess
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    port = s.getsockname()[1]
    s.close()
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node=%d" % args.gpus, "--master-addr", "127.0.0.1",
           "--master-port", str(port), BENCH_PY] + sys.argv[1:]
    sys.exit(subprocess.call(cmd, env=rank_env(os.environ)))


def rank_env(base):
    """Environment of the ranks: dmabuf IPC for RCCL across processes on this driver, a bounded OpenMP pool per rank, RCCL warnings on."""
    env = dict(base)
    env.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
    env.setdefault("OMP_NUM_THREADS", "8")
    env.setdefault("NCCL_DEBUG", "WARN")
    return env


def pin_rank_to_gpu_numa(dev):
    """Bind this rank's host threads to the CPUs of its GPU's NUMA node (PCI bus id -> /sys/bus/pci/devices/<bdf>/numa_node ->
    /sys/devices/system/node/nodeN/cpulist): launch latency and pinned-memory copies stay on the GPU's socket. Best effort: returns a
    description for the JSON line, never raises."""
    try:
        prop = torch.cuda.get_device_properties(dev)
        bdf = "%04x:%02x:%02x.0" % (getattr(prop, "pci_domain_id", 0), prop.pci_bus_id, prop.pci_device_id)
        node = int(open("/sys/bus/pci/devices/%s/numa_node" % bdf).read().strip())
        if node < 0:
            return {"pci": bdf, "numa_node": node, "pinned": False, "reason": "no NUMA information"}
        cpus = set()
        for part in open("/sys/devices/system/node/node%d/cpulist" % node).read().strip().split(","):
            lo, _, hi = part.partition("-")
            cpus.update(range(int(lo), int(hi or lo) + 1))
        cpus &= os.sched_getaffinity(0)
        if not cpus:
            return {"pci": bdf, "numa_node": node, "pinned": False, "reason": "node CPUs outside the allowed set"}
        os.sched_setaffinity(0, cpus)
        return {"pci": bdf, "numa_node": node, "pinned": True, "cpus": len(cpus)}
    except Exception as e:                                        # containers without sysfs, exotic topologies: run unpinned
        return {"pinned": False, "reason": repr(e)[:120]}
