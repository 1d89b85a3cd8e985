#!/usr/bin/env python
"""Why the CPU baseline does not scale with processes: report the container's CPU budget (cgroup v2 cpu.max / v1 cfs quota, cpuset, affinity)
and MEASURE the effective parallelism - k single-thread integer spin loops of 1 s each, k = 1 ... 128: aggregate iterations / s relative
to one worker. A cgroup quota of Q cores shows up as a plateau at ~Q regardless of the 256 hardware threads the box lists."""
import multiprocessing as mp
import os
import time


def read(path):
    try:
        return open(path).read().strip()
    except Exception:
        return None


def spin(seconds, q):
    t0, n, x = time.perf_counter(), 0, 1
    while time.perf_counter() - t0 < seconds:
        for _ in range(20000):
            x = (x * 1103515245 + 12345) & 0x7fffffff
        n += 20000
    q.put(n)


def budget():
    info = {"cpu_count": os.cpu_count(), "affinity": len(os.sched_getaffinity(0)),
            "cgroup_v2_cpu_max": read("/sys/fs/cgroup/cpu.max"), "cgroup_v2_cpuset": read("/sys/fs/cgroup/cpuset.cpus.effective"),
            "cgroup_v1_quota_us": read("/sys/fs/cgroup/cpu/cpu.cfs_quota_us"), "cgroup_v1_period_us": read("/sys/fs/cgroup/cpu/cpu.cfs_period_us"),
            "cpu_stat": read("/sys/fs/cgroup/cpu.stat")}
    q = None
    if info["cgroup_v2_cpu_max"] and info["cgroup_v2_cpu_max"].split()[0] != "max":
        a, b = info["cgroup_v2_cpu_max"].split()
        q = float(a) / float(b)
    elif info["cgroup_v1_quota_us"] and int(info["cgroup_v1_quota_us"]) > 0:
        q = int(info["cgroup_v1_quota_us"]) / float(info["cgroup_v1_period_us"])
    info["quota_cores"] = q
    return info


def effective_parallelism(ks=(1, 4, 8, 16, 32, 64, 128), seconds=1.0):
    out = {}
    base = None
    ctx = mp.get_context("fork")
    for k in ks:
        q = ctx.Queue()
        ps = [ctx.Process(target=spin, args=(seconds, q)) for _ in range(k)]
        t0 = time.perf_counter()
        for p in ps:
            p.start()
        tot = sum(q.get() for _ in ps)
        for p in ps:
            p.join()
        rate = tot / (time.perf_counter() - t0)
        base = base or rate
        out[k] = round(rate / base, 2)
    return out


if __name__ == "__main__":
    import json
    b = budget()
    print(json.dumps(b, indent=1))
    print("effective parallelism (aggregate spin rate / one worker):", json.dumps(effective_parallelism()))
