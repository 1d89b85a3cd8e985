"""Build libforge_hip.so (the C-ABI of include/forge_hip.h) with hipcc for gfx950, in-tree.

    python -m forge_amd.build [--force]

The .so is git-ignored but travels to the GPU box with the repo snapshot. There is exactly one
target architecture (MI355X / gfx950): no multi-arch fat binaries, no CUDA shims.
"""
import os
import shutil
import subprocess
import sys

HERE = os.path.dirname(os.path.abspath(__file__))
CSRC = os.path.join(HERE, "csrc")
LIB = os.path.join(HERE, "libforge_hip.so")
ARCH = "gfx950"


def sources():
    return sorted(os.path.join(CSRC, f) for f in os.listdir(CSRC) if f.endswith((".hip", ".cpp")))


def _deps():
    d = sources() + [os.path.join(CSRC, f) for f in os.listdir(CSRC) if f.endswith(".h")]
    d.append(os.path.join(os.path.dirname(HERE), "include", "forge_hip.h"))
    return d


def up_to_date():
    if not os.path.exists(LIB):
        return False
    t = os.path.getmtime(LIB)
    return all(os.path.getmtime(p) <= t for p in _deps())


def hipcc():
    for c in (os.environ.get("HIPCC"), shutil.which("hipcc"), "/opt/rocm/bin/hipcc"):
        if c and os.path.exists(c):
            return c
    raise RuntimeError("hipcc not found (ROCm toolchain is required to build libforge_hip.so)")


def build(force=False, verbose=True):
    if not force and up_to_date():
        return LIB
    objdir = os.path.join(HERE, "csrc", "_obj")
    os.makedirs(objdir, exist_ok=True)
    cc = hipcc()
    flags = ["--offload-arch=" + ARCH, "-O3", "-std=c++17", "-fPIC", "-munsafe-fp-atomics",
             "-Wall", "-Wno-unused-function", "-DNDEBUG"]
    objs = []
    procs = []
    for src in sources():
        obj = os.path.join(objdir, os.path.basename(src) + ".o")
        objs.append(obj)
        if not force and os.path.exists(obj) and all(os.path.getmtime(p) <= os.path.getmtime(obj)
                                                     for p in _deps() if not p.endswith((".hip", ".cpp")) or p == src):
            continue
        cmd = [cc] + flags + ["-x", "hip", "-c", src, "-o", obj]
        if verbose:
            print("[forge_amd.build]", " ".join(cmd), flush=True)
        procs.append((src, subprocess.Popen(cmd, stdout=subprocess.PIPE, stderr=subprocess.STDOUT, text=True)))
    for src, p in procs:
        out, _ = p.communicate()
        if p.returncode != 0:
            raise RuntimeError("hipcc failed on %s:\n%s" % (src, out))
        if verbose and out.strip():
            print(out)
    cmd = [cc, "--offload-arch=" + ARCH, "-shared", "-fPIC", "-o", LIB] + objs
    if verbose:
        print("[forge_amd.build]", " ".join(cmd), flush=True)
    subprocess.check_call(cmd)
    return LIB


if __name__ == "__main__":
    print(build(force="--force" in sys.argv))
