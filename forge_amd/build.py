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


LIB_ASAN = os.path.join(HERE, "libforge_hip_asan.so")


def build_sanitized(verbose=False):
    """libforge_hip_asan.so: the same sources with the HOST side (argument checks, launch-plan code, kernel-argument marshalling of the 51 entry
    points) under AddressSanitizer + UndefinedBehaviorSanitizer (-fsanitize=address,undefined -fno-gpu-sanitize: device code is compiled as in the
    product). Test infrastructure (SURVEY.md section 5's sanitizer plan; tests/test_gpu_c_host.py runs the plain-C host program against it)."""
    deps = _deps()
    if os.path.exists(LIB_ASAN) and all(os.path.getmtime(p) <= os.path.getmtime(LIB_ASAN) for p in deps):
        return LIB_ASAN
    objdir = os.path.join(HERE, "csrc", "_obj_asan")
    os.makedirs(objdir, exist_ok=True)
    cc = hipcc()
    flags = ["--offload-arch=" + ARCH, "-O1", "-g", "-std=c++17", "-fPIC", "-munsafe-fp-atomics", "-DNDEBUG", "-fsanitize=address,undefined",
             "-fno-gpu-sanitize", "-fno-omit-frame-pointer", "-shared-libsan"]
    procs, objs = [], []
    for src in sources():
        obj = os.path.join(objdir, os.path.basename(src) + ".o")
        objs.append(obj)
        procs.append((src, subprocess.Popen([cc] + flags + ["-x", "hip", "-c", src, "-o", obj], stdout=subprocess.PIPE, stderr=subprocess.STDOUT, text=True)))
    for src, p in procs:
        out, _ = p.communicate()
        if p.returncode != 0:
            raise RuntimeError("hipcc (sanitized) failed on %s:\n%s" % (src, out))
    cmd = [cc, "--offload-arch=" + ARCH, "-shared", "-fPIC", "-fsanitize=address,undefined", "-fno-gpu-sanitize", "-shared-libsan", "-o", LIB_ASAN] + objs
    if verbose:
        print("[forge_amd.build]", " ".join(cmd), flush=True)
    subprocess.check_call(cmd)
    return LIB_ASAN


if __name__ == "__main__":
    print(build_sanitized(verbose=True) if "--asan" in sys.argv else build(force="--force" in sys.argv))
