"""GPU (-m gpu): the C-ABI driven by a plain C program (no Python, no torch in the process): tests/c_host/c_abi_smoke.c is compiled
with gcc (C11, no HIP compiler involved) against include/forge_hip.h + the HIP runtime API and linked to forge_amd/libforge_hip.so, then run. Proves the boundary of INTEGRATION.md §2."""
import os
import shutil
import subprocess

import pytest

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def test_c_host_program_builds_and_runs(tmp_path):
    gcc = shutil.which("gcc") or "gcc"
    rocm = os.environ.get("ROCM_PATH", "/opt/rocm")
    lib_dir = os.path.join(ROOT, "forge_amd")
    assert os.path.exists(os.path.join(lib_dir, "libforge_hip.so")), "build libforge_hip.so first (python -m forge_amd.build)"
    exe = str(tmp_path / "c_abi_smoke")
    cmd = [gcc, "-std=c11", "-O2", os.path.join(ROOT, "tests", "c_host", "c_abi_smoke.c"), "-I", os.path.join(rocm, "include"),
           "-I", os.path.join(ROOT, "include"), "-D__HIP_PLATFORM_AMD__", "-L", os.path.join(rocm, "lib"), "-lamdhip64", "-L", lib_dir,
           "-lforge_hip", "-Wl,-rpath," + lib_dir, "-Wl,-rpath," + os.path.join(rocm, "lib"), "-lm", "-o", exe]
    build = subprocess.run(cmd, capture_output=True, text=True, timeout=300)
    assert build.returncode == 0, build.stdout + build.stderr
    run = subprocess.run([exe], capture_output=True, text=True, timeout=120)
    assert run.returncode == 0, run.stdout + run.stderr
    assert "C host: rotate + render + conv_igemm (explicit plan) + resize + attention + error path OK" in run.stdout


def test_c_host_program_under_address_and_ub_sanitizers(tmp_path):
    """SURVEY.md section 5's sanitizer plan / VERDICT r4 weak item 15: the same plain-C host program against libforge_hip_asan.so - the library's HOST
    side (argument checks, launch planning, kernel-argument marshalling, the tap tables and plan caches of the entry points the program drives:
    rotate, ray-march, implicit-GEMM convolution with planned and explicit tiles, plan query, bilinear resize, the error paths) compiled with
    -fsanitize=address,undefined; the program itself is built with the same sanitizers. Any heap / stack / global overflow, use-after-free or
    undefined behaviour on the host side aborts the run. (Leak detection is off: the HIP runtime keeps process-lifetime allocations.)"""
    from forge_amd import build
    rocm = os.environ.get("ROCM_PATH", "/opt/rocm")
    clang = os.path.join(rocm, "lib", "llvm", "bin", "clang")
    if not os.path.exists(clang):
        pytest.skip("no ROCm clang on this box")
    try:
        lib = build.build_sanitized()                     # built lazily here (the product build must not depend on the sanitizer runtimes)
    except Exception as e:
        pytest.skip("sanitized library not buildable on this host: %r" % (e,))
    lib_dir = os.path.dirname(lib)
    rt = subprocess.run([clang, "-print-file-name=libclang_rt.asan-x86_64.so"], capture_output=True, text=True).stdout.strip()
    exe = str(tmp_path / "c_abi_smoke_asan")
    cmd = [clang, "-std=c11", "-O1", "-g", "-fsanitize=address,undefined", "-shared-libsan", "-fno-omit-frame-pointer",
           os.path.join(ROOT, "tests", "c_host", "c_abi_smoke.c"), "-I", os.path.join(rocm, "include"), "-I", os.path.join(ROOT, "include"),
           "-D__HIP_PLATFORM_AMD__", "-L", os.path.join(rocm, "lib"), "-lamdhip64", "-L", lib_dir, "-l:" + os.path.basename(lib),
           "-Wl,-rpath," + lib_dir, "-Wl,-rpath," + os.path.join(rocm, "lib"), "-Wl,-rpath," + os.path.dirname(rt), "-lm", "-o", exe]
    b = subprocess.run(cmd, capture_output=True, text=True, timeout=300)
    assert b.returncode == 0, b.stdout + b.stderr
    env = dict(os.environ, ASAN_OPTIONS="detect_leaks=0:protect_shadow_gap=0:abort_on_error=0:halt_on_error=1", UBSAN_OPTIONS="halt_on_error=1:print_stacktrace=1")
    run = subprocess.run([exe], capture_output=True, text=True, timeout=300, env=env)
    assert run.returncode == 0, run.stdout[-2000:] + run.stderr[-4000:]
    assert "ERROR: AddressSanitizer" not in run.stderr and "runtime error:" not in run.stderr, run.stderr[-4000:]
    assert "C host: rotate + render + conv_igemm (explicit plan) + resize + attention + error path OK" in run.stdout


def test_lds_dma_hardware_assumptions_probe(tmp_path):
    """The K loops of conv_igemm / conv_wgrad stage their operands with `buffer_load_dwordx4 ... offen lds` from inline assembly (csrc/common.h:
    lds_dma16). What they rely on - lane L of a wave lands at LDS byte M0 + 16 L, lanes whose offset is beyond the buffer write zeros, M0 beyond
    the first KB works - is checked by a stand-alone HIP program (tools/experiments/dma_probe) compiled here with hipcc for gfx950 and run."""
    hipcc = shutil.which("hipcc") or os.path.join(os.environ.get("ROCM_PATH", "/opt/rocm"), "bin", "hipcc")
    if not os.path.exists(hipcc):
        pytest.skip("no hipcc on this box")
    exe = str(tmp_path / "dma_probe")
    build = subprocess.run([hipcc, "--offload-arch=gfx950", "-O2", os.path.join(ROOT, "tools", "experiments", "dma_probe", "dma_probe.hip"), "-o", exe],
                           capture_output=True, text=True, timeout=600)
    assert build.returncode == 0, build.stdout + build.stderr
    run = subprocess.run([exe], capture_output=True, text=True, timeout=120)
    assert run.returncode == 0 and "dma_probe: OK" in run.stdout, run.stdout + run.stderr
