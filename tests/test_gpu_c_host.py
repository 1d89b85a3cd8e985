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
    assert "C host: rotate + render + conv_igemm (explicit plan) + resize + error path OK" in run.stdout
