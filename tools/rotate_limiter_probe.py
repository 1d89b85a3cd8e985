#!/usr/bin/env python
"""What limits rotate_fwd_kernel (0.40-0.47 of 8 TB/s in the step, L2->fabric traffic = algorithmic, wait_any 0.52)?

The SAME kernel (same instruction stream per output voxel: affine, three IEEE divisions, 8 x NQ 16-byte gathers, 2 stores) on a working set far
beyond the 256 MB Infinity Cache (a ring of source / destination sets), under transforms that change ONLY the source access pattern:
  copy      mode 0 for every volume: the kernel's own streaming copy (no taps)
  identity  A = I: the reference's 31/32 shrink - the 8 taps of neighbouring outputs are neighbouring rows, source walked in output order
  rot20y / rot45y / rot90y   rotation about y: the source footprint of a 16^3 output cube is a tilted cube (x rows of the output walk x-z diagonals)
  shift     identity + half a volume along x: half the outputs are out of the grid (zero stores, no loads)
If `identity` runs near the copy rate while the rotations do not, the limiter is the gather's locality in L2 / DRAM, not VALU issue; if all of them
sit at the same rate below the copy, it is the per-voxel arithmetic. ROTATE_PROBE_CASE=name restricts the run to one case (PMC passes)."""
import math
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch  # noqa: E402

from forge_amd import _lib  # noqa: E402

dev = torch.device("cuda:0")
lib, st = _lib.lib(), _lib.current_stream()
D, C, n = int(os.environ.get("ROTATE_PROBE_D", "32")), 128, 5
only = os.environ.get("ROTATE_PROBE_CASE")
iters = int(os.environ.get("ROTATE_PROBE_ITERS", "24"))
set_bytes = n * C * D ** 3 * 4 * 2
nbuf = max(2, min(12, -(-(1024 << 20) // set_bytes)))
srcs = [torch.randn(n, D, D, D, C, device=dev) for _ in range(nbuf)]
dsts = [torch.empty_like(srcs[0]) for _ in range(nbuf)]


def roty(deg):
    a = math.radians(deg)
    return [math.cos(a), 0, math.sin(a), 0.0, 0, 1, 0, 0.0, -math.sin(a), 0, math.cos(a), 0.0]


cases = {"copy": (None, 0), "identity": ([1, 0, 0, 0.0, 0, 1, 0, 0.0, 0, 0, 1, 0.0], 1), "rot20y": (roty(20), 1), "rot45y": (roty(45), 1),
         "rot90y": (roty(90), 1), "shift": ([1, 0, 0, 1.0, 0, 1, 0, 0.0, 0, 0, 1, 0.0], 1),
         "bench": ([1, 0, 0, 0.02, 0, 0.8, -0.6, 0, 0, 0.6, 0.8, 0.01], 1)}
print("rotate_fwd_kernel, %d volumes of %d^3 x %d channels per launch (%.0f MB read + written), ring of %d sets = %.0f MB (Infinity Cache 256 MB)"
      % (n, D, C, set_bytes / 1e6, nbuf, nbuf * set_bytes / 2 ** 20))
for name, (A, md) in cases.items():
    if only and name != only:
        continue
    xf = torch.tensor(A if A else cases["identity"][0], device=dev, dtype=torch.float32).repeat(n, 1).contiguous()
    mode = torch.full((n,), md, dtype=torch.int32, device=dev)
    k = [0]

    def run():
        i = k[0] % nbuf
        k[0] += 1
        _lib.check(lib.forge_rotate_fwd(_lib.ptr(srcs[i]), _lib.ptr(xf), _lib.ptr(mode), _lib.ptr(dsts[i]), n, C, D, D, D, st), "rotate")
    for _ in range(nbuf):
        run()
    torch.cuda.synchronize()
    a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    a.record()
    for _ in range(iters):
        run()
    b.record()
    torch.cuda.synchronize()
    ms = a.elapsed_time(b) / iters
    nz = (dsts[0] != 0).float().mean().item()
    print("%-9s %.1f us  %.2f TB/s algorithmic (%.2f of 8 TB/s)   non-zero outputs %.0f %%" % (name, ms * 1e3, set_bytes / ms / 1e9, set_bytes / ms / 1e9 / 8.0, 100 * nz))
