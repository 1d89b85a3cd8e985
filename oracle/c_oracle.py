"""TEST INFRASTRUCTURE — ctypes wrapper of oracle/c/forge_oracle.c (plain-C restatement of rotate/render)."""
import ctypes
import os
import subprocess

import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))
_SO = os.path.join(_HERE, "_build", "liboracle_c.so")


def lib():
    if not os.path.exists(_SO) or os.path.getmtime(_SO) < os.path.getmtime(os.path.join(_HERE, "c", "forge_oracle.c")):
        subprocess.check_call(["make", "-s", "-C", os.path.join(_HERE, "c")])
    return ctypes.CDLL(_SO)


def _p(a):
    return a.ctypes.data_as(ctypes.c_void_p)


def rotate(vox, T, mode, e):
    """vox [n,C,D,H,W] float32, T [n,4,4], mode [n] -> out"""
    vox = np.ascontiguousarray(vox, np.float32)
    T = np.ascontiguousarray(T, np.float32)
    mode = np.ascontiguousarray(mode, np.int32)
    out = np.empty_like(vox)
    n, C, D, H, W = vox.shape
    lib().oracle_rotate(_p(vox), _p(T), _p(mode), _p(out), n, C, D, H, W, ctypes.c_float(e))
    return out


def render(feat, dens, R, T, Kh, Hr, Wr, S, zmin, zmax, half):
    """feat [V,C,D,H,W], dens [V,1,D,H,W], R [V,3,3], T [V,3], Kh [V,3,3] half-res -> [V,Hr,Wr,C+2]"""
    feat = np.ascontiguousarray(feat, np.float32)
    dens = np.ascontiguousarray(dens, np.float32)
    R = np.ascontiguousarray(R, np.float32)
    T = np.ascontiguousarray(T, np.float32)
    k4 = np.ascontiguousarray(np.stack([Kh[:, 0, 0], Kh[:, 1, 1], Kh[:, 0, 2], Kh[:, 1, 2]], axis=1), np.float32)
    V, C, D, H, W = feat.shape
    out = np.empty((V, Hr, Wr, C + 2), np.float32)
    f = ctypes.c_float
    lib().oracle_render(_p(feat), _p(dens), _p(R), _p(T), _p(k4), _p(out), V, C, D, H, W, Hr, Wr, S,
                        f(zmin), f(zmax), f(half[0]), f(half[1]), f(half[2]))
    return out
