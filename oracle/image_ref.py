"""TEST INFRASTRUCTURE ONLY - numpy front-end of the plain-C oracle (oracle/c/vlg_oracle.c).

Loads oracle/_build/libvlg_oracle.so (built by __graft_entry__.build() or `make -C oracle/c`).
Pinned against tests/golden/*.npz, which the reference itself produced (oracle/make_golden.py).
Only tests/ import this module.
"""
import ctypes
import os
import subprocess

import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))
_SO = os.path.join(_HERE, "_build", "libvlg_oracle.so")
_lib = None

F32P = np.ctypeslib.ndpointer(dtype=np.float32, flags="C_CONTIGUOUS")
I64P = np.ctypeslib.ndpointer(dtype=np.int64, flags="C_CONTIGUOUS")


def lib():
    global _lib
    if _lib is None:
        if not os.path.exists(_SO):
            subprocess.check_call(["make"], cwd=os.path.join(_HERE, "c"))
        L = ctypes.CDLL(_SO)
        L.oracle_ce_nchw.restype = ctypes.c_double
        L.oracle_ce_nchw.argtypes = [F32P, I64P, F32P, ctypes.c_int, ctypes.c_int, ctypes.c_int64]
        L.oracle_l1_mean.restype = ctypes.c_double
        L.oracle_l1_mean.argtypes = [F32P, F32P, F32P, ctypes.c_int64]
        L.oracle_gradient_loss.restype = ctypes.c_double
        L.oracle_gradient_loss.argtypes = [F32P, F32P, F32P, ctypes.c_int, ctypes.c_int, ctypes.c_int]
        L.oracle_ssim_loss.restype = ctypes.c_double
        L.oracle_ssim_loss.argtypes = [F32P, F32P, F32P, ctypes.c_int, ctypes.c_int, ctypes.c_int, ctypes.c_int]
        L.oracle_adam_step.restype = None
        L.oracle_adam_step.argtypes = [F32P, F32P, F32P, F32P, ctypes.c_int64, ctypes.c_int, ctypes.c_double,
                                       ctypes.c_double, ctypes.c_double, ctypes.c_double]
        L.oracle_prep_input.restype = None
        L.oracle_prep_input.argtypes = [F32P] * 7 + [I64P, F32P, F32P, I64P] + [ctypes.c_int] * 4
        _lib = L
    return _lib


def _c(a, dt=np.float32):
    return np.ascontiguousarray(a, dtype=dt)


def ce_nchw(logits, target):
    logits, target = _c(logits), _c(target, np.int64)
    b, C = logits.shape[:2]
    g = np.empty_like(logits)
    v = lib().oracle_ce_nchw(logits, target, g, b, C, int(np.prod(logits.shape[2:])))
    return v, g


def l1_mean(a, b):
    a, b = _c(a), _c(b)
    g = np.empty_like(a)
    return lib().oracle_l1_mean(a, b, g, a.size), g


def gradient_loss(a, b):
    a, b = _c(a), _c(b)
    g = np.empty_like(a)
    return lib().oracle_gradient_loss(a, b, g, a.shape[0] * a.shape[1], a.shape[2], a.shape[3]), g


def ssim_loss(x, y):
    x, y = _c(x), _c(y)
    g = np.empty_like(x)
    return lib().oracle_ssim_loss(x, y, g, *x.shape), g


def adam_step(p, g, m, v, step, lr=2e-4, beta1=0.5, beta2=0.999, eps=1e-8):
    lib().oracle_adam_step(p, _c(g), m, v, p.size, step, lr, beta1, beta2, eps)


def prep_input(e1, seg1, f1, f2, seg2, e2, f3, seg3, flip):
    b, _, H, W = f1.shape
    x10 = np.empty((b, 10, H, W), np.float32)
    f3o = np.empty((b, 3, H, W), np.float32)
    s3o = np.empty((b, H, W), np.int64)
    lib().oracle_prep_input(_c(e1), _c(seg1), _c(f1), _c(f2), _c(seg2), _c(e2), _c(f3), _c(seg3, np.int64), x10, f3o,
                            s3o, b, H, W, int(flip))
    return x10, f3o, s3o
