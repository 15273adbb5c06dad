"""ctypes binding of oracle/convref.c (TEST INFRASTRUCTURE).  Built by oracle/Makefile
(`__graft_entry__.build()` runs it); built on first use if missing."""
import ctypes
import os
import subprocess

import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))
_SO = os.path.join(_HERE, '_build', 'libconvref.so')
_lib = None


def build(force=False):
    if force or not os.path.exists(_SO) or os.path.getmtime(_SO) < os.path.getmtime(os.path.join(_HERE, 'convref.c')):
        subprocess.check_call(['make', '-s', '-C', _HERE, '_build/libconvref.so'] + (['-B'] if force else []))
    return _SO


def lib():
    global _lib
    if _lib is None:
        _lib = ctypes.CDLL(build())
        fp = ctypes.POINTER(ctypes.c_float)
        _lib.moe_oracle_conv2d_f32.argtypes = [fp, fp, fp, fp] + [ctypes.c_int] * 7
        _lib.moe_oracle_conv2d_f32.restype = None
        _lib.moe_oracle_prelu_f32.argtypes = [fp, ctypes.c_int64, ctypes.c_float]
    return _lib


def _p(a):
    return a.ctypes.data_as(ctypes.POINTER(ctypes.c_float))


def conv2d(x, w, b=None, pad=1):
    x = np.ascontiguousarray(x, np.float32)
    w = np.ascontiguousarray(w, np.float32)
    B, C, H, W = x.shape
    O, C2, k, _ = w.shape
    assert C == C2
    out = np.empty((B, O, H + 2 * pad - k + 1, W + 2 * pad - k + 1), np.float32)
    bb = None if b is None else np.ascontiguousarray(b, np.float32)
    lib().moe_oracle_conv2d_f32(_p(x), _p(w), None if bb is None else _p(bb), _p(out), B, C, H, W, O, k, pad)
    return out
