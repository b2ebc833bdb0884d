"""Loader for the host-fiber emulation build of the kernel sources (TEST INFRASTRUCTURE).

``tests/hipemu/_build/libdensereg_emu.so`` is ``densereg_amd/csrc`` compiled with ``-DDR_EMU``
(see tests/hipemu/hip_emu.h): the same kernels, executed thread-by-thread on CPU fibers, behind the
same C ABI.  "Device" pointers are host pointers.  Only CPU-side tests use it; the product loader
(``densereg_amd._lib.load``) refuses anything but the HIP build.
"""
from __future__ import annotations

import ctypes as C
import os
import subprocess

import numpy as np

from densereg_amd import _lib

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
EMU_SO = os.path.join(ROOT, 'tests', 'hipemu', '_build', 'libdensereg_emu.so')

_emu = None


def _stale() -> bool:
    if not os.path.exists(EMU_SO):
        return True
    t = os.path.getmtime(EMU_SO)
    srcs = [os.path.join(ROOT, 'tests', 'hipemu', f) for f in ('hip_emu.h', 'hip_emu.cpp')]
    d = os.path.join(ROOT, 'densereg_amd', 'csrc')
    srcs += [os.path.join(d, f) for f in os.listdir(d)]
    srcs += [os.path.join(ROOT, 'include', f) for f in os.listdir(os.path.join(ROOT, 'include'))]
    return any(os.path.getmtime(s) > t for s in srcs)


def load_emu() -> C.CDLL:
    global _emu
    if _emu is None:
        if _stale():
            subprocess.check_call([os.path.join(ROOT, 'build.sh'), '--emu'], cwd=ROOT)
        _emu = _lib.bind(C.CDLL(EMU_SO), debug=True)
        assert _emu.dr_backend() == b'hipemu'
    return _emu


def ptr(a):
    """address of a numpy array (or None)."""
    if a is None:
        return None
    assert a.flags['C_CONTIGUOUS']
    return a.ctypes.data


def dbg_conv2d(lib, x, w, scale=None, shift=None, relu=False, res=None, rowmask=None, thresh=0.0, want_stats=False,
               x_cs=None, y_cs=None):
    """x (B,H,W,Cin) float32 host; runs dr_dbg_conv2d; returns y (B,H,W,Cout) [, stats (2,Cout)]."""
    B, H, W, Cin = x.shape
    k, _, _, Cout = w.shape
    x_cs = x_cs or -(-Cin // 4) * 4
    y_cs = y_cs or Cout
    xp = np.full((B, H, W, x_cs), np.nan, np.float32)      # NaN in pad channels: the kernel must not read them
    xp[..., :Cin] = x
    yp = np.full((B, H, W, y_cs), -777.0, np.float32)
    stat = np.zeros((2, Cout), np.float64) if want_stats else None
    resp = None
    if res is not None:
        resp = np.ascontiguousarray(res, np.float32)
    rc = lib.dr_dbg_conv2d(B, H, W, Cin, Cout, k, ptr(xp), x_cs, ptr(np.ascontiguousarray(w, np.float32)),
                           ptr(scale), ptr(shift), int(relu), ptr(resp), 0 if res is None else res.shape[-1],
                           ptr(rowmask), thresh, ptr(yp), y_cs, ptr(stat), None)
    assert rc == 0, rc
    assert np.all(yp[..., Cout:] == -777.0), 'kernel wrote outside its channel range'
    y = yp[..., :Cout].copy()
    return (y, stat) if want_stats else y
