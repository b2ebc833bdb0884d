"""ctypes binding of ``libdensereg_hip.so`` (the C ABI of ``include/densereg.h``).

The product path has no CPU fallback: if the HIP library is missing this module raises at load
time with build instructions, and ``dr_create`` itself fails when no gfx950 device is present.
"""
from __future__ import annotations

import ctypes as C
import os

_HERE = os.path.dirname(os.path.abspath(__file__))
# DR_LIB_VARIANT=<name> (measurement only): load lib/variants/<name>/ instead -- the same sources built with other -D switches
# (tools/build_variants.sh), so that compile-time experiments can be A/B-ed in one GPU visit.  Unset = the product.
_LIBDIR = os.path.join(_HERE, 'lib', 'variants', os.environ['DR_LIB_VARIANT']) if os.environ.get('DR_LIB_VARIANT') else os.path.join(_HERE, 'lib')
LIB_PATH = os.path.join(_LIBDIR, 'libdensereg_hip.so')
DEBUG_LIB_PATH = os.path.join(_LIBDIR, 'libdensereg_hip_dbg.so')      # product sources + dr_dbg_* hooks (tests/, tools/)

DR_OK = 0
DROPOUT_OFF, DROPOUT_MASK, DROPOUT_RNG = 0, 1, 2


class DrConfig(C.Structure):
    _fields_ = [('num_stack', C.c_int32), ('num_fea', C.c_int32), ('num_jnt', C.c_int32), ('in_hw', C.c_int32),
                ('kernel_size', C.c_int32), ('max_batch', C.c_int32), ('device', C.c_int32), ('training', C.c_int32)]


class KernelStat(C.Structure):
    _fields_ = [('name', C.c_char * 64), ('launches', C.c_int64), ('total_ms', C.c_double), ('flops', C.c_double),
                ('bytes', C.c_double)]


class DbgBnArgs(C.Structure):
    """include/densereg_debug.h: dr_dbg_bn_args (test hook)."""
    _fields_ = [('B', C.c_int), ('H', C.c_int), ('W', C.c_int), ('Cin', C.c_int), ('Cout', C.c_int), ('k', C.c_int),
                ('x', C.c_void_p), ('x_cs', C.c_int), ('w', C.c_void_p),
                ('gamma', C.c_void_p), ('beta', C.c_void_p), ('mm', C.c_void_p), ('mv', C.c_void_p),
                ('r_max', C.c_float), ('d_max', C.c_float), ('relu', C.c_int),
                ('res', C.c_void_p), ('dout', C.c_void_p),
                ('gr', C.c_void_p), ('gr_cs', C.c_int), ('wr', C.c_void_p), ('kr', C.c_int), ('Cr', C.c_int),
                ('y', C.c_void_p), ('raw', C.c_void_p), ('bnc', C.c_void_p), ('mm_next', C.c_void_p), ('mv_next', C.c_void_p),
                ('dout_used', C.c_void_p), ('draw', C.c_void_p), ('dgamma', C.c_void_p), ('dbeta', C.c_void_p),
                ('dres', C.c_void_p), ('fwd_rows', C.c_int), ('bwd_rows', C.c_int)]


class DenseRegError(RuntimeError):
    def __init__(self, code, msg):
        super().__init__('densereg error %d: %s' % (code, msg))
        self.code = code


_vp, _fp, _i, _sz = C.c_void_p, C.c_void_p, C.c_int, C.c_size_t

# name -> (restype, argtypes); every symbol include/densereg.h declares
SIGNATURES = {
    'dr_abi_version': (_i, []),
    'dr_backend': (C.c_char_p, []),
    'dr_create': (_i, [C.POINTER(DrConfig), C.POINTER(_vp)]),
    'dr_destroy': (None, [_vp]),
    'dr_last_error': (C.c_char_p, [_vp]),
    'dr_param_count': (_i, [_vp]),
    'dr_param_info': (_i, [_vp, _i, C.POINTER(C.c_char_p), C.POINTER(C.c_int32 * 4), C.POINTER(C.c_int32),
                           C.POINTER(C.c_int32)]),
    'dr_load_param': (_i, [_vp, C.c_char_p, _fp, _sz]),
    'dr_read_param': (_i, [_vp, C.c_char_p, _fp, _sz]),
    'dr_finalize_params': (_i, [_vp, _vp]),
    'dr_norm_dm': (_i, [_vp, _i, _fp, _fp, _fp, _vp]),
    'dr_forward_eval': (_i, [_vp, _i, _fp, _fp, _fp, _fp, _vp]),
    'dr_read_maps': (_i, [_vp, _i, _i, _fp, _fp, _fp, _vp]),
    'dr_vote': (_i, [_vp, _i, _fp, _fp, _fp, _fp, _fp, _fp, _fp, _vp]),
    'dr_infer': (_i, [_vp, _i, _fp, _fp, _fp, _fp, _vp]),
    'dr_forward_train': (_i, [_vp, _i, _fp, _i, _vp, C.c_uint64, _vp]),
    'dr_loss': (_i, [_vp, _i, _fp, _fp, _fp, _fp, _fp, _vp]),
    'dr_backward': (_i, [_vp, _i, _vp]),
    'dr_zero_grad': (_i, [_vp, _vp]),
    'dr_set_pipeline': (_i, [_vp, _i]),
    'dr_set_groups': (_i, [_vp, _i]),
    'dr_sync_grads': (_i, [_vp, _vp]),
    'dr_flat_grad': (_i, [_vp, C.POINTER(_vp), C.POINTER(_sz)]),
    'dr_flat_param': (_i, [_vp, C.POINTER(_vp), C.POINTER(_sz)]),
    'dr_apply_adam': (_i, [_vp, C.c_float, C.c_float, C.c_float, C.c_int64, _vp]),
    'dr_read_activation': (_i, [_vp, C.c_char_p, _i, _fp, _sz]),
    'dr_conv_flops_per_crop': (C.c_double, [_vp]),
    'dr_set_precision': (_i, [_vp, _i]),
    'dr_set_fusion': (_i, [_vp, _i]),
    'dr_png_unfilter': (_i, [_vp, _i, _i, _i, _vp]),
    'dr_depth_from_samples': (_i, [_vp, C.c_long, _i, _vp, _vp]),
    'dr_flat_adam': (_i, [_vp, C.POINTER(_vp), C.POINTER(_vp), C.POINTER(C.c_size_t)]),
    'dr_crc32c': (C.c_uint32, [C.c_uint32, _vp, C.c_size_t]),
    'dr_crop_from_pose': (_i, [_i, _vp, _i, _i, _vp, _i, _vp, _i, C.c_float, _i, _vp, _vp, _vp, _vp]),
    'dr_crop_from_bbx': (_i, [_i, _vp, _i, _i, _vp, _vp, _i, _vp, _vp, _vp, _vp]),
    'dr_data_aug': (_i, [_i, _vp, _i, _i, _vp, _i, _vp, _vp, _vp, _vp, _vp, _vp]),
    'dr_profile_enable': (_i, [_vp, _i]),
    'dr_profile_read': (_i, [_vp, C.POINTER(KernelStat), _i, C.POINTER(_i)]),
    'dr_profile_detail': (_i, [_vp, C.POINTER(KernelStat), _i, C.POINTER(_i)]),
    'dr_lookback_expired': (_i, [_vp]),        # diagnostic of the opt-in BatchReNorm look-back (DR_BN_LOOKBACK=1)
}

# include/densereg_debug.h: test / micro-benchmark hooks.  NOT in the product library: libdensereg_hip_dbg.so (same sources
# + -DDR_DEBUG_HOOKS) and the host emulator export them; tests/ and tools/ load that library for these calls only.
DEBUG_SIGNATURES = {
    'dr_dbg_conv_bench': (_i, [_i, _i, _i, _i, _i, _i, _i, _i, _i, C.POINTER(C.c_float)]),
    'dr_dbg_force_tile': (_i, [_i]),
    'dr_dbg_force_bf16': (_i, [_i]),
    'dr_dbg_force_bf16_storage': (_i, [_i]),
    'dr_dbg_force_x3': (_i, [_i]),
    'dr_dbg_p3_launches': (C.c_long, []),
    'dr_dbg_x3h_launches': (C.c_long, []),
    'dr_dbg_bn_finalize_rows': (_i, [_i]),
    'dr_dbg_wgrad_bench': (_i, [_i, _i, _i, _i, _i, _i, _i, _i, _i, C.POINTER(C.c_float), C.POINTER(_i)]),
    'dr_dbg_bn_bench': (_i, [C.c_long, _i, _i, _i, C.POINTER(C.c_float)]),
    'dr_dbg_wgrad': (_i, [_i, _i, _i, _i, _i, _i, _vp, _i, _vp, _i, _vp, C.c_float, _i, _i, _vp, _vp]),
    'dr_dbg_mfma_peak': (_i, [_i, _i, _i, C.POINTER(C.c_float)]),
    'dr_dbg_bn_layer': (_i, [C.POINTER(DbgBnArgs), _vp]),
    'dr_dbg_maxpool': (_i, [_i, _i, _i, _i, _i, _vp, _vp, _vp, _vp, _i, _vp]),
    'dr_dbg_act_dgrad': (_i, [_i, _i, _i, _i, _i, _i, _vp, _vp, _i, _vp, C.c_float, _vp, _vp, _vp]),
    'dr_dbg_conv2d': (_i, [_i, _i, _i, _i, _i, _i, _fp, _i, _fp, _fp, _fp, _i, _fp, _i, _fp, C.c_float, _fp, _i, _fp, _vp]),
}


def bind(lib: C.CDLL, debug: bool = False) -> C.CDLL:
    sigs = dict(SIGNATURES, **DEBUG_SIGNATURES) if debug else SIGNATURES
    for name, (res, args) in sigs.items():
        fn = getattr(lib, name)          # AttributeError if the library misses a declared symbol
        fn.restype = res
        fn.argtypes = args
    return lib


_lib = None


def load() -> C.CDLL:
    """Load the product library; raise loudly when it has not been built."""
    global _lib
    if _lib is None:
        if not os.path.exists(LIB_PATH):
            raise ImportError(
                'densereg_amd: %s not found. Build it with `python -c "import __graft_entry__ as g; g.build()"` '
                '(hipcc --offload-arch=gfx950). There is no CPU fallback.' % LIB_PATH)
        _lib = bind(C.CDLL(LIB_PATH))
        if _lib.dr_backend() != b'hip-gfx950':
            raise ImportError('densereg_amd: %s is not the HIP build (backend=%r)' % (LIB_PATH, _lib.dr_backend()))
    return _lib


_dbg = None


def load_debug() -> C.CDLL:
    """The debug build (``libdensereg_hip_dbg.so``: the product's sources + the ``dr_dbg_*`` hooks of
    ``include/densereg_debug.h``).  Only tests/ and tools/ call this; nothing in the package does."""
    global _dbg
    if _dbg is None:
        if not os.path.exists(DEBUG_LIB_PATH):
            raise ImportError('densereg_amd: %s not found (./build.sh builds it next to the product library)' % DEBUG_LIB_PATH)
        _dbg = bind(C.CDLL(DEBUG_LIB_PATH), debug=True)
        if _dbg.dr_backend() != b'hip-gfx950':
            raise ImportError('densereg_amd: %s is not the HIP build' % DEBUG_LIB_PATH)
    return _dbg


class Handle:
    """Thin RAII wrapper over ``dr_handle*``; all pointer arguments are raw addresses (ints)."""

    def __init__(self, lib: C.CDLL, num_stack=2, num_fea=128, num_jnt=16, in_hw=128, kernel_size=3, max_batch=40,
                 device=0, training=False):
        self.lib = lib
        self.cfg = DrConfig(num_stack, num_fea, num_jnt, in_hw, kernel_size, max_batch, device, int(training))
        self._h = _vp()
        rc = lib.dr_create(C.byref(self.cfg), C.byref(self._h))
        if rc != DR_OK:
            raise DenseRegError(rc, (lib.dr_last_error(None) or b'').decode())

    def close(self):
        if self._h:
            self.lib.dr_destroy(self._h)
            self._h = _vp()

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass

    def check(self, rc):
        if rc != DR_OK:
            raise DenseRegError(rc, (self.lib.dr_last_error(self._h) or b'').decode())

    def call(self, fn_name, *args):
        self.check(getattr(self.lib, fn_name)(self._h, *args))

    # -- variables ---------------------------------------------------------------------------
    def param_infos(self):
        out = []
        for i in range(self.lib.dr_param_count(self._h)):
            name = C.c_char_p()
            dims = (C.c_int32 * 4)()
            nd, tr = C.c_int32(), C.c_int32()
            self.check(self.lib.dr_param_info(self._h, i, C.byref(name), C.byref(dims), C.byref(nd), C.byref(tr)))
            out.append((name.value.decode(), tuple(dims[:nd.value]), bool(tr.value)))
        return out

    def load_params(self, params):
        """params: dict TF-name -> numpy float32 array (host)."""
        import numpy as np
        for name, shape, _ in self.param_infos():
            a = np.ascontiguousarray(params[name], dtype=np.float32)
            assert tuple(a.shape) == tuple(shape), (name, a.shape, shape)
            self.call('dr_load_param', name.encode(), a.ctypes.data, a.size)

    def read_params(self):
        import numpy as np
        out = {}
        for name, shape, _ in self.param_infos():
            a = np.empty(shape, np.float32)
            self.call('dr_read_param', name.encode(), a.ctypes.data, a.size)
            out[name] = a
        return out

    def flat(self, which='param'):
        p, n = _vp(), _sz()
        self.call('dr_flat_' + which, C.byref(p), C.byref(n))
        return p.value, n.value

    def profile(self, on: bool):
        self.call('dr_profile_enable', int(on))

    def profile_read(self):
        arr = (KernelStat * 32)()
        n = _i()
        self.call('dr_profile_read', arr, 32, C.byref(n))
        return [dict(name=arr[i].name.decode(), launches=arr[i].launches, total_ms=arr[i].total_ms,
                     flops=arr[i].flops, bytes=arr[i].bytes) for i in range(n.value)]

    def profile_detail(self):
        arr = (KernelStat * 2048)()
        n = _i()
        self.call('dr_profile_detail', arr, 2048, C.byref(n))
        return [dict(name=arr[i].name.decode(), launches=arr[i].launches, total_ms=arr[i].total_ms,
                     flops=arr[i].flops, bytes=arr[i].bytes) for i in range(n.value)]

    def conv_flops_per_crop(self):
        return float(self.lib.dr_conv_flops_per_crop(self._h))
