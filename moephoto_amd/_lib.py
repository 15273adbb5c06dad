"""ctypes binding of libmoephoto_amd.so (include/moephoto_amd.h).

There is deliberately NO fallback: if the HIP library is missing or no device is visible the
product path raises.  (The CPU oracle under oracle/ is test infrastructure and is never imported
from here.)
"""
import ctypes
import os

HERE = os.path.dirname(os.path.abspath(__file__))
LIB_PATH = os.path.join(HERE, 'libmoephoto_amd.so')

OK, EINVAL, ENOMEM, EHIP, ESTATE = 0, -1, -2, -3, -4
ARCH_NET2X, ARCH_NET3X, ARCH_NET4X, ARCH_NETDN, ARCH_SEDN, ARCH_LITE = range(6)
F32, F16, U8, U16 = range(4)
PREC_FP16, PREC_FP16X3, PREC_DEBUG_DIRECT, PREC_MIXED, PREC_AUTO = range(5)
ABI_VERSION = 4
FWD_INPUT_SINCE_PREV = 1      # moe_net_forward_ex flag (include/moephoto_amd.h)
RESIZE_MODES = {'nearest': 0, 'bilinear': 1, 'bicubic': 2}
PRECISIONS = {'fp16': PREC_FP16, 'fp16x3': PREC_FP16X3, 'debug_direct': PREC_DEBUG_DIRECT, 'mixed': PREC_MIXED}       # ('auto' = PREC_AUTO is resolved by the library)

_lib = None


class EngineError(RuntimeError):
    pass


def lib():
    """Load (once) and return the C-ABI library.  Raises if it has not been built."""
    global _lib
    if _lib is not None:
        return _lib
    if not os.path.exists(LIB_PATH):
        raise EngineError('{} not found: build it with `python -m moephoto_amd.build` (hipcc, gfx950). '
                          'moephoto_amd has no CPU path.'.format(LIB_PATH))
    # torch first: its wheel bundles its own libamdhip64 (+ HSA runtime).  Loaded AFTER this library -- which then has pulled in the system ROCm's copy under the same
    # soname -- torch binds to that copy and finds "No HIP GPUs" (seen with `python __graft_entry__.py smoke`: build() loads the library, then smoke() imports torch).
    # One process, one HIP runtime: whichever torch ships.
    try:
        import torch  # noqa: F401
    except ImportError:
        pass
    L = ctypes.CDLL(LIB_PATH)
    c_int, c_i64, c_vp, c_dbl = ctypes.c_int, ctypes.c_int64, ctypes.c_void_p, ctypes.c_double
    P = ctypes.POINTER
    sig = {
        'moe_last_error': (ctypes.c_char_p, []),
        'moe_abi_version': (c_int, []),
        'moe_device_count': (c_int, []),
        'moe_net_create': (c_int, [c_int, c_int, P(c_vp)]),
        'moe_net_destroy': (None, [c_vp]),
        'moe_net_scale': (c_int, [c_vp]),
        'moe_net_num_params': (c_int, [c_vp]),
        'moe_net_param_info': (c_int, [c_vp, c_int, P(ctypes.c_char_p), P(c_i64), P(c_int)]),
        'moe_net_set_param': (c_int, [c_vp, ctypes.c_char_p, c_vp, P(c_i64), c_int]),
        'moe_net_finalize': (c_int, [c_vp, c_int, c_int]),
        'moe_net_resolved_precision': (c_int, [c_vp, c_int]),
        'moe_net_calibrate': (c_int, [c_vp, c_dbl, P(c_int), P(c_dbl), c_vp]),
        'moe_net_exact_blocks': (c_int, [c_vp]),
        'moe_net_workspace_bytes': (c_i64, [c_vp, c_int, c_int, c_int]),
        'moe_net_max_tile_pixels': (c_i64, [c_vp]),
        'moe_net_forward': (c_int, [c_vp, c_vp, c_int, c_int, c_int, c_int, c_i64, c_i64, c_i64, c_vp, c_vp, c_int, c_vp, c_vp]),
        'moe_net_forward_ex': (c_int, [c_vp, c_vp, c_int, c_int, c_int, c_int, c_i64, c_i64, c_i64, c_vp, c_vp, c_int, c_vp, c_vp, ctypes.c_uint]),
        'moe_net_set_profile': (c_int, [c_vp, ctypes.c_char_p]),
        'moe_net_get_profile': (c_int, [c_vp, P(c_dbl), P(c_i64), P(c_dbl)]),
        'moe_net_get_profile_at': (c_int, [c_vp, c_int, P(c_dbl), P(c_i64), P(c_dbl)]),
        'moe_net_set_exact_blocks': (c_int, [c_vp, c_int]),
        'moe_net_set_debug': (c_int, [c_vp, c_int]),
        'moe_net_set_option': (c_int, [c_vp, ctypes.c_char_p, ctypes.c_char_p]),
        'moe_device_info': (c_int, [c_int, P(c_i64)]),
        'moe_blend_tile': (c_int, [c_vp, c_i64, c_i64, c_vp, c_i64, c_i64, c_int, c_int, c_int, c_int, c_int, c_int, c_int, c_int, c_int, c_vp, c_vp]),
        'moe_stitch_dev': (c_int, [c_vp, c_int, c_vp, c_vp, c_int, c_vp, c_int, c_vp]),
        'moe_stitch_band': (c_int, [c_vp, c_int, c_vp, c_vp, c_int, c_vp, c_int, c_int, c_int, c_int, c_vp]),
        'moe_net_debug_tap': (c_i64, [c_vp, ctypes.c_char_p, c_vp, c_i64, P(c_i64), c_vp]),
        'moe_plan_create': (c_int, [P(c_i64), c_dbl, c_dbl, c_int, c_int, c_int, c_int, P(c_vp)]),
        'moe_plan_destroy': (None, [c_vp]),
        'moe_plan_info': (c_int, [c_vp, P(c_i64)]),
        'moe_plan_tiles': (c_int, [c_vp, P(ctypes.c_int32)]),
        'moe_plan_ramp': (c_int, [c_vp, P(ctypes.c_float)]),
        'moe_plan_rows': (c_int, [c_vp, P(ctypes.c_int32)]),
        'moe_plan_seams': (c_int, [c_vp, P(ctypes.c_int32)]),
        'moe_wire_words': (c_i64, [c_vp]),
        'moe_wire_pack': (c_int, [c_vp, c_vp, c_vp, c_int, c_i64, c_vp]),
        'moe_wire_unpack': (c_int, [c_vp, c_vp, c_vp, c_int, c_i64, c_vp]),
        'moe_plan_pool_elems': (c_i64, [c_vp, c_int]),
        'moe_plan_tile_offsets': (c_int, [c_vp, c_int, P(c_i64)]),
        'moe_stitch': (c_int, [c_vp, c_int, c_vp, c_vp, c_int, c_vp, c_int, c_vp]),
        'moe_run_plan': (c_int, [c_vp, c_vp, c_vp, c_int, c_i64, c_i64, c_i64, c_vp, c_int, c_int, c_vp]),
        'moe_run_plan_ex': (c_int, [c_vp, c_vp, c_vp, c_int, c_i64, c_i64, c_i64, c_vp, c_int, c_int, c_vp, c_int, c_int, c_int, c_vp]),
        'moe_run_plan_frames': (c_int, [c_vp, c_vp, c_vp, c_int, c_i64, c_i64, c_i64, c_i64, c_int, c_vp, c_i64, c_int, c_int, c_int, c_vp]),
        'moe_run_plan_tiles': (c_int, [c_vp, c_vp, c_vp, c_int, c_i64, c_i64, c_i64, c_i64, c_int, c_vp, P(c_i64), c_int, c_vp]),
        'moe_to_float': (c_int, [c_vp, c_int, c_int, c_int, c_int, c_int, c_vp, c_int, c_int, c_vp]),
        'moe_to_output': (c_int, [c_vp, c_int, c_int, c_int, c_int, c_int, c_vp, c_int, c_int, c_vp]),
        'moe_resize': (c_int, [c_vp, c_vp, c_int, c_int, c_int, c_int, c_int, c_int, c_int, c_int, c_vp]),
    }
    for name, (res, args) in sig.items():
        fn = getattr(L, name)   # AttributeError here = header/library mismatch: fail loudly
        fn.restype, fn.argtypes = res, args
    if L.moe_abi_version() != ABI_VERSION:
        raise EngineError('libmoephoto_amd.so ABI version {} != {}: rebuild it (python -m moephoto_amd.build)'.format(L.moe_abi_version(), ABI_VERSION))
    _lib = L
    return L


EXPORTS = ['moe_last_error', 'moe_abi_version', 'moe_device_count', 'moe_net_create', 'moe_net_destroy', 'moe_net_scale',
           'moe_net_num_params', 'moe_net_param_info', 'moe_net_set_param', 'moe_net_finalize', 'moe_net_resolved_precision', 'moe_net_calibrate', 'moe_net_exact_blocks', 'moe_net_workspace_bytes',
           'moe_net_max_tile_pixels', 'moe_net_forward', 'moe_net_forward_ex', 'moe_net_set_profile', 'moe_net_get_profile', 'moe_net_get_profile_at', 'moe_net_set_exact_blocks', 'moe_net_set_debug', 'moe_net_set_option', 'moe_device_info', 'moe_blend_tile', 'moe_stitch_dev', 'moe_stitch_band', 'moe_net_debug_tap', 'moe_plan_create', 'moe_plan_destroy', 'moe_plan_info',
           'moe_plan_tiles', 'moe_plan_ramp', 'moe_plan_rows', 'moe_plan_seams', 'moe_wire_words', 'moe_wire_pack', 'moe_wire_unpack', 'moe_plan_pool_elems', 'moe_plan_tile_offsets', 'moe_stitch', 'moe_run_plan',
           'moe_run_plan_ex', 'moe_run_plan_frames', 'moe_run_plan_tiles', 'moe_to_float', 'moe_to_output', 'moe_resize']


def check(rc):
    """Translate a C status into the exception the reference's callers expect
    (worker.enhance catches everything and reports the traceback: python/worker.py:52-74;
    MemoryError is the planner's "tile does not fit" convention: python/imageProcess.py:58-59)."""
    if rc is not None and rc < 0:
        msg = lib().moe_last_error().decode('utf-8', 'replace')
        if rc == ENOMEM:
            raise MemoryError(msg)
        raise EngineError(msg)
    return rc


def device_info(device=0):
    """dict of the device properties the roofline figures are derived from (moe_device_info)."""
    info = (ctypes.c_int64 * 8)()
    check(lib().moe_device_info(int(device), info))
    keys = ('compute_units', 'clock_khz', 'mem_clock_khz', 'mem_bus_bits', 'l2_bytes', 'total_mem_bytes', 'wall_clock_khz', 'lds_bytes_per_cu')
    return dict(zip(keys, [int(v) for v in info]))


def require_device():
    n = lib().moe_device_count()
    if n < 1:
        raise EngineError('no HIP device visible: moephoto_amd runs on MI355X (gfx950) only and has no CPU path')
    return n
