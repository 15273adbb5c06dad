"""Engine-backed stand-ins for the reference's model classes -- the drop-in boundary.

The reference builds its nets in the "constructor slot" of the plugin tables
(python/runSR.py:10-24, python/runDN.py:10-21) and then does, in initModel
(python/imageProcess.py:319-334):

    m = ctor(); m.load_state_dict(sd); [p.requires_grad_(False) for p in m.parameters()]; m.eval()
    m = m.to(dtype=config.dtype(), device=config.device())        # castModel, :309-317
    y = m(x)                                                        # Option.__call__, :391-395

The classes here accept exactly that protocol, with the same names, and run the forward on the
HIP engine (libmoephoto_amd.so) on torch's current stream:

    Net2x / Net3x / Net4x   python/models.py:125-154     (a2/a3/a4, p2/p3/p4)
    NetDN                   python/models.py:158-164     (dn_lite5/10/15)
    SEDN                    python/models.py:215-224     (l15/l25/l50)
    Net(upscale)            python/MoeNet_lite2.py:22-54 (lite2/4/8; re-exported by moephoto_amd.MoeNet_lite2)

x: torch tensor (B,1,h,w), fp16 or fp32, on a HIP device, any strides (doCrop hands in a slice view).
Returns a one-element list [y], y: (B,1,scale*h,scale*w) in x's dtype -- the reference's forwards return
lists and Option.__call__ takes the last entry.
"""
import ctypes
import os
import weakref
from collections import OrderedDict

import numpy as np
import torch

from . import _lib

_DT = {torch.float32: _lib.F32, torch.float16: _lib.F16}


class EngineModule(object):
    ARCH = None
    SCALE = 0
    castDtype = 'float16'     # class attribute castModel looks up (python/imageProcess.py:310)

    def __init__(self, scale=None):
        L = _lib.lib()
        self._h = ctypes.c_void_p()
        _lib.check(L.moe_net_create(self.ARCH, int(scale if scale is not None else self.SCALE), ctypes.byref(self._h)))
        self.scale = L.moe_net_scale(self._h)
        self._params = OrderedDict()
        self._device = None
        self._dtype = torch.float32
        self._finalized_key = None
        self._last_input = None
        self._last_flag = 0
        self.training = False
        # 'auto' = the cheapest arithmetic that stays within 1e-3 of the fp32 reference on every input class:
        #   'mixed'  Net2x/3x/4x, NetDN: fp16 MFMA operands, hi+lo trunk stream, split operands on the few layers that set the error
        #   'fp16'   SEDN (4-6e-4 as it is)
        #   'fp16x3' lite* (every layer of these shallow 48-channel nets is error-critical)
        # MOE_PRECISION=fp16 forces the single-pass mode everywhere (the arithmetic of the reference's own GPU fp16 mode).
        self.precision = os.environ.get('MOE_PRECISION', 'auto')

    def __del__(self):
        try:
            if getattr(self, '_h', None) is not None and self._h.value:
                _lib.lib().moe_net_destroy(self._h)
                self._h = ctypes.c_void_p()
        except Exception:
            pass

    # ---- nn.Module protocol used by initModel / castModel ----------------------------------------
    def expected_keys(self):
        L = _lib.lib()
        out = OrderedDict()
        name, shape, nd = ctypes.c_char_p(), (ctypes.c_int64 * 4)(), ctypes.c_int()
        for i in range(L.moe_net_num_params(self._h)):
            _lib.check(L.moe_net_param_info(self._h, i, ctypes.byref(name), shape, ctypes.byref(nd)))
            out[name.value.decode()] = tuple(shape[d] for d in range(nd.value))
        return out

    def load_state_dict(self, state_dict, strict=True):
        L = _lib.lib()
        exp = self.expected_keys()
        keys = list(state_dict.keys())
        missing = [k for k in exp if k not in state_dict]
        unexpected = [k for k in keys if k not in exp]
        if strict and (missing or unexpected):
            msg = 'Error(s) in loading state_dict for {}:'.format(type(self).__name__)
            if missing:
                msg += '\n\tMissing key(s) in state_dict: {}. '.format(', '.join('"{}"'.format(k) for k in missing))
            if unexpected:
                msg += '\n\tUnexpected key(s) in state_dict: {}. '.format(', '.join('"{}"'.format(k) for k in unexpected))
            raise RuntimeError(msg)
        for k in keys:
            if k not in exp:
                continue
            v = state_dict[k]
            a = v.detach().cpu().numpy() if isinstance(v, torch.Tensor) else np.asarray(v)
            a = np.ascontiguousarray(a, dtype=np.float32)
            shape = (ctypes.c_int64 * max(1, a.ndim))(*a.shape)
            try:
                _lib.check(L.moe_net_set_param(self._h, k.encode(), a.ctypes.data_as(ctypes.c_void_p), shape, a.ndim))
            except _lib.EngineError as e:
                raise RuntimeError('Error(s) in loading state_dict for {}:\n\t{}'.format(type(self).__name__, e))
            self._params[k] = torch.from_numpy(a.copy())
        self._finalized_key = None
        return self

    def state_dict(self):
        return OrderedDict((k, v.clone()) for k, v in self._params.items())

    def parameters(self):
        return iter(self._params.values())

    def named_parameters(self):
        return iter(self._params.items())

    def eval(self):
        self.training = False
        return self

    def train(self, mode=True):
        if mode:
            raise NotImplementedError('moephoto_amd models are inference-only')
        return self.eval()

    def requires_grad_(self, flag=False):
        return self

    def to(self, *args, **kwargs):
        dtype, device = kwargs.get('dtype'), kwargs.get('device')
        for a in args:
            if isinstance(a, torch.dtype):
                dtype = a
            elif a is not None:
                device = a
        if dtype is not None:
            if dtype not in _DT:
                raise TypeError('moephoto_amd models run with fp16 or fp32 I/O, not {}'.format(dtype))
            self._dtype = dtype
        if device is not None:
            device = torch.device(device)
            if device.type != 'cuda':
                raise _lib.EngineError('moephoto_amd models run on a HIP device only (got device "{}"); there is no CPU path'.format(device))
            self._device = torch.device('cuda', device.index if device.index is not None else torch.cuda.current_device())
        if self._device is not None:
            self._finalize()
        return self

    def half(self):
        return self.to(dtype=torch.float16)

    def float(self):
        return self.to(dtype=torch.float32)

    def cuda(self, device=None):
        return self.to(device=torch.device('cuda', device if device is not None else torch.cuda.current_device()))

    def set_precision(self, precision):
        if precision != 'auto' and precision not in _lib.PRECISIONS:
            raise ValueError('precision must be "auto" or one of {}'.format(sorted(_lib.PRECISIONS)))
        self.precision = precision
        if self._device is not None:
            self._finalize()
        return self

    def resolved_precision(self):
        """The arithmetic 'auto' (or 'mixed' on a family without such a recipe) resolves to.  The per-family policy lives behind the C ABI
        (MOE_PREC_AUTO, moe_net_resolved_precision): this is only its name."""
        names = {v: k for k, v in _lib.PRECISIONS.items()}
        if self.precision == 'auto' or (self.precision == 'mixed' and self.ARCH in (_lib.ARCH_LITE, _lib.ARCH_SEDN)):
            return names[_lib.check(_lib.lib().moe_net_resolved_precision(self._h, _lib.PREC_AUTO))]
        return self.precision

    def max_tile_pixels(self):
        """Largest tile (pixels per plane) one forward accepts; the planner's budget is clamped to it."""
        return int(_lib.check(_lib.lib().moe_net_max_tile_pixels(self._h)))

    def set_exact_blocks(self, blocks):
        """'mixed' precision: number of leading ARSBs computed with split operands (0..6, -1 = architecture default)."""
        _lib.check(_lib.lib().moe_net_set_exact_blocks(self._h, int(blocks)))
        return self

    def calibrate(self, target=0.0):
        """Measure the number of split-operand ARSBs THESE weights need (moe_net_calibrate: uint8-noise tiles through the exact mode and through 'mixed' with
        n = default .. 6 blocks, on the device) and keep it.  Returns (n, predicted worst-tile error of a full frame at n); n = -1 when six blocks do not reach `target`
        (<= 0: the library's default).  With precision 'auto' the module is finalized again on the result (n blocks, or 'fp16x3' when n = -1); with an explicit 'mixed' the
        count applies (exact_blocks()) and n = -1 leaves the architecture's count in force.  `.to(device)` with precision 'auto' already does this once per checkpoint (moe_net_finalize(MOE_PREC_AUTO)); this is the explicit call, e.g. with
        another target.  None for families without the knob (SEDN, lite) or when the module runs in another arithmetic."""
        if self._device is None:
            raise _lib.EngineError('calibrate: move the module to its device first')
        if self.ARCH in (_lib.ARCH_LITE, _lib.ARCH_SEDN) or self.resolved_precision() not in ('mixed', 'fp16x3') or self.precision not in ('auto', 'mixed'):
            return None
        n, err = ctypes.c_int(), ctypes.c_double()
        stream = torch.cuda.current_stream(self._device).cuda_stream
        try:
            _lib.check(_lib.lib().moe_net_calibrate(self._h, float(target), ctypes.byref(n), ctypes.byref(err), stream))
        except Exception:
            self._finalized_key = None      # the C side un-finalizes the net on a failed measurement: the next .to() / forward must finalize again, not report 'not finalized' for ever
            raise
        if self.precision == 'auto':
            # the measurement is what moe_net_finalize(MOE_PREC_AUTO) decides on: finalize again so that the arithmetic (mixed with n blocks, or fp16x3 when n = -1) and
            # exact_blocks() follow THIS result -- otherwise a net that had resolved to fp16x3 would stay there with n >= 0, and one that got n = -1 would keep running
            # 'mixed' with the architecture's count
            self._finalized_key = None
            self._finalize()
        return n.value, err.value

    def exact_blocks(self):
        """The count of split-operand ARSBs the next forward runs with (moe_net_exact_blocks)."""
        return int(_lib.check(_lib.lib().moe_net_exact_blocks(self._h)))

    def set_option(self, key, value):
        """A kernel-form switch of this net (moe_net_set_option): e.g. ('sp_impl', 'rw'), ('arsb_fuse', 0).  Takes effect at the next forward."""
        v = value if isinstance(value, str) else str(int(value))
        _lib.check(_lib.lib().moe_net_set_option(self._h, str(key).encode(), v.encode()))
        return self

    def _finalize(self):
        key = (self._device.index, self.precision)      # (the REQUESTED arithmetic: what 'auto' resolves to may depend on the loaded weights -- moe_net_calibrate)
        if self._finalized_key == key:
            return
        _lib.require_device()
        prec = _lib.PREC_AUTO if self.precision == 'auto' else _lib.PRECISIONS[self.resolved_precision()]
        _lib.check(_lib.lib().moe_net_finalize(self._h, self._device.index, prec))
        self._finalized_key = key

    # ---- forward -------------------------------------------------------------------------------------
    def forward(self, x):
        if not isinstance(x, torch.Tensor) or x.dim() != 4 or x.shape[1] != 1:
            raise ValueError('expected a (B,1,h,w) tensor, got {}'.format(tuple(getattr(x, 'shape', ()))))
        if x.device.type != 'cuda':
            raise _lib.EngineError('input must live on a HIP device (moephoto_amd has no CPU path)')
        if x.dtype not in _DT:
            raise TypeError('input must be fp16 or fp32')
        if self._device is None or self._device != x.device:
            self.to(device=x.device)
        B, _, h, w = x.shape
        y = torch.empty((B, 1, h * self.scale, w * self.scale), dtype=x.dtype, device=x.device)
        sB, _, sH, sW = x.stride()
        stream = torch.cuda.current_stream(x.device).cuda_stream
        _lib.check(_lib.lib().moe_net_forward_ex(self._h, x.data_ptr(), _DT[x.dtype], B, h, w, sB, sH, sW, None,
                                                 y.data_ptr(), _DT[y.dtype], None, stream, self._input_since_prev(x, stream)))
        return [y]

    def _input_since_prev(self, x, stream):
        """MOE_FWD_INPUT_SINCE_PREV for this call: was x complete on the stream when the PREVIOUS forward was enqueued?  Yes when x is a view of the same LIVE storage
        object as the previous call's input, at the same version counter, on the same stream -- torch bumps a storage's version on every in-place write, and an
        out-of-place result would be another storage (a dead weak reference: its address may have been handed out again).  That is the reference's tile loop
        (python/imageProcess.py:164-170: every tile is a slice of one padded image that exists before the loop); consecutive forwards then overlap inside the engine
        (include/moephoto_amd.h).  Anything else -- a new image, an input produced between the calls, another stream -- gets plain stream order."""
        base = x._base if x._base is not None else x
        key = (base._version, stream)
        last = self._last_input
        flag = _lib.FWD_INPUT_SINCE_PREV if (last is not None and last[0]() is base and last[1] == key) else 0
        self._last_input = (weakref.ref(base), key)
        self._last_flag = flag          # (tests)
        return flag

    __call__ = forward

    # ---- live kernel timing (bench.py roofline leg) -----------------------------------------------------
    def set_profile(self, layer_substrings):
        """Comma-separated layer-key substrings to time (None: off), e.g. 'up1,c2_'."""
        self._prof_n = len([k for k in (layer_substrings or '').split(',') if k])
        _lib.check(_lib.lib().moe_net_set_profile(self._h, layer_substrings.encode() if layer_substrings else None))
        return self

    def get_profile(self, all_keys=False):
        """Summed kernel time / launches / algorithmic FLOPs of the first profiled substring (a dict), or of each one (a list,
        all_keys=True), since set_profile; resets the recording."""
        L = _lib.lib()
        out = []
        for i in range(max(1, getattr(self, '_prof_n', 1)) if all_keys else 1):
            ms, n, fl = ctypes.c_double(), ctypes.c_int64(), ctypes.c_double()
            _lib.check(L.moe_net_get_profile_at(self._h, i, ctypes.byref(ms), ctypes.byref(n), ctypes.byref(fl)))
            out.append(dict(total_ms=ms.value, launches=n.value, flops=fl.value))
        ms, n, fl = ctypes.c_double(), ctypes.c_int64(), ctypes.c_double()
        _lib.check(L.moe_net_get_profile(self._h, ctypes.byref(ms), ctypes.byref(n), ctypes.byref(fl)))     # (resets)
        return out if all_keys else out[0]

    # ---- debugging -----------------------------------------------------------------------------------
    def set_debug(self, flag=True):
        _lib.check(_lib.lib().moe_net_set_debug(self._h, 1 if flag else 0))
        return self

    def debug_tap(self, name):
        """fp32 NCHW copy of a named intermediate of the last forward (after set_debug(True))."""
        L = _lib.lib()
        shape = (ctypes.c_int64 * 4)()
        stream = torch.cuda.current_stream(self._device).cuda_stream
        n = _lib.check(L.moe_net_debug_tap(self._h, name.encode(), None, 0, shape, stream))
        out = np.empty(tuple(shape), np.float32)
        _lib.check(L.moe_net_debug_tap(self._h, name.encode(), out.ctypes.data_as(ctypes.c_void_p), n, shape, stream))
        return out


class Net2x(EngineModule):
    ARCH, SCALE = _lib.ARCH_NET2X, 2


class Net3x(EngineModule):
    ARCH, SCALE = _lib.ARCH_NET3X, 3


class Net4x(EngineModule):
    ARCH, SCALE = _lib.ARCH_NET4X, 4


class NetDN(EngineModule):
    ARCH, SCALE = _lib.ARCH_NETDN, 1


class SEDN(EngineModule):
    ARCH, SCALE = _lib.ARCH_SEDN, 1


class Net(EngineModule):
    """MoeNet_lite2.Net(upscale=2|4|8)."""
    ARCH = _lib.ARCH_LITE

    def __init__(self, upscale=2):
        super(Net, self).__init__(scale=upscale)
        self.upscale = upscale
