"""Host-side mirror of the tiling runtime of python/imageProcess.py, on top of the HIP engine.

Same names and argument meaning as the reference so callers (runSR/runDN adapters, the pipeline
builder) read unchanged; the work itself is done on the device by libmoephoto_amd.so:

  Option, initModel, getStateDict      python/imageProcess.py:304-334, 379-395
  prepare / getAnchors / TilePlan      :19-35, 73-118   -> C planner (moe_plan_create)
  prepareOpt, doCrop                   :133-172         -> moe_run_plan (tile gather, net, stitch on device)
  ensemble, trans/transInv             :563-572
  RGBFilter, strengthOp, alpha helpers :350-377, 562
  toTorch / toFloat / toOutput         :238-263         -> moe_to_float / moe_to_output kernels
  readFile / writeFile                 :265-302         (PIL, host)

There is no CPU path: models must be moephoto_amd.models.EngineModule instances on a HIP device.
"""
import ctypes
import logging
import time
from functools import reduce

import numpy as np
import torch
import torch.nn.functional as F

from . import _lib
from .config import config
from .models import EngineModule
from .weights import load_state_dict_file

log = logging.getLogger('Moe')
modelCache = {}
weightCache = {}
minSize = 28
PLAN_CACHE = 4           # tile plans (with their device pools) kept per Option: an image shape and its transposed twin for the ensemble, x2
identity = lambda x, *_, **__: x
apply = lambda v, f: f(v)
_DT = {torch.float32: _lib.F32, torch.float16: _lib.F16}


def ceilBy(d):
    return lambda x: ((int(x) + d - 1) // d) * d


alignF = {1: identity}
alignF.update((1 << k, ceilBy(1 << k)) for k in (3, 4, 5, 6, 7, 9))


# ---------------------------------------------------------------------------------------------------
# planner
# ---------------------------------------------------------------------------------------------------
class TilePlan(object):
    """A tile plan for one image shape: the C planner's result plus its device tables."""

    def __init__(self, shape, ram, ramCoef, pad, sc, align=8, cropsize=0):
        L = _lib.lib()
        self.shape = tuple(int(v) for v in shape[-3:])
        self._h = ctypes.c_void_p()
        _lib.check(L.moe_plan_create((ctypes.c_int64 * 3)(*self.shape), float(ram), float(ramCoef), int(pad), int(sc),
                                     int(align), int(cropsize), ctypes.byref(self._h)))
        info = (ctypes.c_int64 * 12)()
        _lib.check(L.moe_plan_info(self._h, info))
        (self.n_tiles, self.stepH, self.stepW, self.outH, self.outW, self.padHTo, self.padWTo, self.padSc,
         self.tileH, self.tileW, self.clipH, self.clipW) = [int(v) for v in info]
        t = (ctypes.c_int32 * (8 * self.n_tiles))()
        _lib.check(L.moe_plan_tiles(self._h, t))
        self.tiles = [tuple(t[k * 8:(k + 1) * 8]) for k in range(self.n_tiles)]
        r = (ctypes.c_float * max(1, self.padSc))()
        _lib.check(L.moe_plan_ramp(self._h, r))
        self.ramp = np.array(r[:self.padSc], np.float32)
        rt = (ctypes.c_int32 * (4 * self.stepH))()
        _lib.check(L.moe_plan_rows(self._h, rt))
        self.rows = [tuple(rt[i * 4:(i + 1) * 4]) for i in range(self.stepH)]      # (first written row, first un-blended row, row of the tiles' row 0, tiles' height) per tile row
        self.sc, self.pad, self.align = int(sc), int(pad), int(align)

    def __del__(self):
        try:
            if self._h.value:
                _lib.lib().moe_plan_destroy(self._h)
                self._h = ctypes.c_void_p()
        except Exception:
            pass

    def pool_elems(self, C):
        return int(_lib.check(_lib.lib().moe_plan_pool_elems(self._h, int(C))))

    def tile_offsets(self, C):
        off = (ctypes.c_int64 * self.n_tiles)()
        _lib.check(_lib.lib().moe_plan_tile_offsets(self._h, int(C), off))
        return list(off)

    def seams(self):
        """per tile (ra0, ra1, rb0, rb1, ca0, ca1, cb0, cb1): the rows / columns of the tile a blend reads (moe_plan_seams; the wire format of dist.py)"""
        t = (ctypes.c_int32 * (8 * self.n_tiles))()
        _lib.check(_lib.lib().moe_plan_seams(self._h, t))
        return [tuple(t[k * 8:(k + 1) * 8]) for k in range(self.n_tiles)]

    def padImage(self, x):
        """getPad (python/imageProcess.py:47-56) restricted to the region a single-tile axis reads:
        reflect (edge not repeated) up to len-1, then zeros."""
        def pad_axis(t, from_end, to):      # from_end: 1 = width, 2 = height
            n = t.shape[-from_end]
            if not to or to <= n:
                return t
            refl = min(n - 1, to - n)
            if refl > 0:
                p = (0, refl, 0, 0) if from_end == 1 else (0, 0, 0, refl)
                t = F.pad(t.unsqueeze(0), p, mode='reflect').squeeze(0)
            rest = to - n - refl
            if rest > 0:
                t = F.pad(t, (0, rest, 0, 0) if from_end == 1 else (0, 0, 0, rest))
            return t
        x = pad_axis(x, 1, self.padWTo)
        x = pad_axis(x, 2, self.padHTo)
        return x

    def unpad(self, im):
        return im[..., :self.outH, :self.outW]


def getAnchors(s, ns, l, pad, af, sc):
    """python/imageProcess.py:19-35 (host arithmetic; the C planner computes the same per axis)."""
    n = l - 2 * pad
    step = 1 if l >= af(s) else max(2, -(-int(ns) // n))
    start = [k * n + pad for k in range(step)]
    start[0] = 0
    end = [a + l for a in start]
    endSc = [e * sc for e in end]
    if step > 1:
        start[-1] = s - af(s - end[-2] + pad)
        end[-1] = s
        clip = int((end[-2] - s) * sc)
    else:
        end[-1] = af(s)
        clip = 0
    endSc[-1] = s * sc
    return start, end, clip, step, [int(e) for e in endSc]


def prepare(shape, ram, opt, pad, sc, align=8, cropsize=0):
    """python/imageProcess.py:73-118.  Returns (iterClip, padImage, unpad, outShape, blend) like the
    reference; the TilePlan itself is attached to the returned iterClip as `.plan`."""
    plan = TilePlan(shape, ram, opt.ramCoef, pad, sc, align, cropsize)

    def iterClip():
        for t in plan.tiles:
            yield t
    iterClip.plan = plan
    b = torch.from_numpy(plan.ramp.copy()).view(1, -1)
    return iterClip, plan.padImage, plan.unpad, (*shape[:-2], plan.outH, plan.outW), b


class Option(object):
    """The per-op settings bag of python/imageProcess.py:379-395: the adapters (runSR / runDN .getOpt) fill in the fields, doCrop and
    the pipeline builder read them, and calling it runs the net on one tile batch (a list result means "take the last entry")."""
    _DEFAULTS = dict(ramCoef=1e-3, count=0, padding=1, cropsize=0, align=8, fixChannel=1, scale=1, ensemble=0, strength=1.0,
                     outShape=None, oShape=None, iterClip=None, modelCached=None)

    def __init__(self, path=''):
        for name, value in self._DEFAULTS.items():
            setattr(self, name, value)
        self.model = path
        self.prepare = identity
        # planes-as-batch adapters; the SR / DN tables replace them with dim-1 versions
        self.squeeze = lambda t: t.squeeze(0)
        self.unsqueeze = lambda t: t.unsqueeze(0)
        self._plans = {}          # image shape -> [TilePlan, uses]  (LRU, see _plan_for)

    def __call__(self, x, *args, **kwargs):
        result = self.modelCached(x, *args, **kwargs)
        return result[-1] if isinstance(result, list) else result


def getStateDict(path):
    """python/imageProcess.py:304-307, with our own closed parser of the legacy zoo format."""
    if path not in weightCache:
        sd = load_state_dict_file(path)
        weightCache[path] = type(sd)((k, torch.from_numpy(v)) for k, v in sd.items())
    return weightCache[path]


def castModel(model):
    """python/imageProcess.py:309-317.  Every engine model computes with fp16 MFMA operands and fp32
    accumulation whatever castDtype says; the dtype only selects the I/O element type."""
    return model.to(dtype=config.dtype(), device=config.device())


def initModel(opt, weights=None, key=None, f=lambda opt: opt.modelDef(), args=[]):
    """python/imageProcess.py:319-334: one instance per cache key for the life of the process; construction = constructor slot of
    the plugin table, strict load_state_dict (a path is read through the zoo parser), frozen, eval; every call re-casts to the
    configured dtype / device (castModel)."""
    model = modelCache.get(key) if key else None
    if model is None:
        log.info('loading model {}'.format(opt.model))
        model = f(opt, *args)
        if weights:
            log.info('reloading weights')
            model.load_state_dict(getStateDict(weights) if isinstance(weights, str) else weights)
        for p in model.parameters():
            p.requires_grad_(False)
        model.eval()
        if key:
            modelCache[key] = model
    return castModel(model)


def _plan_for(opt, shape):
    """prepareOpt (python/imageProcess.py:133-155): plans are cached per image shape; the reference
    re-plans every 29 calls to follow free memory -- here the plan only depends on free memory when
    cropsize is 'auto', and is then re-derived with the same cadence."""
    key = tuple(int(v) for v in shape[-3:])
    ent = opt._plans.pop(key, None)
    if ent is None or (opt.cropsize <= 0 and ent[1] > 28):
        model = opt.modelCached

        def plan_with(emptyCache):
            try:
                freeMem = config.calcFreeMem(emptyCache=emptyCache)
            except Exception:
                raise MemoryError('Can not calculate free memory.')
            if isinstance(model, EngineModule):
                # the planner's pixel budget is ram * ramCoef / C^2 (solveRam with fixChannel = 0): keep it within the largest
                # tile the convolution kernels address, however much memory is free (288 GB would otherwise plan whole 8K images)
                freeMem = min(freeMem, model.max_tile_pixels() * key[0] * key[0] / opt.ramCoef)
            return prepare(key, freeMem, opt, opt.padding, opt.scale, opt.align, opt.cropsize)
        try:
            it, padImage, unpad, outShape, bl = plan_with(False)
            # 'auto' tile size: memory torch has reserved but not handed out is invisible to hipMemGetInfo and unusable by the engine (hipMalloc).  The
            # reference counts it as free (python/config.py:61-71); here it is RELEASED when it is a large share of what is free -- otherwise the planner
            # would silently settle for smaller tiles while gigabytes sit in torch's cache (once per re-plan, i.e. every 29 calls at most)
            if opt.cropsize <= 0:
                slack = torch.cuda.memory_reserved(config.deviceId) - torch.cuda.memory_allocated(config.deviceId)
                if slack > (1 << 30) and slack > 0.1 * max(1, config.getFreeMem()):
                    it, padImage, unpad, outShape, bl = plan_with(True)
        except MemoryError:          # the smallest tile did not fit: hand torch's cached blocks back to the driver and ask again, once
            it, padImage, unpad, outShape, bl = plan_with(True)
        ent = [it.plan, 0]
        while len(opt._plans) >= PLAN_CACHE:       # each plan owns a device tile pool of ~1.1x its output image: keep a few, drop the
            opt._plans.pop(next(iter(opt._plans)))   # least recently used (the reference keeps exactly one, re-planning on a shape change)
        opt._plans[key] = ent
        opt.iterClip, opt.padImage, opt.unpad, opt.blend = it, padImage, unpad, bl
        opt.outShape = list(outShape)
    else:
        ent[1] += 1
        opt._plans[key] = ent                      # re-inserted last = most recently used
    return ent[0]


def prepareOpt(opt, shape):
    _plan_for(opt, shape)
    return opt.scale, int(opt.padding * opt.scale)


def doCrop(opt, x, *args, **_):
    """python/imageProcess.py:157-172, entirely on the device: the planes of x (C,H,W) become the batch
    (runSR.py:37-40), tiles are gathered straight from x by the stem kernel, same-shaped tiles are batched
    through the net, and the gather-stitch kernel folds them with the reference's sequential blend."""
    model = opt.modelCached
    if not isinstance(model, EngineModule):
        raise TypeError('doCrop needs an engine-backed model (moephoto_amd.models.*), got {}'.format(type(model).__name__))
    if x.device.type != 'cuda':
        raise _lib.EngineError('doCrop: input must live on a HIP device (moephoto_amd has no CPU path)')
    if x.dim() != 3:
        raise ValueError('doCrop expects a (C,H,W) image')
    plan = _plan_for(opt, x.shape)
    xp = plan.padImage(x)
    if xp.dtype not in _DT:
        xp = xp.to(config.dtype())
    model.to(device=x.device)
    C = xp.shape[0]
    out = xp.new_empty((C, plan.outH, plan.outW))
    sC, sH, sW = xp.stride()
    stream = torch.cuda.current_stream(x.device).cuda_stream

    def run():
        _lib.check(_lib.lib().moe_run_plan(model._h, plan._h, xp.data_ptr(), _DT[xp.dtype], sC, sH, sW,
                                           out.data_ptr(), _DT[out.dtype], int(config.tilesPerBatch), stream))
    try:
        run()
    except MemoryError:              # the engine's workspace / tile pool is hipMalloc'ed beside torch's caching allocator: release the
        torch.cuda.empty_cache()     # cache and try once more before reporting "does not fit" (python/imageProcess.py:58-59)
        run()
    # the engine reads xp asynchronously: keep it alive until the stream has consumed it
    xp.record_stream(torch.cuda.current_stream(x.device))
    return out


# ---- resize step (python/imageProcess.py:174-214, 555-556) --------------------------------------------------
def blendTile(r, canvas, tile, sc, padSc, ramp):
    """The two `blend` calls and the slice-assign of the reference's tile loop (python/imageProcess.py:120-131,167-170) for ONE tile, as one kernel
    (moe_blend_tile) -- for a caller that keeps MoePhoto's own doCrop loop around the drop-in model class:

        for tile in opt.iterClip():                          # tile = (top, bottom, left, right, topT, leftT, bsc, rsc)
            r = opt.squeeze(opt(x[..., tile[0]:tile[1], tile[2]:tile[3]]))
            blendTile(r, tmp_image, tile, sc, padSc, opt.blend)      # instead of: t = ...; q, _ = blend(*blend(...)); tmp_image[...] = q

    r: (C, h, w) tile result (rows / columns beyond the window are ignored: opt.unpad); canvas: (C, H, W); ramp: opt.blend (padSc values); all three of one
    dtype (fp16 / fp32), on the device, unit stride along the last axis.  The canvas receives the bits the reference's torch expression produces."""
    top, bottom, left, right, topT, leftT, bsc, rsc = [int(v) for v in tile]
    canvas_in = canvas
    # moe_blend_tile takes raw pointers and (plane, row) strides: this wrapper is the only guard against a canvas / tile of another rank (the reference's
    # tmp_image is (1, C, H, W) when opt.oShape is set; a (C, 1, H, W) result is the net's own output shape).  Singleton leading / second axes are dropped, anything
    # else is refused -- stride(1) of a 4-D tensor is NOT the row stride
    if r.dim() == 4 and r.shape[1] == 1:
        r = r.squeeze(1)
    while r.dim() > 3 and r.shape[0] == 1:
        r = r[0]
    while canvas.dim() > 3 and canvas.shape[0] == 1:
        canvas = canvas[0]
    if canvas.dim() == 4 and canvas.shape[1] == 1:
        canvas = canvas[:, 0]
    if r.dim() != 3 or canvas.dim() != 3:
        raise ValueError('blendTile: a (C, h, w) tile result and a (C, H, W) canvas expected, got {} and {}'.format(tuple(r.shape), tuple(canvas.shape)))
    if not (r.dtype == canvas.dtype and (ramp is None or ramp.dtype == canvas.dtype)) or r.dtype not in _DT:
        raise TypeError('blendTile: tile, canvas and ramp must share one dtype (fp16 or fp32)')
    if r.stride(-1) != 1 or canvas.stride(-1) != 1 or r.device != canvas.device or r.shape[0] != canvas.shape[0]:
        raise ValueError('blendTile: unit stride along the last axis, one device and one plane count expected')
    if r.shape[-2] < bsc - top * sc or r.shape[-1] < rsc - left * sc or bsc > canvas.shape[-2] or rsc > canvas.shape[-1]:
        raise ValueError('blendTile: the window {}..{} x {}..{} does not fit the tile result {} / the canvas {}'.format(top * sc, bsc, left * sc, rsc, tuple(r.shape), tuple(canvas.shape)))
    rp = ramp.reshape(-1) if ramp is not None else None
    if rp is not None and (rp.numel() < padSc or rp.stride(0) != 1):
        raise ValueError('blendTile: ramp must hold padSc contiguous values')
    stream = torch.cuda.current_stream(canvas.device).cuda_stream
    _lib.check(_lib.lib().moe_blend_tile(r.data_ptr(), r.stride(-3), r.stride(-2), canvas.data_ptr(), canvas.stride(-3), canvas.stride(-2), _DT[canvas.dtype], int(canvas.shape[0]),
                                         int(top * sc), int(left * sc), bsc, rsc, topT, leftT, int(padSc), rp.data_ptr() if rp is not None else None, stream))
    return canvas_in


def resizeByTorch(x, width, height, mode='bilinear'):
    """The reference's `resizeByTorch` = F.interpolate(x[None], size=(height, width), mode=mode, align_corners=False)[0]; the name
    is kept for the callers, the work is moe_resize on the device (nearest / bilinear / bicubic)."""
    if mode not in _lib.RESIZE_MODES:
        raise ValueError('resize: interpolation method "{}" is not supported (nearest, bilinear, bicubic)'.format(mode))
    if x.device.type != 'cuda':
        raise _lib.EngineError('resize: input must live on a HIP device (moephoto_amd has no CPU path)')
    if x.dtype not in _DT:
        x = x.to(config.dtype())
    x = x.contiguous()
    C, H, W = x.shape
    out = torch.empty((C, int(height), int(width)), dtype=x.dtype, device=x.device)
    stream = torch.cuda.current_stream(x.device).cuda_stream
    _lib.check(_lib.lib().moe_resize(x.data_ptr(), out.data_ptr(), _DT[x.dtype], C, H, W, int(height), int(width),
                                     _lib.RESIZE_MODES[mode], x.device.index or 0, stream))
    x.record_stream(torch.cuda.current_stream(x.device))
    return out


def resize(opt, out=None, pos=0, nodes=(), h=1, w=1):
    """The `resize` step: target size from scaleH / scaleW (rounded like the reference) or height / width; once the size is known
    for a stream source it is kept (opt['update'])."""
    opt['update'] = True
    opt.setdefault('method', 'bilinear')

    def f(im):
        nonlocal h, w
        if opt['update']:
            _, ih, iw = im.shape
            h = round(ih * opt['scaleH']) if 'scaleH' in opt else opt['height']
            w = round(iw * opt['scaleW']) if 'scaleW' in opt else opt['width']
            if out and out.get('source'):
                opt['update'] = False
        return resizeByTorch(im, w, h, opt['method'])
    return f


def restrictSize(width, height=0, method='bilinear'):
    """Shrink to fit width x height keeping the aspect ratio; images that already fit pass through (python/imageProcess.py:197-214)."""
    height = height or width
    st = {}

    def f(im):
        if not st:
            _, ih, iw = im.shape
            st['fit'] = ih <= height and iw <= width
            sh, sw = height / ih, width / iw
            st['hw'] = (height, round(iw * sh)) if sh < sw else (round(ih * sw), width)
        return im if st['fit'] else resizeByTorch(im, st['hw'][1], st['hw'][0], method)
    return f


# ---- self-ensemble (python/imageProcess.py:563-572) -------------------------------------------------
def _dihedral(t, fh, fv):
    """One element of the square's symmetry group as (transpose first?, flip width?, flip height?), and its inverse."""
    def fwd(x):
        if t:
            x = x.transpose(-1, -2)
        dims = ([-1] if fh else []) + ([-2] if fv else [])
        return x.flip(*dims) if dims else x

    def inv(x):
        dims = ([-1] if fh else []) + ([-2] if fv else [])
        if dims:
            x = x.flip(*dims)
        return x.transpose(-1, -2) if t else x
    return fwd, inv


# the seven non-identity symmetries in the reference's order (python/imageProcess.py:563-569: transpose, flip, flip2, flip.transpose,
# transpose.flip, transpose.flip.transpose, flip2.transpose -- "a.b" applies a first); transpose.flip.transpose is a height flip
_SYMMETRIES = [_dihedral(True, False, False), _dihedral(False, True, False), _dihedral(False, True, True), _dihedral(True, False, True),
               _dihedral(True, True, False), _dihedral(False, False, True), _dihedral(True, True, True)]
trans = [f for f, _ in _SYMMETRIES]
transInv = [g for _, g in _SYMMETRIES]


def ensemble(opt):
    """x -> doCrop(x) + sum_i transInv_i(doCrop(trans_i(x))) for the first opt.ensemble transforms.
    (The reference keeps a second Option for transposed shapes; here plans are keyed by shape.)"""
    def f(x):
        v = doCrop(opt, x)
        for i in range(opt.ensemble):
            v = v + transInv[i](doCrop(opt, trans[i](x)))
        return v
    return f


# ---- DN wrapper (python/imageProcess.py:336-377, 562) -----------------------------------------------
def strengthOp(x, inp, s=1):
    """Blend the denoised image with its input: s * x + (1 - s) * inp (python/imageProcess.py:562)."""
    return x if s == 1 else s * x + (1 - s) * inp


def extractAlpha(t):
    """Splits a 4-plane image into RGB (returned) and alpha (kept in `t`); other plane counts pass through."""
    def f(im):
        if im.shape[0] != 4:
            return im
        t['im'] = im[3]
        return im[:3]
    return f


def mergeAlpha(t):
    """Re-attaches the alpha plane extractAlpha put aside (none: the image passes through)."""
    def f(im):
        if 'im' not in t:
            return im
        return torch.cat([im, t['im'].unsqueeze(0).to(im.dtype)], 0)
    return f


def _RGBFilter(opt, img):
    """The DN wrapper (python/imageProcess.py:350-377): the denoiser sees the colour planes only, `strength` blends its result
    with the input, alpha rides around it."""
    alpha = {}
    rgb = opt.prepare(extractAlpha(alpha)(img))
    return mergeAlpha(alpha)(strengthOp(doCrop(opt, rgb), rgb, opt.strength))


RGBFilter = lambda opt: lambda img: _RGBFilter(opt, img)


# ---- image edges (python/imageProcess.py:238-302) ----------------------------------------------------
def toTorch(bitDepth, dtype=None, device=None):
    """HWC uint8 / uint16 numpy image -> (C,H,W) tensor on the device, v/255 (8 bit) or v/2^bits."""
    def f(image):
        dt = dtype if dtype is not None else config.dtype()
        dev = torch.device(device if device is not None else config.device())
        a = np.ascontiguousarray(image)
        if a.ndim == 2:
            a = a[:, :, None]
        if bitDepth <= 8:
            a = a.astype(np.uint8, copy=False)
            src_dt = _lib.U8
        else:
            a = a.astype(np.uint16)
            src_dt = _lib.U16
        H, W, C = a.shape
        src = torch.from_numpy(a).to(dev)
        dst = torch.empty((C, H, W), dtype=dt, device=dev)
        stream = torch.cuda.current_stream(dev).cuda_stream
        _lib.check(_lib.lib().moe_to_float(src.data_ptr(), src_dt, int(bitDepth), H, W, C, dst.data_ptr(), _DT[dt], dev.index or 0, stream))
        src.record_stream(torch.cuda.current_stream(dev))
        return dst
    return f


def toFloat(image):
    """python/imageProcess.py:238-243: (C,H,W) -> (H,W,C) fp32 (a permuted view when already fp32)."""
    if len(image.shape) == 3:
        image = image.permute(1, 2, 0)
    else:
        image = image.squeeze(0)
    return image.to(dtype=torch.float)


def toOutput(bitDepth):
    """python/imageProcess.py:245-257: x * 2^bits, clamp [0, 2^bits - 1], truncate, to host numpy (H,W,C)."""
    def f(image):
        if image.dim() == 2:
            image = image.unsqueeze(2)
        H, W, C = image.shape
        planar = image.permute(2, 0, 1)
        if not planar.is_contiguous():
            planar = planar.contiguous()
        if planar.dtype not in _DT:
            planar = planar.float()
        dev = planar.device
        out_dt, np_dt, lib_dt = (torch.uint8, np.uint8, _lib.U8) if bitDepth <= 8 else (torch.int16, np.uint16, _lib.U16)
        dst = torch.empty((H, W, C), dtype=out_dt, device=dev)
        stream = torch.cuda.current_stream(dev).cuda_stream
        _lib.check(_lib.lib().moe_to_output(planar.data_ptr(), _DT[planar.dtype], H, W, C, int(bitDepth), dst.data_ptr(), lib_dt, dev.index or 0, stream))
        planar.record_stream(torch.cuda.current_stream(dev))
        res = dst.cpu().numpy()
        return res.view(np_dt) if bitDepth > 8 else res
    return f


toOutput8 = toOutput(8)


def toNumPy(bitDepth):
    """python/imageProcess.py:216-229: (raw frame bytes, height, width) -> (H,W,3) array (the video path's `buffer` source;
    ffmpeg delivers bgr24 / bgr48le).  The reference widens to fp32 here and divides in toTorch; the integer view is kept
    instead and `toTorch` does the division on the device -- same values, one host copy less."""
    dtype = np.uint8 if bitDepth <= 8 else np.uint16

    def f(args):
        buffer, height, width = args
        if not buffer:
            return None
        return np.frombuffer(buffer, dtype=dtype).reshape((height, width, 3))
    return f


def toBuffer(bitDepth):
    """python/imageProcess.py:231-236: quantised (H,W,C) array -> raw bytes for the encoder pipe."""
    dtype = np.uint8 if bitDepth <= 8 else np.uint16
    return lambda im: np.ascontiguousarray(im, dtype=dtype).tobytes() if im is not None else None


def readFile(nodes=[], context=None):
    from PIL import Image

    def f(file):
        image = Image.open(file)
        if context is not None:
            context.imageMode = image.mode
        if image.mode == 'P':
            if context is not None:
                context.palette = image
            image = image.convert('RGB')
        image = np.array(image)
        if len(image.shape) == 2:
            return image.reshape(*image.shape, 1)
        if image.shape[2] in (3, 4):
            return image
        raise RuntimeError('Unknown image format')
    return f


def writeFile(image, name, context=None, *args):
    from PIL import Image
    if not name:
        name = 'output_{}.png'.format(int(time.time()))
    elif hasattr(name, 'seek'):
        name.seek(0)
    if image.shape[2] == 1:
        image = image.squeeze(2)
    Image.fromarray(image).save(name, *args)
    return name
