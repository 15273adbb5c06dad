"""Oracle: tile planner.  TEST INFRASTRUCTURE (see oracle/__init__.py).

Restates, in plain Python/numpy, what the reference computes in
  getAnchors   python/imageProcess.py:19-35
  getPad       python/imageProcess.py:47-56   (effective region only, see pad_plan)
  solveRam     python/imageProcess.py:61-71   (scalar-coefficient branch, the only one on this path)
  prepare      python/imageProcess.py:73-118
  ceilBy/alignF/minSize  python/imageProcess.py:550-554
Pinned by tests/golden/planner.json (tools/gen_golden.py).
"""
import math
from dataclasses import dataclass, field
from typing import List, Tuple

import numpy as np

MIN_SIZE = 28  # imageProcess.py:552


def ceil_by(d: int, x) -> int:
    """Round int(x) up to a multiple of d (d a power of two); alignF[d] of imageProcess.py:550-554."""
    if d == 1:
        return x
    v = int(x)
    return ((v + d - 1) // d) * d


@dataclass
class Anchors:
    start: List[int]
    end: List[int]
    clip: int      # <=0: minus the count of genuinely new HR rows of the last tile (0 if single tile)
    step: int      # number of tiles along this axis
    end_sc: List[int]


def get_anchors(s: int, ns: int, l: int, pad: int, align: int, sc) -> Anchors:
    """imageProcess.py:19-35.  s: image extent, ns: extent used to count tiles (s-3*pad),
    l: tile extent, pad: overlap padding, sc: scale."""
    stride = l - 2 * pad
    if l >= ceil_by(align, s):
        step = 1
    else:
        step = max(2, int(math.ceil(ns / stride)))
    start = [k * stride + pad for k in range(step)]
    start[0] = 0
    end = [a + l for a in start]
    end_sc = [e * sc for e in end]
    if step > 1:
        # last tile is re-anchored flush with the image end; its extent is the aligned remainder
        start[-1] = s - ceil_by(align, s - end[-2] + pad)
        end[-1] = s
        clip = int((int(end[-2]) - s) * sc)
    else:
        end[-1] = ceil_by(align, s)
        clip = 0
    end_sc[-1] = s * sc
    return Anchors(start, end, clip, step, [int(e) for e in end_sc])


def pixel_budget(ram: float, channels: int, ram_coef: float, lead: int) -> float:
    """solveRam (imageProcess.py:61-71) with a scalar coefficient, as called from prepare
    (imageProcess.py:75): m / c * (ramCoef / shape[0])."""
    k = ram_coef / lead if lead else 1.0
    return ram / channels * k


@dataclass
class PadPlan:
    """Which axes are padded (reflect, then zeros) to an aligned single tile: imageProcess.py:98-108."""
    pad_w_to: int = 0   # 0: untouched
    pad_h_to: int = 0
    crop_w: int = 0     # 0: no crop of the HR result
    crop_h: int = 0


@dataclass
class Plan:
    tiles: List[Tuple[int, int, int, int, int, int, int, int]]  # (top,bottom,left,right,topT,leftT,bsc,rsc)
    pad: PadPlan
    out_shape: Tuple[int, ...]
    pad_sc: int
    step_h: int
    step_w: int
    tile_h: int
    tile_w: int
    anchors_h: Anchors = field(repr=False, default=None)
    anchors_w: Anchors = field(repr=False, default=None)


def prepare(shape, ram, ram_coef, pad, sc, align=8, cropsize=0, fix_channel=0) -> Plan:
    """imageProcess.py:73-118 without the torch closures: returns the tile list in raster order,
    the padding plan and the HR output shape."""
    c, h, w = shape[-3], shape[-2], shape[-1]
    n = pixel_budget(ram, fix_channel or c, ram_coef, shape[0])
    s = ceil_by(align, MIN_SIZE + pad * 2)
    if n < s * s:
        raise MemoryError('Free memory space is {} bytes, which is not enough.'.format(ram))
    ph, pw = max(1, h - pad * 3), max(1, w - pad * 3)
    # candidate tile heights ns (multiples of align) and the widest width ms that keeps ns*ms <= n
    ns = np.arange(s / align, int(n / (align * s)) + 1, dtype=int)
    ms = (n / (align * align) / ns).astype(int)
    ns, ms = ns * align, ms * align
    nn = np.ceil(ph / (ns - 2 * pad)).clip(2)
    mn = np.ceil(pw / (ms - 2 * pad)).clip(2)
    nn[ns >= h] = 1
    mn[ms >= w] = 1
    ds = nn * mn
    ind = np.argwhere(ds == ds.min()).squeeze(1)
    mina = ind[np.abs(ind - len(ds) / 2).argmin()]
    ah, aw, acs = ceil_by(align, h), ceil_by(align, w), ceil_by(align, cropsize)
    if cropsize > 0:
        ih, iw = min(acs, int(ns[mina])), min(acs, int(ms[mina]))
    else:
        ih, iw = int(ns[mina]), int(ms[mina])
    ih, iw = min(ah, ih), min(aw, iw)
    a_h = get_anchors(h, ph, ih, pad, align, sc)
    a_w = get_anchors(w, pw, iw, pad, align, sc)
    pad_sc, outh, outw = int(pad * sc), int(h * sc), int(w * sc)
    pp = PadPlan()
    if a_h.step > 1 and a_w.step > 1:
        pass
    elif a_h.step > 1:
        pp.pad_w_to, pp.crop_w = aw, outw
    elif a_w.step > 1:
        pp.pad_h_to, pp.crop_h = ah, outh
    else:
        pp.pad_w_to, pp.crop_w, pp.pad_h_to, pp.crop_h = aw, outw, ah, outh
    tiles = []
    for i in range(a_h.step):
        top_t = a_h.clip if i == a_h.step - 1 else (0 if i == 0 else pad_sc)
        for j in range(a_w.step):
            left_t = a_w.clip if j == a_w.step - 1 else (0 if j == 0 else pad_sc)
            tiles.append((a_h.start[i], a_h.end[i], a_w.start[j], a_w.end[j], top_t, left_t,
                          a_h.end_sc[i], a_w.end_sc[j]))
    return Plan(tiles, pp, tuple(shape[:-2]) + (outh, outw), pad_sc, a_h.step, a_w.step, ih, iw, a_h, a_w)


def reflect_then_zero(x: np.ndarray, axis: int, to: int) -> np.ndarray:
    """Effective result of getPad (imageProcess.py:47-56) on one axis, restricted to the first `to`
    entries (the only ones a single-tile axis ever reads): reflect (edge not repeated) by
    min(len-1, to-len), zeros after that.  The reference's two-stage branch over-allocates zeros
    beyond `to` (rw = aw - tw); they are never read."""
    n = x.shape[axis]
    if to <= n:
        return x
    refl = min(n - 1, to - n)
    idx = list(range(n)) + [n - 2 - k for k in range(refl)]
    out = np.take(x, idx, axis=axis)
    rest = to - n - refl
    if rest > 0:
        zshape = list(out.shape)
        zshape[axis] = rest
        out = np.concatenate([out, np.zeros(zshape, dtype=x.dtype)], axis=axis)
    return out


def blend_ramp(pad_sc: int) -> np.ndarray:
    """imageProcess.py:109: sigmoid(9*(k/padSc - 0.5)), k < padSc, evaluated by torch in fp32 (the
    authority for the last ulp of the sigmoid)."""
    import torch
    if pad_sc <= 0:
        return np.zeros((0,), np.float32)
    return ((torch.arange(pad_sc, dtype=torch.float32) / pad_sc - .5) * 9).sigmoid().numpy()
