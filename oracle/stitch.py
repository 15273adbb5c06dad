"""Oracle: the doCrop tile loop and its order-dependent blend.  TEST INFRASTRUCTURE.

Sequential numpy (fp32) restatement of
  blend    python/imageProcess.py:120-131
  doCrop   python/imageProcess.py:157-172
plus `fold_stitch`, the per-pixel closed form the HIP stitch kernel implements, so that the
closed form itself is pinned against the sequential loop (and through it against the reference).
"""
import numpy as np

from .planner import Plan, blend_ramp, reflect_then_zero


def blend(r, ex, lt, pad, axis, ramp):
    """imageProcess.py:120-131.  r: fresh tile result, ex: what the canvas window already holds.
    Entries [lt-pad, lt) of r are cross-faded into ex with `ramp`, entries >= lt are taken from r,
    entries before lt-pad are dropped.  Returns (merged, ex narrowed to the same extent)."""
    l = r.shape[axis]
    if lt < 0:
        lt = l + lt
    if lt < 1:
        return r, ex
    start = lt - pad
    sl = [slice(None)] * r.ndim
    sl[axis] = slice(start, lt)
    band_r, band_x = r[tuple(sl)], ex[tuple(sl)]
    shp = [1] * r.ndim
    shp[axis] = pad
    band = band_x + ramp.reshape(shp) * (band_r - band_x)
    sl[axis] = slice(lt, l)
    merged = np.concatenate([band, r[tuple(sl)]], axis=axis)
    sl[axis] = slice(start, l)
    return merged, ex[tuple(sl)]


def pad_image(x, plan: Plan):
    if plan.pad.pad_w_to:
        x = reflect_then_zero(x, x.ndim - 1, plan.pad.pad_w_to)
    if plan.pad.pad_h_to:
        x = reflect_then_zero(x, x.ndim - 2, plan.pad.pad_h_to)
    return x


def unpad(r, plan: Plan):
    if plan.pad.crop_h:
        r = r[..., :plan.pad.crop_h, :]
    if plan.pad.crop_w:
        r = r[..., :plan.pad.crop_w]
    return r


def do_crop(x, plan: Plan, sc, model, collect=None):
    """imageProcess.py:157-172.  x: (C,H,W) fp32; model: callable (C,1,h,w)->(C,1,sc*h,sc*w)
    (planes as batch: runSR.py:37-40).  Returns the stitched (C,sc*H,sc*W) canvas."""
    x = np.asarray(x, np.float32)
    ramp = blend_ramp(plan.pad_sc)
    xp = pad_image(x, plan)
    out = np.full((x.shape[0],) + tuple(plan.out_shape[-2:]), np.nan, np.float32)  # new_empty: never read before written
    for (top, bottom, left, right, top_t, left_t, bsc, rsc) in plan.tiles:
        s = xp[:, None, top:bottom, left:right]
        r = np.asarray(model(s), np.float32)[:, 0]
        if collect is not None:
            collect.append(r)
        r = unpad(r, plan)
        t = out[:, int(top * sc):bsc, int(left * sc):rsc]
        q, t2 = blend(r, t, top_t, plan.pad_sc, 1, ramp)
        q, _ = blend(q, t2, left_t, plan.pad_sc, 2, ramp)
        h, w = q.shape[-2:]
        out[:, bsc - h:bsc, rsc - w:rsc] = q
    return out


def axis_cover(anchors, sc, pad_sc, out_len):
    """For one axis: per tile index, (first HR index written, first HR index taken un-blended,
    HR index of the tile's origin).  Derived from blend()'s lt/start arithmetic."""
    res = []
    for i in range(anchors.step):
        origin = int(anchors.start[i] * sc)
        end = anchors.end_sc[i]
        l = min(end, out_len) - origin
        lt = anchors.clip if i == anchors.step - 1 else (0 if i == 0 else pad_sc)
        if lt < 0:
            lt = l + lt
        if lt < 1:
            first, solid = origin, origin
        else:
            first, solid = origin + lt - pad_sc, origin + lt
        res.append((first, solid, origin, end))
    return res


def fold_stitch(tile_results, plan: Plan, sc):
    """Per-pixel closed form of do_crop's stitch (what the HIP gather-stitch kernel computes):
    every HR pixel folds, in raster tile order, over the tiles whose written region covers it:
        v1 = ex + wH*(r-ex)   (row inside the tile's blend band, else v1 = r)
        v  = ex + wW*(v1-ex)  (column inside the band, else v = v1)
    tile_results: list of (C, th*sc, tw*sc) arrays in plan.tiles order (before unpad)."""
    ramp = blend_ramp(plan.pad_sc)
    C = tile_results[0].shape[0]
    H, W = plan.out_shape[-2:]
    out = np.full((C, H, W), np.nan, np.float32)
    rows = axis_cover(plan.anchors_h, sc, plan.pad_sc, H)
    cols = axis_cover(plan.anchors_w, sc, plan.pad_sc, W)
    for i, (fy, sy, oy, ey) in enumerate(rows):
        for j, (fx, sx, ox, ex_) in enumerate(cols):
            r = tile_results[i * plan.step_w + j]
            y1, x1 = min(ey, H), min(ex_, W)
            ys, xs = np.arange(fy, y1), np.arange(fx, x1)
            rr = r[:, fy - oy:y1 - oy, fx - ox:x1 - ox]
            cur = out[:, fy:y1, fx:x1]
            wh = np.where(ys < sy, ramp[np.clip(ys - fy, 0, max(plan.pad_sc - 1, 0))] if plan.pad_sc else 1.0, np.float32(np.nan))
            ww = np.where(xs < sx, ramp[np.clip(xs - fx, 0, max(plan.pad_sc - 1, 0))] if plan.pad_sc else 1.0, np.float32(np.nan))
            band_y = (ys < sy)[None, :, None]
            band_x = (xs < sx)[None, None, :]
            with np.errstate(invalid='ignore'):
                v1 = np.where(band_y, cur + wh[None, :, None].astype(np.float32) * (rr - cur), rr)
                v = np.where(band_x, cur + ww[None, None, :].astype(np.float32) * (v1 - cur), v1)
            out[:, fy:y1, fx:x1] = v.astype(np.float32)
    return out
