"""Oracle: the pipeline's `resize` step.  TEST INFRASTRUCTURE (see oracle/__init__.py).

resizeByTorch (python/imageProcess.py:555-556) = F.interpolate(x[None], size=(h, w), mode=mode, align_corners=False)[0].
Restated in numpy float32 in the operation order of torch's CPU upsample kernels (aten/native UpSample*.h:
area_pixel_compute_source_index, compute_indices_weights): scale = in / out as float32, src = scale * (dst + 0.5) - 0.5,
bilinear clamps src at 0 and the upper neighbour at in - 1, bicubic (A = -0.75) clamps the four tap indices, nearest takes
min(floor(dst * scale), in - 1).  Pinned against outputs of the imported reference (tests/golden/resize/*.npz)."""
import numpy as np

f32 = np.float32


def _src(dst, scale):
    return (scale * (dst.astype(f32) + f32(0.5)) - f32(0.5)).astype(f32)


def _cubic_coeffs(t):
    A = f32(-0.75)
    c1 = lambda x: ((A + f32(2)) * x - (A + f32(3))) * x * x + f32(1)
    c2 = lambda x: ((A * x - f32(5) * A) * x + f32(8) * A) * x - f32(4) * A
    return [c2(t + f32(1)), c1(t), c1(f32(1) - t), c2(f32(2) - t)]


def resize(x, width, height, mode='bilinear'):
    x = np.asarray(x, dtype=np.float32)
    C, H, W = x.shape
    sy, sx = f32(H) / f32(height), f32(W) / f32(width)
    ys, xs = np.arange(height), np.arange(width)
    if mode == 'nearest':
        iy = np.minimum(np.floor(ys.astype(f32) * sy).astype(np.int64), H - 1)
        ix = np.minimum(np.floor(xs.astype(f32) * sx).astype(np.int64), W - 1)
        return x[:, iy][:, :, ix].copy()
    if mode == 'bilinear':
        fy, fx = np.maximum(_src(ys, sy), f32(0)), np.maximum(_src(xs, sx), f32(0))
        y0, x0 = fy.astype(np.int64), fx.astype(np.int64)
        y1, x1 = y0 + (y0 < H - 1), x0 + (x0 < W - 1)
        ly, lx = (fy - y0.astype(f32)).astype(f32), (fx - x0.astype(f32)).astype(f32)
        hy, hx = (f32(1) - ly).astype(f32), (f32(1) - lx).astype(f32)
        r0 = hx[None, None, :] * x[:, y0][:, :, x0] + lx[None, None, :] * x[:, y0][:, :, x1]
        r1 = hx[None, None, :] * x[:, y1][:, :, x0] + lx[None, None, :] * x[:, y1][:, :, x1]
        return (hy[None, :, None] * r0 + ly[None, :, None] * r1).astype(np.float32)
    if mode == 'bicubic':
        fy, fx = _src(ys, sy), _src(xs, sx)
        gy, gx = np.floor(fy), np.floor(fx)
        cy, cx = _cubic_coeffs((fy - gy).astype(f32)), _cubic_coeffs((fx - gx).astype(f32))
        iy, ix = gy.astype(np.int64), gx.astype(np.int64)
        out = np.zeros((C, height, width), np.float32)
        for i in range(4):
            yy = np.clip(iy - 1 + i, 0, H - 1)
            row = np.zeros((C, height, width), np.float32)
            for j in range(4):
                xx = np.clip(ix - 1 + j, 0, W - 1)
                row += x[:, yy][:, :, xx] * cx[j][None, None, :]
            out += row * cy[i][None, :, None]
        return out
    raise ValueError('unknown mode ' + mode)
