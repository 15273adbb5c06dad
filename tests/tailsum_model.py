"""CPU model of the fused 64->1 tail's "phase-class sums" data path (conv3x3_rw.hip EPI 3/7 -> tapsum4_kernel), index for index.

The last upsampler conv of a branch (python/models.py:29-36: conv 64->256, PixelShuffle(2), PReLU) is followed by the 3x3 64->1 tail
conv (models.py:145-154).  The fused kernel never stores the 64-channel HR tensor: per HR pixel P it forms the nine products
T[k][P] = sum_c Wt[k][c] * act[c][P] (k = 3 dy + dx) and the tail conv is   out[Y][X] = sum_k T[k][Y + dy - 1][X + dx - 1].

One workgroup owns one pixel-shuffle phase (i, j) of an 8 x 32 patch of conv-input pixels, so of the nine values an HR pixel sends
to its neighbours, those that land on output pixels of the SAME parity class can be pre-summed inside the workgroup.  For the phase
(i, j) and an output parity class (ci, cj) (ci = 1: the other row parity, cj = 1: the other column parity):

    sv = +1 if i == 0 else -1       the conv-input row whose taps reach the class from "far":  y + sv
    dyn = 0 if i == 0 else 2        tap row sent from the pixel's own conv-input row to the other row parity;  dyf = 2 - dyn from row y + sv
    (sh, dxn, dxf likewise for columns with j)

    S[0][0][y][x] = T[1][1](y, x)
    S[0][1][y][x] = T[1][dxn](y, x) + T[1][dxf](y, x + sh)
    S[1][0][y][x] = T[dyn][1](y, x) + T[dyf][1](y + sv, x)
    S[1][1][y][x] = T[dyn][dxn](y, x) + T[dyn][dxf](y, x + sh) + T[dyf][dxn](y + sv, x) + T[dyf][dxf](y + sv, x + sh)

    out[2 y' + i'][2 x' + j'] = sum over the four phases (i, j) of S_(i,j)[i' ^ i][j' ^ j][y'][x']          (and over both branches)

Terms whose source pixel lies in ANOTHER patch cannot be added by the workgroup: the patch that owns the source exports them as
aprons (RA: its row facing -sv, CA: its column facing -sh, CO: the corner), and the gather adds them in a fixed order
(S + RA + CA + CO).  Everything is fp32 adds of the same nine products per output pixel as the nine-plane form; only the association
differs.  Used by tests/test_tailsum_model.py; the HIP kernels implement exactly these loops.
"""
import numpy as np

PH, PW = 8, 32          # patch of conv-input pixels (kTileH x kTileW)


def phase_consts(i, j):
    sv, sh = (1 if i == 0 else -1), (1 if j == 0 else -1)
    dyn, dxn = (0 if i == 0 else 2), (0 if j == 0 else 2)
    return sv, sh, dyn, 2 - dyn, dxn, 2 - dxn


def layout(B, H, W):
    """Element offsets of the arrays one branch writes (fp32), in this order: S, RA, CA, CO; returns (offsets dict, total)."""
    py, px = (H + PH - 1) // PH, (W + PW - 1) // PW
    off, pos = {}, 0
    for name, n in (('S', 16 * B * H * W), ('RA', 8 * B * py * W), ('CA', 8 * B * H * px), ('CO', 4 * B * py * px)):
        off[name] = pos
        pos += (n + 63) // 64 * 64
    return off, pos, py, px


def producer(T, B, H, W):
    """T: [4 phases][9 taps][B][H][W] fp32 (zero where the kernel masks: it never sees pixels outside the image).  Returns the flat
    buffer the conv kernel leaves behind, patch by patch like the workgroups do."""
    off, total, py, px = layout(B, H, W)
    buf = np.zeros(total, np.float32)
    S = buf[off['S']:off['S'] + 16 * B * H * W].reshape(4, 4, B, H, W)
    RA = buf[off['RA']:off['RA'] + 8 * B * py * W].reshape(4, 2, B, py, W)
    CA = buf[off['CA']:off['CA'] + 8 * B * H * px].reshape(4, 2, B, H, px)
    CO = buf[off['CO']:off['CO'] + 4 * B * py * px].reshape(4, B, py, px)
    f32 = np.float32
    for ph in range(4):
        i, j = ph >> 1, ph & 1
        sv, sh, dyn, dyf, dxn, dxf = phase_consts(i, j)
        r_exp, c_exp = (0 if sv == 1 else PH - 1), (0 if sh == 1 else PW - 1)
        for b in range(B):
            for pyi in range(py):
                for pxi in range(px):
                    # the patch's tap image, zero outside the image (the kernel masks such lanes)
                    t = np.zeros((9, PH, PW), np.float32)
                    y0, x0 = pyi * PH, pxi * PW
                    hh, ww = min(PH, H - y0), min(PW, W - x0)
                    t[:, :hh, :ww] = T[ph, :, b, y0:y0 + hh, x0:x0 + ww]
                    tap = lambda dy, dx: t[dy * 3 + dx]

                    def sh_c(a):        # a(rho, chi + sh) where inside the patch, else 0
                        o = np.zeros_like(a)
                        if sh == 1:
                            o[:, :-1] = a[:, 1:]
                        else:
                            o[:, 1:] = a[:, :-1]
                        return o

                    def sh_r(a):        # a(rho + sv, chi) where inside the patch, else 0
                        o = np.zeros_like(a)
                        if sv == 1:
                            o[:-1] = a[1:]
                        else:
                            o[1:] = a[:-1]
                        return o
                    s00 = tap(1, 1)
                    s01 = (tap(1, dxn) + sh_c(tap(1, dxf))).astype(f32)
                    s10 = (tap(dyn, 1) + sh_r(tap(dyf, 1))).astype(f32)
                    s11 = (((tap(dyn, dxn) + sh_c(tap(dyn, dxf))).astype(f32) + sh_r(tap(dyf, dxn))).astype(f32) + sh_r(sh_c(tap(dyf, dxf)))).astype(f32)
                    for cls, v in enumerate((s00, s01, s10, s11)):
                        S[ph, cls, b, y0:y0 + hh, x0:x0 + ww] = v[:hh, :ww]
                    # aprons: what the neighbours' edge pixels are missing, from this patch's row r_exp / column c_exp
                    ra0 = tap(dyf, 1)[r_exp]
                    ra1 = (tap(dyf, dxn)[r_exp] + sh_c(tap(dyf, dxf))[r_exp]).astype(f32)
                    RA[ph, 0, b, pyi, x0:x0 + ww] = ra0[:ww]
                    RA[ph, 1, b, pyi, x0:x0 + ww] = ra1[:ww]
                    ca0 = tap(1, dxf)[:, c_exp]
                    ca1 = (tap(dyn, dxf)[:, c_exp] + sh_r(tap(dyf, dxf))[:, c_exp]).astype(f32)
                    CA[ph, 0, b, y0:y0 + hh, pxi] = ca0[:hh]
                    CA[ph, 1, b, y0:y0 + hh, pxi] = ca1[:hh]
                    CO[ph, b, pyi, pxi] = tap(dyf, dxf)[r_exp, c_exp]
    return buf


def gather(bufs, B, H, W):
    """tapsum4: out [B][2H][2W] from the buffers of the branches (list), summation order: branch, phase; per value S + RA + CA + CO."""
    off, total, py, px = layout(B, H, W)
    out = np.zeros((B, 2 * H, 2 * W), np.float32)
    for buf in bufs:
        S = buf[off['S']:off['S'] + 16 * B * H * W].reshape(4, 4, B, H, W)
        RA = buf[off['RA']:off['RA'] + 8 * B * py * W].reshape(4, 2, B, py, W)
        CA = buf[off['CA']:off['CA'] + 8 * B * H * px].reshape(4, 2, B, H, px)
        CO = buf[off['CO']:off['CO'] + 4 * B * py * px].reshape(4, B, py, px)
        for ph in range(4):
            i, j = ph >> 1, ph & 1
            sv, sh = phase_consts(i, j)[:2]
            r_edge, c_edge = (PH - 1 if sv == 1 else 0), (PW - 1 if sh == 1 else 0)
            for ip in range(2):
                for jp in range(2):
                    ci, cj = ip ^ i, jp ^ j
                    v = S[ph, 2 * ci + cj].copy()                       # [B][H][W]
                    ys, xs = np.arange(H), np.arange(W)
                    row_fix = (ys % PH == r_edge) & (ys // PH + sv >= 0) & (ys // PH + sv < py)
                    col_fix = (xs % PW == c_edge) & (xs // PW + sh >= 0) & (xs // PW + sh < px)
                    if ci == 1:
                        for y in ys[row_fix]:
                            v[:, y, :] = (v[:, y, :] + RA[ph, cj, :, y // PH + sv, :]).astype(np.float32)
                    if cj == 1:
                        for x in xs[col_fix]:
                            v[:, :, x] = (v[:, :, x] + CA[ph, ci, :, :, x // PW + sh]).astype(np.float32)
                    if ci == 1 and cj == 1:
                        for y in ys[row_fix]:
                            for x in xs[col_fix]:
                                v[:, y, x] = (v[:, y, x] + CO[ph, :, y // PH + sv, x // PW + sh]).astype(np.float32)
                    out[:, ip::2, jp::2] = (out[:, ip::2, jp::2] + v).astype(np.float32)
    return out


def direct(Ts, B, H, W):
    """The tail conv itself from per-phase taps: out[Y][X] = sum_k T_hr[k][Y + dy - 1][X + dx - 1] (float64 reference)."""
    out = np.zeros((B, 2 * H, 2 * W), np.float64)
    for T in Ts:
        hr = np.zeros((9, B, 2 * H, 2 * W), np.float64)
        for ph in range(4):
            hr[:, :, (ph >> 1)::2, (ph & 1)::2] = T[ph]
        p = np.pad(hr, ((0, 0), (0, 0), (1, 1), (1, 1)))
        for dy in range(3):
            for dx in range(3):
                out += p[dy * 3 + dx, :, dy:dy + 2 * H, dx:dx + 2 * W]
    return out
