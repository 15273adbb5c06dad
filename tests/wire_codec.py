"""numpy restatement of the wire format's pack / unpack passes (moe_wire_pack / moe_wire_unpack, include/moephoto_amd.h): the checker of the HIP kernels in
the GPU tests, and the codec the gloo CPU tests plug into moephoto_amd.dist (dist.CPU_CODEC) -- test infrastructure, the product packs on the GPU."""
import numpy as np


def _views(w, r):
    C, th, tw = int(r['C']), int(r['th']), int(r['tw'])
    rows = list(range(int(r['ra0']), int(r['ra1']))) + list(range(int(r['rb0']), int(r['rb1'])))
    cols = list(range(int(r['ca0']), int(r['ca1']))) + list(range(int(r['cb0']), int(r['cb1'])))
    n = C * th * tw
    raw = len(rows) >= th
    wo = int(r['wire_off'])
    hw = 0 if raw else (n + 1) // 2
    h16 = w[wo:wo + hw].view(np.float16)[:n]
    o = wo + hw
    rv = w[o:o + C * len(rows) * tw].view(np.float32).reshape(C, len(rows), tw)
    o += C * len(rows) * tw
    nc = 0 if raw else len(cols)
    cv = w[o:o + C * th * nc].view(np.float32).reshape(C, th, nc)
    return (C, th, tw), rows, cols if not raw else [], raw, h16, rv, cv


def codec(pack, buf, words, recs):
    """buf: 1-D fp32 CPU tensor (or array) of tiles, words: 1-D int32 CPU tensor (or array) of the wire, recs: numpy records (dist.WIRE_REC)"""
    b = buf.numpy() if hasattr(buf, 'numpy') else buf
    w = words.numpy() if hasattr(words, 'numpy') else words
    for r in recs:
        shape, rows, cols, raw, h16, rv, cv = _views(w, r)
        n = shape[0] * shape[1] * shape[2]
        t = b[int(r['tile_off']):int(r['tile_off']) + n].reshape(shape)
        if pack:
            if not raw:
                with np.errstate(over='ignore'):
                    h16[:] = t.reshape(-1).astype(np.float16)
                cv[:] = t[:, :, cols]
            rv[:] = t[:, rows, :]
        else:
            if not raw:
                t[:] = h16.astype(np.float32).reshape(shape)
                t[:, :, cols] = cv
            t[:, rows, :] = rv
