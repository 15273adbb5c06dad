"""Helpers shared by the tests (oracle-side compositions of doCrop: ensemble, planes)."""
import numpy as np

from oracle import planner as oplanner, stitch as ostitch

_T = lambda x: np.swapaxes(x, -1, -2)
_F = lambda x: x[..., ::-1]
_F2 = lambda x: x[..., ::-1, ::-1]
# python/imageProcess.py:568-569 trans / transInv
TRANS = [_T, _F, _F2, lambda x: _T(_F(x)), lambda x: _F(_T(x)), lambda x: _T(_F(_T(x))), lambda x: _T(_F2(x))]
TRANS_INV = [_T, _F, _F2, TRANS[4], TRANS[3], TRANS[5], TRANS[6]]


def oracle_do_crop(x, sc, pad, crop, model, align=8):
    pl = oplanner.prepare(tuple(x.shape), 1 << 40, 1e-3, pad, sc, align, crop)
    return ostitch.do_crop(np.ascontiguousarray(x), pl, sc, model)


def oracle_ensemble(x, n, sc, pad, crop, model):
    """ensemble (python/imageProcess.py:572): doCrop(x) + sum_i transInv_i(doCrop(trans_i(x)))."""
    v = oracle_do_crop(x, sc, pad, crop, model)
    for i in range(n):
        v = v + TRANS_INV[i](oracle_do_crop(TRANS[i](x), sc, pad, crop, model))
    return np.ascontiguousarray(v, dtype=np.float32)
