"""The phase-class-sums form of the fused tail (tests/tailsum_model.py = what conv3x3_rw.hip EPI 3/7 and tapsum4_kernel compute) equals
the 3x3 tail conv of python/models.py:145-154 on random per-tap products, for patch-aligned, ragged and tiny shapes."""
import numpy as np
import pytest

import tailsum_model as tm


@pytest.mark.parametrize('B,H,W', [(1, 8, 32), (2, 16, 64), (1, 24, 40), (2, 9, 36), (1, 8, 8), (1, 40, 100), (3, 17, 33)])
def test_sums_and_aprons_equal_the_tail_conv(B, H, W):
    rng = np.random.default_rng(H * 1000 + W)
    Ts = [rng.standard_normal((4, 9, B, H, W)).astype(np.float32) for _ in range(2)]
    bufs = [tm.producer(T, B, H, W) for T in Ts]
    got = tm.gather(bufs, B, H, W)
    want = tm.direct(Ts, B, H, W)
    assert np.abs(got - want).max() <= 2e-5, float(np.abs(got - want).max())


def test_layout_is_disjoint_and_aligned():
    off, total, py, px = tm.layout(3, 24, 40)
    assert (py, px) == (3, 2)
    names = ['S', 'RA', 'CA', 'CO']
    sizes = {'S': 16 * 3 * 24 * 40, 'RA': 8 * 3 * 3 * 40, 'CA': 8 * 3 * 24 * 2, 'CO': 4 * 3 * 3 * 2}
    for a, b in zip(names, names[1:]):
        assert off[a] + sizes[a] <= off[b] and off[b] % 64 == 0
    assert off['CO'] + sizes['CO'] <= total
