"""The oracle (oracle/) against the golden vectors generated from the reference (tests/golden/,
tools/gen_golden.py).  CPU only.  This is what pins the oracle; the GPU tests then pin the HIP path
to the oracle and to the same goldens."""
import glob
import json
import os

import numpy as np
import pytest

import golden_defs as gd
from moephoto_amd.weights import load_state_dict_file
from oracle import imageio as oio, nets as onets, planner as oplanner, stitch as ostitch

G = gd.GOLDEN
PLANNER = json.load(open(os.path.join(G, 'planner.json')))


def test_get_anchors_known_answers():
    # answers probed from the reference
    a = oplanner.get_anchors(1080, 1080 - 15, 256, 5, 8, 4)
    assert a.start == [0, 251, 497, 743, 992] and a.end == [256, 507, 753, 999, 1080] and a.clip == -324
    assert a.end_sc == [1024, 2028, 3012, 3996, 4320]
    a = oplanner.get_anchors(1920, 1920 - 15, 256, 5, 8, 4)
    assert a.start == [0, 251, 497, 743, 989, 1235, 1481, 1728] and a.clip == -732
    a = oplanner.get_anchors(100, 85, 48, 5, 8, 2)
    assert (a.start, a.end, a.clip) == ([0, 43, 84], [48, 91, 100], -18)
    a = oplanner.get_anchors(250, 235, 256, 5, 8, 2)
    assert a.step == 1 and a.end == [256] and a.end_sc == [500]


@pytest.mark.parametrize('case', PLANNER['anchors'], ids=lambda c: 's{s}_l{l}_p{pad}_a{align}_x{sc}'.format(**c))
def test_get_anchors_golden(case):
    a = oplanner.get_anchors(case['s'], case['ns'], case['l'], case['pad'], case['align'], case['sc'])
    assert (a.start, a.end, a.clip, a.step, a.end_sc) == (case['start'], case['end'], case['clip'], case['step'], case['end_sc'])


@pytest.mark.parametrize('case', PLANNER['prepare'], ids=lambda c: 'x'.join(map(str, c['shape'])) + '_c{}'.format(c['cropsize']))
def test_prepare_golden(case):
    pl = oplanner.prepare(tuple(case['shape']), case['ram'], case['ram_coef'], case['pad'], case['sc'], case['align'], case['cropsize'])
    assert [list(t) for t in pl.tiles] == case['tiles']
    assert list(pl.out_shape) == case['out_shape']
    np.testing.assert_array_equal(oplanner.blend_ramp(pl.pad_sc), np.array(case['ramp'], np.float32))
    # our padded extent covers everything the tiles read (the reference over-allocates zeros beyond it)
    ph = pl.pad.pad_h_to or case['shape'][1]
    pw = pl.pad.pad_w_to or case['shape'][2]
    assert max(t[1] for t in pl.tiles) <= ph <= case['padded_shape'][0]
    assert max(t[3] for t in pl.tiles) <= pw <= case['padded_shape'][1]


def test_config2_grid():
    # 1080p, 256-px tiles, pad 5, x4: 40 tiles (5 x 8), 28 full ones
    pl = oplanner.prepare((3, 1080, 1920), 1 << 40, 1e-3, 5, 4, 8, 256)
    assert len(pl.tiles) == 40 and pl.step_h == 5 and pl.step_w == 8
    assert sum(1 for t in pl.tiles if t[1] - t[0] == 256 and t[3] - t[2] == 256) == 28
    assert pl.tiles[-1][:4] == (992, 1080, 1728, 1920)


NET_KEYS = [os.path.basename(p)[:-4] for p in sorted(glob.glob(os.path.join(G, 'nets', '*.npz')))]


@pytest.mark.parametrize('key', NET_KEYS)
def test_net_forward_golden(key):
    z = np.load(os.path.join(G, 'nets', key + '.npz'))
    arch = gd.MODELS[key][0]
    sd = gd.state_dict_for(key, load_state_dict_file)
    h, w = [int(v) for v in z['hw']]
    seed = int(z['seed'])
    y = onets.forward(arch, sd, gd.noise_image(seed, (3, 1, h, w))).numpy()
    assert np.abs(y - z['y_noise']).max() <= 1e-5
    y = onets.forward(arch, sd, gd.natural_image(seed, (3, h, w))[:, None]).numpy()
    assert np.abs(y - z['y_natural']).max() <= 1e-5


@pytest.mark.parametrize('key', ['a2', 'dn_lite5', 'lite2', 'l25'])
def test_c_backend_matches_torch_backend(key):
    arch = gd.MODELS[key][0]
    sd = gd.state_dict_for(key, load_state_dict_file)
    x = gd.noise_image(3, (2, 1, 20, 24))
    if key == 'l25':
        x = x[:1, :, :16, :16]
    a = onets.forward(arch, sd, x, 'torch').numpy()
    b = onets.forward(arch, sd, x, 'c').numpy()
    assert np.abs(a - b).max() <= 2e-5


def _fake_model(sc):
    calls = []

    def f(s):
        k = len(calls)
        calls.append(0)
        B, _, h, w = s.shape
        return np.random.default_rng(9000 + k).random((B, 1, h * sc, w * sc), dtype=np.float32)
    return f


STITCH_ONLY = [os.path.basename(p)[:-4] for p in sorted(glob.glob(os.path.join(G, 'stitch_only', '*.npz')))]


@pytest.mark.parametrize('name', STITCH_ONLY)
def test_stitch_only_golden(name):
    z = np.load(os.path.join(G, 'stitch_only', name + '.npz'))
    shape = tuple(int(v) for v in z['shape'])
    sc = int(z['sc'])
    pl = oplanner.prepare(shape, 1 << 40, 1e-3, int(z['pad']), sc, int(z['align']), int(z['crop']))
    tiles = []
    y = ostitch.do_crop(gd.noise_image(5, shape), pl, sc, _fake_model(sc), collect=tiles)
    np.testing.assert_array_equal(y, z['y'])           # the sequential restatement is bit-exact
    yf = ostitch.fold_stitch(tiles, pl, sc)
    np.testing.assert_array_equal(yf, z['y'])          # and so is the per-pixel closed form the HIP kernel uses


STITCHED = {'a2_noise': 'a2', 'a2_natural': 'a2', 'a4_natural': 'a4', 'lite2_natural': 'lite2', 'a2_onetile_pad': 'a2'}


@pytest.mark.parametrize('name', sorted(STITCHED))
def test_stitched_golden(name):
    z = np.load(os.path.join(G, 'stitched', name + '.npz'))
    key = STITCHED[name]
    arch, _, sc = gd.MODELS[key]
    shape = tuple(int(v) for v in z['shape'])
    x = gd.noise_image(101, shape) if str(z['kind']) == 'noise' else gd.natural_image(101, shape)
    pl = oplanner.prepare(shape, 1 << 40, 1e-3, 5, sc, 8, int(z['crop']))
    y = ostitch.do_crop(x, pl, sc, onets.model_fn(arch, gd.state_dict_for(key, load_state_dict_file)))
    assert np.abs(y - z['y']).max() <= 1e-5


def test_dn_rgba_strength_golden():
    z = np.load(os.path.join(G, 'stitched', 'dn5_rgba_s06.npz'))
    x = gd.natural_image(101, (4, 64, 80))
    sd = gd.state_dict_for('dn_lite5', load_state_dict_file)
    pl = oplanner.prepare((3, 64, 80), 1 << 40, 1e-3, 7, 1, 8, 48)
    y = ostitch.do_crop(x[:3], pl, 1, onets.model_fn('netdn', sd))
    y = np.float32(0.6) * y + np.float32(1 - 0.6) * x[:3]
    out = np.concatenate([y, x[3:]], 0)
    assert np.abs(out - z['y']).max() <= 1e-5


def test_ensemble_golden():
    from tests_util import oracle_ensemble
    z = np.load(os.path.join(G, 'stitched', 'a2_ens3.npz'))
    x = gd.natural_image(101, (3, 60, 72))
    sd = gd.state_dict_for('a2', load_state_dict_file)
    y = oracle_ensemble(x, 3, 2, 5, 48, onets.model_fn('net2x', sd)) / 4
    assert np.abs(y - z['y']).max() <= 1e-5


def test_e2e_uint8_golden():
    z = np.load(os.path.join(G, 'e2e', 'a2_256_natural.npz'))
    img = gd.to_u8(gd.natural_image(7, (3, 256, 256)))
    x = oio.to_float_image(img)
    sd = gd.state_dict_for('a2', load_state_dict_file)
    pl = oplanner.prepare((3, 256, 256), 1 << 40, 1e-3, 5, 2, 8, 0)
    assert len(pl.tiles) == 1
    y = ostitch.do_crop(x, pl, 2, onets.model_fn('net2x', sd))
    out = oio.to_output(oio.to_hwc(y))
    assert out.dtype == np.uint8 and out.shape == (512, 512, 3)
    # fp32 conv results may differ in the last ulp between hosts: allow the odd pixel to flip by one level
    d = np.abs(out.astype(np.int32) - z['out'].astype(np.int32))
    assert d.max() <= 1 and (d > 0).mean() < 1e-3


def test_to_output_known_answers():
    m = json.load(open(os.path.join(G, 'manifest.json')))
    v = np.array([[[0.998, 0.5, -0.2, 1.7, 0.00390625, 0.0039]]], np.float32)
    assert oio.to_output(v, 8).reshape(-1).tolist() == m['to_output_known'] == [255, 128, 0, 255, 1, 0]
    v16 = np.array([[[0.5, 0.99999, 1.2]]], np.float32)
    assert oio.to_output(v16, 16).reshape(-1).tolist() == m['to_output16_known']


RESIZE = sorted(os.path.basename(p)[:-4] for p in glob.glob(os.path.join(gd.GOLDEN, 'resize', '*.npz')) if 'scale_factors' not in p)


@pytest.mark.parametrize('name', RESIZE)
def test_resize_oracle_vs_reference_golden(name):
    """oracle.resize (numpy restatement of torch's upsample arithmetic) against the reference's resizeByTorch outputs."""
    from oracle import resize as oresize
    z = np.load(os.path.join(gd.GOLDEN, 'resize', name + '.npz'))
    shape = tuple(int(v) for v in z['shape'])
    h, w = [int(v) for v in z['hw']]
    x = gd.natural_image(77, shape) if str(z['kind']) == 'natural' else gd.noise_image(77, shape)
    y = oresize.resize(x, w, h, str(z['method']))
    assert y.shape == z['y'].shape
    assert np.abs(y - z['y']).max() <= (3e-6 if str(z['method']) == 'bicubic' else 2e-7)
