#!/usr/bin/env python
"""Generate tests/golden/* by running the MoePhoto *reference* (read-only, /root/reference) on
seeded inputs.  Runs ONLY in the build container (the GPU box has no reference).  The reference's
own tests pin nothing on this path, so these vectors -- outputs of the
reference's PyTorch-CPU fp32 path -- are what the oracle, and through it the HIP path, is pinned to.

    python tools/gen_golden.py            # regenerate everything (about a minute)
    python tools/gen_golden.py nets:p2 resize   # only the named sections (manifest.json is merged, not rewritten)

Stored: outputs only (inputs/weights are re-derived from seeds, see tests/golden_defs.py).
While generating, the oracle restatement is compared against the reference (final outputs and
layer-by-layer intermediates); the observed max-abs differences are recorded in manifest.json.
"""
import hashlib
import json
import os
import sys
import tempfile

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, 'tools'))
sys.path.insert(0, os.path.join(ROOT, 'tests'))

import numpy as np  # noqa: E402
import torch  # noqa: E402

import golden_defs as gd  # noqa: E402
from ref_import import load_reference  # noqa: E402
from moephoto_amd.weights import load_state_dict_file, save_state_dict_file  # noqa: E402
from oracle import nets as onets, planner as oplanner, stitch as ostitch, imageio as oio  # noqa: E402

OUT = gd.GOLDEN
manifest = {'torch': torch.__version__, 'threads': torch.get_num_threads(), 'files': {}, 'oracle_vs_reference': {}}


def save_npz(name, **arrs):
    p = os.path.join(OUT, name)
    os.makedirs(os.path.dirname(p), exist_ok=True)
    np.savez_compressed(p, **arrs)
    manifest['files'][name] = hashlib.sha256(open(p, 'rb').read()).hexdigest()[:16]


def ref_model(r, key, sd):
    arch = gd.MODELS[key][0]
    ctor = {'net2x': r.models.Net2x, 'net3x': r.models.Net3x, 'net4x': r.models.Net4x, 'netdn': r.models.NetDN,
            'sedn': r.models.SEDN, 'lite2': r.MoeNet_lite2.Net, 'lite4': lambda: r.MoeNet_lite2.Net(upscale=4),
            'lite8': lambda: r.MoeNet_lite2.Net(upscale=8)}[arch]
    m = ctor()
    m.load_state_dict(OrderedDictT(sd))
    for p in m.parameters():
        p.requires_grad_(False)
    return m.eval()


def OrderedDictT(sd):
    from collections import OrderedDict
    return OrderedDict((k, torch.from_numpy(np.array(v))) for k, v in sd.items())


def gen_planner(r):
    ip = r.imageProcess
    cases = []
    fixed = [(1080, 256, 5, 8, 4), (1920, 256, 5, 8, 4), (100, 48, 5, 8, 2), (140, 48, 5, 8, 2), (293, 256, 5, 8, 2),
             (250, 256, 5, 8, 2), (2160, 256, 7, 8, 1), (3840, 256, 7, 8, 1), (4320, 512, 5, 8, 4), (7680, 512, 5, 8, 4),
             (64, 48, 9, 8, 3), (41, 40, 5, 8, 2), (48, 48, 7, 8, 1), (49, 48, 7, 8, 1)]
    rng = np.random.default_rng(77)
    for _ in range(40):
        pad = int(rng.choice([5, 7, 9]))
        align = int(rng.choice([8, 16]))
        l = int(oplanner.ceil_by(align, int(rng.integers(28 + 2 * pad, 300))))
        s = int(rng.integers(8, 1200))
        fixed.append((s, l, pad, align, int(rng.choice([1, 2, 3, 4]))))
    for s, l, pad, align, sc in fixed:
        ns = max(1, s - 3 * pad)
        start, end, clip, step, endsc = ip.getAnchors(s, ns, l, pad, ip.alignF[align], sc)
        cases.append(dict(s=s, ns=ns, l=l, pad=pad, align=align, sc=sc, start=start, end=end, clip=clip, step=step, end_sc=endsc))
    prep = []
    shapes = [((3, 1080, 1920), 256, 5, 4, 8, 1 << 40, 1e-3), ((3, 100, 140), 48, 5, 2, 8, 1 << 40, 1e-3),
              ((3, 2160, 3840), 256, 7, 1, 8, 1 << 40, 1e-3), ((3, 4320, 7680), 512, 5, 4, 8, 1 << 40, 1e-3),
              ((3, 250, 300), 256, 5, 2, 8, 1 << 40, 1e-3), ((3, 300, 250), 256, 5, 2, 8, 1 << 40, 1e-3),
              ((3, 250, 250), 256, 5, 2, 8, 1 << 40, 1e-3), ((4, 77, 33), 0, 5, 2, 8, 1 << 40, 1e-3),
              ((3, 1080, 1920), 0, 5, 4, 8, 48 << 30, float(r.runSR.ramCoef[2][0])),
              ((3, 1080, 1920), 0, 5, 4, 8, 200 << 30, float(r.runSR.ramCoef[2][2])),
              ((3, 720, 1280), 0, 7, 1, 8, 8 << 30, float(r.runDN.ramCoef[0][2])),
              ((3, 600, 800), 128, 9, 3, 8, 1 << 40, 1e-3), ((1, 64, 64), 256, 5, 2, 8, 1 << 40, 1e-3),
              ((3, 30, 500), 64, 5, 2, 8, 1 << 40, 1e-3)]
    for shape, crop, pad, sc, align, ram, coef in shapes:
        opt = ip.Option()
        opt.ramCoef, opt.fixChannel = coef, 0
        it, pad_image, unpad, out_shape, b = ip.prepare(shape, ram, opt, pad, sc, align, crop)
        probe = torch.zeros(shape)
        prep.append(dict(shape=list(shape), cropsize=crop, pad=pad, sc=sc, align=align, ram=ram, ram_coef=coef,
                         tiles=[list(map(int, t)) for t in it()], out_shape=list(map(int, out_shape)),
                         padded_shape=list(pad_image(probe.unsqueeze(1)).shape[-2:]),
                         ramp=[float(v) for v in b.reshape(-1).tolist()]))
    with open(os.path.join(OUT, 'planner.json'), 'w') as f:
        json.dump(dict(anchors=cases, prepare=prep), f, indent=0)
    # oracle check
    for c in cases:
        a = oplanner.get_anchors(c['s'], c['ns'], c['l'], c['pad'], c['align'], c['sc'])
        assert (a.start, a.end, a.clip, a.step, a.end_sc) == (c['start'], c['end'], c['clip'], c['step'], c['end_sc']), c
    for p in prep:
        pl = oplanner.prepare(tuple(p['shape']), p['ram'], p['ram_coef'], p['pad'], p['sc'], p['align'], p['cropsize'])
        assert [list(t) for t in pl.tiles] == p['tiles'], p
    print('planner: {} anchor cases, {} prepare cases; oracle identical'.format(len(cases), len(prep)))


NET_CASES = [  # key, seed, (h, w)
    ('a2', 11, (40, 48)), ('p2', 21, (40, 48)), ('a3', 12, (40, 48)), ('a4', 13, (40, 48)), ('dn_lite5', 14, (40, 48)), ('dn_lite10', 15, (40, 48)),
    ('dn_lite15', 16, (40, 48)), ('l25', 17, (40, 48)), ('lite2', 18, (40, 48)), ('lite4', 19, (40, 48)), ('lite8', 20, (16, 24)),
]


def gen_nets(r, only=None):
    for key, seed, (h, w) in NET_CASES:
        if only and key not in only:
            continue
        arch = gd.MODELS[key][0]
        sd = gd.state_dict_for(key, load_state_dict_file)
        m = ref_model(r, key, sd)
        x = gd.noise_image(seed, (3, 1, h, w))
        xs = gd.natural_image(seed, (3, h, w))[:, None]
        taps_ref = {}
        hooks = []
        if arch in ('net2x', 'net3x', 'net4x', 'netdn'):
            for i, f in enumerate(m.convt_F):
                hooks.append(f.register_forward_hook(lambda mod, inp, out, i=i: taps_ref.__setitem__('arsb{}'.format(i + 1), out.detach().clone())))
            hooks.append(m.conv_input2.register_forward_hook(lambda mod, inp, out: taps_ref.__setitem__('input2', out.detach().clone())))
        with torch.no_grad():
            y = m(torch.from_numpy(x))[-1].numpy()
        for hk in hooks:
            hk.remove()
        with torch.no_grad():
            ys = m(torch.from_numpy(xs))[-1].numpy()
        save_npz('nets/{}.npz'.format(key), y_noise=y, y_natural=ys, seed=np.int64(seed), hw=np.array([h, w]))
        taps = {}
        yo = onets.forward(arch, sd, x, 'torch', taps).numpy()
        d = float(np.abs(yo - y).max())
        dt = max([float((taps[k] - taps_ref[k]).abs().max()) for k in taps_ref] + [0.0])
        yc = onets.forward(arch, sd, x[:1, :, :24, :24], 'c').numpy()
        with torch.no_grad():
            dc = float(np.abs(yc - m(torch.from_numpy(x[:1, :, :24, :24]))[-1].numpy()).max())
        manifest['oracle_vs_reference']['net:' + key] = dict(out=d, intermediates=dt, c_backend=dc,
                                                             range=[float(y.min()), float(y.max())])
        print('net {:10s} oracle-vs-ref out {:.2e} taps {:.2e} c-backend {:.2e} range [{:.3f},{:.3f}]'.format(key, d, dt, dc, y.min(), y.max()))


def with_synth_zoo(r, tmp):
    """Point the plugin tables at the fixture zoo + synthetic files written by OUR legacy writer
    (so the reference's own torch.load path reads them: format compatibility)."""
    for key in ('a3', 'a4'):
        p = os.path.join(tmp, key + '.pth')
        save_state_dict_file(gd.synth_state_dict(key, load_state_dict_file), p)
        old = r.runSR.mode_switch[key]
        r.runSR.mode_switch[key] = (p,) + tuple(old[1:])
    p = os.path.join(tmp, 'l25.pth')
    save_state_dict_file(gd.synth_state_dict('l25', load_state_dict_file), p)
    old = r.runDN.mode_switch['25']
    r.runDN.mode_switch['25'] = (p,) + tuple(old[1:])


def plan_from_opt(shape, opt, r):
    pl = oplanner.prepare(tuple(shape), 1 << 40, opt.ramCoef, opt.padding, opt.scale, opt.align, opt.cropsize)
    assert [tuple(t) for t in opt.iterClip()] == pl.tiles
    return pl


def gen_stitched(r):
    cases = [  # name, op, step dict, crop, image (C,H,W), kind
        ('a2_noise', 'SR', dict(model='a', scale=2, ensemble=0), 48, (3, 100, 140), 'noise'),
        ('a2_natural', 'SR', dict(model='a', scale=2, ensemble=0), 48, (3, 100, 140), 'natural'),
        ('dn10_noise', 'DN', dict(model='lite10'), 48, (3, 100, 140), 'noise'),
        ('a4_natural', 'SR', dict(model='a', scale=4, ensemble=0), 48, (3, 60, 100), 'natural'),
        ('lite2_natural', 'SR', dict(model='lite', scale=2, ensemble=0), 48, (3, 60, 100), 'natural'),
        ('a2_onetile_pad', 'SR', dict(model='a', scale=2, ensemble=0), 64, (3, 45, 100), 'natural'),
        ('a2_ens3', 'SR', dict(model='a', scale=2, ensemble=3), 48, (3, 60, 72), 'natural'),
        ('a2_ens7', 'SR', dict(model='a', scale=2, ensemble=7), 48, (3, 52, 60), 'noise'),
        ('dn5_rgba_s06', 'DN', dict(model='lite5', strength=0.6), 48, (4, 64, 80), 'natural'),
        ('l25_natural', 'DN', dict(model='25'), 48, (3, 60, 72), 'natural'),
    ]
    for name, op, step, crop, shape, kind in cases:
        load_reference(crop_sr=crop, crop_dn=crop, crop_dns=crop)
        x = (gd.noise_image(101, shape) if kind == 'noise' else gd.natural_image(101, shape))
        xt = torch.from_numpy(x)
        if op == 'SR':
            opt = r.runSR.getOpt(dict(op='SR', **step))
            y = r.runSR.sr(opt)(xt)
        else:
            opt = r.runDN.getOpt(dict(op='DN', **step))
            y = r.imageProcess.RGBFilter(opt)(xt)
        y = y.numpy()
        save_npz('stitched/{}.npz'.format(name), y=y, crop=np.int64(crop), shape=np.array(shape), kind=np.array(kind),
                 step=np.array(json.dumps(step)), op=np.array(op))
        # oracle check (plain doCrop cases only; ensemble / RGBFilter are checked by the tests via the host mirror)
        if step.get('ensemble', 0) == 0 and op == 'SR':
            key = {('a', 2): 'a2', ('a', 4): 'a4', ('lite', 2): 'lite2'}[(step['model'], step['scale'])]
            sd = gd.state_dict_for(key, load_state_dict_file)
            pl = plan_from_opt(shape, opt, r)
            yo = ostitch.do_crop(x, pl, opt.scale, onets.model_fn(gd.MODELS[key][0], sd))
            d = float(np.abs(yo - y).max())
            manifest['oracle_vs_reference']['stitched:' + name] = d
            print('stitched {:16s} tiles {:3d} oracle-vs-ref {:.2e} mean {:.6f}'.format(name, len(pl.tiles), d, y.mean()))
        else:
            print('stitched {:16s} mean {:.6f}'.format(name, y.mean()))


def gen_stitch_only(r):
    """doCrop with a fake model (seeded random tile outputs): isolates planner + blend/stitch."""
    ip = r.imageProcess
    cases = [('s2_multi', (2, 70, 90), 40, 5, 2, 8), ('s4_multi', (3, 61, 83), 40, 5, 4, 8), ('s1_dn', (3, 90, 75), 48, 7, 1, 8),
             ('s3_multi', (1, 80, 100), 48, 9, 3, 8), ('s2_onetile_w', (2, 100, 37), 40, 5, 2, 8), ('s2_onetile_h', (2, 30, 100), 40, 5, 2, 8),
             ('s2_onetile', (3, 33, 37), 40, 5, 2, 8), ('s2_tiny_last', (1, 83, 83), 48, 5, 2, 8), ('s4_three', (1, 120, 130), 48, 5, 4, 8)]
    for name, shape, crop, pad, sc, align in cases:
        load_reference(crop_sr=crop)
        opt = ip.Option()
        opt.fixChannel, opt.padding, opt.scale, opt.align, opt.cropsize, opt.ramCoef = 0, pad, sc, align, crop, 1e-3
        opt.squeeze = lambda t: t.squeeze(1)
        opt.unsqueeze = lambda t: t.unsqueeze(1)
        calls = []

        def fake(s):
            k = len(calls)
            B, _, h, w = s.shape
            out = torch.from_numpy(np.random.default_rng(9000 + k).random((B, 1, h * sc, w * sc), dtype=np.float32))
            calls.append(out.numpy()[:, 0].copy())
            return out
        opt.modelCached = fake
        x = torch.from_numpy(gd.noise_image(5, shape))
        y = ip.doCrop(opt, x).numpy()
        save_npz('stitch_only/{}.npz'.format(name), y=y, shape=np.array(shape), crop=np.int64(crop), pad=np.int64(pad),
                 sc=np.int64(sc), align=np.int64(align))
        pl = oplanner.prepare(shape, 1 << 40, 1e-3, pad, sc, align, crop)
        assert [tuple(t) for t in opt.iterClip()] == pl.tiles
        kk = []

        def fake_o(s):
            k = len(kk)
            kk.append(0)
            B, _, h, w = s.shape
            return np.random.default_rng(9000 + k).random((B, 1, h * sc, w * sc), dtype=np.float32)
        tile_out = []
        yo = ostitch.do_crop(x.numpy(), pl, sc, fake_o, collect=tile_out)
        yf = ostitch.fold_stitch(tile_out, pl, sc)
        d, df = float(np.abs(yo - y).max()), float(np.abs(yf - y).max())
        assert not np.isnan(yo).any() and not np.isnan(yf).any()
        manifest['oracle_vs_reference']['stitch_only:' + name] = dict(sequential=d, fold=df, tiles=len(pl.tiles))
        print('stitch-only {:14s} tiles {:3d} seq {:.2e} fold {:.2e}'.format(name, len(pl.tiles), d, df))


def gen_e2e(r):
    """Config 1: 256x256 RGB uint8 -> a2 x2 -> uint8 (toTorch -> sr -> toFloat -> toOutput)."""
    ip = r.imageProcess
    load_reference(crop_sr=0)
    opt = r.runSR.getOpt(dict(op='SR', model='a', scale=2, ensemble=0))
    for kind in ('natural', 'noise'):
        img = gd.to_u8(gd.natural_image(7, (3, 256, 256))) if kind == 'natural' else gd.noise_u8(0, (256, 256, 3))
        x = ip.toTorch(8, torch.float, torch.device('cpu'))(img)
        y = r.runSR.sr(opt)(x)
        o = ip.toOutput(8)(ip.toFloat(y))
        save_npz('e2e/a2_256_{}.npz'.format(kind), out=o)
        xo = oio.to_float_image(img)
        assert np.array_equal(xo, x.numpy())
        oo = oio.to_output(oio.to_hwc(y.numpy()))
        assert np.array_equal(oo, o)
        print('e2e', kind, o.shape, o.dtype, 'sha', hashlib.sha256(o.tobytes()).hexdigest()[:16], 'tiles', len(list(opt.iterClip())))
    # quantisation known answers (SURVEY section 8a row IO)
    q = ip.toOutput(8)(torch.tensor([[[0.998, 0.5, -0.2, 1.7, 0.00390625, 0.0039]]]))
    manifest['to_output_known'] = q.reshape(-1).tolist()
    q16 = ip.toOutput(16)(torch.tensor([[[0.5, 0.99999, 1.2]]]))
    manifest['to_output16_known'] = q16.reshape(-1).tolist()


RESIZE_CASES = [  # name, (C, H, W), (h, w), method, input kind
    ('bilinear_up', (3, 24, 40), (50, 77), 'bilinear', 'natural'), ('bilinear_down', (3, 45, 64), (20, 31), 'bilinear', 'noise'),
    ('bilinear_2x', (3, 16, 24), (32, 48), 'bilinear', 'natural'), ('nearest_up', (3, 17, 23), (40, 50), 'nearest', 'noise'),
    ('nearest_down', (4, 40, 56), (13, 19), 'nearest', 'natural'), ('bicubic_up', (3, 20, 28), (45, 61), 'bicubic', 'natural'),
    ('bicubic_down', (3, 48, 40), (21, 18), 'bicubic', 'noise'), ('bilinear_720p_row', (1, 9, 1920), (6, 1280), 'bilinear', 'natural'),
]


def gen_resize(r):
    """The `resize` step (python/imageProcess.py:174-195 through resizeByTorch :555-556) on seeded inputs, plus one pass through the
    reference's own `resize(opt, out)` closure with scale factors (rounding of the target size)."""
    from oracle import resize as oresize
    ip = r.imageProcess
    worst = 0.0
    for name, shape, (h, w), method, kind in RESIZE_CASES:
        x = gd.natural_image(77, shape) if kind == 'natural' else gd.noise_image(77, shape)
        if method == 'nearest':
            # the reference passes align_corners=False with every method, which torch refuses for 'nearest' (ValueError): its UI's
            # "nearest" choice cannot run.  The vector is torch's own nearest (no align_corners), which is what the option means.
            try:
                ip.resizeByTorch(torch.from_numpy(x), w, h, method)
                manifest['reference_nearest_raises'] = False
            except ValueError:
                manifest['reference_nearest_raises'] = True
            y = torch.nn.functional.interpolate(torch.from_numpy(x).unsqueeze(0), size=(h, w), mode='nearest').squeeze(0).numpy()
        else:
            y = ip.resizeByTorch(torch.from_numpy(x), w, h, method).numpy().reshape(shape[0], h, w)
        save_npz('resize/{}.npz'.format(name), y=y, shape=np.array(shape), hw=np.array([h, w]), method=np.array(method), kind=np.array(kind))
        d = float(np.abs(oresize.resize(x, w, h, method) - y).max())
        worst = max(worst, d)
        print('resize {:18s} oracle-vs-ref {:.2e}'.format(name, d))
    x = gd.natural_image(78, (3, 33, 47))
    f = ip.resize({'scaleH': 1.7, 'scaleW': 0.6}, {'source': False})
    y = f(torch.from_numpy(x)).numpy()
    save_npz('resize/scale_factors.npz', y=y, shape=np.array([3, 33, 47]), scale=np.array([1.7, 0.6]))
    manifest['oracle_vs_reference']['resize'] = worst


def main():
    os.makedirs(OUT, exist_ok=True)
    r = load_reference()
    tmp = tempfile.mkdtemp()
    with_synth_zoo(r, tmp)
    sel = sys.argv[1:]
    if sel:       # partial regeneration: merge into the existing manifest
        old = json.load(open(os.path.join(OUT, 'manifest.json')))
        for s_ in sel:
            if s_.startswith('nets:'):
                gen_nets(r, only=s_[5:].split(','))
            elif s_ == 'resize':
                gen_resize(r)
            else:
                raise SystemExit('unknown section ' + s_)
        old['files'].update(manifest['files'])
        old['oracle_vs_reference'].update(manifest['oracle_vs_reference'])
        with open(os.path.join(OUT, 'manifest.json'), 'w') as f:
            json.dump(old, f, indent=1, sort_keys=True)
        print('updated', OUT)
        return
    gen_planner(r)
    gen_nets(r)
    gen_stitch_only(r)
    gen_stitched(r)
    gen_e2e(r)
    gen_resize(r)
    with open(os.path.join(OUT, 'manifest.json'), 'w') as f:
        json.dump(manifest, f, indent=1, sort_keys=True)
    print('wrote', OUT)


if __name__ == '__main__':
    main()
