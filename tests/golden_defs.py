"""Shared definitions of the seeded inputs and synthetic weights behind tests/golden/.

Used by tools/gen_golden.py (which runs the *reference* on these inputs, in the build container)
and by the tests (which run the oracle / the HIP path on the same inputs).  Everything is derived
from numpy `default_rng` seeds or closed forms, so only outputs are stored as fixtures.
"""
import os
from collections import OrderedDict

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
GOLDEN = os.path.join(HERE, 'golden')
ZOO = os.path.join(GOLDEN, 'zoo')            # copies of the reference's weight *data* files

# model key -> (arch name used by oracle/engine, zoo-relative path or None when synthetic, scale)
MODELS = OrderedDict([
    ('a2', ('net2x', 'model/a2/model_new.pth', 2)),
    ('p2', ('net2x', 'model/p2/model_new.pth', 2)),
    ('a3', ('net3x', None, 3)),
    ('a4', ('net4x', None, 4)),
    ('dn_lite5', ('netdn', 'model/dn_lite5/model_new.pth', 1)),
    ('dn_lite10', ('netdn', 'model/dn_lite10/model_new.pth', 1)),
    ('dn_lite15', ('netdn', 'model/dn_lite15/model_new.pth', 1)),
    ('l25', ('sedn', None, 1)),
    ('lite2', ('lite2', 'model/lite/model.pth', 2)),
    ('lite4', ('lite4', 'model/lite/model_4.pth', 4)),
    ('lite8', ('lite8', 'model/lite/model_8.pth', 8)),
])


def noise_image(seed, shape):
    """Uniform [0,1) fp32 -- the adversarial input (output range of a2 reaches [-0.5, 1.8])."""
    return np.random.default_rng(seed).random(shape, dtype=np.float32)


def noise_u8(seed, shape):
    return np.random.default_rng(seed).integers(0, 256, size=shape, dtype=np.uint8)


def smooth_image(shape):
    """Closed-form smooth image in [0.1, 0.9]; channel c is phase shifted."""
    c, h, w = shape
    y, x = np.meshgrid(np.linspace(0, 1, h, dtype=np.float64), np.linspace(0, 1, w, dtype=np.float64), indexing='ij')
    return np.stack([0.5 + 0.4 * np.sin(9 * x + 3 * y + 0.7 * k) for k in range(c)]).astype(np.float32)


def natural_image(seed, shape):
    """Natural-image-like synthetic: low-frequency shading + hard edges + fine texture + mild
    sensor noise, clipped to [0,1].  This is the regime the SR/DN nets are trained for."""
    c, h, w = shape
    rng = np.random.default_rng(seed)
    y, x = np.meshgrid(np.arange(h, dtype=np.float64), np.arange(w, dtype=np.float64), indexing='ij')
    planes = []
    for k in range(c):
        base = 0.5 + 0.22 * np.sin(x / 37.0 + 0.3 * k) * np.cos(y / 29.0 - 0.2 * k)
        edges = 0.18 * (((x * 0.8 + y * 0.45 + 11 * k) % 64) > 32)
        blobs = 0.12 * (np.sin(x / 5.0 + k) * np.sin(y / 7.0) > 0.55)
        tex = 0.04 * np.sin(x * 1.7 + y * 2.3 + k)
        noise = 0.012 * rng.standard_normal((h, w))
        planes.append(base + edges + blobs + tex + noise)
    return np.clip(np.stack(planes), 0, 1).astype(np.float32)


def to_u8(img_chw):
    return np.clip(np.round(np.asarray(img_chw).transpose(1, 2, 0) * 255), 0, 255).astype(np.uint8)


def zoo_path(rel):
    return os.path.join(ZOO, rel)


def _he(rng, shape):
    # initConvParameters (models.py:21-27): normal(0, sqrt(2 / (k*k*out_channels)))
    o, _, k, _ = shape
    return (rng.standard_normal(shape) * np.sqrt(2.0 / (k * k * o))).astype(np.float32)


def synth_state_dict(key, load):
    """Deterministic stand-ins, in the zoo's exact key schema, for weight files that are absent
    from the reference mount (.MISSING_LARGE_BLOBS): a3, a4 (Net3x/Net4x) and l25 (SEDN).
    `load(path)` reads a zoo file into an OrderedDict of fp32 arrays."""
    if key == 'a4':
        # a2's trunk; its upsampler stage duplicated as stage 1; its tail moved to index 2
        a2 = load(zoo_path(MODELS['a2'][1]))
        sd = OrderedDict()
        for k, v in a2.items():
            if k.startswith(('u.', 'convt_R1.')):
                br, rest = k.split('.', 1)
                if rest.startswith('0.'):
                    sd[k] = v.copy()
                    sd['{}.1.{}'.format(br, rest[2:])] = (v * np.float32(0.9)).astype(np.float32)
                elif rest == '1.weight':
                    sd['{}.2.weight'.format(br)] = v.copy()
            else:
                sd[k] = v.copy()
        # zoo ordering: conv_input, conv_input2, relu, u.*, convt_R1.*, convt_F*
        return _ordered(sd, 3)
    if key == 'a3':
        a2 = load(zoo_path(MODELS['a2'][1]))
        rng = np.random.default_rng(303)
        sd = OrderedDict()
        for k, v in a2.items():
            if k.endswith('.0.0.weight'):
                sd[k] = (_he(rng, (576, 64, 3, 3)) * np.float32(0.8)).astype(np.float32)
            elif k.endswith('.0.0.bias'):
                sd[k] = (rng.standard_normal(576) * 0.01).astype(np.float32)
            else:
                sd[k] = v.copy()
        return sd
    if key == 'l25':
        rng = np.random.default_rng(2525)
        sd = OrderedDict()
        sd['conv_input.weight'] = _he(rng, (64, 1, 3, 3))
        sd['convt_R1.weight'] = (_he(rng, (1, 64, 3, 3)) * np.float32(0.05)).astype(np.float32)
        for b in range(16):
            p = 'convt_F1.{}.'.format(b)
            sd[p + 'rblock.0.weight'] = _he(rng, (64, 64, 3, 3))
            sd[p + 'rblock.2.weight'] = _he(rng, (64, 64, 3, 3))
            sd[p + 'rblock.4.weight'] = _he(rng, (256, 64, 3, 3))
            sd[p + 'trans.0.weight'] = (_he(rng, (64, 256, 1, 1)) * np.float32(0.35)).astype(np.float32)
            sd[p + 'conv_down.weight'] = _he(rng, (16, 256, 1, 1))
            sd[p + 'conv_up.weight'] = _he(rng, (256, 16, 1, 1))
        return sd
    raise KeyError(key)


def _ordered(sd, stages):
    order = ['conv_input.weight', 'conv_input2.weight', 'relu.weight']
    for br in ('u', 'convt_R1'):
        for s in range(stages - 1):
            order += ['{}.{}.0.weight'.format(br, s), '{}.{}.0.bias'.format(br, s), '{}.{}.2.weight'.format(br, s)]
        order.append('{}.{}.weight'.format(br, stages - 1))
    rest = [k for k in sd if k not in order]
    return OrderedDict((k, sd[k]) for k in order + rest)


def state_dict_for(key, load):
    arch, rel, _ = MODELS[key]
    return load(zoo_path(rel)) if rel else synth_state_dict(key, load)
