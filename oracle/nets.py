"""Oracle: fp32 forwards of the SR/DN nets.  TEST INFRASTRUCTURE (see oracle/__init__.py).

Functional restatements over a plain state dict (name -> fp32 array), one per reference module:
  net_sr      MyNet.forward + Net2x/Net3x/Net4x   python/models.py:108-154 (eval branch of multiConvt, :41-43)
  net_dn      NetDN                                python/models.py:158-164
  sedn        SEDN + _Conv_Block                   python/models.py:166-223
  lite        MoeNet_lite2.Net / LB                python/MoeNet_lite2.py:5-54, FRM models.py:270-287
Two conv backends: "torch" (F.conv2d, oneDNN -- what the reference itself runs on CPU) and "c"
(oracle/convref.c, plain loops) so that the convolution semantics are pinned independently of torch.
Every forward takes x as (B,1,h,w) fp32 and returns (B,1,sc*h,sc*w) fp32; `taps` (a dict) receives
named intermediates in NCHW for layer-by-layer localisation of kernel bugs.
"""
import numpy as np
import torch
import torch.nn.functional as F

from . import convref

ARCHS = ('net2x', 'net3x', 'net4x', 'netdn', 'sedn', 'lite2', 'lite4', 'lite8')


def _t(a):
    return a if isinstance(a, torch.Tensor) else torch.from_numpy(np.array(a, dtype=np.float32))


class Ops:
    def __init__(self, backend='torch'):
        self.backend = backend

    def conv(self, x, w, b=None, pad=None):
        w = _t(w)
        k = w.shape[-1]
        p = (k // 2) if pad is None else pad
        if self.backend == 'c':
            y = convref.conv2d(x.numpy(), w.numpy(), None if b is None else _t(b).numpy(), p)
            return torch.from_numpy(y)
        return F.conv2d(x, w, None if b is None else _t(b), padding=p)

    @staticmethod
    def prelu(x, a):
        a = float(_t(a).reshape(-1)[0]) if not isinstance(a, float) else a
        return torch.where(x >= 0, x, x * a)

    @staticmethod
    def shuffle(x, r):
        # PixelShuffle: out[c, r*h+i, r*w+j] = in[c*r*r + i*r + j, h, w]
        B, C, H, W = x.shape
        c = C // (r * r)
        return x.view(B, c, r, r, H, W).permute(0, 1, 4, 2, 5, 3).reshape(B, c, H * r, W * r)


def _tap(taps, name, v):
    if taps is not None:
        taps[name] = v.detach().clone()
    return v


def _upsampler(ops, sd, pre, x, r, taps, tag):
    """nn.Sequential([conv+bias, PixelShuffle(r), PReLU] * k, conv C->1): models.py:29-36,125-154."""
    k = 0
    while '{}{}.0.weight'.format(pre, k) in sd:
        p = '{}{}.'.format(pre, k)
        x = ops.conv(x, sd[p + '0.weight'], sd[p + '0.bias'])
        x = ops.prelu(ops.shuffle(x, r), sd[p + '2.weight'])
        _tap(taps, '{}.up{}'.format(tag, k), x)
        k += 1
    return ops.conv(x, sd['{}{}.weight'.format(pre, k)])


def net_sr(sd, x, r=2, backend='torch', taps=None):
    ops = Ops(backend)
    x = _t(x)
    out = _tap(taps, 'stem', ops.prelu(ops.conv(x, sd['conv_input.weight']), sd['relu.weight']))
    t = _tap(taps, 'input2', ops.conv(out, sd['conv_input2.weight']))
    for i in range(1, 7):
        p = 'convt_F{}.0.'.format(i)
        m = ops.prelu(ops.conv(t, sd[p + 'conv_1.weight']), sd[p + 'relu.weight'])
        t = _tap(taps, 'arsb{}'.format(i), t + float(_t(sd[p + 'scale.scale'])[0]) * ops.conv(m, sd[p + 'conv_2.weight']))
    u = _upsampler(ops, sd, 'u.', out, r, taps, 'u')
    rr = _upsampler(ops, sd, 'convt_R1.', t, r, taps, 'r')
    return rr + u


def net_dn(sd, x, backend='torch', taps=None):
    ops = Ops(backend)
    x = _t(x)
    out = _tap(taps, 'stem', ops.prelu(ops.conv(x, sd['conv_input.weight']), sd['relu.weight']))
    t = _tap(taps, 'input2', ops.conv(out, sd['conv_input2.weight']))
    for i in range(1, 7):
        p = 'convt_F{}.0.'.format(i)
        m = ops.prelu(ops.conv(t, sd[p + 'conv_1.weight']), sd[p + 'relu.weight'])
        t = _tap(taps, 'arsb{}'.format(i), t + float(_t(sd[p + 'scale.scale'])[0]) * ops.conv(m, sd[p + 'conv_2.weight']))
    return ops.conv(t, sd['convt_R1.weight']) + ops.conv(out, sd['u.weight'])


def sedn(sd, x, backend='torch', taps=None, blocks=16):
    ops = Ops(backend)
    x = _t(x)
    t = _tap(taps, 'stem', ops.prelu(ops.conv(x, sd['conv_input.weight']), 0.2))
    for k in range(blocks):
        p = 'convt_F1.{}.'.format(k)
        o = ops.prelu(ops.conv(t, sd[p + 'rblock.0.weight']), 0.2)
        o = ops.prelu(ops.conv(o, sd[p + 'rblock.2.weight']), 0.2)
        o = ops.conv(o, sd[p + 'rblock.4.weight'])
        g = o.mean(dim=(2, 3), keepdim=True)
        g = ops.prelu(ops.conv(g, sd[p + 'conv_down.weight'], pad=0), 0.2)
        g = torch.sigmoid(ops.conv(g, sd[p + 'conv_up.weight'], pad=0))
        o = ops.prelu(ops.conv(o * g, sd[p + 'trans.0.weight'], pad=0), 0.2)
        t = _tap(taps, 'block{}'.format(k), t + o)
    return ops.conv(t, sd['convt_R1.weight']) + x


def lite(sd, x, upscale=2, backend='torch', taps=None):
    ops = Ops(backend)
    x = _t(x)
    stages = int(upscale).bit_length() - 1
    out = _tap(taps, 'stem', ops.prelu(ops.conv(x, sd['conv_input.weight']), sd['relu.weight']))
    t = _tap(taps, 'input2', ops.conv(out, sd['conv_input2.weight']))
    for k in (1, 2, 3):
        p = 'convt_F1{}.'.format(k)
        o = ops.prelu(ops.conv(t, sd[p + 'conv_1.weight']), sd[p + 'relu.weight'])
        o = ops.conv(o, sd[p + 'conv_2.weight'])
        g = o.mean(dim=(2, 3), keepdim=True)
        g = torch.relu(ops.conv(g, sd[p + 'se.conv_du.0.weight'], sd[p + 'se.conv_du.0.bias']))
        g = torch.sigmoid(ops.conv(g, sd[p + 'se.conv_du.2.weight'], sd[p + 'se.conv_du.2.bias']))
        t = _tap(taps, 'lb{}'.format(k), o * g + t)

    def branch(v, pre, tag):
        for k in range(stages):
            p = '{}.{}.'.format(pre, k)
            v = ops.conv(v, sd[p + '0.weight'], sd[p + '0.bias'])
            v = _tap(taps, '{}.up{}'.format(tag, k), ops.prelu(ops.shuffle(v, 2), sd[p + '2.weight']))
        return v
    res = branch(t, 'ures', 'r')
    im = branch(out, 'uim', 'u')
    return ops.conv(res, sd['convt_R1.weight']) + ops.conv(im, sd['convt_I1.weight'])


def forward(arch, sd, x, backend='torch', taps=None):
    arch = arch.lower()
    with torch.no_grad():
        if arch in ('net2x', 'net3x', 'net4x'):
            return net_sr(sd, x, 3 if arch == 'net3x' else 2, backend, taps)
        if arch == 'netdn':
            return net_dn(sd, x, backend, taps)
        if arch == 'sedn':
            return sedn(sd, x, backend, taps)
        if arch in ('lite2', 'lite4', 'lite8'):
            return lite(sd, x, int(arch[4:]), backend, taps)
    raise ValueError('unknown arch ' + arch)


def scale_of(arch):
    return {'net2x': 2, 'net3x': 3, 'net4x': 4, 'netdn': 1, 'sedn': 1, 'lite2': 2, 'lite4': 4, 'lite8': 8}[arch.lower()]


def model_fn(arch, sd, backend='torch'):
    """numpy-in / numpy-out callable for oracle.stitch.do_crop."""
    def f(s):
        return forward(arch, sd, torch.from_numpy(np.ascontiguousarray(s, dtype=np.float32)), backend).numpy()
    return f
