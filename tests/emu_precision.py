"""CPU emulation of the engine's arithmetic modes -- the error budget behind MOE_PREC_MIXED.

Test infrastructure (like oracle/): a functional fp32 forward of Net2x/3x/4x and NetDN (python/models.py:108-164 of the
reference) in which each convolution's weights and/or input activations can be rounded to fp16, and the trunk stream
(x + s*conv2(PReLU(conv1(x))), models.py:76-80) can be stored as fp16 or kept in fp32.  fp32 accumulation is what the MFMA
kernels do, so `F.conv2d` on rounded operands models them up to summation order.

  python tests/emu_precision.py layers a2      per-layer contribution of weight / activation rounding
  python tests/emu_precision.py budget         max-abs error of fp16 / mixed(n) on noise and natural tiles, all ARSB nets
  python tests/emu_precision.py wino a4        the upsampler convs as Winograd F(2x2, 3x3) with fp16 transformed operands, per branch (round 4 study)

Used by tests/test_precision_budget.py (CPU suite) to pin the defaults of `exact_blocks_of` in engine.cpp.
"""
import os
import sys

import numpy as np
import torch
import torch.nn.functional as F

HERE = os.path.dirname(os.path.abspath(__file__))
for _p in (HERE, os.path.dirname(HERE)):
    if _p not in sys.path:
        sys.path.insert(0, _p)
import golden_defs as gd  # noqa: E402


def r16(x):
    return x.half().float()


def q8(x, shift):
    """x rounded to OCP fp8 e4m3 after scaling by 2^shift (saturating at +-448), as a scaled-MFMA operand would carry it."""
    v = torch.clamp(x * float(2 ** shift), -448.0, 448.0)
    return v.to(torch.float8_e4m3fn).float() * float(2.0 ** -shift)


def q6_block(x, dim, fmt='e2m3'):
    """x in an MX-style block format: blocks of 32 along `dim` share a power-of-two scale (chosen so that the block's largest magnitude fits the element format),
    elements are OCP fp6 e2m3 (1, 1.125 .. 7.5; subnormals in steps of 0.125) or e3m2 (0.25 .. 28; subnormals in steps of 0.0625), round to nearest even.
    What v_mfma_scale_f32_32x32x64_f8f6f4 would read with fp6 operands and per-block scales (4x the fp16 rate instead of fp8's 2x)."""
    mb, emax, emin = {'e2m3': (3, 2, 0), 'e3m2': (2, 4, -2), 'e2m1': (1, 2, 0)}[fmt]      # (e2m1 = OCP fp4: 0.5, 1, 1.5, 2, 3, 4, 6)
    top = (2.0 - 2.0 ** -mb) * 2.0 ** emax
    xs = x.movedim(dim, -1)
    n0 = xs.shape[-1]
    if n0 % 32:                                                      # (48-channel nets: the engine pads to 64 with zeros)
        xs = F.pad(xs, (0, 32 - n0 % 32))
    shp = xs.shape
    b = xs.reshape(shp[:-1] + (shp[-1] // 32, 32))
    amax = b.abs().amax(dim=-1, keepdim=True).clamp_min(1e-30)
    sc = torch.exp2(torch.ceil(torch.log2(amax / top)))              # smallest power of two with amax / sc <= top
    v = b / sc
    e = torch.floor(torch.log2(v.abs().clamp_min(2.0 ** emin))).clamp(emin, emax)
    step = torch.exp2(e - mb)
    q = torch.round(v / step) * step                                 # (torch.round: half to even)
    q = q.clamp(-top, top)
    return (q * sc).reshape(shp)[..., :n0].movedim(-1, dim)


_WBT = torch.tensor([[1, 0, -1, 0], [0, 1, 1, 0], [0, -1, 1, 0], [0, 1, 0, -1]], dtype=torch.float32)
_WG = torch.tensor([[1, 0, 0], [.5, .5, .5], [.5, -.5, .5], [0, 0, 1]], dtype=torch.float32)
_WAT = torch.tensor([[1, 1, 1, 0], [0, 1, -1, -1]], dtype=torch.float32)


def wino_conv(v, w, b, mode='v1'):
    """3x3 conv (zero pad 1, even H and W) as Winograd F(2x2, 3x3): Y = A^T [(G g G^T) . (B^T d B)] A with the transformed weights U and the transformed
    4x4 input tiles V rounded to fp16 (the MFMA operands of a Winograd-domain kernel), fp32 accumulation over the input channels and an fp32 output transform.
    `v` is expected to hold fp16 values already.  mode 'v1': V formed in fp32, one rounding; 'v2': two 1-D passes in fp16 arithmetic (a rounding after
    each); 'v32': nothing rounded (checks the transform itself)."""
    Bn, C, H, W = v.shape
    if mode == 'x1':        # F(2, 3) along x only, the three rows of the kernel as direct taps (round 6 study: 1.5x fewer MFMAs, one 1-D transform each side)
        d = F.pad(v, (1, 1, 1, 1)).unfold(3, 4, 2)                      # (B, C, H + 2, W/2, 4)
        V = r16(torch.einsum('bchwk,lk->bchwl', d, _WBT))               # (B, C, H + 2, W/2, 4 positions)
        U = r16(torch.einsum('ocjk,lk->ocjl', w, _WG))                  # (O, C, 3 rows, 4 positions)
        M = sum(torch.einsum('ocl,bchwl->bohwl', U[:, :, dy], V[:, :, dy:dy + H]) for dy in range(3))
        Y = torch.einsum('bohwk,lk->bohwl', M, _WAT).reshape(Bn, w.shape[0], H, W)
        return Y if b is None else Y + b.view(1, -1, 1, 1)
    d = F.pad(v, (1, 1, 1, 1)).unfold(2, 4, 2).unfold(3, 4, 2)          # (B, C, H/2, W/2, 4, 4)
    if mode == 'v2':
        V = r16(torch.einsum('bchwik,lk->bchwil', r16(torch.einsum('ij,bchwjk->bchwik', _WBT, d)), _WBT))
    else:
        V = torch.einsum('ij,bchwjk,lk->bchwil', _WBT, d, _WBT)
        V = V if mode == 'v32' else r16(V)
    U = torch.einsum('ij,ocjk,lk->ocil', _WG, w, _WG)
    U = U if mode == 'v32' else r16(U)
    M = torch.einsum('ocil,bchwil->bohwil', U, V)
    Y = torch.einsum('ij,bohwjk,lk->bohwil', _WAT, M, _WAT).permute(0, 1, 2, 4, 3, 5).reshape(Bn, w.shape[0], H, W)
    return Y if b is None else Y + b.view(1, -1, 1, 1)


def prelu(x, a):
    return torch.where(x >= 0, x, x * a)


def shuffle(x, r):
    B, C, H, W = x.shape
    c = C // (r * r)
    return x.view(B, c, r, r, H, W).permute(0, 1, 4, 2, 5, 3).reshape(B, c, H * r, W * r)


def layer_names(arch):
    trunk = ['input2'] + [n for i in range(1, 7) for n in ('c1_%d' % i, 'c2_%d' % i)]
    if arch == 'netdn':
        return trunk + ['r.tail', 'u.tail']
    st = 2 if arch == 'net4x' else 1
    out = list(trunk)
    for tag in ('r', 'u'):
        out += ['%s.up%d' % (tag, k) for k in range(st)] + [tag + '.tail']
    return out


def forward(arch, sd, x, w16=(), a16=(), stream16=False, corr8=(), shifts=(8, 4), lo8=False, corr6=None, wino=(), wino_mode='v1'):
    """w16 / a16: names of the convs whose weights / input activations are rounded to fp16 ('all' = every conv);
    stream16: the trunk stream is stored as fp16 after conv_input2 and after every ARSB.
    corr8: convs computed as  conv(w16, a16) + conv(fp8(w - w16), fp8(a16)) + conv(fp8(w16), fp8(a - a16))  -- the split-operand form with its two
    correction products on fp8 operands (scaled by 2^shifts[0] for weights, 2^shifts[1] for activations): a study for the block-scaled fp8 MFMA, which
    runs at twice the fp16 rate (`python tests/emu_precision.py corr8`).
    lo8: the trunk stream is stored as fp16 + an fp8 e4m3 low part ((t - fp16(t)) 2^9, the operand the fp8 correction product reads anyway): 192 instead of
    256 bytes a pixel through HBM, ~15 bits of the value (`python tests/emu_precision.py lo8`)."""
    T = lambda k: torch.from_numpy(np.asarray(sd[k], dtype=np.float32))
    r = 3 if arch == 'net3x' else 2

    def conv(name, v, w, b=None):
        if name in wino:                 # Winograd-domain kernel: fp16 activations in, U and V rounded to fp16
            return wino_conv(r16(v), w, b, wino_mode)
        if name in corr8 and corr6:      # the two correction products on block-scaled fp6 operands (blocks of 32 input channels)
            wh, vh = r16(w), r16(v)
            return (F.conv2d(vh, wh, b, padding=1) + F.conv2d(q6_block(vh, 1, corr6), q6_block(w - wh, 1, corr6), None, padding=1)
                    + F.conv2d(q6_block(v - vh, 1, corr6), q6_block(wh, 1, corr6), None, padding=1))
        if name in corr8:
            wh, vh = r16(w), r16(v)
            sw, sa = shifts
            return F.conv2d(vh, wh, b, padding=1) + F.conv2d(q8(vh, sa), q8(w - wh, sw + 11), None, padding=1) + F.conv2d(q8(v - vh, sa + 11), q8(wh, sw), None, padding=1)
        if name in w16 or 'all' in w16:
            w = r16(w)
        if name in a16 or 'all' in a16:
            v = r16(v)
        return F.conv2d(v, w, b, padding=1)
    x = torch.from_numpy(np.asarray(x, dtype=np.float32))
    out = prelu(F.conv2d(x, T('conv_input.weight'), padding=1), float(T('relu.weight')[0]))       # the stem is fp32 in every mode
    t = conv('input2', out, T('conv_input2.weight'))
    s8 = (lambda v: r16(v) + q8(v - r16(v), 9)) if lo8 else (lambda v: v)
    if stream16:
        t = r16(t)
    t = s8(t)
    for i in range(1, 7):
        p = 'convt_F{}.0.'.format(i)
        m = prelu(conv('c1_%d' % i, t, T(p + 'conv_1.weight')), float(T(p + 'relu.weight')[0]))
        t = t + conv('c2_%d' % i, m, T(p + 'conv_2.weight') * float(T(p + 'scale.scale')[0]))    # ScaleLayer folded into the weights (fp32 product)
        if stream16:
            t = r16(t)
        t = s8(t)
    if arch == 'netdn':
        return conv('r.tail', t, T('convt_R1.weight')) + conv('u.tail', out, T('u.weight'))

    def ups(pre, tag, v):
        k = 0
        while '{}{}.0.weight'.format(pre, k) in sd:
            p = '{}{}.'.format(pre, k)
            v = conv('%s.up%d' % (tag, k), v, T(p + '0.weight'), T(p + '0.bias'))
            v = prelu(shuffle(v, r), float(T(p + '2.weight')[0]))
            k += 1
        return conv('%s.tail' % tag, v, T('{}{}.weight'.format(pre, k)))
    return ups('convt_R1.', 'r', t) + ups('u.', 'u', out)


def mode_sets(arch, mode, n_exact=0):
    """(w16, a16, stream16) of an engine precision mode."""
    L = set(layer_names(arch))
    if mode == 'fp32':
        return set(), set(), False
    if mode == 'fp16':
        return L, L, True
    if mode == 'mixed':
        # exact: conv_input2 and the first n ARSBs (split operands); the tails' WEIGHTS (low-order rows of the fused tail GEMM, or the
        # split-operand tail kernel of NetDN, which also takes the hi+lo activations); the R-branch tail's ACTIVATIONS (EPI 7 of
        # conv3x3_sp.hip: a second small GEMM on their low parts); hi+lo stream
        ex = {'input2'} | {'c%d_%d' % (j, i) for i in range(1, n_exact + 1) for j in (1, 2)}
        tails = {'r.tail', 'u.tail'}
        w16 = L - ex - tails
        a16 = L - ex - (tails if arch == 'netdn' else {'r.tail'})
        return w16, a16, False
    raise ValueError(mode)


def max_err(arch, sd, x, mode, n_exact=0, want=None):
    with torch.no_grad():
        if want is None:
            want = forward(arch, sd, x)
        w16, a16, s16 = mode_sets(arch, mode, n_exact)
        return float((forward(arch, sd, x, w16, a16, s16) - want).abs().max())


def lite_layer_names(scale):
    st = int(scale).bit_length() - 1
    return ['input2'] + ['lb%d.c%d' % (k, j) for k in (1, 2, 3) for j in (1, 2)] + ['%s.up%d' % (t, k) for t in ('r', 'u') for k in range(st)]


def forward_lite(sd, x, scale, exact=(), pairs=True):
    """MoeNet_lite2.Net (python/MoeNet_lite2.py:22-54 of the reference) as the engine computes it: a conv in `exact` (or 'all') uses
    fp32 operands (split operands on the GPU) and its output keeps its low part; any other conv rounds weights, input and output to fp16.
    The stem is fp32 in every mode; `pairs`: the stem output and the stream t = o * g + t are kept as hi + lo (else fp16).  The last
    upsampler stage feeds the 48->1 tail in fp32 (one kernel), so the tails add no rounding."""
    T = lambda k: torch.from_numpy(np.asarray(sd[k], dtype=np.float32))
    stages = int(scale).bit_length() - 1
    ex = lambda name: 'all' in exact or name in exact

    def conv(name, v, w, b=None, pad=1, keep32=False):
        if ex(name):
            return F.conv2d(v, w, b, padding=pad)
        o = F.conv2d(r16(v), r16(w), b, padding=pad)
        return o if keep32 else r16(o)
    x = torch.from_numpy(np.asarray(x, dtype=np.float32))
    out = prelu(F.conv2d(x, T('conv_input.weight'), padding=1), float(T('relu.weight')[0]))
    if not pairs:
        out = r16(out)
    w = T('conv_input2.weight')
    t = conv('input2', out, w, pad=w.shape[-1] // 2)
    for k in (1, 2, 3):
        p = 'convt_F1{}.'.format(k)
        o = prelu(conv('lb%d.c1' % k, t, T(p + 'conv_1.weight')), float(T(p + 'relu.weight')[0]))
        if not ex('lb%d.c1' % k):
            o = r16(o)
        o = conv('lb%d.c2' % k, o, T(p + 'conv_2.weight'))
        g = o.mean(dim=(2, 3), keepdim=True)
        g = torch.relu(F.conv2d(g, T(p + 'se.conv_du.0.weight'), T(p + 'se.conv_du.0.bias')))
        g = torch.sigmoid(F.conv2d(g, T(p + 'se.conv_du.2.weight'), T(p + 'se.conv_du.2.bias')))
        t = o * g + t
        if not pairs:
            t = r16(t)

    def branch(v, pre, tag):
        for k in range(stages):
            p = '{}.{}.'.format(pre, k)
            w = T(p + '0.weight')
            last = k == stages - 1
            v = conv('%s.up%d' % (tag, k), v, w, T(p + '0.bias'), pad=w.shape[-1] // 2, keep32=last)
            v = prelu(shuffle(v, 2), float(T(p + '2.weight')[0]))
            if not last and not ex('%s.up%d' % (tag, k)):
                v = r16(v)
        return v
    wr, wi = T('convt_R1.weight'), T('convt_I1.weight')
    return F.conv2d(branch(t, 'ures', 'r'), wr, padding=wr.shape[-1] // 2) + F.conv2d(branch(out, 'uim', 'u'), wi, padding=wi.shape[-1] // 2)


LITE_RECIPES = [('fp16', None), ('pairs', ()), ('+input2', ('input2',)), ('+lb1', ('input2', 'lb1.c1', 'lb1.c2')),
                ('+lb1+lb2', ('input2', 'lb1.c1', 'lb1.c2', 'lb2.c1', 'lb2.c2')),
                ('+input2+r.up*', ('input2', 'r.up0', 'r.up1', 'r.up2')), ('+lb1+r.up*', ('input2', 'lb1.c1', 'lb1.c2', 'r.up0', 'r.up1', 'r.up2')),
                ('+all c2', ('input2', 'lb1.c2', 'lb2.c2', 'lb3.c2')), ('+lb1+up*', ('input2', 'lb1.c1', 'lb1.c2', 'r.up0', 'r.up1', 'r.up2', 'u.up0', 'u.up1', 'u.up2')),
                ('fp16x3', ('all',))]


def lite_budget(keys=('lite2', 'lite4', 'lite8')):
    for key in keys:
        sd, scale = _load(key), int(key[4:])
        for kind, shape, seed in (('natural', (3, 40, 264), 5), ('noise', (3, 96, 96), 5), ('noise-u8', (3, 128, 128), 0)):
            x = gd.natural_image(seed, shape) if kind == 'natural' else gd.noise_image(seed, shape) if kind == 'noise' else gd.noise_u8(seed, shape).astype(np.float32) / 255.0
            x = x[:, None]
            with torch.no_grad():
                want = forward_lite(sd, x, scale, ('all',))
                line = '%-6s %-8s' % (key, kind)
                for name, ex in LITE_RECIPES:
                    got = forward_lite(sd, x, scale, (), pairs=False) if ex is None else forward_lite(sd, x, scale, ex, pairs=True)
                    line += ' | %s %.2e' % (name, float((got - want).abs().max()))
            print(line, flush=True)


DEFAULT_EXACT = {'net2x': 4, 'net3x': 2, 'net4x': 1, 'netdn': 1}      # == exact_blocks_of() in engine.cpp


def _load(key):
    from moephoto_amd.weights import load_state_dict_file
    return gd.state_dict_for(key, load_state_dict_file)


def main(argv):
    torch.set_num_threads(max(1, min(16, os.cpu_count() or 1)))
    cmd = argv[1] if len(argv) > 1 else 'budget'
    if cmd == 'lite':
        lite_budget(tuple(argv[2:]) or ('lite2', 'lite4', 'lite8'))
        return
    if cmd == 'wino1d':     # the same study for F(2, 3) along x only (wino_conv mode 'x1')
        for key in (argv[2:] or ['a4']):
            arch, sd = gd.MODELS[key][0], _load(key)
            ups = [l for l in layer_names(arch) if '.up' in l]
            for kind, shape, seed in (('noise-u8', (3, 256, 256), 0), ('noise-u8', (3, 256, 256), 1), ('noise-u8', (3, 256, 256), 2), ('natural', (3, 40, 264), 5)):
                x = gd.natural_image(seed, shape) if kind == 'natural' else gd.noise_u8(seed, shape).astype(np.float32) / 255.0
                x = x[:, None]
                with torch.no_grad():
                    want = forward(arch, sd, x)
                    n = DEFAULT_EXACT[arch]
                    ex = ['input2'] + ['c%d_%d' % (j, i) for i in range(1, n + 1) for j in (1, 2)]
                    w16, a16, s16 = mode_sets(arch, 'mixed', n)
                    line = '%-4s %-8s seed %d n=%d: direct %.3e' % (key, kind, seed, n, float((forward(arch, sd, x, w16, a16, s16, corr8=ex, lo8=True) - want).abs().max()))
                    for tag, ws in (('R up1', ['r.up1']), ('U up1', ['u.up1']), ('U up0 + up1', [u for u in ups if u.startswith('u.')]), ('R + U up1', ['r.up1', 'u.up1']), ('all four', ups)):
                        ws = [u for u in ws if u in ups]
                        e = float((forward(arch, sd, x, w16, a16, s16, corr8=ex, lo8=True, wino=ws, wino_mode='x1') - want).abs().max())
                        line += ' | 1-D Winograd on %s %.3e' % (tag, e)
                    print(line, flush=True)
        return
    if cmd == 'wino':       # the upsampler convs as Winograd F(2x2, 3x3) with fp16 U and V (2.25x fewer MFMAs): what would it cost in precision, per branch?
        for key in (argv[2:] or ['a4']):
            arch, sd = gd.MODELS[key][0], _load(key)
            ups = [l for l in layer_names(arch) if '.up' in l]
            for kind, shape, seed in (('noise-u8', (3, 256, 256), 0), ('noise-u8', (3, 256, 256), 1), ('noise-u8', (3, 256, 256), 2), ('natural', (3, 40, 264), 5)):
                x = gd.natural_image(seed, shape) if kind == 'natural' else gd.noise_u8(seed, shape).astype(np.float32) / 255.0
                x = x[:, None]
                with torch.no_grad():
                    want = forward(arch, sd, x)
                    for n in (DEFAULT_EXACT[arch], DEFAULT_EXACT[arch] + 2):
                        ex = ['input2'] + ['c%d_%d' % (j, i) for i in range(1, n + 1) for j in (1, 2)]
                        w16, a16, s16 = mode_sets(arch, 'mixed', n)
                        line = '%-4s %-8s seed %d n=%d: direct %.3e' % (key, kind, seed, n, float((forward(arch, sd, x, w16, a16, s16, corr8=ex, lo8=True) - want).abs().max()))
                        for tag, ws in (('R branch', [u for u in ups if u.startswith('r.')]), ('U branch', [u for u in ups if u.startswith('u.')]), ('both', ups)):
                            e = float((forward(arch, sd, x, w16, a16, s16, corr8=ex, lo8=True, wino=ws) - want).abs().max())
                            line += ' | Winograd on the %s %.3e' % (tag, e)
                        print(line, flush=True)
        return
    if cmd == 'corr6':      # the correction products on block-scaled fp6 instead of fp8 (4x instead of 2x the fp16 MFMA rate): does the budget hold?
        for key in (argv[2:] or ['a2', 'a3', 'a4', 'dn_lite5']):
            arch, sd = gd.MODELS[key][0], _load(key)
            n = DEFAULT_EXACT[arch]
            ex = ['input2'] + ['c%d_%d' % (j, i) for i in range(1, n + 1) for j in (1, 2)]
            for kind, shape, seed in (('noise', (3, 96, 96), 5), ('noise-u8', (3, 256, 256), 0), ('noise-u8', (3, 256, 256), 1), ('natural', (3, 40, 264), 5)):
                x = gd.natural_image(seed, shape) if kind == 'natural' else gd.noise_image(seed, shape) if kind == 'noise' else gd.noise_u8(seed, shape).astype(np.float32) / 255.0
                x = x[:, None]
                with torch.no_grad():
                    want = forward(arch, sd, x)
                    w16, a16, s16 = mode_sets(arch, 'mixed', n)
                    e8 = float((forward(arch, sd, x, w16, a16, s16, corr8=ex, lo8=True) - want).abs().max())
                    e6 = [float((forward(arch, sd, x, w16, a16, s16, corr8=ex, lo8=True, corr6=f) - want).abs().max()) for f in ('e2m3', 'e3m2', 'e2m1')]
                    e0 = float((forward(arch, sd, x, w16 | set(ex), a16 | set(ex), s16, lo8=True) - want).abs().max())
                print('%-9s %-8s n=%d: fp8 corrections %.3e | fp6 e2m3 blocks %.3e | fp6 e3m2 blocks %.3e | fp4 e2m1 blocks %.3e | no corrections %.3e' % (key, kind, n, e8, e6[0], e6[1], e6[2], e0), flush=True)
        return
    if cmd == 'lo8':        # the trunk stream's low part stored as fp8: error against fp32 beside the present form (both with fp8 corrections on the exact layers)
        for key in (argv[2:] or ['a2', 'a3', 'a4', 'dn_lite5']):
            arch, sd = gd.MODELS[key][0], _load(key)
            n = DEFAULT_EXACT[arch]
            ex = ['input2'] + ['c%d_%d' % (j, i) for i in range(1, n + 1) for j in (1, 2)]
            for kind, shape, seed in (('noise', (3, 96, 96), 5), ('noise-u8', (3, 256, 256), 0), ('noise-u8', (3, 256, 256), 1), ('natural', (3, 40, 264), 5)):
                x = gd.natural_image(seed, shape) if kind == 'natural' else gd.noise_image(seed, shape) if kind == 'noise' else gd.noise_u8(seed, shape).astype(np.float32) / 255.0
                x = x[:, None]
                with torch.no_grad():
                    want = forward(arch, sd, x)
                    w16, a16, s16 = mode_sets(arch, 'mixed', n)
                    e = [float((forward(arch, sd, x, w16, a16, s16, corr8=ex, lo8=l) - want).abs().max()) for l in (False, True)]
                    e16 = float((forward(arch, sd, x, w16, a16, True, corr8=ex) - want).abs().max())
                print('%-9s %-8s n=%d: hi+lo stream %.3e | hi + fp8 lo %.3e | fp16 stream %.3e' % (key, kind, n, e[0], e[1], e16), flush=True)
        return
    if cmd == 'corr8':      # the exact layers of 'mixed' with fp8 correction products: error against fp32, beside the present form
        for key in (argv[2:] or ['a2', 'a4', 'dn_lite5']):
            arch, sd = gd.MODELS[key][0], _load(key)
            n = DEFAULT_EXACT[arch]
            ex = ['input2'] + ['c%d_%d' % (j, i) for i in range(1, n + 1) for j in (1, 2)]
            for kind, shape, seed in (('noise', (3, 96, 96), 5), ('noise-u8', (3, 256, 256), 0), ('natural', (3, 40, 264), 5)):
                x = gd.natural_image(seed, shape) if kind == 'natural' else gd.noise_image(seed, shape) if kind == 'noise' else gd.noise_u8(seed, shape).astype(np.float32) / 255.0
                x = x[:, None]
                with torch.no_grad():
                    want = forward(arch, sd, x)
                    w16, a16, s16 = mode_sets(arch, 'mixed', n)
                    line = '%-9s %-8s n=%d: split fp16x3 %.3e' % (key, kind, n, float((forward(arch, sd, x, w16, a16, s16) - want).abs().max()))
                    for sh in ((8, 4), (6, 2), (10, 6)):
                        line += ' | fp8 corrections 2^%d/2^%d %.3e' % (sh[0], sh[1], float((forward(arch, sd, x, w16, a16, s16, corr8=ex, shifts=sh) - want).abs().max()))
                    line += ' | no corrections (n=0, input2 fp16) %.3e' % float((forward(arch, sd, x, set(layer_names(arch)) - {'r.tail', 'u.tail'}, set(layer_names(arch)) - ({'r.tail', 'u.tail'} if arch == 'netdn' else {'r.tail'}), False) - want).abs().max())
                print(line, flush=True)
        return
    if cmd == 'layers':
        key = argv[2] if len(argv) > 2 else 'a2'
        arch, sd = gd.MODELS[key][0], _load(key)
        for kind, shape in (('natural', (3, 40, 264)), ('noise', (3, 96, 96))):
            x = (gd.natural_image(5, shape) if kind == 'natural' else gd.noise_image(5, shape))[:, None]
            with torch.no_grad():
                want = forward(arch, sd, x)
                print(key, kind, 'fp16 everywhere %.2e' % max_err(arch, sd, x, 'fp16', want=want),
                      ' stream rounding alone %.2e' % float((forward(arch, sd, x, stream16=True) - want).abs().max()))
                for L in layer_names(arch):
                    ew = float((forward(arch, sd, x, w16={L}) - want).abs().max())
                    ea = float((forward(arch, sd, x, a16={L}) - want).abs().max())
                    print('  %-8s weights fp16: %.2e   input activations fp16: %.2e' % (L, ew, ea), flush=True)
        return
    for key in ('a2', 'a3', 'a4', 'dn_lite5', 'dn_lite10', 'dn_lite15'):
        arch, sd = gd.MODELS[key][0], _load(key)
        for kind, shape, seed in (('noise', (3, 96, 96), 5), ('noise-u8', (3, 256, 256), 0), ('natural', (3, 40, 264), 5)):
            if kind == 'natural':
                x = gd.natural_image(seed, shape)
            elif kind == 'noise':
                x = gd.noise_image(seed, shape)
            else:
                x = (gd.noise_u8(seed, shape).astype(np.float32) / 255.0)
            x = x[:, None]
            with torch.no_grad():
                want = forward(arch, sd, x)
            line = '%-9s %-8s %s: fp16 %.2e' % (key, kind, shape[1:], max_err(arch, sd, x, 'fp16', want=want))
            for n in (0, 1, 2, 3, 6):
                line += ' | mixed n=%d %.2e' % (n, max_err(arch, sd, x, 'mixed', n, want))
            print(line + '   (default n=%d)' % DEFAULT_EXACT[arch], flush=True)


if __name__ == '__main__':
    main(sys.argv)
