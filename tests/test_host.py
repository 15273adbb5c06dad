"""Host-side logic of the product (no GPU): zoo reader/writer, the C-ABI library (loads, exports every
symbol the header declares, planner entry points), the nn.Module protocol of the engine-backed models,
and the rule that the product never touches the oracle."""
import ctypes
import glob
import json
import os
import re
import subprocess
import sys

import numpy as np
import pytest
import torch

import golden_defs as gd
from moephoto_amd import _lib
from moephoto_amd.weights import load_state_dict_file, save_state_dict_file

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
PLANNER = json.load(open(os.path.join(gd.GOLDEN, 'planner.json')))
ZOO_FILES = sorted(glob.glob(os.path.join(gd.ZOO, 'model', '*', '*.pth')))


@pytest.mark.parametrize('path', ZOO_FILES, ids=lambda p: '/'.join(p.split(os.sep)[-2:]))
def test_zoo_files_load_unchanged(path):
    sd = load_state_dict_file(path)
    ref = torch.load(path, map_location='cpu', weights_only=False)   # the reference's loader (imageProcess.py:306)
    assert list(sd.keys()) == list(ref.keys())
    for k in ref:
        assert sd[k].dtype == np.float32 and np.array_equal(sd[k], ref[k].numpy())


def test_zoo_schemas():
    # SURVEY.md section 8a row W
    counts = {'a2/model_new.pth': (35, 776399), 'dn_lite5/model_new.pth': (29, 270877), 'lite/model.pth': (32, 146703),
              'lite/model_4.pth': (38, 165521), 'lite/model_8.pth': (44, 184339)}
    for rel, (n, params) in counts.items():
        sd = load_state_dict_file(os.path.join(gd.ZOO, 'model', rel))
        assert len(sd) == n and sum(v.size for v in sd.values()) == params


def test_reader_rejects_foreign_globals(tmp_path):
    import pickle
    p = tmp_path / 'evil.pth'
    with open(p, 'wb') as f:
        pickle.dump(0x1950a86a20f9469cfc6c, f, protocol=2)
        pickle.dump(1001, f, protocol=2)
        pickle.dump({'little_endian': True}, f, protocol=2)
        pickle.dump(os.getcwd, f, protocol=2)
    with pytest.raises(pickle.UnpicklingError):
        load_state_dict_file(str(p))
    # ... nor in the auxiliary pickles around the state dict (magic / protocol / sys_info / key list)
    for slot in range(3):
        e = tmp_path / 'evil_aux{}.pth'.format(slot)
        with open(e, 'wb') as f:
            for i, good in enumerate((0x1950a86a20f9469cfc6c, 1001, {'little_endian': True})):
                pickle.dump(os.getcwd if i == slot else good, f, protocol=2)
        with pytest.raises(pickle.UnpicklingError):
            load_state_dict_file(str(e))
    # a tensor view that reaches past its storage is refused (as_strided would read out of bounds)
    good = os.path.join(gd.ZOO, 'model', 'dn_lite5', 'model_new.pth')
    raw = bytearray(open(good, 'rb').read())
    sd = load_state_dict_file(good)
    first = next(iter(sd.values()))
    import struct as _st
    # the first storage's element count sits right after the key-list pickle: shrink it -> the announced and stored counts disagree
    idx = raw.rfind(_st.pack('<q', first.size))
    assert idx > 0
    bad = tmp_path / 'short.pth'
    bad.write_bytes(bytes(raw[:idx]) + _st.pack('<q', first.size - 1) + bytes(raw[idx + 8:]))
    with pytest.raises(ValueError):
        load_state_dict_file(str(bad))
    from moephoto_amd.weights import _LazyStorage, _LazyTensor, _StorageType
    st = _LazyStorage(_StorageType('FloatStorage'), 'k', 6)
    st.data = np.arange(6, dtype='<f4')
    assert _LazyTensor(st, 0, (2, 3), (3, 1)).materialize().tolist() == [[0, 1, 2], [3, 4, 5]]
    for off, size, stride in ((1, (2, 3), (3, 1)), (0, (2, 4), (3, 1)), (0, (2, 3), (-3, 1)), (7, (), ())):
        with pytest.raises(ValueError):
            _LazyTensor(st, off, size, stride).materialize()
    q = tmp_path / 'notzoo.pth'
    q.write_bytes(b'PK\x03\x04 zip')
    with pytest.raises(ValueError):
        load_state_dict_file(str(q))


@pytest.mark.parametrize('key', ['a3', 'a4', 'l25'])
def test_writer_roundtrip_synthetic(key, tmp_path):
    sd = gd.synth_state_dict(key, load_state_dict_file)
    p = save_state_dict_file(sd, str(tmp_path / (key + '.pth')))
    back = torch.load(p, map_location='cpu', weights_only=False)     # what the reference would do with it
    ours = load_state_dict_file(p)
    assert list(back.keys()) == list(sd.keys()) == list(ours.keys())
    for k in sd:
        assert np.array_equal(back[k].numpy(), sd[k]) and np.array_equal(ours[k], sd[k])


def test_library_exports_every_header_symbol():
    hdr = open(os.path.join(ROOT, 'include', 'moephoto_amd.h')).read()
    declared = sorted(set(re.findall(r'\b(moe_[a-z0-9_]+)\s*\(', hdr)))
    assert len(declared) >= 20
    L = ctypes.CDLL(_lib.LIB_PATH)
    for name in declared:
        assert hasattr(L, name), name
    assert sorted(_lib.EXPORTS) == declared
    assert _lib.lib().moe_abi_version() == _lib.ABI_VERSION == 4


def _plan(case):
    L = _lib.lib()
    h = ctypes.c_void_p()
    rc = L.moe_plan_create((ctypes.c_int64 * 3)(*case['shape']), float(case['ram']), float(case['ram_coef']), case['pad'], case['sc'],
                           case['align'], case['cropsize'], ctypes.byref(h))
    assert rc == 0, L.moe_last_error()
    return h


@pytest.mark.parametrize('case', PLANNER['prepare'], ids=lambda c: 'x'.join(map(str, c['shape'])) + '_c{}'.format(c['cropsize']))
def test_c_planner_golden(case):
    from moephoto_amd.imageProcess import TilePlan
    pl = TilePlan(case['shape'], case['ram'], case['ram_coef'], case['pad'], case['sc'], case['align'], case['cropsize'])
    assert [list(t) for t in pl.tiles] == case['tiles']
    assert [pl.outH, pl.outW] == case['out_shape'][-2:]
    assert np.abs(pl.ramp - np.array(case['ramp'], np.float32)).max() <= 1.2e-7
    off = pl.tile_offsets(case['shape'][0])
    sizes = [case['shape'][0] * (t[1] - t[0]) * (t[3] - t[2]) * case['sc'] ** 2 for t in pl.tiles]
    assert off == list(np.cumsum([0] + sizes[:-1])) and pl.pool_elems(case['shape'][0]) == sum(sizes)


@pytest.mark.parametrize('case', PLANNER['anchors'][:30], ids=lambda c: 's{s}_l{l}_p{pad}_a{align}_x{sc}'.format(**c))
def test_python_get_anchors_golden(case):
    from moephoto_amd.imageProcess import alignF, getAnchors
    got = getAnchors(case['s'], case['ns'], case['l'], case['pad'], alignF[case['align']], case['sc'])
    assert got == (case['start'], case['end'], case['clip'], case['step'], case['end_sc'])


def test_planner_memory_error():
    from moephoto_amd.imageProcess import TilePlan
    with pytest.raises(MemoryError):      # imageProcess.py:58-59,78-80: even a minimal tile does not fit
        TilePlan((3, 100, 100), 1000.0, 1e-3, 5, 2, 8, 0)


def test_module_protocol_without_gpu():
    from moephoto_amd.models import Net, Net2x, NetDN
    m = Net2x()
    sd = load_state_dict_file(gd.zoo_path('model/a2/model_new.pth'))
    assert list(m.expected_keys().keys()) == list(sd.keys()) or set(m.expected_keys()) == set(sd)
    for k, shp in m.expected_keys().items():
        assert tuple(sd[k].shape) == shp, k
    m.load_state_dict({k: torch.from_numpy(v) for k, v in sd.items()})
    for p in m.parameters():
        p.requires_grad_(False)
    assert m.eval() is m and len(list(m.parameters())) == 35
    bad = dict(sd)
    bad.pop('relu.weight')
    bad['bogus.weight'] = np.zeros(1, np.float32)
    with pytest.raises(RuntimeError) as e:
        Net2x().load_state_dict(bad)
    assert 'Missing key(s)' in str(e.value) and 'Unexpected key(s)' in str(e.value)
    wrong = dict(sd)
    wrong['conv_input.weight'] = np.zeros((64, 1, 5, 5), np.float32)
    with pytest.raises(RuntimeError) as e:
        Net2x().load_state_dict(wrong)
    assert 'size mismatch' in str(e.value)
    for cls, rel in ((NetDN, 'model/dn_lite10/model_new.pth'), (Net, 'model/lite/model.pth')):
        sd2 = load_state_dict_file(gd.zoo_path(rel))
        assert set(cls().expected_keys()) == set(sd2)
    assert set(Net(upscale=8).expected_keys()) == set(load_state_dict_file(gd.zoo_path('model/lite/model_8.pth')))
    with pytest.raises(_lib.EngineError):
        Net(upscale=3)


@pytest.mark.skipif(torch.cuda.is_available(), reason='checks the no-GPU failure mode')
def test_fails_loudly_without_device():
    from moephoto_amd.models import Net2x
    m = Net2x()
    m.load_state_dict(load_state_dict_file(gd.zoo_path('model/a2/model_new.pth')))
    with pytest.raises(_lib.EngineError):
        m.to(dtype=torch.float16, device='cuda:0')
    with pytest.raises(_lib.EngineError):
        m(torch.zeros(1, 1, 8, 8))
    with pytest.raises(_lib.EngineError):
        m.to(device='cpu')


def test_product_never_imports_oracle():
    """The oracle is test infrastructure: nothing under moephoto_amd/ may import, link or load it."""
    imp = re.compile(r'^\s*(from|import)\s+\.*(oracle|tests|golden_defs)\b')
    n = 0
    for p in glob.glob(os.path.join(ROOT, 'moephoto_amd', '**', '*'), recursive=True):
        if p.endswith(('.py', '.cpp', '.hip', '.h')):
            txt = open(p).read()
            n += 1
            assert not [l for l in txt.splitlines() if imp.match(l)], p
            assert 'convref' not in txt and '_build' not in txt, p
    assert n >= 10


def test_build_entry_and_oracle_build():
    # build() compiles everything in-tree; here only check it is importable and idempotent (fast when fresh)
    out = subprocess.run([sys.executable, '-c', 'import __graft_entry__ as g; g.build(); print("built")'], cwd=ROOT,
                         capture_output=True, text=True, timeout=900)
    assert out.returncode == 0 and 'built' in out.stdout, out.stderr[-2000:]
    assert os.path.exists(_lib.LIB_PATH) and os.path.exists(os.path.join(ROOT, 'oracle', '_build', 'libconvref.so'))


def test_video_buffer_edges_and_frame_loop():
    """toNumPy / toBuffer (python/imageProcess.py:216-236) and the per-frame loop of SR_vid (python/video.py:349-360):
    raw bgr48le frames in, raw frames out, `start` frames skipped, short reads rejected."""
    import io
    from moephoto_amd import imageProcess as ip, procedure
    h, w = 4, 6
    rng = np.random.default_rng(7)
    frames = [rng.integers(0, 65536, (h, w, 3), dtype=np.uint16) for _ in range(3)]
    raw = b''.join(f.tobytes() for f in frames)
    a = ip.toNumPy(16)((frames[1].tobytes(), h, w))
    assert a.dtype == np.uint16 and a.shape == (h, w, 3) and np.array_equal(a, frames[1])
    assert ip.toNumPy(16)((b'', h, w)) is None
    assert ip.toNumPy(8)((bytes(range(72)), h, w)).dtype == np.uint8
    assert ip.toBuffer(16)(frames[2]) == frames[2].tobytes() and ip.toBuffer(8)(None) is None
    seen, out = [], []

    def process(frame):          # stand-in for a genProcess 'buffer' pipeline: doubles every sample
        seen.append(frame[1:])
        return [ip.toBuffer(16)(ip.toNumPy(16)(frame) // 2)]
    n = procedure.runFrames(process, io.BytesIO(raw).read, out.append, w, h, bitDepth=16, start=1)
    assert n == 2 and seen == [(h, w), (h, w)]
    assert np.array_equal(np.frombuffer(out[1], np.uint16).reshape(h, w, 3), frames[2] // 2)
    assert procedure.runFrames(process, io.BytesIO(raw).read, out.append, w, h, bitDepth=16, stop=0) == 1
    with pytest.raises(ValueError):
        procedure.runFrames(process, io.BytesIO(raw[:-5]).read, out.append, w, h, bitDepth=16)


@pytest.mark.parametrize('case', [c for c in PLANNER['prepare'] if len(c['tiles']) > 1 and int(np.prod(c['out_shape'])) <= 3e7], ids=lambda c: 'x'.join(map(str, c['shape'])) + '_c{}'.format(c['cropsize']))
def test_wire_format_seams_keep_fp16_canvases_bit_identical(case):
    """The inter-rank wire format (include/moephoto_amd.h: moe_plan_seams / moe_wire_*; moephoto_amd/dist.py wire='f16s') sends fp16 values plus the fp32
    values of the seam rows / columns.  Host-side proof on the planner's golden cases, with the numpy restatement of the pack / unpack passes and the
    oracle's fold: a canvas folded from round-tripped tiles equals, after rounding to fp16, the canvas folded from the fp32 tiles -- and rounding the
    WHOLE tile to fp16 does not (the seams are needed)."""
    import wire_codec
    from moephoto_amd import dist as mdist
    from moephoto_amd.imageProcess import TilePlan
    from oracle import planner as oplanner, stitch as ostitch
    C, sc = case['shape'][0], case['sc']
    pl = TilePlan(case['shape'], case['ram'], case['ram_coef'], case['pad'], sc, case['align'], case['cropsize'])
    opl = oplanner.prepare(tuple(case['shape']), case['ram'], case['ram_coef'], case['pad'], sc, case['align'], case['cropsize'])
    seams = pl.seams()
    dims = [(C, (t[1] - t[0]) * sc, (t[3] - t[2]) * sc) for t in pl.tiles]
    rng = np.random.default_rng(3)
    tiles = [(rng.standard_normal(d) * 0.7 + 0.4).astype(np.float32) for d in dims]
    recs = np.zeros(len(dims), mdist.WIRE_REC)
    off = wpos = 0
    for k, d in enumerate(dims):
        recs[k] = (off, wpos) + d + tuple(seams[k]) + (0,)
        words = mdist.wire_words(recs[k])
        assert words == _lib.lib().moe_wire_words(recs[k:k + 1].ctypes.data)
        s = seams[k]
        assert 0 <= s[0] <= s[1] <= s[2] <= s[3] <= d[1] and 0 <= s[4] <= s[5] <= s[6] <= s[7] <= d[2]
        off += d[0] * d[1] * d[2]
        wpos += words
    buf = np.concatenate([t.reshape(-1) for t in tiles])
    wire = np.zeros(wpos, np.int32)
    wire_codec.codec(True, buf, wire, recs)
    back = np.full_like(buf, np.nan)
    wire_codec.codec(False, back, wire, recs)
    got = [back[int(r['tile_off']):int(r['tile_off']) + d[0] * d[1] * d[2]].reshape(d) for r, d in zip(recs, dims)]
    for k, (t, g) in enumerate(zip(tiles, got)):
        s = seams[k]
        m = np.zeros(t.shape, bool)
        m[:, s[0]:s[1]] = m[:, s[2]:s[3]] = True
        m[:, :, s[4]:s[5]] = m[:, :, s[6]:s[7]] = True
        assert np.array_equal(g[m], t[m]) and np.array_equal(g[~m], t.astype(np.float16).astype(np.float32)[~m])
    want = ostitch.fold_stitch(tiles, opl, sc)
    assert np.array_equal(ostitch.fold_stitch(got, opl, sc).astype(np.float16), want.astype(np.float16))
    if case['pad'] > 0:
        crude = ostitch.fold_stitch([t.astype(np.float16).astype(np.float32) for t in tiles], opl, sc)
        assert not np.array_equal(crude.astype(np.float16), want.astype(np.float16))
    assert wpos * 4 < 0.95 * buf.nbytes or min(d[1] for d in dims) < 6 * case['pad'] * sc


# ---- which kernel instantiation does each (zoo key, precision, shape class) resolve to? ------------------------------------------------
# tests/golden/kernel_resolution.json is generated on the GPU (tools/kernel_table.sh: one short process per case under rocprofv3 --kernel-trace).  Here, on the CPU, the
# library's compiled instantiations (its host launch stubs: nm -C) are held against it: every one of them is either launched by a case of the table or listed below WITH
# the reason it exists -- a new instantiation that nothing resolves to, or one that lost its last user, fails this test instead of riding along in the build.
NOT_IN_THE_TABLE = {
    # reached by the defaults on other shapes than the table's small cases (profiles/r05/h_kernel_census_defaults.txt: the 1080p frame)
    'conv3x3_ps4_kernel<0, true>': 'store form of the x2 stages: launches with >= 32 four-row blocks per workgroup (a 1080p frame of a4)',
    # option forms the form-vs-form GPU tests compare, and fallbacks for shapes the fast kernels refuse (profiles/r05/h_kernel_census_gpu_test_suite.txt: all launched by the GPU suite)
    'conv64_x3_kernel<3>': 'plain + pooled epilogue: lite with the gate pooled from conv_2\'s OUTPUT (option frm_pre = 0: test_lite_frm_gate_from_conv2_input compares the forms)',
    'frm_gate_kernel': 'the FRM gate from pooled sums of conv_2\'s output: frm_pre = 0, and lite under fp16 / mixed',
    'conv1x1_kernel<true, 1, false, 4>': 'lite\'s conv_input2 as a launched 1x1 conv (option stem2 = 0: test_lite_conv_input2_in_closed_form compares it with the stem\'s closed form)',
    'conv1x1_kernel<false, 1, false, 4>': 'the same layer on plain fp16 operands (precision fp16, stem2 = 0: the same test)',
    'sedn_xsum_kernel': 'the pass over x when the producing conv did not form the channel totals (option pool_fuse = 0: test_sedn_fused_block_tail_shapes compares the forms); since round 6 sedn_fmean visits the border itself when it did',
    'conv3x3_rw_kernel<3, false>': 'phase-class-sums fused tail on patch-aligned images: option up_impl = rw (A/B of conv3x3_ps4)',
    'conv3x3_rw_kernel<7, false>': 'the same with split tail activations',
    'conv3x3_sp_kernel<1>': 'PReLU epilogue with the weights in LDS: option sp_impl = sp',
    'conv3x3_sp_kernel<5>': 'split-precision pass of the three-launch form: x3_fuse = 0',
    'conv3x3_sp_kernel<6>': "SEDN's fused block tail with the weights in LDS: option s64 = 0",
    'conv64_q8_kernel<0, false, false>': 'patch form of the split-operand layers: option q8_impl = p / shapes conv64_sq refuses; lo8 = off',
    'conv64_q8_kernel<0, false, true>': 'patch form, conv_input2 behind an fp16 low part writing fp8 (no caller in the current chains: the stem writes fp8 itself)',
    'conv64_q8_kernel<0, true, true>': 'patch form, conv_input2 of the fp8 chain',
    'conv64_q8_kernel<1, false, false>': 'patch form, conv_1, lo8 = off',
    'conv64_q8_kernel<1, true, true>': 'patch form, conv_1 of the fp8 chain',
    'conv64_q8_kernel<2, false, false>': 'patch form, conv_2 + residual, lo8 = off',
    'conv64_q8_kernel<2, true, false>': 'patch form, last conv_2 of the fp8 chain (fp16 low part out)',
    'conv64_q8_kernel<2, true, true>': 'patch form, conv_2 inside the fp8 chain',
    'conv64_sq_kernel<1, true>': 'conv_1 of an exact ARSB as its own launch: option exact_fuse = 0 (A/B of arsb_sq)',
    'conv64_sq_kernel<2, false>': 'conv_2 of the last exact ARSB as its own launch',
    'conv64_sq_kernel<2, true>': 'conv_2 of an exact ARSB inside the chain as its own launch',
    'conv1x1_kernel<true, 4, false, 4>': "lite's split-operand upsampler stage with all four k-slices, weights in LDS: option k48 = 0 (A/B of the register-weight form), 64-channel inputs",
    'conv1x1_kernel<true, 4, true, 4>': 'the same with the folded 48->1 tail',
    'conv_direct_kernel': "precision 'debug_direct': the scalar device convolution (kernel debugging)",
    'conv_mfma_kernel<1, 1>': 'generic 1x1 conv: option conv1x1 = 0, SEDN unfused trans',
    'conv_mfma_kernel<1, 3>': 'generic 1x1 conv with split operands: shapes beyond conv1x1.hip (32-bit offsets)',
    'conv_mfma_kernel<9, 1>': 'generic 3x3 conv: conv_impl = v1 and 64-bit shapes',
    'nhwc_to_nchw_kernel': 'debug taps (moe_net_debug_tap)',
    'stitch_kernel': 'stitch fallback for canvases the vector forms do not take',
    'tail_kernel<1>': 'unfused 1x1 tail (lite with fuse_tail = 0)',
    'tail_kernel<9>': 'first-generation 3x3 tail: MOE_TAIL_V1',
    'tailadd_kernel<false>': 'branch sum into output planes that are not 16-byte aligned',
    'tailadd3_kernel<false>': 'the same for x3 nets (conv3x3_ps9)',
    'tapsum2_kernel': 'nine-plane fused tail of x2 nets: tail_form = planes',
    'tapsum4_kernel<false>': 'phase-class sums into unaligned output planes',
    'tapsum_kernel<2>': 'nine-plane fused tail, scalar form',
}
NOT_THE_NET = ('blend_tile_kernel', 'resize_kernel', 'to_float', 'to_output', 'wire_kernel', 'stitch', 'maxabsdiff_kernel')      # edges / callers of the path: exercised by their own tests, not by a forward


def test_every_compiled_kernel_instantiation_has_a_user():
    import subprocess
    table = json.load(open(os.path.join(ROOT, 'tests', 'golden', 'kernel_resolution.json')))
    sys.path.insert(0, os.path.join(ROOT, 'tools'))
    import kernel_census
    compiled = kernel_census.compiled()
    assert len(compiled) >= 90, 'nm -C found no launch stubs in the library?'
    launched = set(table['launched'])
    assert launched <= compiled, 'the table names instantiations the library no longer has (regenerate it: tools/kernel_table.sh): {}'.format(sorted(launched - compiled))
    for case, ks in table['cases'].items():      # every default-arithmetic case of the table resolves its layers to library kernels, and the families to their own trunk kernel
        assert ks, case
    assert 'arsb32c_kernel<true, 4>' in table['cases']['a4/auto/frame'] and 'conv3x3_ps4_kernel<2, false>' in table['cases']['a4/auto/frame']
    assert 'conv3x3_ps9_kernel<true, false>' in table['cases']['a3/auto/frame'] and 'tapsum_kernel<3>' not in table['cases']['a3/auto/frame']      # (round 6: x3 nets off the per-phase form)
    assert 'arsb32c_kernel<true, 3>' in table['cases']['dn_lite5/auto/frame'] and 'conv64_s_kernel<6>' in table['cases']['l25/auto/frame']
    assert 'conv1x1_f2_kernel' in table['cases']['lite4/auto/frame'] and 'conv1x1_kernel<true, 4, true, 3>' in table['cases']['lite2/auto/frame'] and not any('conv_mfma' in k for k in table['cases']['lite8/auto/frame'])      # (round 5: lite8 off the generic kernel)
    orphans = sorted(k for k in compiled - launched if k not in NOT_IN_THE_TABLE and not k.startswith(NOT_THE_NET))
    assert not orphans, 'compiled, launched by no case of tests/golden/kernel_resolution.json and not explained in NOT_IN_THE_TABLE: {}'.format(orphans)
    stale = sorted(k for k in NOT_IN_THE_TABLE if k not in compiled)
    assert not stale, 'NOT_IN_THE_TABLE explains instantiations that are no longer compiled: {}'.format(stale)


def test_round5_entry_points_validate_their_arguments_without_a_device():
    """moe_blend_tile and moe_net_calibrate / moe_net_exact_blocks (ABI version 3): argument errors are reported before anything touches a device -- the reference's callers
    get an exception with a message (python/worker.py:52-74), not a launch on bad pointers."""
    L = _lib.lib()
    buf = (ctypes.c_float * 16)()
    p = ctypes.cast(buf, ctypes.c_void_p)
    blend = lambda *a: L.moe_blend_tile(*a)
    assert blend(None, 0, 0, p, 0, 0, _lib.F32, 1, 0, 0, 4, 4, 0, 0, 2, p, None) == _lib.EINVAL and b'NULL' in L.moe_last_error()
    assert blend(p, 16, 4, p, 16, 4, _lib.U8, 1, 0, 0, 4, 4, 0, 0, 2, p, None) == _lib.EINVAL and b'dtype' in L.moe_last_error()
    assert blend(p, 16, 4, p, 16, 4, _lib.F32, 1, 4, 0, 4, 4, 0, 0, 2, p, None) == _lib.EINVAL and b'window' in L.moe_last_error()          # empty window (top_sc == bsc)
    assert blend(p, 16, 4, p, 16, 4, _lib.F32, 1, 0, 0, 4, 4, 1, 0, 2, p, None) == _lib.EINVAL and b'band' in L.moe_last_error()            # first new row 1, two ramp rows in front of it: outside
    assert blend(p, 16, 4, p, 16, 4, _lib.F32, 1, 0, 0, 4, 4, 2, 0, 2, None, None) == _lib.EINVAL and b'ramp' in L.moe_last_error()
    h = ctypes.c_void_p()
    _lib.check(L.moe_net_create(_lib.ARCH_NET2X, 2, ctypes.byref(h)))
    try:
        n, e = ctypes.c_int(), ctypes.c_double()
        assert L.moe_net_calibrate(h, 0.0, ctypes.byref(n), ctypes.byref(e), None) == _lib.ESTATE and b'finalized' in L.moe_last_error()
        assert L.moe_net_calibrate(None, 0.0, ctypes.byref(n), ctypes.byref(e), None) == _lib.EINVAL
        assert L.moe_net_exact_blocks(h) == 0                          # (not in MOE_PREC_MIXED yet)
        assert L.moe_net_set_option(h, b'branch_streams', b'0') == 0 and L.moe_net_set_option(h, b'branch_groups', b'64') == 0 and L.moe_net_set_option(h, b'auto_calibrate', b'off') == 0
        assert L.moe_net_set_option(h, b'repeat', b'arsb3:20') == 0 and L.moe_net_set_option(h, b'repeat', b'0') == 0 and L.moe_net_set_option(h, b'repeat', b'arsb3:0') == _lib.EINVAL
        assert L.moe_net_set_option(h, b'arsb_impl', b's') == _lib.EINVAL       # (the streamed ARSB left the build in round 5)
        # a refused `repeat` value leaves the option as it was ("up1:0" used to store a count of zero and then report EINVAL -- the next forward issued no launch
        # for the matching layers); the state is visible through what the next valid / invalid calls return
        assert L.moe_net_set_option(h, b'repeat', b'up1:3') == 0
        for bad in (b'up1:0', b'k:-3', b'nonsense', b':4'):
            assert L.moe_net_set_option(h, b'repeat', bad) == _lib.EINVAL, bad
        assert L.moe_net_set_option(h, b'repeat', b'') == 0 and L.moe_net_set_option(h, b'calib_log', b'1') == 0 and L.moe_net_set_option(h, b'calib_log', b'x') == _lib.EINVAL
    finally:
        L.moe_net_destroy(h)


def test_blend_tile_wrapper_refuses_tensors_of_another_rank():
    """imageProcess.blendTile is the only guard in front of moe_blend_tile's raw pointers: a canvas or tile result whose extra axes are not singletons must be
    refused before strides of the wrong axes reach the kernel; singleton axes -- the reference's (1, C, H, W) tmp_image with opt.oShape, the net's (C, 1, h, w) result --
    are dropped (the accepted forms run in tests/test_gpu_parity.py)."""
    import torch
    from moephoto_amd import imageProcess as ip
    tile = (0, 8, 0, 8, 0, 0, 16, 16)
    ramp = torch.zeros(4)
    for r, canvas in ((torch.zeros(3, 16, 16), torch.zeros(3, 2, 16, 16)), (torch.zeros(3, 2, 16, 16), torch.zeros(3, 16, 16)),
                      (torch.zeros(2, 3, 16, 16), torch.zeros(3, 16, 16)), (torch.zeros(16, 16), torch.zeros(3, 16, 16))):
        with pytest.raises(ValueError, match='expected'):
            ip.blendTile(r, canvas, tile, 2, 4, ramp)
