"""The JSON line bench.py prints is a contract with the driver (metric / value / unit / n_gpus / steps / warmup / ms_per_step / higher_is_better / scaling /
vs_baseline / dtype / data / config + the roofline and cpu_baseline objects).  bench.py itself needs the GPU; here the lines committed under profiles/ (the
closing pass of the latest round) are held against that contract and against BASELINE.json, so that a change of the line's shape shows up on CPU."""
import glob
import json
import os

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
BASE = json.load(open(os.path.join(ROOT, 'BASELINE.json')))
ROUNDS = sorted(glob.glob(os.path.join(ROOT, 'profiles', 'r[0-9][0-9]')))


def _line(path):
    return json.loads(open(path).read().strip().splitlines()[-1])


def _latest(name):
    for d in reversed(ROUNDS):
        p = os.path.join(d, name)
        if os.path.exists(p):
            return p
    return None


def test_headline_line_follows_the_contract():
    p = _latest('bench_c2.json')
    if p is None:
        pytest.skip('no committed bench line')
    r = _line(p)
    for k, t in (('metric', str), ('value', (int, float)), ('unit', str), ('n_gpus', int), ('steps', int), ('warmup', int), ('ms_per_step', (int, float)),
                 ('higher_is_better', bool), ('scaling', str), ('dtype', str), ('data', str), ('config', dict), ('roofline', dict), ('cpu_baseline', dict)):
        assert k in r and isinstance(r[k], t), k
    assert 'vs_baseline' in r and r['vs_baseline'] is None          # BASELINE.md holds no published number for this metric on this hardware
    # BASELINE.json: "megapixels/sec, 1080p -> 4K 4x SR (... a4), tiled" -- the line names the same metric on the same config (a4 is Net4x: SURVEY.md section 8)
    assert BASE['metric'].startswith('megapixels/sec') and r['metric'].startswith('megapixels/sec') and r['unit'] == 'MP/s'
    assert all(w in r['metric'] for w in ('1080p', '4x', 'a4')) and all(w in BASE['metric'] for w in ('1080p', 'a4')), (r['metric'], BASE['metric'])
    assert r['higher_is_better'] is True and r['scaling'] == 'weak' and r['n_gpus'] == 1 and r['data'].startswith('synthetic')
    assert isinstance(r['config'].get('workload'), str) and 'model' not in r['config']
    assert abs(r['value'] - 1920 * 1080 / 1e6 / (r['ms_per_step'] / 1e3)) / r['value'] < 2e-3      # value = input megapixels of the frame over the step time
    rf = r['roofline']
    for k in ('bound', 'achieved', 'peak', 'unit', 'frac', 'traffic'):
        assert k in rf, k
    assert rf['bound'] in ('hbm', 'mfma') and rf['unit'] in ('GB/s', 'TFLOP/s')
    assert abs(rf['frac'] - rf['achieved'] / rf['peak']) < 1e-3 and 0 < rf['frac'] < 1
    assert rf['traffic'] is None or rf['traffic'] > 0
    cb = r['cpu_baseline']
    for k in ('value', 'unit', 'cores', 'kind', 'sample'):
        assert k in cb, k
    assert cb['kind'] in ('reference', 'port') and cb['unit'] == r['unit'] and cb['cores'] >= 1
    par = r['config'].get('parity_max_abs_vs_oracle')
    assert par is not None and par <= r['config']['parity_tolerance'] == 1e-3 and r['config']['parity_ok'] is True
    # round 5: the other BASELINE configs ride on the default command's line (children of it), the split-operand layers are an MFMA object, the drop-in loop is taken apart
    if 'configs' in r:
        for c in (3, 4, 5):
            e = r['configs']['config%d' % c]
            assert 'error' not in e, e
            assert e['ms_per_step'] > 0 and e['value'] > 0 and e['parity_ok'] is True and e['parity_max_abs_vs_oracle'] <= 1e-3
            assert 'configs[%d]' % (c - 1) in e['workload']
        so = r['roofline_split_operand']
        assert so['bound'] == 'mfma' and so['unit'] == 'TFLOP/s' and abs(so['frac'] - so['achieved'] / so['peak']) < 1e-3
        assert rf.get('traffic_source') in (None, 'this run', 'committed')
        d = r['dropin_loop']
        assert d['breakdown']['engine_forwards_only_ms'] > 0 and d['with_moe_blend_tile']['ms_per_step'] <= d['ms_per_step'] * 1.02


@pytest.mark.parametrize('cfg', [3, 4, 5])
def test_other_config_lines_follow_the_contract(cfg):
    p = _latest('bench_c%d.json' % cfg)
    if p is None:
        pytest.skip('no committed line for config %d' % cfg)
    r = _line(p)
    for k in ('metric', 'value', 'unit', 'n_gpus', 'steps', 'warmup', 'ms_per_step', 'higher_is_better', 'scaling', 'vs_baseline', 'dtype', 'data', 'config'):
        assert k in r, k
    assert r['n_gpus'] == 1 and r['higher_is_better'] is True and r['vs_baseline'] is None
    assert 'configs[%d]' % (cfg - 1) in r['config']['workload']
    assert r['config'].get('parity_ok') is True and r['config']['parity_max_abs_vs_oracle'] <= 1e-3


def test_pmc_summary_names_its_tree():
    p = os.path.join(ROOT, 'profiles', 'pmc_bench.json')
    if not os.path.exists(p):
        pytest.skip('no PMC summary')
    d = json.load(open(p))
    assert isinstance(d.get('source_sha256'), str) and len(d['source_sha256']) == 64
    for g in ('convt_R1.up1', 'u.up1', 'arsb', 'exact'):            # the groups bench.py looks its `traffic` fields up under
        assert g in d['groups'] and d['groups'][g]['hbm_bytes_per_frame'] > 0, g
