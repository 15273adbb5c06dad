#!/bin/bash
# round 5, call B: where do arsb32c's joules go?  Ablation builds (results wrong by design, same instruction stream), each looped by itself (option repeat) with rocm-smi
# sampling power / clock: ms per launch and GHz per variant.  Then the blend test + bench line with the new objects.   -> gpurun_out/r05b/
set -u
cd "$(dirname "$0")/.."
export TMPDIR=/tmp
OUT=gpurun_out/r05b
mkdir -p $OUT
cp moephoto_amd/libmoephoto_amd.so /tmp/lib_product.so
for rep in 1 2; do
for v in product nolo nost nolost nodma noepi nofrag noall; do
  if [ $v = product ]; then cp /tmp/lib_product.so moephoto_amd/libmoephoto_amd.so; else cp moephoto_amd/_abl/lib_a32_$v.so moephoto_amd/libmoephoto_amd.so; fi
  timeout 120 python tools/kernel_power.py 3 arsb3 $v 2>/dev/null | grep arsb32c
done
done > $OUT/arsb32c_ablations.txt 2>&1
cp /tmp/lib_product.so moephoto_amd/libmoephoto_amd.so
cat $OUT/arsb32c_ablations.txt
timeout 300 python -m pytest tests/test_gpu_parity.py -m gpu -x -q -k "blend_tile" > $OUT/pytest_blend.txt 2>&1; echo "blend rc=$?"; tail -3 $OUT/pytest_blend.txt
(time timeout 900 python bench.py) > $OUT/bench_c2.json 2> $OUT/bench_c2.err; echo "bench rc=$?"; tail -3 $OUT/bench_c2.err
python - <<P
import json
d = json.loads(open('$OUT/bench_c2.json').read().strip().splitlines()[-1])
print('value', d['value'], 'ms', d['ms_per_step'], 'parity', d['config'].get('parity_max_abs_vs_oracle'))
for k in d.get('roofline_kernels', []): print(k['layer_key'], k['ms_per_frame'], k['frac'], k.get('traffic'), k.get('traffic_source'), k.get('mfma_busy_pmc'))
print('split', {k: d['roofline_split_operand'].get(k) for k in ('bound', 'achieved', 'frac', 'frac_algorithmic', 'ms_per_frame', 'traffic_source')} if 'roofline_split_operand' in d else None)
print('dropin', json.dumps(d.get('dropin_loop', {}).get('breakdown')), d.get('dropin_loop', {}).get('ms_per_step'), json.dumps({k: v for k, v in d.get('dropin_loop', {}).get('with_moe_blend_tile', {}).items() if k != 'what'}))
for k, c in d.get('configs', {}).items(): print(k, json.dumps({a: b for a, b in c.items() if a in ('value', 'ms_per_step', 'parity_max_abs_vs_oracle', 'parity_ok', 'child_wall_s', 'error')}) if isinstance(c, dict) else c)
P
