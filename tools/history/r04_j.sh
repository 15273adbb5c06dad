#!/bin/bash
# bench.py: the default line with its new objects, then configs 3, 4, 5 (short)
set -u
cd "$(dirname "$0")/.."
export TMPDIR=/tmp
OUT=gpurun_out/${R04_TAG:-r04j}
mkdir -p $OUT
timeout 600 python bench.py --steps 10 --warmup 2 --sustain 3 --cpu-tiles 2 > $OUT/bench_c2.json 2> $OUT/bench_c2.err; echo "c2 rc=$?"; tail -3 $OUT/bench_c2.err
python - $OUT/bench_c2.json <<'PY'
import json, sys
r = json.loads(open(sys.argv[1]).read().strip().splitlines()[-1])
print(r['ms_per_step'], r['value'], [(k['kernel'][:24], k['achieved'], k['frac'], k['ms']) for k in r.get('roofline_hbm_kernels', []) if 'achieved' in k], {k: (v['achieved'], v['ms']) for k, v in r.get('io_edges', {}).items() if isinstance(v, dict)})
print([k for k in r.get('roofline_hbm_kernels', []) if 'error' in k])
PY
for c in 3 4 5; do
  timeout 900 python bench.py --config $c > $OUT/bench_c$c.json 2> $OUT/bench_c$c.err; echo "c$c rc=$?"; tail -3 $OUT/bench_c$c.err
  python - $OUT/bench_c$c.json <<'PY'
import json, sys
try:
    r = json.loads(open(sys.argv[1]).read().strip().splitlines()[-1])
    print(r['metric'][:60], r['ms_per_step'], r['value'], r.get('roofline', {}).get('frac'), r.get('roofline_hbm', {}).get('achieved'), r.get('cpu_baseline', {}).get('value'), r['config'].get('parity_max_abs_vs_oracle'))
except Exception as e:
    print('no line', e)
PY
done
