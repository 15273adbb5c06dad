#!/bin/bash
# Round-6 closing pass on ONE binary in ONE call: all GPU tests, the benchmark line (default command = config 2 incl. its live PMC child and the configs 3 / 4 / 5 children, then the
# full lines of configs 3 / 4 / 5), the committed PMC table over the bench's own launches, the kernel-trace summary of the same command, frame times of every model family, the
# margin sweep, per-kernel power.  Results under gpurun_out/$TAG; the caller copies them to profiles/r06.
set -u
cd "$(dirname "$0")/.."
export TMPDIR=/tmp
TAG=${R06_TAG:-r06z}
OUT=gpurun_out/$TAG
mkdir -p $OUT
python - > $OUT/source_digest.txt <<'PY'
import sys; sys.path.insert(0, '.')
from moephoto_amd.build import source_digest
import hashlib
print('sources', source_digest()); print('library', hashlib.sha256(open('moephoto_amd/libmoephoto_amd.so', 'rb').read()).hexdigest())
PY
cat $OUT/source_digest.txt
timeout 1500 python -m pytest tests -x -q -m gpu -rs > $OUT/pytest_gpu.log 2>&1; echo "pytest rc=$?"; tail -4 $OUT/pytest_gpu.log
PMC_TAG=$TAG PMC_STEPS=2 bash tools/pmc_bench.sh > $OUT/pmc_stdout.log 2>&1; tail -14 $OUT/pmc_stdout.log
cp $OUT/pmc_bench.json profiles/pmc_bench.json 2>/dev/null      # (bench.py's fallback table: same sources, same box)
( time timeout 1200 python bench.py ) > $OUT/bench_c2.json 2> $OUT/bench_c2.err; echo "bench c2 rc=$?"; grep real $OUT/bench_c2.err
for c in 3 4 5; do timeout 900 python bench.py --config $c > $OUT/bench_c$c.json 2> $OUT/bench_c$c.err; echo "bench c$c rc=$?"; done
timeout 300 rocprofv3 --kernel-trace --stats -d $OUT/stats -o bench -f csv -- python bench.py --steps 5 --warmup 2 --no-cpu-baseline --sustain 0 --no-noise-input --no-dropin-loop --no-extras --no-configs --no-pmc --no-floor > $OUT/stats_stdout.log 2>&1; echo "stats rc=$?"
find $OUT/stats -name '*kernel_stats.csv' -exec cp {} $OUT/bench_kernel_stats.csv \;
rm -rf $OUT/stats $OUT/pmc_fetch $OUT/pmc_write $OUT/pmc_sq $OUT/pmc_grbm
head -14 $OUT/bench_kernel_stats.csv
TM_PREC=auto timeout 600 python tools/time_models.py > $OUT/time_models.txt 2>&1; grep -v "^$" $OUT/time_models.txt | grep -v amdgpu | tail -9
timeout 900 python tools/margin_sweep.py a4 a2 dn_lite5 > $OUT/margin_sweep.txt 2>&1; grep -v amdgpu $OUT/margin_sweep.txt | tail -30
FUZZ_N=16 FUZZ_KEYS=a2,a4,a3,dn_lite5,lite2,lite8,l25 FUZZ_SEED=29 FUZZ_CROPS=8 timeout 900 python tools/fuzz_gpu.py > $OUT/fuzz.txt 2>&1; echo "fuzz rc=$?"; grep -v amdgpu $OUT/fuzz.txt | tail -14
timeout 200 python tools/kernel_power.py 4 > $OUT/kernel_power.txt 2>&1; grep -v amdgpu $OUT/kernel_power.txt
timeout 200 python tools/prof_dropin.py 8 > $OUT/prof_dropin.txt 2>&1; grep prof_dropin $OUT/prof_dropin.txt
timeout 300 python tools/calib_report.py 2>&1 | grep -v amdgpu > $OUT/calib_report.txt; cat $OUT/calib_report.txt
timeout 300 python tools/graph_probe.py 2>&1 | grep -v amdgpu > $OUT/graph_probe.txt; cat $OUT/graph_probe.txt
for g in 2; do MOE_FORCE_DEVICE=0 MOE_DIST_BACKEND=gloo timeout 600 python bench.py --gpus $g --steps 3 --warmup 1 --no-cpu-baseline --sustain 0 --no-noise-input > $OUT/bench_gpus${g}_shared_gpu.json 2> $OUT/bench_gpus${g}_shared_gpu.err; echo "bench --gpus $g (ranks sharing the GPU, gloo) rc=$?"; done
MOE_DIST_EXCHANGE=p2p MOE_FORCE_DEVICE=0 MOE_DIST_BACKEND=gloo timeout 600 python bench.py --gpus 2 --steps 3 --warmup 1 --no-cpu-baseline --sustain 0 --no-noise-input > $OUT/bench_gpus2_shared_gpu_p2p.json 2> $OUT/bench_gpus2_shared_gpu_p2p.err; echo "bench --gpus 2 p2p rc=$?"
cp gpurun_out/fullsize_report.json $OUT/fullsize_report.json 2>/dev/null
python - $OUT <<'PY'
import json, sys
for c in (2, 3, 4, 5):
    try:
        r = json.loads(open('%s/bench_c%d.json' % (sys.argv[1], c)).read().strip().splitlines()[-1])
        print('config', c, r['ms_per_step'], 'ms', r['value'], r['unit'], '| roofline', r.get('roofline', {}).get('layer_key'), r.get('roofline', {}).get('frac'), '| parity', r['config'].get('parity_max_abs_vs_oracle'), r['config'].get('parity_ok'))
        if c == 2:
            print('   summary', json.dumps(r.get('summary')))
            print('   dropin', r['dropin_loop']['ms_per_step'], r['dropin_loop']['ratio_to_value'], 'without overlap', r['dropin_loop']['without_overlap_calls']['ms_per_step'], 'with blend_tile', r['dropin_loop']['with_moe_blend_tile']['ms_per_step'], r['dropin_loop']['with_moe_blend_tile']['ratio_to_value'], json.dumps(r['dropin_loop']['breakdown']))
            for k in r['roofline_kernels']: print('   ', k['layer_key'], k['ms_per_frame'], k['frac'], k.get('frac_of_power_roofline'), k.get('traffic'), k.get('traffic_source'), k.get('mfma_busy_pmc'))
            print('   configs', {k: (v.get('ms_per_step'), v.get('value'), v.get('parity_max_abs_vs_oracle'), v.get('child_wall_s')) for k, v in r.get('configs', {}).items() if isinstance(v, dict)})
    except Exception as e:
        print('config', c, 'no line', e)
PY
