#!/bin/bash
# A/B of the fused-tail up-conv forms inside one call: bench (short) with up_impl = rw and ps4, then the SQ / GRBM counter passes and the kernel-trace summary of the ps4 run
set -u
cd "$(dirname "$0")/.."
export TMPDIR=/tmp
TAG=${R04_TAG:-r04b}
OUT=gpurun_out/$TAG
mkdir -p $OUT
for impl in rw ps4 rw ps4; do
  MOE_UP_IMPL=$impl timeout 300 python bench.py --steps 20 --warmup 3 --no-cpu-baseline --sustain 4 --no-noise-input --no-dropin-loop > $OUT/bench_$impl.json 2> $OUT/bench_$impl.err
  python - $OUT/bench_$impl.json $impl <<'PY'
import json, sys
r = json.loads(open(sys.argv[1]).read().strip().splitlines()[-1])
ks = {k['layer_key']: k for k in r.get('roofline_kernels', [])}
c = r.get('clock', {})
print('%-4s %.3f ms/frame %.2f MP/s sustained %.3f | R %.3f ms %.3f | U %.3f ms %.3f | arsb %.3f ms %.3f | sclk %s GHz %s W' % (sys.argv[2], r['ms_per_step'], r['value'], r.get('sustained', {}).get('ms_per_step', 0),
      ks['convt_R1.up1']['ms_per_frame'], ks['convt_R1.up1']['frac'], ks['u.up1']['ms_per_frame'], ks['u.up1']['frac'], ks['arsb']['ms_per_frame'], ks['arsb']['frac'], c.get('sclk_ghz_mean'), c.get('power_w_mean')))
PY
done
PMC_TAG=$TAG PMC_STEPS=2 bash tools/pmc_bench.sh > $OUT/pmc_stdout.log 2>&1; tail -25 $OUT/pmc_stdout.log
timeout 300 rocprofv3 --kernel-trace --stats -d $OUT/stats -o bench -f csv -- python bench.py --steps 5 --warmup 2 --no-cpu-baseline --sustain 0 --no-noise-input --no-dropin-loop > $OUT/stats_stdout.log 2>&1; echo "stats rc=$?"
find $OUT/stats -name '*kernel_stats.csv' -exec cp {} $OUT/bench_kernel_stats.csv \;
head -14 $OUT/bench_kernel_stats.csv
