#!/bin/bash
# round 6, call V: bench.py --gpus N with the f16s wire as default (config 2 honours --wire): the multi-rank GPU tests (ranks sharing the GPU) and one line of each form
set -u
cd "$(dirname "$0")/.."
export TMPDIR=/tmp
OUT=gpurun_out/r06v
mkdir -p $OUT
timeout 1500 python -m pytest tests -q -m gpu -x -k "bench_gpus or dist_ or config4" 2>&1 | tail -6 > $OUT/pytest_dist.txt; cat $OUT/pytest_dist.txt
for w in f16s f32; do
  MOE_FORCE_DEVICE=0 MOE_DIST_BACKEND=gloo timeout 600 python -m torch.distributed.run --nnodes=1 --nproc-per-node 2 --master-addr 127.0.0.1 --master-port 29533 bench.py --gpus 2 --steps 3 --warmup 1 --no-cpu-baseline --sustain 0 --no-noise-input --wire $w > $OUT/bench_gpus2_$w.json 2> $OUT/bench_gpus2_$w.err; echo "wire $w rc=$?"
  tail -1 $OUT/bench_gpus2_$w.json | python -c "
import json,sys
d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print(d['n_gpus'], d['ms_per_step'], d['value'], d['config'].get('wire'), d.get('first_contact'))"
done
