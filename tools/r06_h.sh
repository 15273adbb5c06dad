#!/bin/bash
# round 6, call H: the whole GPU suite and the default bench line on the tree so far; the XCD-paired 1-D probe
set -u
cd "$(dirname "$0")/.."
export TMPDIR=/tmp
OUT=gpurun_out/r06h
mkdir -p $OUT
python -m pytest tests -q -m gpu -x 2>&1 | tail -25 > $OUT/pytest_gpu_tail.txt; cat $OUT/pytest_gpu_tail.txt
python bench.py --steps 20 --warmup 5 > $OUT/bench_c2.json 2> $OUT/bench_c2.err; tail -c 2500 $OUT/bench_c2.json; tail -3 $OUT/bench_c2.err
{
for v in v4 v4x0 v4 v4x0; do echo "== wino1d_probe_$v"; timeout 300 tools/micro/bin/wino1d_probe_$v 96 512 512 60 | grep -v "^reference"; done
} > $OUT/wino1d_probe_v4.txt 2>&1
cat $OUT/wino1d_probe_v4.txt
for pass in "fetch FETCH_SIZE"; do
  set -- $pass; name=$1; shift
  timeout 300 rocprofv3 --pmc "$@" --kernel-include-regex wino_kernel -d $OUT/v4/pmc_$name -o pmc -f csv -- tools/micro/bin/wino1d_probe_v4 96 512 512 3 > $OUT/pmc_v4_$name.log 2>&1
done
python tools/wino_pmc_report.py $OUT/v4 | tail -3
