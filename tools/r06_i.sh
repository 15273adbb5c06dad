#!/bin/bash
set -u
cd "$(dirname "$0")/.."
export TMPDIR=/tmp
OUT=gpurun_out/r06i
mkdir -p $OUT
timeout 600 python tools/graph_probe.py 2>&1 | grep -v amdgpu.ids > $OUT/graph_probe.txt; cat $OUT/graph_probe.txt
