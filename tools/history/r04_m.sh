#!/bin/bash
# the reference-style per-tile loop (3 planes per forward) with up_impl = rw and ps4, same box
cd "$(dirname "$0")/.."
for impl in rw ps4 rw ps4; do
  MOE_UP_IMPL=$impl timeout 300 python bench.py --steps 10 --warmup 3 --no-cpu-baseline --sustain 0 --no-noise-input --no-extras 2>/dev/null | python -c "
import json,sys
r=json.loads(sys.stdin.read().strip().splitlines()[-1])
print('$impl', 'frame', r['ms_per_step'], 'ms | drop-in loop', r['dropin_loop']['ms_per_step'], 'ms', r['dropin_loop']['value'], 'MP/s ratio', r['dropin_loop']['ratio_to_value'])"
done
