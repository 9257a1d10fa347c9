#!/bin/bash
# A/B of two builds of the library on the same box: bench.py per-step time and per-kernel event times.
# usage: tools/ab_bench.sh old.so new.so   (paths relative to the repo root)
for lib in "$@"; do
  for rep in 1 2; do
    NFMC_LIB=$PWD/$lib timeout -k 10 200 python bench.py --no-cpu-baseline --steps 40 | python -c "
import json,sys
l=json.loads(sys.stdin.read()); r=l['roofline']
print('$lib', 'value %.4g' % l['value'], 'ms/step %.4f' % l['ms_per_step'], 'mala %.4f ms' % r['mean_launch_ms'])
"
  done
done
