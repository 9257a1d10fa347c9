#!/bin/bash
# A/B of library builds on one box: C3 bench line + C5 probe for each library given (paths relative to repo root)
for lib in "$@"; do
  echo "== $lib"
  for g in 512 1024 2048; do
    echo -n "C3 grid $g: "
    NFMC_FLOWB_GRID=$g NFMC_LIB=$PWD/$lib timeout -k 10 200 python bench.py --no-cpu-baseline --steps 40 2>/dev/null | python -c "
import json,sys
l=json.loads(sys.stdin.read()); r=l['roofline']
print('value %.4g' % l['value'], 'ms/step %.4f' % l['ms_per_step'], 'rest %.1f us' % ((l['ms_per_step']-r['mean_launch_ms'])*1e3))
"
    echo -n "C5 grid $g: "
    NFMC_FLOWB_GRID=$g NFMC_LIB=$PWD/$lib timeout -k 10 120 python tools/probe_c5.py 2>/dev/null
  done
done
