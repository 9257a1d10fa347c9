#!/bin/bash
# timing only (ablation builds compute garbage): C4 bench line per library
for lib in "$@"; do
  echo "== $lib"
  NFMC_LIB=$PWD/$lib timeout -k 10 200 python bench.py --config C4 --no-cpu-baseline --steps 10 --reps 5 2>/dev/null | python3 -c "
import json,sys
l=json.loads(sys.stdin.read()); r=l['roofline']
print('C4 ms/step %.3f  reps %s  acc %.3f' % (l['ms_per_step'], [round(v,1) for v in l['rep_ms']], l['parity']['mcmc_acceptance']))
"
done
