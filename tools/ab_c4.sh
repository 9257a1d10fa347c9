#!/bin/bash
# A/B of library builds on ONE box for the matrix-core NeuTra path: parity tests, then the C4 bench line of each.
# usage: tools/ab_c4.sh lib1.so lib2.so ...   (paths relative to the repo root)
for lib in "$@"; do
  echo "== $lib"
  NFMC_LIB=$PWD/$lib timeout -k 10 400 python -m pytest tests -x -q -m gpu -k "C4 or mfma or matrix_cores" 2>&1 | tail -1
  for rep in $(seq 1 ${AB_REPS:-2}); do
    NFMC_LIB=$PWD/$lib timeout -k 10 200 python bench.py --config C4 --no-cpu-baseline --steps 10 --reps 5 2>/dev/null | python3 -c "
import json,sys
l=json.loads(sys.stdin.read()); r=l['roofline']
print('C4 value %.4g  ms/step %.3f  reps %s  TF/s %.1f  frac %.3f  acc %.3f' % (l['value'], l['ms_per_step'], [round(v,1) for v in l['rep_ms']], r['achieved'], r['frac'], l['parity']['mcmc_acceptance']))
"
  done
done
