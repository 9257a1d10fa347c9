#!/bin/bash
# sweep an environment knob on one box: tools/sweep_env.sh VAR "v1 v2 ..." [bench args...]
var=$1; vals=$2; shift 2
for rep in 1 2; do
for val in $vals; do
  env $var=$val timeout -k 10 200 python bench.py --no-cpu-baseline "$@" 2>/dev/null | python -c "
import json,sys
l=json.loads(sys.stdin.read()); r=l['roofline']
print('$var=$val', 'value %.4g' % l['value'], 'ms/step %.4f' % l['ms_per_step'], 'mala %.4f ms' % r['mean_launch_ms'], 'rest %.1f us' % ((l['ms_per_step']-r['mean_launch_ms'])*1e3))
"
done
done
