#!/bin/bash
# A/B of an environment knob on the same box: tools/ab_env.sh VAR val1 val2 [bench args...]
var=$1; v1=$2; v2=$3; shift 3
for val in $v1 $v2 $v1 $v2; do
  env $var=$val timeout -k 10 200 python bench.py --no-cpu-baseline "$@" | python -c "
import json,sys
l=json.loads(sys.stdin.read()); r=l['roofline']
print('$var=$val', 'value %.4g' % l['value'], 'ms/step %.4f' % l['ms_per_step'], 'mala %.4f ms' % r['mean_launch_ms'], l['parity'])
"
done
