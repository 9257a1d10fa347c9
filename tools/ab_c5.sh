#!/bin/bash
# C5 A/B on one box: rocprofv3 kernel times of the bench command (hmc_kernel, the jump) and the bench value
set -o pipefail
mkdir -p gpurun_out/prof_c5q
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
timeout -k 10 300 rocprofv3 --kernel-trace --stats --output-format csv -d gpurun_out/prof_c5q/trace -- python3 bench.py --config C5 --steps 20 --warmup 2 --min-busy-s 0 --no-other-configs --no-cpu-baseline > gpurun_out/prof_c5q/line.json 2> gpurun_out/prof_c5q/err.txt || exit 1
f=$(find gpurun_out/prof_c5q/trace -name "*kernel_stats.csv" | head -1)
python3 - "$f" <<'PY'
import csv, sys
for r in csv.DictReader(open(sys.argv[1])):
    if 'hmc_kernel' in r['Name'] or 'flow_mh' in r['Name']:
        print(r['Name'][11:60], 'calls', r['Calls'], 'avg us %.1f' % (float(r['AverageNs']) / 1e3), 'min us %.1f' % (float(r['MinNs']) / 1e3))
PY
rm -rf gpurun_out/prof_c5q/trace
python3 -c "import json; d=json.load(open('gpurun_out/prof_c5q/line.json')); print('value %.4g  ms/step %.4f' % (d['value'], d['ms_per_step']))"
