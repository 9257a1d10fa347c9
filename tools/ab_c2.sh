#!/bin/bash
# C2 (imh) A/B on one box: the data-parallel IMH parity tests, then the rocprofv3 kernel times of the bench command
# (eval / scan / replay per 1000-transition call = the MaxNs column) and the bench value.
set -o pipefail
mkdir -p gpurun_out/prof_c2q
timeout -k 10 500 python -m pytest tests/test_gpu_parity.py tests/test_gpu_configs.py -x -q -k "imh or C2 or independence or data_parallel" > gpurun_out/imh_tests.log 2>&1
echo "tests rc=$?"; tail -2 gpurun_out/imh_tests.log
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
timeout -k 10 300 rocprofv3 --kernel-trace --stats --output-format csv -d gpurun_out/prof_c2q/trace -- python3 bench.py --config C2 --steps 20 --warmup 2 --min-busy-s 0 --no-other-configs --no-cpu-baseline > gpurun_out/prof_c2q/line.json 2> gpurun_out/prof_c2q/err.txt || exit 1
f=$(find gpurun_out/prof_c2q/trace -name "*kernel_stats.csv" | head -1)
python3 - "$f" <<'PY'
import csv, sys
for r in csv.DictReader(open(sys.argv[1])):
    if 'imh_' in r['Name']:
        print(r['Name'][11:36], 'calls', r['Calls'], 'avg us %.1f' % (float(r['AverageNs']) / 1e3), 'max us %.1f' % (float(r['MaxNs']) / 1e3))
PY
rm -rf gpurun_out/prof_c2q/trace
python3 -c "import json; d=json.load(open('gpurun_out/prof_c2q/line.json')); print('value %.4g  call ms %.3f' % (d['value'], d['roofline']['mean_launch_ms']))"
