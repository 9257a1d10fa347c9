#!/bin/bash
# All rocprofv3 artefacts of profiles/ in one gpurun call (from the repo root on the GPU box): tools/profile_round.sh
set -o pipefail
tools/profile_bench.sh || exit 1
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
O=gpurun_out/prof
rocprofv3 --kernel-trace --stats --output-format csv -d $O/c5 -- python3 tools/probe_c5.py > $O/c5.txt 2> $O/c5.err || exit 1
cp $(find $O/c5 -name "*kernel_stats.csv" | head -1) $O/c5_kernel_stats.csv
C2_N=8192 rocprofv3 --kernel-trace --stats --output-format csv -d $O/c2 -- python3 tools/probe_c2.py > $O/c2.txt 2> $O/c2.err || exit 1
cp $(find $O/c2 -name "*kernel_stats.csv" | head -1) $O/c2_kernel_stats.csv
python3 tools/bench_configs.py > $O/configs.jsonl 2> $O/configs.err
echo round done
