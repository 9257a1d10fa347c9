#!/bin/bash
# Fixed cost of a jump launch vs its per-tile cost: the kernel time of flow_mh_b*_kernel at d = 256 for 1, 2, 4 and 8 chain tiles per
# workgroup (512 workgroups), from rocprofv3 kernel traces.
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
for n in ${NS:-8192 16384 32768 65536}; do
  O=/tmp/abt_$$; rm -rf $O
  NFMC_FLOWB_GRID=${GRID:-512} rocprofv3 --kernel-trace --stats --output-format csv -d $O -- python3 tools/probe_jump.py $n ${D:-256} > /dev/null 2>&1
  f=$(find $O -name "*kernel_stats.csv" | head -1)
  python3 -c "
import csv
for r in csv.DictReader(open('$f')):
    if 'flow_mh_b' in r['Name']: print('n=$n d=${D:-256} grid<=${GRID:-512}: %.1f us (min %.1f, %s calls)  %s' % (float(r['AverageNs'])/1e3, float(r['MinNs'])/1e3, r['Calls'], r['Name'][11:45]))
"
done
