#!/bin/bash
# Kernel time of the jump (flow_mh_b_kernel) per library / grid cap / shape from rocprofv3 kernel traces.
# usage: tools/ab_jump.sh lib1.so lib2.so ...      (paths relative to the repo root; default: the in-tree library)
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
LIBS=${@:-nfmc_amd/libnfmc_hip.so}
for lib in $LIBS; do
  for shape in "65536 64" "32768 256" "8192 64"; do
    for g in ${GRIDS:-512 1024 2048}; do
     for dual in ${DUALS:-auto}; do   # DUALS="0 1": one / two chains per lane group (NFMC_FLOWB_DUAL)
      O=/tmp/abj_$$; rm -rf $O
      if [ "$dual" = auto ]; then unset NFMC_FLOWB_DUAL; else export NFMC_FLOWB_DUAL=$dual; fi
      NFMC_FLOWB_GRID=$g NFMC_LIB=$PWD/$lib rocprofv3 --kernel-trace --stats --output-format csv -d $O -- python3 tools/probe_jump.py $shape > /dev/null 2>&1
      f=$(find $O -name "*kernel_stats.csv" | head -1)
      us=$(python3 -c "
import csv,sys
for r in csv.DictReader(open('$f')):
    if 'flow_mh_b' in r['Name']: print('%.1f us (min %.1f, %s calls)' % (float(r['AverageNs'])/1e3, float(r['MinNs'])/1e3, r['Calls']))
")
      echo "$lib  n,d=$shape  grid<=$g  dual=$dual  $us"
     done
    done
  done
done
