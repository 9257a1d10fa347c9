#!/bin/bash
# copy what tools/profile_all.sh left under gpurun_out/prof_<cfg>/ (merged back by gpurun) into profiles/ under the round's tag
R=${NFMC_ROUND_TAG:-r04}
for c in "$@"; do
  lc=$(echo $c | tr A-Z a-z)
  d=gpurun_out/prof_$c
  [ -f $d/pmc_summary.json ] || { echo "no $d/pmc_summary.json"; continue; }
  cp $d/pmc_summary.json profiles/${R}_${lc}_pmc_summary.json
  cp $d/kernel_stats.csv profiles/${R}_${lc}_kernel_stats.csv
  cp $d/bench_line_under_rocprof.json profiles/${R}_${lc}_bench_line_under_rocprof.json
  if [ -f $d/bench_line_fresh.json ]; then cp $d/bench_line_fresh.json profiles/${R}_${lc}_bench_line.json; else cp $d/bench_line.json profiles/${R}_${lc}_bench_line.json; fi
  echo "collected $c"
done
