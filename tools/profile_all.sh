#!/bin/bash
# profiles/ refresh for the configs given (default: all four): tools/profile_bench.sh per config (a failed counter pass is
# fatal), then the bench line again with the fresh counter summary in place (so `roofline.pmc_stale` is false in the
# committed line); summary, kernel statistics and both lines are copied into profiles/ under the round's tag.
R=${NFMC_ROUND_TAG:-r04}
set -o pipefail
for CFG in "${@:-C3 C5 C4 C2}"; do
  for c in $CFG; do
    lc=$(echo $c | tr A-Z a-z)
    bash tools/profile_bench.sh $c || exit 1
    cp gpurun_out/prof_$c/pmc_summary.json profiles/${R}_${lc}_pmc_summary.json
    cp gpurun_out/prof_$c/kernel_stats.csv profiles/${R}_${lc}_kernel_stats.csv
    cp gpurun_out/prof_$c/bench_line_under_rocprof.json profiles/${R}_${lc}_bench_line_under_rocprof.json
    python3 bench.py --config $c --no-other-configs > gpurun_out/prof_$c/bench_line_fresh.json 2> gpurun_out/prof_$c/bench_fresh.err || exit 1
    cp gpurun_out/prof_$c/bench_line_fresh.json profiles/${R}_${lc}_bench_line.json
    tail -c 400 gpurun_out/prof_$c/bench_line_fresh.json | head -c 400; echo
    # the raw traces and per-dispatch counter tables are summarised above; dropped so that gpurun_out stays under what
    # gpurun copies back (64 MiB)
    rm -rf gpurun_out/prof_$c/trace
    find gpurun_out/prof_$c -name "*.db" -delete
    find gpurun_out/prof_$c -name "*counter_collection.csv" -delete
  done
done
