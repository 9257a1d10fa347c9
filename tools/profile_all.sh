#!/bin/bash
# profiles/ refresh for the configs given (default: all four): tools/profile_bench.sh per config, then the bench line again
# with the fresh counter summary in place (so `roofline.pmc_stale` is false in the committed line).
set -o pipefail
for CFG in "${@:-C3 C5 C4 C2}"; do
  for c in $CFG; do
    lc=$(echo $c | tr A-Z a-z)
    bash tools/profile_bench.sh $c || exit 1
    cp gpurun_out/prof_$c/pmc_summary.json profiles/r03_${lc}_pmc_summary.json
    python3 bench.py --config $c > gpurun_out/prof_$c/bench_line_fresh.json 2> gpurun_out/prof_$c/bench_fresh.err || exit 1
    tail -c 400 gpurun_out/prof_$c/bench_line_fresh.json | head -c 400; echo
    # the raw traces and per-dispatch counter tables are summarised above; dropped so that gpurun_out stays under what
    # gpurun copies back (64 MiB)
    rm -rf gpurun_out/prof_$c/trace
    find gpurun_out/prof_$c -name "*.db" -delete
    find gpurun_out/prof_$c -name "*counter_collection.csv" -delete
  done
done
