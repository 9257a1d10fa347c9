#!/bin/bash
# rocprofv3 evidence for profiles/: per-kernel times of the bench command, then separate PMC passes (never combined
# with trace domains; see the gpurun rules).  Output under gpurun_out/prof_<config>/.
# usage (on the GPU box, from the repo root): tools/profile_bench.sh [C3|C2|C4|C5]
#   NFMC_BENCH_EXTRA="--fit-nf" NFMC_PROF_TAG=C5fit tools/profile_bench.sh C5   profiles the same config with extra bench flags
set -o pipefail
CFG=${1:-C3}
X=${NFMC_BENCH_EXTRA:-}
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
O=gpurun_out/prof_${NFMC_PROF_TAG:-$CFG}
rm -rf $O; mkdir -p $O
rocprofv3 --kernel-trace --stats --output-format csv -d $O/trace -- python3 bench.py --config $CFG $X --steps 20 --warmup 2 --min-busy-s 0 --no-other-configs > $O/bench_line_under_rocprof.json 2> $O/trace.err || exit 1
cp $(find $O/trace -name "*kernel_stats.csv" | head -1) $O/kernel_stats.csv
PASSES=("FETCH_SIZE" "WRITE_SIZE" "SQ_WAVES SQ_INSTS_VALU SQ_ACTIVE_INST_VALU SQ_WAVE_CYCLES SQ_BUSY_CYCLES GRBM_GUI_ACTIVE" "SQ_INSTS_SALU SQ_INSTS_LDS SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY" "SQ_INSTS_VALU_ADD_F32 SQ_INSTS_VALU_MUL_F32 SQ_INSTS_VALU_FMA_F32 SQ_INSTS_VALU_TRANS_F32" "SQ_INSTS_VALU_INT32 SQ_INSTS_VALU_INT64 SQ_INSTS_VALU_CVT")
if [ "$CFG" = "C4" ]; then PASSES+=("SQ_INSTS_MFMA SQ_VALU_MFMA_BUSY_CYCLES"); fi
for pass in "${PASSES[@]}"; do
  tag=$(echo $pass | cut -d' ' -f1)
  rocprofv3 --pmc $pass --output-format csv -d $O/pmc_$tag -- python3 bench.py --config $CFG $X --steps 3 --warmup 1 --reps 2 --min-busy-s 0 --no-other-configs --no-cpu-baseline > $O/pmc_$tag.json 2> $O/pmc_$tag.err || { echo "pass $tag failed"; exit 1; }
done
python3 tools/summarize_pmc.py $O > $O/pmc_summary.json
python3 bench.py --config $CFG $X --no-other-configs > $O/bench_line.json 2> $O/bench.err
echo done $CFG
