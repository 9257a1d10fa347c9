#!/bin/bash
# rocprofv3 evidence for the streamed matrix-core kernels (csrc/mfma_wide.hip) at ONE shape (D, default 256; conditioner 128 x 2,
# 65536 rows; tools/probe_wide.py): per-kernel times, then separate PMC passes (never combined with trace domains).
# Output: gpurun_out/prof_wide/{kernel_stats.csv,pmc_summary.json}
set -o pipefail
export D=${D:-256}
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
O=gpurun_out/prof_wide
rm -rf $O; mkdir -p $O
rocprofv3 --kernel-trace --stats --output-format csv -d $O/trace -- python3 tools/probe_wide.py > $O/probe_under_rocprof.txt 2> $O/trace.err || exit 1
cp $(find $O/trace -name "*kernel_stats.csv" | head -1) $O/kernel_stats_all.csv
(head -1 $O/kernel_stats_all.csv; grep "_wide_kernel" $O/kernel_stats_all.csv) > $O/kernel_stats.csv
for pass in "SQ_INSTS_MFMA SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CYCLES" "SQ_WAVES SQ_WAVE_CYCLES SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY GRBM_GUI_ACTIVE" "SQ_INSTS_VALU SQ_INSTS_LDS SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_INSTS_SALU" "FETCH_SIZE" "WRITE_SIZE"; do
  tag=$(echo $pass | cut -d' ' -f1)
  rocprofv3 --pmc $pass --output-format csv -d $O/pmc_$tag -- python3 tools/probe_wide.py > $O/pmc_$tag.txt 2> $O/pmc_$tag.err || { echo "pass $tag failed"; exit 1; }
done
python3 tools/summarize_pmc.py $O > $O/pmc_summary.json
rm -rf $O/trace; find $O -name "*.db" -delete; find $O -name "*counter_collection.csv" -delete
echo done wide D=$D
