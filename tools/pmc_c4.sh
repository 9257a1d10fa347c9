#!/bin/bash
# PMC passes on the C4 NeuTra probe (matrix-core path): tools/pmc_c4.sh  -> gpurun_out/pmc_c4/summary.txt
set -o pipefail
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
O=gpurun_out/pmc_c4; rm -rf $O; mkdir -p $O
i=0
for pass in "SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_VALU_MFMA_BUSY_CYCLES SQ_INSTS_MFMA GRBM_GUI_ACTIVE SQ_WAVES" \
            "SQ_WAIT_INST_ANY SQ_WAIT_INST_LDS SQ_WAIT_ANY SQ_ACTIVE_INST_LDS SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_ANY" \
            "SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_LDS_ADDR_CONFLICT SQ_INSTS_LDS SQ_LDS_DATA_FIFO_FULL SQ_LDS_CMD_FIFO_FULL" \
            "SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_INST_CYCLES_VMEM_RD SQ_INSTS_FLAT_NO_LDS"; do
  i=$((i+1))
  rocprofv3 --pmc $pass --output-format csv -d $O/p$i -- python3 tools/probe_c4.py > $O/p$i.txt 2> $O/p$i.err || echo "pass $i failed"
done
python3 - <<'PY' > $O/summary.txt
import csv, glob, collections
acc=collections.defaultdict(lambda: collections.defaultdict(list))
for path in glob.glob('gpurun_out/pmc_c4/p*/**/*counter_collection.csv', recursive=True):
    for row in csv.DictReader(open(path)):
        k=row['Kernel_Name']
        if 'leapfrog' not in k: continue
        acc[k[:70]][row['Counter_Name']].append(float(row['Counter_Value']))
for k,c in acc.items():
    print(k)
    for n,v in sorted(c.items()): print('   %-28s %16.1f  (n=%d)' % (n, sum(v)/len(v), len(v)))
PY
cat $O/summary.txt
