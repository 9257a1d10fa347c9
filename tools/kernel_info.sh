#!/bin/bash
# Register / scratch / LDS usage of the kernels of one translation unit (compiler remarks):
#   tools/kernel_info.sh <unit-without-.hip> [grep pattern] [extra hipcc flags...]
unit=$1; pat=${2:-.}; shift 2
cd "$(dirname "$0")/.."
/opt/rocm/bin/hipcc -O3 --offload-arch=gfx950 -std=c++17 -fPIC -fno-gpu-rdc -Iinclude -Infmc_amd/csrc -Wno-unused-result "$@" \
  -Rpass-analysis=kernel-resource-usage -c nfmc_amd/csrc/$unit.hip -o /tmp/ki_$unit.o 2>&1 | python3 -c "
import re,sys,subprocess
cur=None; rows={}
for line in sys.stdin:
    m=re.search(r'remark: .*Function Name: (\S+)',line)
    if m:
        cur=subprocess.run(['c++filt',m.group(1)],capture_output=True,text=True).stdout.strip(); rows[cur]={}; continue
    m=re.search(r'remark: .*?\s+(\w[\w ]+?): (\d+)',line)
    if m and cur: rows[cur][m.group(1).strip()]=m.group(2)
for k,v in rows.items():
    if re.search(r'''$pat''',k):
        print(k[:120]); print('    ', ', '.join('%s %s'%(a,b) for a,b in v.items()))
"
