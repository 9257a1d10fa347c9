#!/bin/bash
# Build an experimental variant of the matrix-core units only (neutra_mfma.hip, flow_mfma.hip, neutra_kernels.hip) and link it with the
# product build's other objects:  tools/build_mfma_variant.sh <name> "-DFLAG ..."  ->  nfmc_amd/libnfmc_hip.<name>.so
set -e
name=$1; flags=$2
cd "$(dirname "$0")/.."
python -m nfmc_amd.build >/dev/null
out=nfmc_amd/csrc/build_$name; mkdir -p $out
PL=${NFMC_NO_PROMOTE_LIMIT:+}; [ -z "$NFMC_NO_PROMOTE_LIMIT" ] && PL="-mllvm -amdgpu-promote-alloca-to-vector-limit=16"
for u in neutra_mfma flow_mfma neutra_kernels; do
  /opt/rocm/bin/hipcc -O3 --offload-arch=gfx950 -std=c++17 -fPIC -fno-gpu-rdc -Iinclude -Infmc_amd/csrc -Wno-unused-result -w $PL $flags \
    -c nfmc_amd/csrc/$u.hip -o $out/$u.o &
done
wait
objs=$(ls nfmc_amd/csrc/build/*.o | grep -v -e neutra_mfma.o -e flow_mfma.o -e neutra_kernels.o)
/opt/rocm/bin/hipcc -shared -fPIC --offload-arch=gfx950 -o nfmc_amd/libnfmc_hip.$name.so $objs $out/neutra_mfma.o $out/flow_mfma.o $out/neutra_kernels.o
echo nfmc_amd/libnfmc_hip.$name.so
