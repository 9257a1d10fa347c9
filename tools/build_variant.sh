#!/bin/bash
# Experimental variant of the library: recompile ONE translation unit with extra flags and relink against the
# objects of the main build.   tools/build_variant.sh <unit-without-.hip> <out.so> <extra hipcc flags...>
set -e
unit=$1; out=$2; shift 2
cd "$(dirname "$0")/.."
B=nfmc_amd/csrc/build
/opt/rocm/bin/hipcc -O3 --offload-arch=gfx950 -std=c++17 -fPIC -fno-gpu-rdc -Iinclude -Infmc_amd/csrc -Wno-unused-result "$@" \
  -c nfmc_amd/csrc/$unit.hip -o /tmp/variant_$unit.o
objs=$(ls $B/*.o | grep -v "/$unit.o")
/opt/rocm/bin/hipcc -shared -fPIC --offload-arch=gfx950 -o $out $objs /tmp/variant_$unit.o
echo built $out
