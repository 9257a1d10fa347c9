#!/bin/bash
# Build an experimental variant of the library next to the product build:
#   tools/build_mfma_variant.sh <name> "-DFLAG ..."  ->  nfmc_amd/libnfmc_hip.<name>.so   (objects in csrc/build_<name>/)
# through nfmc_amd/build.py's variant switch (every unit is compiled with the flags; A/B runs select the library with NFMC_LIB).
set -e
name=$1; flags=$2
cd "$(dirname "$0")/.."
NFMC_BUILD_VARIANT=$name NFMC_EXTRA_FLAGS="$flags" python3 -c "import nfmc_amd.build as b; print(b.build())"
