#!/bin/bash
# Builds variants of libdhts.so that differ in -D switches of the macro kernels (tuning experiments), next to the product
# build: diff-hybrid-traffic-sim_amd/csrc/variants/libdhts_<name>.so.  Select one with DHTS_LIB=<path>.
#   tools/build_variants.sh name1:"-DFLAG1 -DFLAG2" name2:"-DFLAG3" ...
set -eu
REPO=$(cd "$(dirname "$0")/.." && pwd)
C=$REPO/diff-hybrid-traffic-sim_amd/csrc
make -C "$C" libdhts.so > /dev/null
mkdir -p "$C/variants"
FLAGS="-O3 -std=c++17 -fPIC -ffp-contract=off -Wall -Wno-unused-function -Wno-pass-failed"
for spec in "$@"; do
  name=${spec%%:*}; defs=${spec#*:}
  /opt/rocm/bin/hipcc --offload-arch=gfx950 $FLAGS $defs -c -o "$C/variants/macro_$name.o" "$C/macro_kernels.hip"
  /opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -fPIC -o "$C/variants/libdhts_$name.so" "$C/variants/macro_$name.o" \
      "$C/dhts_common.o" "$C/micro_kernels.o" "$C/network_kernels.o" "$C/hybrid_kernels.o"
  echo "built $C/variants/libdhts_$name.so ($defs)"
done
