#!/bin/bash
# Builds variants of libdhts.so that differ in -D switches of the macro kernels (tuning experiments), next to the product
# build: diff-hybrid-traffic-sim_amd/csrc/variants/libdhts_<name>.so.  Select one with DHTS_LIB=<path>.
#   tools/build_variants.sh name1:"-DFLAG1 -DFLAG2" name2:"-DFLAG3" ...
#   SRC=hybrid_kernels tools/build_variants.sh os:"-Os" ...      (another translation unit; default macro_kernels; the flags come
#                                                                  after the product's, so a later -O level wins)
set -eu
REPO=$(cd "$(dirname "$0")/.." && pwd)
C=$REPO/diff-hybrid-traffic-sim_amd/csrc
make -C "$C" libdhts.so > /dev/null
mkdir -p "$C/variants"
FLAGS="-O3 -std=c++17 -fPIC -ffp-contract=off -Wall -Wno-unused-function -Wno-pass-failed"
SRC=${SRC:-macro_kernels}
for spec in "$@"; do
  name=${spec%%:*}; defs=${spec#*:}
  EXTRA=""; [ "$SRC" = hybrid_kernels ] && EXTRA="-mllvm -disable-lsr"      # (the product's per-file flags, csrc/Makefile)
  /opt/rocm/bin/hipcc --offload-arch=gfx950 $FLAGS $EXTRA $defs -c -o "$C/variants/${SRC}_$name.o" "$C/$SRC.hip"
  OBJS=""
  for u in dhts_common macro_kernels micro_kernels network_kernels netstep_kernels netstep_hybrid hybrid_kernels; do
    if [ "$u" = "$SRC" ]; then OBJS="$OBJS $C/variants/${SRC}_$name.o"; else OBJS="$OBJS $C/$u.o"; fi
  done
  /opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -fPIC -o "$C/variants/libdhts_$name.so" $OBJS
  echo "built $C/variants/libdhts_$name.so ($SRC: $defs)"
done
