#!/bin/bash
# The oracle's golden tests under AddressSanitizer + UndefinedBehaviorSanitizer (CPU build; GPU sanitizers are not available on this pool).
#   bash tools/oracle_sanitizers.sh        -> rebuilds oracle/libdhts_oracle.so instrumented, runs tests/test_oracle_golden.py, restores it
set -eu
cd "$(dirname "$0")/.."
ASAN=$(gcc -print-file-name=libasan.so)
cp oracle/libdhts_oracle.so /tmp/libdhts_oracle_plain.so
trap 'cp /tmp/libdhts_oracle_plain.so oracle/libdhts_oracle.so; touch oracle/libdhts_oracle.so' EXIT
gcc -O1 -g -mfma -ffp-contract=off -fno-builtin-pow -fPIC -std=c11 -fopenmp -fsanitize=address,undefined -fno-omit-frame-pointer \
    -shared -o oracle/libdhts_oracle.so oracle/dhts_oracle.c -lm
ASAN_OPTIONS=detect_leaks=0 LD_PRELOAD=$ASAN python -m pytest tests/test_oracle_golden.py -x -q -p no:cacheprovider
