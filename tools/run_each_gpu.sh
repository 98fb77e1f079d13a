#!/bin/bash
# Runs the given pytest node ids one process each (a GPU memory fault aborts the whole interpreter: isolation keeps the others'
# results).  Prints every test's exit code and exits non-zero if any test failed or timed out.
set -o pipefail
cd "$(dirname "$0")/.."
bad=0
for t in "$@"; do
  echo "=== $t"
  timeout 900 python -m pytest "$t" -m gpu -q -s -x 2>&1 | grep -v "^  File\|^Extension\|^$" | tail -40
  rc=${PIPESTATUS[0]}
  echo "--- rc=$rc  $t"
  [ "$rc" -ne 0 ] && bad=1
done
exit $bad
