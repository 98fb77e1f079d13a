#!/bin/bash
# Runs the given pytest node ids one process each (a GPU memory fault aborts the whole interpreter: isolation keeps the others' results)
cd "$(dirname "$0")/.."
for t in "$@"; do
  echo "=== $t"
  timeout 900 python -m pytest "$t" -m gpu -q -s -x 2>&1 | grep -v "^  File\|^Extension\|^$" | tail -25
done
