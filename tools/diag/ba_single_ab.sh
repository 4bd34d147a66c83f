#!/bin/bash
# GPU box: one window-8 window and one dense window end to end under two or more builds, alternating: tools/diag/ba_single_ab.sh libA.so libB.so
export PYTHONPATH=$PWD
for rep in 1 2 3; do
  for lib in "$@"; do
    echo "$lib: $(SLAMIT_LIB=$PWD/$lib python3 tools/diag/ba_w8.py | tail -1) | $(SLAMIT_LIB=$PWD/$lib python3 tools/diag/ba_dense.py | tail -1)"
  done
done
