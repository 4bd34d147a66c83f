#!/bin/bash
# GPU box: batch-64 phases, pose blocks fused into the Schur launch or launched on their own (SLAMIT_BA_POSE_SEPARATE), k_schur at 3 or 4 waves per SIMD
export PYTHONPATH=$PWD
for rep in 1 2; do
  for v in "sf 0" "sf 1" "sf4 1"; do
    set -- $v
    echo "lib$1 separate=$2"; SLAMIT_BA_POSE_SEPARATE=$2 SLAMIT_LIB=$PWD/tools/diag/lib$1.so python3 tools/diag/ba_batch_phases.py 64 2>&1 | tail -1 | cut -c1-230
  done
done
