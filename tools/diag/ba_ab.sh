#!/bin/bash
# GPU box: window-8 / dense single-window local BA with and without the fused point pass: tools/diag/ba_ab.sh
export PYTHONPATH=$PWD
for i in 1 2; do
  python3 tools/diag/ba_w8.py; SLAMIT_BA_NO_FUSE=1 python3 tools/diag/ba_w8.py
done
python3 tools/diag/ba_dense.py; SLAMIT_BA_NO_FUSE=1 python3 tools/diag/ba_dense.py
