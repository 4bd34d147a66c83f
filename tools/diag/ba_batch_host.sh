export PYTHONPATH=$PWD
SLAMIT_BA_TIMING=1 python3 tools/diag/ba_batch_phases.py 64 2>&1 | tail -6
