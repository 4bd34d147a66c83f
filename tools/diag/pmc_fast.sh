#!/bin/bash
# GPU box: SQ counter passes over bench.py's extract+match step; prints per-kernel averages.  tools/diag/pmc_fast.sh [kernel-substring]
export PYTHONPATH=$PWD TMPDIR=/tmp
rm -rf gpurun_out/pmc_a gpurun_out/pmc_b
rocprofv3 --pmc SQ_WAVES SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_WAVE_CYCLES SQ_BUSY_CYCLES --output-format csv -d gpurun_out/pmc_a -- python3 bench.py --no-cpu --no-ba --no-extras --steps 4 --warmup 1 > /dev/null 2>&1
rocprofv3 --pmc SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_INSTS_VMEM_RD SQ_WAIT_INST_LDS SQ_ACTIVE_INST_SCA --output-format csv -d gpurun_out/pmc_b -- python3 bench.py --no-cpu --no-ba --no-extras --steps 4 --warmup 1 > /dev/null 2>&1
python3 tools/diag/pmc_summary.py "${1:-fast}"
