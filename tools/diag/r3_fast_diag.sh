#!/bin/bash
# GPU box, round 3: per-phase wave cycles (FAST_DIAG builds) and SQ counters of FAST variants.  tools/diag/r3_fast_diag.sh <diag libs> -- <pmc libs>
export PYTHONPATH=$PWD TMPDIR=/tmp
O=gpurun_out/r3_fast; mkdir -p $O
while [ "$1" != "--" ] && [ -n "$1" ]; do
  echo "== phases $1"; SLAMIT_LIB=$PWD/tools/diag/lib$1.so python3 tools/diag/fast_phases.py 2>&1 | tee $O/phases_$1.txt | tail -9; shift
done
shift
for v in "$@"; do
  echo "== pmc $v"
  export SLAMIT_LIB=$PWD/tools/diag/lib$v.so
  rm -rf gpurun_out/pmc_a gpurun_out/pmc_b
  rocprofv3 --pmc SQ_WAVES SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_WAVE_CYCLES SQ_BUSY_CYCLES --output-format csv -d gpurun_out/pmc_a -- python3 bench.py --no-cpu --no-ba --no-extras --steps 4 --warmup 1 > /dev/null 2>&1
  rocprofv3 --pmc SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_INSTS_VMEM_RD SQ_WAIT_INST_LDS SQ_ACTIVE_INST_SCA --output-format csv -d gpurun_out/pmc_b -- python3 bench.py --no-cpu --no-ba --no-extras --steps 4 --warmup 1 > /dev/null 2>&1
  python3 tools/diag/pmc_summary.py fast | tee $O/pmc_$v.txt
done
