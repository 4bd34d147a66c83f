#!/bin/bash
# GPU box, round 3: LDS-DMA alignment probe, parity of the FAST variants, same-box A/B against the round-2 kernel.
export PYTHONPATH=$PWD TMPDIR=/tmp
O=gpurun_out/r3_fast; mkdir -p $O
tools/diag/ubench/glds_unaligned > $O/glds_unaligned.txt 2>&1; tail -3 $O/glds_unaligned.txt
for v in "$@"; do
  echo "== parity $v"
  SLAMIT_LIB=$PWD/tools/diag/lib$v.so timeout -k 10 600 python3 -m pytest tests/test_gpu_orb.py -x -q -m gpu > $O/pytest_$v.txt 2>&1
  rc=$?; tail -3 $O/pytest_$v.txt
  if [ $rc -ne 0 ]; then echo "parity of $v FAILED (rc $rc): not benchmarked"; continue; fi
  ok="$ok tools/diag/lib$v.so"
done
tools/diag/ab.sh tools/diag/libbase.so $ok 2>&1 | tee $O/ab.txt
for lib in tools/diag/libbase.so $ok; do
  SLAMIT_LIB=$PWD/$lib python3 bench.py --config 720p --no-ba --no-cpu --steps 10 2>/dev/null | tail -1 | python3 -c "import json,sys; d=json.loads(sys.stdin.read()); print('720p $lib', round(d['value']), d['ms_per_step'], d['roofline'])" | tee -a $O/ab.txt
done
