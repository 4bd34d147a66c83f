#!/bin/bash
# GPU box: the FAST launch with its workgroups' LDS padded (SLAMIT_FAST_LDS_PAD bytes, a diagnostic build: tools/diag/build_here.sh pad with the
# pad added to orbk_fast's smem) -- how much of the kernel's speed hangs on waves per SIMD (4.6 KB per wave now; a two-cell wave would need 8.4).
export PYTHONPATH=$PWD TMPDIR=/tmp
for rep in 1 2; do
  for pad in 0 2000 3800 6000; do
    SLAMIT_FAST_LDS_PAD=$pad SLAMIT_LIB=$PWD/tools/diag/libpad.so python3 bench.py --no-ba --no-cpu --no-extras --steps 40 2>/dev/null | tail -1 | python3 -c "import json,sys; d=json.loads(sys.stdin.read()); print('pad $pad', round(d['value']), d['ms_per_step'], d['roofline']['kernel_ms'] if 'kernel_ms' in d['roofline'] else d['roofline'])"
  done
done
