# octree LDS budget A/B: bash tools/diag/oct_lds_ab.sh  (on the GPU box)
for rep in 1 2; do
for cfg in "SLAMIT_LIB=tools/diag/liboctnew.so SLAMIT_OCT_KEYS=2048" "SLAMIT_LIB=tools/diag/liboctnew2.so SLAMIT_OCT_KEYS=2048" "SLAMIT_LIB=tools/diag/liboctnew2.so SLAMIT_OCT_KEYS=1536" "SLAMIT_LIB=tools/diag/liboctnew2.so SLAMIT_OCT_KEYS=2560"  "SLAMIT_LIB=tools/diag/liboctnew2.so SLAMIT_OCT_KEYS=3072"; do
  env $cfg timeout -k 10 200 python3 bench.py --no-ba --no-cpu --no-extras --steps 40 --reps 3 > gpurun_out/o.json 2>/dev/null || exit 1
  python3 -c "
import json,sys
d=json.loads([l for l in open('gpurun_out/o.json') if l.startswith('{')][-1])
print('$cfg', d['ms_per_step'], d['stage_ms_per_step']['octree'], d['stage_ms_per_step']['blur'])"
done
done
