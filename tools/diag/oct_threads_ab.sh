# octree workgroup size for small batches: bash tools/diag/oct_threads_ab.sh (GPU box)
for rep in 1 2; do for nt in 512 1024; do
  SLAMIT_OCT_THREADS=$nt python3 tools/diag/host_api_time.py 2>/dev/null | head -3 | tr '\n' ' '; echo " [threads $nt]"
  SLAMIT_OCT_THREADS=$nt timeout -k 10 200 python3 bench.py --config pipeline --no-cpu --reps 2 2>/dev/null | python3 -c "
import json,sys
d=json.loads([l for l in sys.stdin if l.startswith('{')][-1]); print('pipeline threads $nt', d['ms_per_step'], d['stage_ms_per_step'])"
done; done
