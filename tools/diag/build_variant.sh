#!/bin/bash
# Build the library of another git revision for same-box A/B runs: tools/diag/build_variant.sh <rev> <name> [-DMACRO ... for orb_kernels.hip] -> tools/diag/lib<name>.so
set -e
cd "$(dirname "$0")/../.."
rev=$1; name=$2; shift 2
tmp=/tmp/variant_$name; rm -rf $tmp; mkdir -p $tmp
git archive $rev weiner_slamit_v2_amd/csrc include | tar -x -C $tmp
objs=""
for f in $tmp/weiner_slamit_v2_amd/csrc/*.hip; do
  extra=""; case $(basename $f) in ba_kernels.hip|pose.hip) extra="-ffp-contract=fast";; hamming.hip) extra="-mllvm -amdgpu-mfma-vgpr-form";; orb_kernels.hip) extra="$*";; esac
  /opt/rocm/bin/hipcc -O3 -std=c++17 --offload-arch=gfx950 -fPIC -ffp-contract=off -fno-fast-math -fno-gpu-rdc -Wno-unused-value $extra -c $f -o ${f%.hip}.o &
  objs="$objs ${f%.hip}.o"
done
wait
/opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -fPIC -o tools/diag/lib$name.so $objs
echo tools/diag/lib$name.so
