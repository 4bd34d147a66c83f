#!/bin/bash
# Build the WORKING TREE's library with extra flags for orb_kernels.hip into tools/diag/lib<name>.so (same-box A/B runs with
# tools/diag/ab.sh; SLAMIT_LIB selects the library):  tools/diag/build_here.sh <name> [-DMACRO ...]   (HERE_TU=ba_kernels: the flags go to that file instead)
set -e
cd "$(dirname "$0")/../.."
name=$1; shift
tmp=/tmp/here_$name; rm -rf $tmp; mkdir -p $tmp
objs=""
for f in weiner_slamit_v2_amd/csrc/*.hip; do
  b=$(basename $f .hip)
  extra=""
  case $b in ba_kernels|pose) extra="-ffp-contract=fast";; hamming) extra="-mllvm -amdgpu-mfma-vgpr-form";; esac
  if [ "$b" = "${HERE_TU:-orb_kernels}" ]; then extra="$extra $*"; fi
  /opt/rocm/bin/hipcc -O3 -std=c++17 --offload-arch=gfx950 -fPIC -ffp-contract=off -fno-fast-math -fno-gpu-rdc -Wno-unused-value $extra -c $f -o $tmp/$b.o &
  objs="$objs $tmp/$b.o"
done
wait
/opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -fPIC -o tools/diag/lib$name.so $objs
echo tools/diag/lib$name.so
