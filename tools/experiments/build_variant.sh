#!/bin/bash
# build_variant.sh NAME "<extra hipcc flags>" tu1 [tu2 ...]: recompiles the named translation units of co-snarks_amd/csrc with the
# extra flags and links them with the other objects of the in-tree build into gpurun_ab/libcosnarks_hip_NAME.so (A/B runs on one
# GPU box: COSNARKS_HIP_LIB=gpurun_ab/libcosnarks_hip_NAME.so python tools/gpu_msm_loop.py ...)
set -e
cd "$(dirname "$0")/../.."
name=$1; flags=$2; shift 2
mkdir -p gpurun_ab/obj_$name
objs=""
for o in co-snarks_amd/build/*.o; do
  b=$(basename $o .o); skip=0; case $b in host_*) skip=1;; esac
  for tu in "$@"; do [ "$tu" = "$b" ] && skip=1; done
  [ $skip = 0 ] && objs="$objs $o"
done
for tu in "$@"; do
  /opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -std=c++17 -fPIC -Wno-unused-value -fno-slp-vectorize $flags -c co-snarks_amd/csrc/$tu.hip -o gpurun_ab/obj_$name/$tu.o &
done
wait
for tu in "$@"; do objs="$objs gpurun_ab/obj_$name/$tu.o"; done
/opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -fPIC -o gpurun_ab/libcosnarks_hip_$name.so $objs
echo built gpurun_ab/libcosnarks_hip_$name.so
