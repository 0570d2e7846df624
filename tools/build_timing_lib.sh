#!/bin/bash
# Debug build of the library with per-workgroup timestamps in the tap-table conv kernel (-DNNL_TAPS_TIMING, igemm_taps.h):
#   tools/ab/libnnl_hip_timing.so  — load with NNL_LIB_PATH (see tools/conv_timing.py).  Not part of the product build.
set -e
cd "$(dirname "$0")/.."
mkdir -p tools/ab/timing_obj
for f in neuralnetworklibrary_amd/csrc/*.hip; do
  o=tools/ab/timing_obj/$(basename ${f%.hip}).o
  if [ ! -f $o ] || [ $f -nt $o ] || [ neuralnetworklibrary_amd/csrc/igemm_taps.h -nt $o ]; then
    /opt/rocm/bin/hipcc -O3 -std=c++17 -fPIC --offload-arch=gfx950 -Wall -Wno-unused-function -ffp-contract=off -DNNL_TAPS_TIMING -c $f -o $o &
  fi
done
wait
/opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -fPIC -o tools/ab/libnnl_hip_timing.so tools/ab/timing_obj/*.o
echo built tools/ab/libnnl_hip_timing.so
