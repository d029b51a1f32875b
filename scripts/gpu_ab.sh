#!/bin/bash
# A/B of two library builds on the same box, interleaved: scripts/gpu_ab.sh "<extra hipcc flags for B>" [bench args]
FLAGS="$1"; shift
cd opticalflow_amd/csrc && /opt/rocm/bin/hipcc -O3 -std=c++17 --offload-arch=gfx950 -Wno-unused-value -shared -fPIC $FLAGS -o libvof_b.so vof.hip && cd ../..
for i in 1 2; do
  echo "--- A (default build)"; python bench.py --steps 2 --no-cpu-baseline --profile-table "$@" 2>&1 | grep -E "^  (gs|apply0|residual) .*L[01] |value" | cut -c1-118
  echo "--- B ($FLAGS)"; VOF_LIB=$PWD/opticalflow_amd/csrc/libvof_b.so python bench.py --steps 2 --no-cpu-baseline --profile-table "$@" 2>&1 | grep -E "^  (gs|apply0|residual) .*L[01] |value" | cut -c1-118
done
