#!/bin/bash
# micro A/B of the level-0 sweep: scripts/gpu_micro_ab.sh "<flags B>" ["<flags C>" ...]; prints per-launch times of vof_bench_sweeps_dev
cd opticalflow_amd/csrc
i=0
LIBS=("")
for F in "$@"; do
  i=$((i+1))
  /opt/rocm/bin/hipcc -O3 -std=c++17 --offload-arch=gfx950 -Wno-unused-value -shared -fPIC -pthread $F -o libvof_x$i.so vof.hip || exit 1
  LIBS+=("$PWD/libvof_x$i.so")
done
cd ../..
for P in ${PAIRS:-255 80 24}; do
  for L in "${LIBS[@]}"; do
    VOF_LIB=$L python scripts/gpu_sweep_micro.py $P ${SWEEPS:-5} 2>&1 | tail -1
  done
done
