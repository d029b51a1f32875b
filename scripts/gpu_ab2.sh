#!/bin/bash
# A/B of the in-tree libvof.so (A) against a pre-built libvof_base.so (B) on the same box, interleaved
for i in 1 2; do
  echo "--- A (current)"; python bench.py --steps 2 --no-cpu-baseline --profile-table "$@" 2>&1 | grep -E "^  (gs0|gs|apply0|residual) .*L[012] |value" | cut -c1-118
  echo "--- B (base)"; VOF_LIB=$PWD/opticalflow_amd/csrc/libvof_base.so python bench.py --steps 2 --no-cpu-baseline --profile-table "$@" 2>&1 | grep -E "^  (gs0|gs|apply0|residual) .*L[012] |value" | cut -c1-118
done
