#!/bin/bash
# level-0 sweep micro-benchmark under environment switches: scripts/gpu_micro_env.sh "VOF_SWEEP0=0" ...
for P in ${PAIRS:-255 80 24}; do
  python scripts/gpu_sweep_micro.py $P ${SWEEPS:-5} 2>&1 | tail -1
  for E in "$@"; do
    echo -n "[$E] "; env $E python scripts/gpu_sweep_micro.py $P ${SWEEPS:-5} 2>&1 | tail -1
  done
done
