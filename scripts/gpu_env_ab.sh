#!/bin/bash
# A/B via an environment switch on the same box: scripts/gpu_env_ab.sh VAR
V=$1
for i in 1 2; do
  echo "--- A (default)"; python bench.py --steps 2 --no-cpu-baseline --profile-table 2>&1 | grep -E "^  (gs0|gs|prolong|apply0|restrict) .*L[01] |value" | cut -c1-118
  echo "--- B ($V=0)"; env $V=0 python bench.py --steps 2 --no-cpu-baseline --profile-table 2>&1 | grep -E "^  (gs0|gs|prolong|apply0|restrict) .*L[01] |value" | cut -c1-118
done
