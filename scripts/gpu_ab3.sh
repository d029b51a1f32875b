#!/bin/bash
# A/B/C of library builds + env on one box: scripts/gpu_ab3.sh "<flags B>" "<flags C>" ...; env EXTRA_ENV="VAR=val" adds a run of the default build under it
cd opticalflow_amd/csrc
i=0; LIBS=("")
for F in "$@"; do i=$((i+1)); /opt/rocm/bin/hipcc -O3 -std=c++17 --offload-arch=gfx950 -Wno-unused-value -shared -fPIC -pthread $F -o libvof_y$i.so vof.hip || exit 1; LIBS+=("$PWD/libvof_y$i.so"); done
cd ../..
show() { grep -E "^  (gs |residual|gs0|apply0) .*L[0-3] " | cut -c1-110; }
for rep in 1 2; do
  for L in "${LIBS[@]}"; do
    echo "--- lib=${L##*/}"; VOF_LIB=$L python bench.py --steps 2 --no-cpu-baseline --no-end-to-end --no-variants --profile-table 2> /tmp/ab.log | python -c "import json,sys; d=json.loads(sys.stdin.read()); print('pairs/s', round(d['value'],1))"; show < /tmp/ab.log
  done
  if [ -n "$EXTRA_ENV" ]; then echo "--- env $EXTRA_ENV"; env $EXTRA_ENV python bench.py --steps 2 --no-cpu-baseline --no-end-to-end --no-variants --profile-table 2> /tmp/ab.log | python -c "import json,sys; d=json.loads(sys.stdin.read()); print('pairs/s', round(d['value'],1))"; show < /tmp/ab.log; fi
done
