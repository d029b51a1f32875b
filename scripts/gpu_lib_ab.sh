#!/bin/bash
# A/B of library builds on one box: usage gpu_lib_ab.sh <suffix> ... (opticalflow_amd/csrc/libvof<suffix>.so; "-" = the default build)
for m in "$@"; do
  [ "$m" = "-" ] && m=""
  for rep in 1 2; do
    VOF_LIB=$PWD/opticalflow_amd/csrc/libvof$m.so timeout -k 10 200 python bench.py --steps 3 --warmup 1 --no-cpu-baseline --no-end-to-end --no-variants --profile-table 2> gpurun_out/lib$m.log | python -c "import json,sys; d=json.loads(sys.stdin.read()); print('lib$m', round(d['value'],1), round(d['ms_per_step'],1), d['config']['iterations_mean'])" || exit 1
  done
  grep -E "^  gs  " gpurun_out/lib$m.log | cut -c1-110
done
