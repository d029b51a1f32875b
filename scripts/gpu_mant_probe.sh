#!/bin/bash
# Probe: iterations / pairs per second if the off-diagonal stencil words kept fewer mantissa bits than bfloat16's 7
# (libraries built with -DVOF_EXP_MANT_BITS=m; storage is still 16 bits - only the convergence is probed).
for m in "" _m5 _m4 _m3; do
  for wl in "" "--wobble 0.5"; do
    VOF_LIB=$PWD/opticalflow_amd/csrc/libvof$m.so timeout -k 10 200 python bench.py --steps 2 --warmup 1 --no-cpu-baseline --no-end-to-end --no-variants $wl 2> gpurun_out/mant$m.log | python -c "import json,sys; d=json.loads(sys.stdin.read()); print('lib$m $wl', round(d['value'],1), round(d['ms_per_step'],1), d['config']['iterations_mean'], d['config']['relres_max'])" || exit 1
  done
done
