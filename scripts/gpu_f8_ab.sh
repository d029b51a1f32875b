#!/bin/bash
# 8-bit float stencils (coarse_precision=3) against the packed bfloat16 default: kernel + parity tests, then the bench
timeout -k 10 600 python -m pytest tests/test_gpu_kernels.py tests/test_gpu_parity.py -x -q -m gpu > gpurun_out/f8_tests.log 2>&1; tail -15 gpurun_out/f8_tests.log
for cp in bfloat16 float8; do
  for wl in "" "--wobble 0.5"; do
    timeout -k 10 200 python bench.py --steps 2 --warmup 1 --no-cpu-baseline --no-end-to-end --no-variants --coarse-precision $cp $wl --profile-table 2> gpurun_out/f8_$cp.log | python -c "import json,sys; d=json.loads(sys.stdin.read()); print('$cp $wl', round(d['value'],1), round(d['ms_per_step'],1), d['config']['iterations_mean'], d['config']['relres_max'])" || exit 1
  done
done
