#!/bin/bash
# k_sweep_st (decoupled coefficient stream of the stored sweeps) against the generic k_sweep on the same box: kernel tests,
# then the bench with VOF_SWEEP_ST=1 / 0
timeout -k 10 300 python -m pytest tests/test_gpu_kernels.py -x -q -m gpu > gpurun_out/r2_t15.log 2>&1; tail -3 gpurun_out/r2_t15.log
for v in 1 0; do
  VOF_SWEEP_ST=$v timeout -k 10 200 python bench.py --steps 2 --warmup 1 --no-cpu-baseline --no-end-to-end --no-variants --profile-table 2> gpurun_out/st$v.log | python -c "import json,sys; d=json.loads(sys.stdin.read()); print('sweep_st=$v', d['value'], d['ms_per_step'], d['config']['iterations_mean'], d['config']['relres_max'])"
  grep -E "^  gs  " gpurun_out/st$v.log | cut -c1-120
done
