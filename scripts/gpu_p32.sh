#!/bin/bash
# packed float32 level-0 smoother (vcycle_precision 4) against the float64 one: bench line + regime table
mkdir -p gpurun_out/r3
for vp in coarse_float32 auto float32; do
  timeout -k 10 300 python bench.py --profile-table --no-variants --no-end-to-end --no-cpu-baseline --vcycle-precision $vp > gpurun_out/r3/bench_p32_$vp.json 2> gpurun_out/r3/bench_p32_$vp.log || exit 1
  echo $vp; grep -E "gs0|apply0|all kernels" gpurun_out/r3/bench_p32_$vp.log
  python3 -c "import json;d=json.load(open('gpurun_out/r3/bench_p32_$vp.json'));print(d['value'], d['config']['iterations_mean'], d['config']['iterations_max'], d['config']['relres_max'], d['roofline']['frac'])"
done
