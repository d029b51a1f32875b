python -m pytest tests/test_gpu_kernels.py -m gpu -q -x > gpurun_out/pytest_gpu.log 2>&1; echo pytest rc=$?; tail -3 gpurun_out/pytest_gpu.log
for geo in AB AA BB; do
  VOF_SWEEP_GEO=$geo python bench.py --steps 2 --no-cpu-baseline --profile-table > gpurun_out/bench_geo_$geo.log 2>&1; echo "== $geo rc=$?"; grep -E "^  gs" gpurun_out/bench_geo_$geo.log | head -4; tail -1 gpurun_out/bench_geo_$geo.log | cut -c1-120
done
VOF_SWEEP_GEO=BB python -m pytest tests/test_gpu_kernels.py -m gpu -q -x -k "sweep or vcycle" 2>&1 | tail -2
