python -m pytest tests/test_gpu_kernels.py -x -q -m gpu -k "level0 or krylov_product or fused_sweep_equals" 2>&1 | tail -2
for lib in "" _d2 _d1; do
  L=$PWD/opticalflow_amd/csrc/libvof$lib.so
  for P in 255 80; do VOF_LIB=$L python scripts/gpu_sweep_micro.py $P 4 2>&1 | tail -1; done
done
