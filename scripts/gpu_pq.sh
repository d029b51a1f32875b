#!/bin/bash
# Does the multigrid preconditioner need the reference's 'dy' == 'dx' quirk?  VOF_PRECOND_QUIRKS=hs (hierarchy, smoother).
mkdir -p gpurun_out/r3
for pq in 11 01 00; do
  VOF_PRECOND_QUIRKS=$pq TAG=pq$pq REGIMES="N,a.1,T,W" SIZES=130,258,514 PYTHONPATH=. timeout -k 10 500 python scripts/gpu_regimes3.py || exit 1
done
