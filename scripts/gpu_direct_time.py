"""Time and memory of the direct preconditioner with the in-house dense inverses:
usage: python scripts/gpu_direct_time.py n [alpha beta [calls]]  (8-bit texture, one pair, use_direct_solver=True)"""
import sys, os, time
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
import numpy as np
from opticalflow_amd import optical_flow as of, _native
from opticalflow_amd.synthetic import texture_stack_numpy
n = int(sys.argv[1])
alpha = float(sys.argv[2]) if len(sys.argv) > 2 else 1e4
beta = float(sys.argv[3]) if len(sys.argv) > 3 else 1e2
mv = np.round(texture_stack_numpy(n, 2, 5) * 255.0)
f0, _ = _native.device_memory(0)
for rep in range(int(sys.argv[4]) if len(sys.argv) > 4 else 2):
    t0 = time.time()
    r = of.variational_optical_flow(mv, speed_alpha=alpha, remodelling_alpha=beta, use_direct_solver=True, return_stats=True, max_pairs_in_flight=1)
    dt = time.time() - t0
    f1, _ = _native.device_memory(0)
    st = r["stats"]
    print(f"n={n} alpha={alpha:g} beta={beta:g}: {dt:.2f} s per pair (call {rep}), iterations {st['iterations'].tolist()}, relres {st['relative_residual'].max():.2e}, "
          f"converged {bool(st['converged'].all())}, device memory in use {(f0 - f1) / 1e9:.2f} GB, batch_ms {st['batch_ms'][0]:.0f}", flush=True)
of.release_device_memory()
