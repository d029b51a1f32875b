"""Direct preconditioner at 514x514 in the regime where the multigrid cycle stagnates (8-bit data, speed_alpha = 1e4):
rocSOLVER path (Schur blocks of 1536 unknowns).  Prints timings; VOF_TRACE=1 shows the library load."""
import os, sys, time
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
import numpy as np
from opticalflow_amd import optical_flow as of
from opticalflow_amd.synthetic import texture_stack_numpy
import sys as _s
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
from oracle import vof_oracle as orc
n = int(os.environ.get("VOF_N", "514"))
mv = texture_stack_numpy(n, 3, 5) * 255.0
for rep in range(2):
    t0 = time.time()
    r = of.variational_optical_flow(mv, speed_alpha=1e4, remodelling_alpha=1e2, use_direct_solver=True, return_stats=True)
    st = r["stats"]
    print(f"n={n} use_direct_solver=True call {rep}: {time.time() - t0:7.2f} s  iterations {st['iterations'].tolist()}  relres {st['relative_residual'].max():.2e}  converged {st['converged'].tolist()}", flush=True)
# independent check of pair 0 with the oracle's matrix-free operator (CPU)
xi = np.stack([r["v_x"][0], r["v_y"][0], r["remodelling"][0]])[:, 1:-1, 1:-1]
b = orc.rhs_interior(mv[0], mv[1])
res = b - orc.apply_operator_interior(mv[0], xi, 1e4, 1e2)
print("CPU-evaluated relative residual of pair 0:", np.linalg.norm(res) / np.linalg.norm(b), flush=True)
