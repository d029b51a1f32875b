"""8-bit against bfloat16 stencil storage over random intensity scales and regularisation weights: iteration counts and
convergence flags side by side (preconditioner only - the answers agree to the stopping rule either way).  GPU box only."""
import os, sys
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
import numpy as np
from opticalflow_amd import optical_flow as of
from opticalflow_amd.synthetic import texture_stack_numpy
worse = 0
for case in range(int(sys.argv[1]) if len(sys.argv) > 1 else 40):
    rng = np.random.default_rng(9000 + case)
    n_i, n_j = int(rng.integers(60, 300)), int(rng.integers(60, 300))
    scale = float(10 ** rng.uniform(-3, 3))
    alpha = float(10 ** rng.uniform(-1, 2)) * scale ** 2
    beta = float(10 ** rng.uniform(-2, 6)) * (1.0 if case % 2 else scale ** 2)
    mv = texture_stack_numpy(max(n_i, n_j), 3, seed=300 + case)[:, :n_i, :n_j] * scale
    out = {}
    for fmt in ("bfloat16", "float8"):
        r = of.variational_optical_flow(mv, speed_alpha=alpha, remodelling_alpha=beta, coarse_precision=fmt, return_stats=True,
                                        max_iterations=300)
        st = r["stats"]
        out[fmt] = (st["iterations"].tolist(), bool(st["converged"].all()), float(st["relative_residual"].max()))
    a, b = out["bfloat16"], out["float8"]
    flag = ""
    if (a[1] and not b[1]) or max(b[0]) > 1.25 * max(a[0]) + 1:
        worse += 1; flag = "  <-- float8 worse"
    print(f"case {case}: {n_i}x{n_j} scale {scale:.1e} alpha {alpha:.1e} beta {beta:.1e}: bfloat16 {a[0]} {a[1]}  float8 {b[0]} {b[1]}{flag}", flush=True)
print("cases where float8 is notably worse:", worse)
