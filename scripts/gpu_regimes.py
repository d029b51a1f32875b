"""Iteration counts of the GPU solver across regimes and preconditioner precisions (diagnostic)."""
import sys, os, time
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
import numpy as np
from opticalflow_amd import optical_flow as of
from opticalflow_amd.synthetic import texture_parameters

def tex(n, T, seed, scale=1.0):
    f, g, a, phi = texture_parameters(n, seed)
    i = np.arange(n, dtype=np.float64)
    out = np.empty((T, n, n)); sc = 0.45 * np.sqrt(64) / (3.0 * a.sum())
    for t in range(T):
        p = 2*np.pi*f[:, None]*(i[None, :]-0.3*t)/n + phi[:, None]; q = 2*np.pi*g[:, None]*(i[None, :]-0.6*t)/n
        A = (a[:, None]*np.cos(p)).T @ np.cos(q) - (a[:, None]*np.sin(p)).T @ np.sin(q)
        out[t] = np.clip(0.5 + sc*A, 0, 1)
    return out*scale

cases = [("N  [0,1] a=1 b=1e4", 1.0, 1.0, 1e4), ("T  8bit a=1e4 b=1e2", 255.0, 1e4, 1e2), ("8bit a=1e5 b=1e3", 255.0, 1e5, 1e3),
         ("[0,1] a=0.1 b=1e2", 1.0, 0.1, 1e2), ("[0,1] a=10 b=1", 1.0, 10.0, 1.0)]
for n in (256, 512):
    for name, scale, al, be in cases:
        mv = tex(n, 4, 5, scale)
        row = []
        for vp in ("float64", "float32"):
            for cp in ("float64", "float32"):
                t = time.time()
                r = of.variational_optical_flow(mv, speed_alpha=al, remodelling_alpha=be, vcycle_precision=vp,
                                                coarse_precision=cp, return_stats=True, max_iterations=300)
                st = r["stats"]
                row.append(f"V{vp[-2:]}/C{cp[-2:]}: its {st['iterations'].max():3d} conv {int(st['converged'].all())} rr {st['relative_residual'].max():.1e}")
        print(n, name, " | ".join(row), flush=True)
