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

cases = [("N  [0,1] a=1 b=1e4", 1.0, 1.0, 1e4), ("8bit a=1e5 b=1e3", 255.0, 1e5, 1e3), ("[0,1] a=10 b=1", 1.0, 10.0, 1.0),
         ("[0,1] a=1 b=1", 1.0, 1.0, 1.0), ("[0,1] a=0.5 b=1e3", 1.0, 0.5, 1e3), ("T  8bit a=1e4 b=1e2", 255.0, 1e4, 1e2)]
sweeps = [((2, 2, 2, 2), -1), ((2, 2, 1, 1), (1, 2)), ((2, 2, 1, 1), (1, 3)), ((2, 2, 1, 1), (1, 4))]
for n in (256, 512):
    for name, scale, al, be in cases:
        mv = tex(n, 4, 5, scale)
        row = []
        for ms, wl in sweeps:
            t = time.time()
            r = of.variational_optical_flow(mv, speed_alpha=al, remodelling_alpha=be, multigrid_sweeps=ms, return_stats=True,
                                            max_iterations=400, w_cycle_level=wl)
            st = r["stats"]
            row.append(f"{ms}W{wl}: its {st['iterations'].max():3d} c{int(st['converged'].all())} {time.time()-t:5.2f}s")
        print(n, name, " | ".join(row), flush=True)
# Gaussian (the reference's enabled test regime)
from opticalflow_amd.optical_flow import make_fake_data_frame
for n in (128, 512):
    fr = [make_fake_data_frame(2.5 + 0.1 * t, 2.5 + 0.2 * t, sigma=3, width=5, dimension=n)[0] + 0.05 * t for t in range(3)]
    mv = np.stack(fr)
    row = []
    for ms, wl in sweeps:
        r = of.variational_optical_flow(mv, speed_alpha=1.0, remodelling_alpha=1e4, multigrid_sweeps=ms, return_stats=True,
                                        w_cycle_level=wl)
        row.append(f"{ms}W{wl}: its {r['stats']['iterations'].max():3d} c{int(r['stats']['converged'].all())}")
    print(n, "G gaussian a=1 b=1e4", " | ".join(row), flush=True)
