"""PCIe-inclusive rate of the drop-in entry point (host numpy in, host numpy out) on the bench workload."""
import os, sys, time
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
import numpy as np
import torch
from opticalflow_amd import optical_flow as of, _native
from opticalflow_amd.synthetic import texture_stack_torch
n, T = 1024, 256
movie = texture_stack_torch(n, T, 1, torch.device("cuda", 0)).cpu().numpy()
for rep in range(2):
    t0 = time.time()
    r = of.variational_optical_flow(movie, speed_alpha=1.0, remodelling_alpha=1e4, return_stats=True)
    dt = time.time() - t0
    print(f"drop-in call, host arrays: {dt:.3f} s -> {(T - 1) / dt:.1f} pairs/s (iterations max {r['stats']['iterations'].max()})", flush=True)
p = _native.default_params(speed_alpha=1.0, remodelling_alpha=1e4)
with _native.Solver(n, n, 255) as s:
    for rep in range(2):
        t0 = time.time(); out = s.solve_host(movie, p); dt = time.time() - t0
        print(f"vof_solve_stack_host only (context re-used): {dt:.3f} s -> {(T - 1) / dt:.1f} pairs/s", flush=True)
