"""End-to-end (PCIe-inclusive) timing of the host API at the bench workload.  GPU box only."""
import os, sys, time
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
import numpy as np, ctypes as C
from opticalflow_amd import _native, optical_flow as of
from opticalflow_amd.synthetic import texture_stack_numpy
n, T = 1024, 256
base = texture_stack_numpy(n, 9, seed=1)
movie = np.concatenate([base] * 29)[:T].copy()      # 8 distinct pairs repeated (a seam pair every 9 frames)
p = _native.default_params(speed_alpha=1.0, remodelling_alpha=1e4)
for B in (96, 64, 128):
    with _native.Solver(n, n, B) as s:
        for rep in range(3):
            outs = [np.empty((T - 1, n, n)) for _ in range(4)]
            st = np.zeros(T - 1, dtype=_native.STATS_DTYPE)
            t0 = time.time()
            rc = s.lib.vof_solve_stack_host(s.h, _native._ptr(movie), T, C.byref(p), *[_native._ptr(o) for o in outs], _native._ptr(st))
            dt = time.time() - t0
            assert rc == 0 and st["converged"].all()
            print(f"B={B} vof_solve_stack_host fresh outputs: {dt:.3f} s = {(T-1)/dt:.0f} pairs/s (iters mean {st['iterations'].mean():.2f})", flush=True)
            del outs
for rep in range(3):
    t0 = time.time(); r = of.variational_optical_flow(movie, remodelling_alpha=1e4); dt = time.time() - t0
    print(f"drop-in call: {dt:.3f} s = {(T-1)/dt:.0f} pairs/s", flush=True)
