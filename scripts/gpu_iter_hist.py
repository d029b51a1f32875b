"""Per-iteration active-pair counts of the default (two-phase warm start) and cold solves of the benchmark stack, and the time
of each BiCGStab iteration (VOF_TRACE=1 makes solve_batch print them)."""
import os, sys, time
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
import numpy as np, torch
from opticalflow_amd import _native
from opticalflow_amd.synthetic import texture_stack_torch
n, T = 1024, 256
P = T - 1
s = _native.Solver(n, n, P)
dev = torch.device("cuda", 0)
movie = texture_stack_torch(n, T, 1, dev, solver=s)
out = [torch.empty((P, n, n), dtype=torch.float64, device=dev) for _ in range(4)]
for stride in (3, 0):
    prm = _native.default_params(remodelling_alpha=1e4, warm_start_stride=stride)
    s.solve_dev(movie, T, prm, *out)
    torch.cuda.synchronize(); t0 = time.perf_counter()
    st = s.solve_dev(movie, T, prm, *out)
    torch.cuda.synchronize(); dt = time.perf_counter() - t0
    it = st["iterations"]
    groups = [("phase1", it[::3]), ("phase2", np.delete(it, np.arange(0, P, 3)))] if stride else [("all", it)]
    print(f"stride {stride}: {P / dt:.1f} pairs/s, mean iterations {it.mean():.2f}")
    for name, g in groups:
        print("   ", name, len(g), "pairs; active at iteration k:", [int((g >= k).sum()) for k in range(1, g.max() + 1)])
