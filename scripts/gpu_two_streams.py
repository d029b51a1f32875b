"""Experiment: the stack solved as two halves by two contexts (two streams, two host threads) at the same time - do the
VALU-bound smoother passes of one half overlap the bandwidth-bound kernels of the other?  1024^2 x 256."""
import os, sys, time, threading
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
import numpy as np, torch
from opticalflow_amd import _native
from opticalflow_amd.synthetic import texture_stack_torch
n, T = 1024, 256
dev = torch.device("cuda", 0)
P = T - 1
prm = _native.default_params(speed_alpha=1.0, remodelling_alpha=1e4)
s0 = _native.Solver(n, n, P)
movie = texture_stack_torch(n, T, 1, dev, solver=s0)
outs = [torch.empty((P, n, n), dtype=torch.float64, device=dev) for _ in range(4)]
torch.cuda.synchronize()
def run_one():
    return s0.solve_dev(movie, T, prm, *outs)
for _ in range(2): st = run_one()
torch.cuda.synchronize(); t = time.perf_counter()
for _ in range(3): st = run_one()
torch.cuda.synchronize(); dt = (time.perf_counter() - t) / 3
print(f"one context: {P / dt:.0f} pairs/s, iterations {st['iterations'].mean():.2f}", flush=True)
s0.close()
for parts in (2, 3):
    cuts = [round(i * P / parts) for i in range(parts + 1)]
    solvers = [_native.Solver(n, n, cuts[i + 1] - cuts[i]) for i in range(parts)]
    stats = [None] * parts
    def work(i):
        a, b = cuts[i], cuts[i + 1]
        stats[i] = solvers[i].solve_dev(movie[a:b + 1], b - a + 1, prm, *[o[a:b] for o in outs])
    def run_all():
        th = [threading.Thread(target=work, args=(i,)) for i in range(parts)]
        [t.start() for t in th]; [t.join() for t in th]
    for _ in range(2): run_all()
    torch.cuda.synchronize(); t = time.perf_counter()
    for _ in range(3): run_all()
    torch.cuda.synchronize(); dt = (time.perf_counter() - t) / 3
    its = np.concatenate([s["iterations"] for s in stats])
    print(f"{parts} contexts at once: {P / dt:.0f} pairs/s, iterations {its.mean():.2f}, converged {all(s['converged'].all() for s in stats)}", flush=True)
    for s in solvers: s.close()
