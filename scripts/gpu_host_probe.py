import os, sys, time
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
import numpy as np, torch, ctypes as C
from opticalflow_amd import _native
from opticalflow_amd.synthetic import texture_stack_torch
n, T = 1024, 256
dev = torch.device("cuda", 0)
movie_d = texture_stack_torch(n, T, 1, dev)
movie = movie_d.cpu().numpy()
def t(f, label):
    torch.cuda.synchronize(); t0 = time.time(); r = f(); torch.cuda.synchronize(); print(f"{label}: {time.time()-t0:.3f} s", flush=True); return r
t(lambda: torch.from_numpy(movie).to(dev), "H2D 2GB pageable (torch)")
t(lambda: torch.from_numpy(movie).to(dev), "H2D 2GB pageable (torch) again")
out = t(lambda: np.empty((4, T - 1, n, n)), "np.empty 8GB")
big = torch.empty((4, T - 1, n, n), dtype=torch.float64, device=dev)
t(lambda: big.cpu(), "D2H 8GB torch .cpu() (fresh pageable)")
host = torch.empty((4, T - 1, n, n), dtype=torch.float64)
t(lambda: host.copy_(big), "D2H 8GB into untouched torch CPU tensor")
t(lambda: host.copy_(big), "D2H 8GB into touched torch CPU tensor")
p = _native.default_params(speed_alpha=1.0, remodelling_alpha=1e4)
with _native.Solver(n, n, 96) as s:
    vx = torch.empty((T - 1, n, n), dtype=torch.float64, device=dev); vy = torch.empty_like(vx); gm = torch.empty_like(vx); sp = torch.empty_like(vx)
    t(lambda: s.solve_dev(movie_d, T, p, vx, vy, gm, sp), "solve_dev B=96")
    t(lambda: s.solve_dev(movie_d, T, p, vx, vy, gm, sp), "solve_dev B=96 again")
    outs = [np.empty((T - 1, n, n)) for _ in range(4)]
    st = np.zeros(T - 1, dtype=_native.STATS_DTYPE)
    def call():
        rc = s.lib.vof_solve_stack_host(s.h, _native._ptr(movie), T, C.byref(p), *[_native._ptr(o) for o in outs], _native._ptr(st)); assert rc == 0
    t(call, "vof_solve_stack_host into untouched outputs")
    t(call, "vof_solve_stack_host into touched outputs")
    t(call, "vof_solve_stack_host into touched outputs again")
