"""Executable specification (numpy/scipy, CPU) of the solver the HIP kernels implement.
TEST INFRASTRUCTURE ONLY - imported by tests/ to check individual kernels (operator, smoother,
Galerkin coarse operators, transfer operators, V-cycle) against an independent implementation.

Algorithm:
  * boundary unknowns of the reference system (OF.py:964-1070) are eliminated: unknowns live on the
    interior grid ``n = N - 2`` per axis; ghost values fold onto interior points
    (edge ghost -> mirror point, corner ghost -> 2 x the diagonal mirror point);
  * the folded operator is a 9-point stencil of 3x3 blocks ``C[a, b, r, c, p, q]``
    (a, b = offset + 1 along axis 0 / 1; r = equation, c = unknown);
  * geometric multigrid: vertex-centred coarsening ``n_c = ceil(n/2)`` (coarse c <-> fine 2c),
    bilinear prolongation P (orphan last odd point copies its left coarse neighbour),
    restriction R = P^T / 4, Galerkin coarse operators ``A_c = R A P`` (again 9-point),
    smoother = 4-colour 3x3-block Gauss-Seidel (colour = 2 (p mod 2) + (q mod 2), order 0,1,2,3
    before and 3,2,1,0 after the coarse-grid correction), dense solve on the coarsest grid
    (``max(n) <= 5``);
  * outer iteration: right-preconditioned BiCGStab (the reference's KSP type, OF.py:1081) with one
    V-cycle as preconditioner; stop when ``||b - A x||_2 <= rtol ||b||_2`` (OF.py:1120,1126).
"""
from __future__ import annotations

import numpy as np
import scipy.sparse as sp

from . import vof_oracle as orc

COARSEST_MAX = 5


# ------------------------------------------------------------------ fine-level folded stencil
def fine_stencil(I, alpha, beta, reference_quirks=True):
    """Folded 9-point block stencil ``C[3,3,3,3,n_i,n_j]`` of the interior (eliminated) system."""
    I = np.asarray(I, dtype=np.float64)
    d = orc.derivatives(I, I, reference_quirks)
    n_i, n_j = d["P"].shape
    C = np.zeros((3, 3, 3, 3, n_i, n_j))
    for (rq, di, dj, cq, coef) in orc._interior_stencil(d, alpha, beta):
        C[di + 1, dj + 1, rq, cq] += coef
    # fold ghosts along axis 0
    if True:
        F = C.copy()
        # p = 0: offset -1 -> point 1 (offset +1); p = n_i-1: offset +1 -> point n_i-2 (offset -1)
        F[2, :, :, :, 0, :] += C[0, :, :, :, 0, :]; F[0, :, :, :, 0, :] = 0
        F[0, :, :, :, -1, :] += C[2, :, :, :, -1, :]; F[2, :, :, :, -1, :] = 0
        C = F
        F = C.copy()
        F[:, 2, :, :, :, 0] += C[:, 0, :, :, :, 0]; F[:, 0, :, :, :, 0] = 0
        F[:, 0, :, :, :, -1] += C[:, 2, :, :, :, -1]; F[:, 2, :, :, :, -1] = 0
        C = F
    # corner ghosts carry a factor 2 (x(0,0) = x(2,0) + x(0,2) = 2 x(2,2)); after the two folds the
    # diagonal-offset coefficient of a corner point sits at the inward diagonal offset, but that
    # slot also holds the genuine inward-diagonal coefficient and the two edge-folded ones.
    # Re-build the corner-ghost contribution explicitly: add it once more.
    raw = np.zeros((3, 3, 3, 3, n_i, n_j))
    for (rq, di, dj, cq, coef) in orc._interior_stencil(d, alpha, beta):
        raw[di + 1, dj + 1, rq, cq] += coef
    for (p, a) in ((0, 0), (n_i - 1, 2)):
        for (q, b) in ((0, 0), (n_j - 1, 2)):
            C[2 - a, 2 - b, :, :, p, q] += raw[a, b, :, :, p, q]
    return C


def stencil_to_sparse(C):
    """Sparse matrix of a block stencil; unknown ordering (field, p, q) -> c * n_i n_j + p n_j + q."""
    n_i, n_j = C.shape[-2:]
    N = n_i * n_j
    pp, qq = np.meshgrid(np.arange(n_i), np.arange(n_j), indexing="ij")
    rows, cols, vals = [], [], []
    for a in range(3):
        for b in range(3):
            tp, tq = pp + a - 1, qq + b - 1
            ok = (tp >= 0) & (tp < n_i) & (tq >= 0) & (tq < n_j)
            for r in range(3):
                for c in range(3):
                    v = C[a, b, r, c]
                    m = ok & (v != 0)
                    rows.append(r * N + (pp * n_j + qq)[m])
                    cols.append(c * N + (tp * n_j + tq)[m])
                    vals.append(v[m])
    return sp.coo_matrix((np.concatenate(vals), (np.concatenate(rows), np.concatenate(cols))),
                         shape=(3 * N, 3 * N)).tocsr()


def sparse_to_stencil(A, n_i, n_j):
    N = n_i * n_j
    A = A.tocoo()
    r, rp = np.divmod(A.row, N); c, cp = np.divmod(A.col, N)
    p, q = np.divmod(rp, n_j); tp, tq = np.divmod(cp, n_j)
    a, b = tp - p + 1, tq - q + 1
    assert a.min() >= 0 and a.max() <= 2 and b.min() >= 0 and b.max() <= 2
    C = np.zeros((3, 3, 3, 3, n_i, n_j))
    np.add.at(C, (a, b, r, c, p, q), A.data)
    return C


def apply_stencil(C, x):
    """y = A x for x of shape (3, n_i, n_j); zero outside the grid."""
    n_i, n_j = x.shape[-2:]
    xp = np.zeros((3, n_i + 2, n_j + 2)); xp[:, 1:-1, 1:-1] = x
    y = np.zeros_like(x)
    for a in range(3):
        for b in range(3):
            nb = xp[:, a:a + n_i, b:b + n_j]
            for r in range(3):
                for c in range(3):
                    y[r] += C[a, b, r, c] * nb[c]
    return y


# ------------------------------------------------------------------ transfer operators
def coarse_size(n):
    return (n + 1) // 2


def prolong_1d(n):
    nc = coarse_size(n)
    P = np.zeros((n, nc))
    for f in range(n):
        if f % 2 == 0:
            P[f, f // 2] = 1.0
        else:
            l, r = (f - 1) // 2, (f + 1) // 2
            if r < nc:
                P[f, l] = 0.5; P[f, r] = 0.5
            else:
                P[f, l] = 1.0
    return P


def prolong(xc, n_i, n_j):
    """Bilinear prolongation of (3, nc_i, nc_j) to (3, n_i, n_j)."""
    Pi, Pj = prolong_1d(n_i), prolong_1d(n_j)
    return np.einsum("pa,cab,qb->cpq", Pi, xc, Pj)


def restrict(xf):
    """R = P^T / 4 applied to (3, n_i, n_j)."""
    n_i, n_j = xf.shape[-2:]
    Pi, Pj = prolong_1d(n_i), prolong_1d(n_j)
    return np.einsum("pa,cpq,qb->cab", Pi, xf, Pj) / 4.0


def galerkin(C):
    """Coarse stencil ``R A P`` (R = P^T/4) of a fine stencil."""
    n_i, n_j = C.shape[-2:]
    Pi, Pj = sp.csr_matrix(prolong_1d(n_i)), sp.csr_matrix(prolong_1d(n_j))
    P2 = sp.kron(Pi, Pj, format="csr")
    P = sp.kron(sp.identity(3), P2, format="csr")
    A = stencil_to_sparse(C)
    Ac = (P.T @ A @ P) / 4.0
    return sparse_to_stencil(Ac, coarse_size(n_i), coarse_size(n_j))


# ------------------------------------------------------------------ smoother
def _solve3(D, r):
    """Solve the 3x3 systems D[r,c,...] x = r[c,...] by Cramer's rule (the kernel's formula)."""
    a, b, c = D[0, 0], D[0, 1], D[0, 2]
    d, e, f = D[1, 0], D[1, 1], D[1, 2]
    g, h, i = D[2, 0], D[2, 1], D[2, 2]
    co00 = e * i - f * h; co01 = -(d * i - f * g); co02 = d * h - e * g
    det = a * co00 + b * co01 + c * co02
    inv = 1.0 / det
    x0 = (r[0] * co00 + r[1] * -(b * i - c * h) + r[2] * (b * f - c * e)) * inv
    x1 = (r[0] * co01 + r[1] * (a * i - c * g) + r[2] * -(a * f - c * d)) * inv
    x2 = (r[0] * co02 + r[1] * -(a * h - b * g) + r[2] * (a * e - b * d)) * inv
    return np.stack([x0, x1, x2])


def gs_colour(C, x, b, colour):
    """In-place block Gauss-Seidel update of one colour."""
    cp, cq = colour >> 1, colour & 1
    n_i, n_j = x.shape[-2:]
    xp = np.zeros((3, n_i + 2, n_j + 2)); xp[:, 1:-1, 1:-1] = x
    sl = (slice(cp, n_i, 2), slice(cq, n_j, 2))
    rhs = b[(slice(None),) + sl].copy()
    for a in range(3):
        for bb in range(3):
            if a == 1 and bb == 1:
                continue
            nb = xp[:, a:a + n_i, bb:bb + n_j][(slice(None),) + sl]
            for r in range(3):
                for c in range(3):
                    rhs[r] -= C[a, bb, r, c][sl] * nb[c]
    D = C[1, 1][(slice(None), slice(None)) + sl]
    x[(slice(None),) + sl] = _solve3(D, rhs)


def smooth(C, x, b, sweeps, reverse=False):
    order = (3, 2, 1, 0) if reverse else (0, 1, 2, 3)
    for _ in range(sweeps):
        for col in order:
            gs_colour(C, x, b, col)


# ------------------------------------------------------------------ hierarchy / V-cycle
class Hierarchy:
    def __init__(self, I, alpha, beta, reference_quirks=True, coarsest_max=COARSEST_MAX):
        self.levels = [fine_stencil(I, alpha, beta, reference_quirks)]
        while max(self.levels[-1].shape[-2:]) > coarsest_max:
            self.levels.append(galerkin(self.levels[-1]))
        Cc = self.levels[-1]
        self.coarse_dense = stencil_to_sparse(Cc).toarray()
        self.coarse_inv = np.linalg.inv(self.coarse_dense)

    def shapes(self):
        return [c.shape[-2:] for c in self.levels]

    def apply(self, x):
        return apply_stencil(self.levels[0], x)

    def vcycle(self, b, nu1=2, nu2=2, level=0):
        C = self.levels[level]
        if level == len(self.levels) - 1:
            return (self.coarse_inv @ b.ravel()).reshape(b.shape)
        x = np.zeros_like(b)
        smooth(C, x, b, nu1)
        r = b - apply_stencil(C, x)
        ec = self.vcycle(restrict(r), nu1, nu2, level + 1)
        x += prolong(ec, *x.shape[-2:])
        smooth(C, x, b, nu2, reverse=True)
        return x


def bicgstab(apply_A, precond, b, x0=None, rtol=1e-6, max_it=1000):
    """Right-preconditioned BiCGStab; returns (x, iterations, relres history)."""
    x = np.zeros_like(b) if x0 is None else x0.copy()
    r = b - apply_A(x)
    bn = np.linalg.norm(b)
    hist = [np.linalg.norm(r) / bn]
    if hist[-1] <= rtol:
        return x, 0, hist
    rh = r.copy()
    rho = alpha = omega = 1.0
    v = np.zeros_like(b); p = np.zeros_like(b)
    for it in range(1, max_it + 1):
        rho_new = np.vdot(rh, r)
        beta = (rho_new / rho) * (alpha / omega)
        rho = rho_new
        p = r + beta * (p - omega * v)
        y = precond(p)
        v = apply_A(y)
        alpha = rho / np.vdot(rh, v)
        s = r - alpha * v
        if np.linalg.norm(s) / bn <= rtol:
            x += alpha * y
            hist.append(np.linalg.norm(s) / bn)
            return x, it, hist
        z = precond(s)
        t = apply_A(z)
        omega = np.vdot(t, s) / np.vdot(t, t)
        x += alpha * y + omega * z
        r = s - omega * t
        hist.append(np.linalg.norm(r) / bn)
        if hist[-1] <= rtol:
            return x, it, hist
    return x, max_it, hist


def solve_pair(I, J, alpha, beta, rtol=1e-6, max_it=1000, nu1=2, nu2=2, reference_quirks=True):
    """Interior solution (3, n_i, n_j), iteration count, residual history."""
    H = Hierarchy(I, alpha, beta, reference_quirks)
    b = orc.rhs_interior(np.asarray(I, float), np.asarray(J, float), reference_quirks)
    x, it, hist = bicgstab(H.apply, lambda r: H.vcycle(r, nu1, nu2), b, rtol=rtol, max_it=max_it)
    return x, it, hist
