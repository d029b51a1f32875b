"""Generate the golden fixtures in this directory from the REFERENCE itself.

Runs only in the build container (needs /root/reference, which never travels to the GPU box).
The reference module ``source/optical_flow.py`` is imported, not copied.  Four of its imports
(numba, cv2, petsc4py, skimage) are not installed here; none of them carries arithmetic of the
path we pin:

* ``numba.jit/njit`` are decorators -> identity (the decorated bodies run as plain numpy);
* ``cv2`` is not used by ``variational_optical_flow``;
* ``petsc4py.PETSc`` objects are constructed unconditionally (OF.py:1081-1126) but never used for
  arithmetic when ``use_direct_solver=True`` (OF.py:1146-1147, the branch we run) -> MagicMock;
* ``skimage.filters.gaussian(img, sigma, preserve_range=True)`` is only reached for the blur
  fixture (G4); skimage implements it as ``scipy.ndimage.gaussian_filter(img, sigma,
  mode='nearest', truncate=4.0)``, which is what the stand-in calls.

Everything else - derivative rules, the 43-entry stencil assembly, boundary rows, the SuperLU
direct solve, mirror fix-up, functionals, unit scaling - executes from the reference's source.
``scipy.sparse.linalg.spsolve`` is wrapped to capture the assembled ``(A, b)``.

Usage:  python tests/golden/make_golden.py   (writes tests/golden/*.npz)
"""
import io
import os
import sys
import types
import contextlib
from unittest import mock

import numpy as np
import scipy.ndimage
import scipy.sparse
import scipy.sparse.linalg

HERE = os.path.dirname(os.path.abspath(__file__))
sys.dont_write_bytecode = True


def import_reference():
    def _ident(*a, **k):
        return a[0] if (len(a) == 1 and callable(a[0]) and not k) else (lambda f: f)
    nb = types.ModuleType("numba"); nb.jit = nb.njit = _ident; sys.modules["numba"] = nb
    sys.modules["cv2"] = types.ModuleType("cv2")
    p4 = types.ModuleType("petsc4py"); p4.PETSc = mock.MagicMock(); sys.modules["petsc4py"] = p4
    sk, skf = types.ModuleType("skimage"), types.ModuleType("skimage.filters")
    skf.gaussian = lambda img, sigma, preserve_range=True: scipy.ndimage.gaussian_filter(
        np.asarray(img, float), sigma, mode="nearest", truncate=4.0)
    sk.filters = skf; sys.modules["skimage"] = sk; sys.modules["skimage.filters"] = skf
    import matplotlib
    matplotlib.use("Agg")
    sys.path.insert(0, "/root/reference/source")
    import optical_flow
    return optical_flow


captured = []
_real_spsolve = scipy.sparse.linalg.spsolve


def _capturing_spsolve(A, b, *a, **k):
    captured.append((A.copy(), np.array(b, copy=True)))
    return _real_spsolve(A, b, *a, **k)


def run_ref(OF, movie, **kw):
    captured.clear()
    scipy.sparse.linalg.spsolve = _capturing_spsolve
    try:
        with contextlib.redirect_stdout(io.StringIO()):
            res = OF.variational_optical_flow(np.array(movie, copy=True), use_direct_solver=True, **kw)
    finally:
        scipy.sparse.linalg.spsolve = _real_spsolve
    return res, list(captured)


def scalars(res):
    return dict(L1_functional=float(res["L1_functional"]),
                remodelling_functional=float(res["remodelling_functional"]),
                speed_functional=float(res["speed_functional"]))


def main():
    sys.path.insert(0, os.path.join(HERE, "..", ".."))
    from oracle import vof_oracle as orc
    OF = import_reference()
    only = [a.split("=", 1)[1] for a in sys.argv if a.startswith("--only=")]   # e.g. --only=g10_ : add one fixture,
    if "--only-g9" in sys.argv:                                                 # leave the others untouched
        only.append("g9_")
    if only:
        np_savez = np.savez
        def _only(path, *a, **k):
            if os.path.basename(path).startswith(tuple(only)):
                np_savez(path, *a, **k)
        np.savez = _only

    # ---- G1: the reference's only enabled experiment, AVOF.py:26-50 -------------------
    f1, dx = OF.make_fake_data_frame(x_position=2.5, y_position=2.5, sigma=3, width=5, dimension=50,
                                     include_noise=False)
    f2, _ = OF.make_fake_data_frame(x_position=2.6, y_position=2.7, sigma=3, width=5, dimension=50,
                                    include_noise=False)
    f2 = f2 + 0.05
    movie = np.stack((f1, f2))
    kw = dict(delta_x=dx, delta_t=1.0, speed_alpha=1.0, remodelling_alpha=10000.0, smoothing_sigma=None)
    res, _ = run_ref(OF, movie, **kw)
    np.savez(os.path.join(HERE, "g1_avof_simple_50.npz"), movie=movie, v_x=res["v_x"], v_y=res["v_y"],
             remodelling=res["remodelling"], speed=res["speed"], **kw_to_np(kw), **scalars(res))
    print("G1 mean v_x %.15e mean v_y %.15e mean gamma %.15e" % (
        res["v_x"].mean(), res["v_y"].mean(), res["remodelling"].mean()))

    # ---- G2: tiny rectangular random pair with the captured matrix --------------------
    rng = np.random.default_rng(42)
    movie = rng.random((2, 6, 7))
    kw = dict(delta_x=1.0, delta_t=1.0, speed_alpha=2.0, remodelling_alpha=3.0)
    res, cap = run_ref(OF, movie, **kw)
    A, b = cap[0]
    A = A.tocsr(); A.sort_indices()
    np.savez(os.path.join(HERE, "g2_matrix_6x7.npz"), movie=movie, indptr=A.indptr, indices=A.indices,
             data=A.data, b=b, v_x=res["v_x"], v_y=res["v_y"], remodelling=res["remodelling"],
             **kw_to_np(kw), **scalars(res))

    # a second captured matrix, 9x11, other parameters
    movie = rng.random((2, 9, 11)) * 3.0
    kw = dict(delta_x=1.0, delta_t=1.0, speed_alpha=2.5, remodelling_alpha=7.0)
    res, cap = run_ref(OF, movie, **kw)
    A, b = cap[0]
    A = A.tocsr(); A.sort_indices()
    np.savez(os.path.join(HERE, "g2b_matrix_9x11.npz"), movie=movie, indptr=A.indptr, indices=A.indices,
             data=A.data, b=b, v_x=res["v_x"], v_y=res["v_y"], remodelling=res["remodelling"],
             **kw_to_np(kw), **scalars(res))

    # ---- G3: 32x48x4 stack, non-zero initial fields, delta_x != 1 != delta_t ----------
    movie = orc.make_texture_stack(48, 4, seed=7)[:, :32, :]
    kw = dict(delta_x=0.25, delta_t=0.5, speed_alpha=1.0, remodelling_alpha=100.0,
              initial_v_x=0.3, initial_v_y=-0.2, initial_remodelling=0.01)
    res, _ = run_ref(OF, movie, **kw)
    np.savez(os.path.join(HERE, "g3_stack_32x48x4.npz"), movie=movie, v_x=res["v_x"], v_y=res["v_y"],
             remodelling=res["remodelling"], speed=res["speed"], **kw_to_np(kw), **scalars(res))

    # ---- G4: 64x64 pair with blur ----------------------------------------------------
    rng = np.random.default_rng(3)
    base = scipy.ndimage.gaussian_filter(rng.random((80, 80)), 1.0, mode="wrap")
    base = (base - base.min()) / (base.max() - base.min())
    movie = np.stack([scipy.ndimage.shift(base, (0.3 * t, 0.6 * t), order=3, mode="wrap")[8:72, 8:72]
                      for t in range(2)])
    kw = dict(delta_x=1.0, delta_t=1.0, speed_alpha=1.0, remodelling_alpha=10000.0, smoothing_sigma=2.0)
    res, _ = run_ref(OF, movie, **kw)
    np.savez(os.path.join(HERE, "g4_blur_64.npz"), movie=movie, v_x=res["v_x"], v_y=res["v_y"],
             remodelling=res["remodelling"], blurred=res["blurred_data"], **kw_to_np(kw), **scalars(res))

    # ---- G5: BASELINE config 1, 128x128x8 translating Gaussian (crops + strided sample) ----
    movie, dx = orc.make_gaussian_stack(128, 8)
    kw = dict(delta_x=float(dx), delta_t=1.0, speed_alpha=1.0, remodelling_alpha=10000.0)
    res, _ = run_ref(OF, movie, **kw)
    c = slice(56, 72)
    np.savez(os.path.join(HERE, "g5_gaussian_128x8.npz"),
             v_x_crop=res["v_x"][:, c, c], v_y_crop=res["v_y"][:, c, c], remodelling_crop=res["remodelling"][:, c, c],
             v_x_sub=res["v_x"][:, ::8, ::8], v_y_sub=res["v_y"][:, ::8, ::8], remodelling_sub=res["remodelling"][:, ::8, ::8],
             v_x_mean=res["v_x"].mean(axis=(1, 2)), v_y_mean=res["v_y"].mean(axis=(1, 2)),
             remodelling_mean=res["remodelling"].mean(axis=(1, 2)),
             v_x_border=res["v_x"][:, :3, :], **kw_to_np(kw), **scalars(res))

    # ---- G6: 8-bit-range texture 64x64, alpha=1e4, beta=1e2 (harder regime) -----------
    movie = np.round(orc.make_texture_stack(64, 2, seed=11) * 255.0)
    kw = dict(delta_x=1.0, delta_t=1.0, speed_alpha=1e4, remodelling_alpha=1e2)
    res, _ = run_ref(OF, movie.astype(np.uint8), **kw)
    np.savez(os.path.join(HERE, "g6_8bit_64.npz"), movie=movie.astype(np.uint8), v_x=res["v_x"], v_y=res["v_y"],
             remodelling=res["remodelling"], **kw_to_np(kw), **scalars(res))

    # ---- G7: texture 64x64x3 (the GPU parity workhorse: C2 recipe, small) --------------
    movie = orc.make_texture_stack(64, 3, seed=0)
    kw = dict(delta_x=1.0, delta_t=1.0, speed_alpha=1.0, remodelling_alpha=10000.0)
    res, _ = run_ref(OF, movie, **kw)
    np.savez(os.path.join(HERE, "g7_texture_64x3.npz"), movie=movie, v_x=res["v_x"], v_y=res["v_y"],
             remodelling=res["remodelling"], **kw_to_np(kw), **scalars(res))

    # ---- G9: vary_regularisation (OF.py:1918-1998), 2x3 grid on a small stack ----------
    movie = orc.make_texture_stack(24, 3, seed=9)
    captured.clear()
    scipy.sparse.linalg.spsolve = _capturing_spsolve
    try:
        with contextlib.redirect_stdout(io.StringIO()):
            vr = OF.vary_regularisation(movie, speed_alpha_values=np.array([1.0, 5.0]),
                                        remodelling_alpha_values=np.array([10.0, 100.0, 1000.0]), filename=None,
                                        use_direct_solver=True, delta_x=0.5, delta_t=1.0)
    finally:
        scipy.sparse.linalg.spsolve = _real_spsolve
    np.savez(os.path.join(HERE, "g9_vary_regularisation.npz"), movie=movie,
             **{k: np.asarray(v) for k, v in vr.items() if k != "converged"})

    # ---- G10: subsample_velocities_for_visualisation (OF.py:1574-1646) on a random result dict ----
    rng = np.random.default_rng(10)
    fr = dict(original_data=rng.random((4, 17, 23)), v_x=rng.standard_normal((3, 17, 23)),
              v_y=rng.standard_normal((3, 17, 23)), delta_x=0.25)
    out = dict(v_x=fr["v_x"], v_y=fr["v_y"], delta_x=np.float64(fr["delta_x"]), n_frames=np.int64(4))
    for box in (1, 2, 3, 4, 5, 7):
        xp, yp, sx, sy = OF.subsample_velocities_for_visualisation(fr, arrow_boxsize=box)
        out.update({f"x_positions_{box}": xp, f"y_positions_{box}": yp, f"v_x_{box}": sx, f"v_y_{box}": sy})
    np.savez(os.path.join(HERE, "g10_subsample.npz"), **out)

    # ---- synthetic generator: make_fake_data_frame itself ------------------------------
    fr, dxx = OF.make_fake_data_frame(1.3, 2.9, sigma=1.7, width=6.0, dimension=37, include_noise=False)
    np.savez(os.path.join(HERE, "g8_fake_frame.npz"), frame=fr, delta_x=dxx)
    print("done")


def kw_to_np(kw):
    return {"kw_" + k: np.float64(v) for k, v in kw.items() if v is not None}


if __name__ == "__main__":
    main()
