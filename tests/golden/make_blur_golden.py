"""Pin the blur fixture on the REAL skimage (run under /opt/conda/bin/python3.9, the one interpreter of the build
container that has skimage 0.18.3; the default python3.10 that produced the other fixtures does not).

The reference's ``blur_movie`` (source/optical_flow.py:282-306) is imported and executed, not copied; it calls
``skimage.filters.gaussian(frame, sigma=..., preserve_range=True)``.  Stand-ins only for imports that carry no arithmetic
of this function: ``numba`` decorators (identity; the conda numba fails to initialise against its numpy), ``cv2`` and
``petsc4py`` (absent, unused here).  Inputs: the G4 movie (float64, what ``variational_optical_flow`` passes after its
cast, OF.py:769) and an 8-bit stack (what a script calling ``blur_movie`` directly would pass).

Usage:  /opt/conda/bin/python3.9 tests/golden/make_blur_golden.py     (writes tests/golden/g4b_blur_skimage.npz)
"""
import io
import os
import sys
import types
import contextlib
from unittest import mock

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
sys.dont_write_bytecode = True


def import_reference():
    def _ident(*a, **k):
        return a[0] if (len(a) == 1 and callable(a[0]) and not k) else (lambda f: f)
    nb = types.ModuleType("numba"); nb.jit = nb.njit = _ident; sys.modules["numba"] = nb
    sys.modules["cv2"] = types.ModuleType("cv2")
    p4 = types.ModuleType("petsc4py"); p4.PETSc = mock.MagicMock(); sys.modules["petsc4py"] = p4
    import matplotlib
    matplotlib.use("Agg")
    import skimage.filters                      # the real one
    sys.path.insert(0, "/root/reference/source")
    import optical_flow
    assert optical_flow.skimage.filters.gaussian is skimage.filters.gaussian
    return optical_flow, skimage.__version__


def main():
    OF, version = import_reference()
    with np.load(os.path.join(HERE, "g4_blur_64.npz"), allow_pickle=False) as z:
        movie, sigma, stand_in = z["movie"], float(z["kw_smoothing_sigma"]), z["blurred"]
    rng = np.random.default_rng(44)
    movie_u8 = rng.integers(0, 256, (3, 40, 52)).astype(np.uint8)
    out = {}
    with contextlib.redirect_stdout(io.StringIO()):
        out["blurred_skimage"] = np.asarray(OF.blur_movie(np.array(movie, copy=True), smoothing_sigma=sigma), dtype=np.float64)
        for s in (1.0, 2.48):
            out["blurred_u8_sigma_%s" % str(s).replace(".", "p")] = np.asarray(
                OF.blur_movie(np.array(movie_u8, copy=True), smoothing_sigma=s), dtype=np.float64)
    diff = float(np.abs(out["blurred_skimage"] - stand_in).max())
    print("skimage", version, "| max |skimage - scipy stand-in of G4| =", diff)
    np.savez(os.path.join(HERE, "g4b_blur_skimage.npz"), movie=movie, sigma=sigma, movie_u8=movie_u8,
             skimage_version=np.array([int(v) for v in version.split(".")[:3]]), **out)


if __name__ == "__main__":
    main()
