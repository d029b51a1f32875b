"""GPU parity tests: the HIP solver, called through the drop-in Python entry point (which goes
through the C ABI), against (1) the golden fixtures produced by the reference itself and (2) the
CPU oracle on fresh seeded inputs.

Tolerances; float64 end to end:
  * tight solve (rtol 1e-10 / use_direct_solver=True): rel-L2 error <= 1e-7 per field versus the exact
    solution of the reference-assembled system;
  * the reference's own setting rtol = 1e-6: stopping rule met and rel-L2 error <= 1e-3 (the band any
    converged PETSc bcgs run lies in).
"""
import os

import numpy as np
import pytest

from conftest import load_golden, golden_kwargs
from oracle import vof_oracle as orc

pytestmark = pytest.mark.gpu

TIGHT = 1e-7
LOOSE = 1e-3


def relerr(a, b):
    return float(np.linalg.norm(np.asarray(a) - np.asarray(b)) / max(np.linalg.norm(b), 1e-300))


@pytest.fixture(scope="module")
def of():
    from opticalflow_amd import optical_flow
    return optical_flow


def check_fields(res, ref, tol, keys=("v_x", "v_y", "remodelling")):
    for k in keys:
        e = relerr(res[k], ref[k])
        assert e < tol, (k, e)


GOLDEN = ["g1_avof_simple_50.npz", "g2_matrix_6x7.npz", "g2b_matrix_9x11.npz", "g3_stack_32x48x4.npz",
          "g7_texture_64x3.npz"]


@pytest.mark.parametrize("name", GOLDEN)
def test_golden_tight(of, name):
    g = load_golden(name)
    kw = golden_kwargs(g)
    res = of.variational_optical_flow(g["movie"], rtol=1e-10, return_stats=True, **kw)
    check_fields(res, g, TIGHT)
    assert res["stats"]["converged"].all()
    assert res["stats"]["relative_residual"].max() < 1e-9
    for key in ("L1_functional", "remodelling_functional", "speed_functional"):
        assert res[key] == pytest.approx(float(g[key]), rel=1e-6, abs=1e-12), key
    assert res["speed_functional"] == res["remodelling_functional"]      # OF.py:1205
    if "speed" in g:
        assert relerr(res["speed"], g["speed"]) < TIGHT
    assert res["converged"] is True


@pytest.mark.parametrize("name", GOLDEN)
def test_golden_reference_tolerance(of, name):
    """The reference's own stopping rule rtol=1e-6 (OF.py:1120)."""
    g = load_golden(name)
    kw = golden_kwargs(g)
    res = of.variational_optical_flow(g["movie"], return_stats=True, **kw)
    st = res["stats"]
    assert st["converged"].all()
    assert st["relative_residual"].max() <= 1.5e-6
    assert st["iterations"].max() <= 60
    check_fields(res, g, LOOSE)


def test_g1_printed_means(of):
    """What the reference's enabled experiment prints (AVOF.py:58-66): expect ~0.1, 0.2, 0.05."""
    g = load_golden("g1_avof_simple_50.npz")
    res = of.variational_optical_flow(g["movie"], use_direct_solver=True, **golden_kwargs(g))
    assert np.mean(res["v_x"]) == pytest.approx(1.049780578641995e-01, rel=1e-7)
    assert np.mean(res["v_y"]) == pytest.approx(1.950243954755939e-01, rel=1e-7)
    assert np.mean(res["remodelling"]) == pytest.approx(4.862204763997428e-02, rel=1e-7)
    assert np.max(res["v_x"]) == pytest.approx(float(g["v_x"].max()), rel=1e-7)


def test_device_blur_matches_reference(of):
    """blur_movie on the GPU (OF.py:282-306) against the reference's blurred stack and fresh scipy results."""
    import scipy.ndimage
    g = load_golden("g4_blur_64.npz")
    np.testing.assert_allclose(of.blur_movie(g["movie"], 2.0), g["blurred"], rtol=0, atol=4.5e-16)
    rng = np.random.default_rng(0)
    for shape, sigma in (((3, 37, 53), 1.3), ((2, 5, 200), 2.48), ((18, 64, 64), 0.6), ((1, 9, 9), 5.0)):
        mv = (rng.random(shape) * 255).astype(np.uint8 if shape[0] == 3 else np.float64)
        ref = np.stack([scipy.ndimage.gaussian_filter(f.astype(np.float64), sigma, mode="nearest", truncate=4.0) for f in mv])
        got = of.blur_movie(mv, sigma)
        assert got.dtype == np.float64 and got.shape == mv.shape
        np.testing.assert_allclose(got, ref, rtol=1e-15, atol=1e-13)


def test_device_blur_against_the_real_skimage(of):
    """The HIP blur against the reference's blur_movie run with the real skimage 0.18.3 (fixture G4b): float64 frames
    and an 8-bit stack, to 2 ulp."""
    gb = load_golden("g4b_blur_skimage.npz")
    np.testing.assert_allclose(of.blur_movie(gb["movie"], float(gb["sigma"])), gb["blurred_skimage"], rtol=0, atol=4.5e-16)
    for key, sigma in (("blurred_u8_sigma_1p0", 1.0), ("blurred_u8_sigma_2p48", 2.48)):
        got = of.blur_movie(gb["movie_u8"], sigma)
        assert got.dtype == np.float64 and got.shape == gb["movie_u8"].shape
        np.testing.assert_allclose(got, gb[key], rtol=9e-16, atol=0)


def test_blur_path(of):
    g = load_golden("g4_blur_64.npz")
    res = of.variational_optical_flow(g["movie"], rtol=1e-10, **golden_kwargs(g))
    np.testing.assert_allclose(res["blurred_data"], g["blurred"], atol=1e-15)
    check_fields(res, g, TIGHT)
    assert res["original_data"].shape == g["movie"].shape


def test_original_data_is_a_copy_made_while_the_solve_runs(of):
    """OF.py:769: the reference works on movie.astype(float64) and returns that COPY as 'original_data' (and as 'blurred_data'
    when nothing is blurred).  A float64 stack large enough for the threaded copy is read in place by the solver while the copy
    is made in the background: the result must still hold an equal array that does not alias the caller's."""
    movie = orc.make_texture_stack(640, 42, seed=3)                # 17.2 M values: above the threaded-copy threshold
    keep = movie.copy()
    res = of.variational_optical_flow(movie, speed_alpha=1.0, remodelling_alpha=1e4, return_stats=True)
    assert res["stats"]["converged"].all()
    assert np.array_equal(movie, keep)                              # the input is only read
    assert res["original_data"] is not movie and not np.shares_memory(res["original_data"], movie)
    assert np.array_equal(res["original_data"], movie) and res["blurred_data"] is res["original_data"]
    res["original_data"][0, 0, 0] += 1.0
    assert movie[0, 0, 0] == keep[0, 0, 0]


def test_8bit_regime_uint8_input(of):
    """uint8 stack, alpha=1e4, beta=1e2 (the harder regime T of SURVEY Appendix B)."""
    g = load_golden("g6_8bit_64.npz")
    assert g["movie"].dtype == np.uint8
    res = of.variational_optical_flow(g["movie"], rtol=1e-10, return_stats=True, **golden_kwargs(g))
    assert res["stats"]["converged"].all()
    check_fields(res, g, 1e-6)
    assert res["original_data"].dtype == np.float64


def test_config1_gaussian_128x8(of):
    """BASELINE config 1 (128x128x8 translating Gaussian) against the reference-generated crops."""
    g = load_golden("g5_gaussian_128x8.npz")
    movie, dx = orc.make_gaussian_stack(128, 8)
    res = of.variational_optical_flow(movie, delta_x=dx, speed_alpha=1.0, remodelling_alpha=10000.0, rtol=1e-10,
                                      return_stats=True)
    c = slice(56, 72)
    assert relerr(res["v_x"][:, c, c], g["v_x_crop"]) < TIGHT
    assert relerr(res["v_y"][:, c, c], g["v_y_crop"]) < TIGHT
    assert relerr(res["remodelling"][:, c, c], g["remodelling_crop"]) < TIGHT
    assert relerr(res["v_y"][:, ::8, ::8], g["v_y_sub"]) < TIGHT
    assert relerr(res["v_x"][:, :3, :], g["v_x_border"]) < TIGHT          # mirror-fixed border rows
    np.testing.assert_allclose(res["remodelling"].mean(axis=(1, 2)), g["remodelling_mean"], rtol=1e-7)
    assert res["L1_functional"] == pytest.approx(float(g["L1_functional"]), rel=1e-6)
    assert res["stats"]["iterations"].max() <= 12


@pytest.mark.parametrize("shape,alpha,beta,seed", [((40, 57), 1.0, 1e4, 5), ((33, 33), 3.0, 10.0, 6),
                                                    ((4, 4), 1.0, 1.0, 7), ((5, 64), 1.0, 100.0, 8),
                                                    ((130, 130), 1.0, 1e4, 9)])
@pytest.mark.parametrize("quirks", [True, False])
def test_fresh_inputs_against_oracle(of, shape, alpha, beta, seed, quirks):
    """Seeded inputs the fixtures do not cover: ragged / minimal / odd sizes, quirk opt-out."""
    n = max(shape)
    if n >= 16:
        movie = orc.make_texture_stack(n, 3, seed=seed)[:, :shape[0], :shape[1]]
    else:
        movie = np.random.default_rng(seed).random((3,) + shape)
    ref = orc.variational_optical_flow(movie, speed_alpha=alpha, remodelling_alpha=beta, delta_x=0.5, delta_t=2.0,
                                       reference_quirks=quirks)
    res = of.variational_optical_flow(movie, speed_alpha=alpha, remodelling_alpha=beta, delta_x=0.5, delta_t=2.0,
                                      reference_quirks=quirks, rtol=1e-10, return_stats=True)
    assert res["stats"]["converged"].all()
    check_fields(res, ref, TIGHT, keys=("v_x", "v_y", "remodelling", "speed"))
    for key in ("L1_functional", "remodelling_functional", "speed_functional"):
        assert res[key] == pytest.approx(ref[key], rel=1e-6, abs=1e-12)
    if not quirks and min(shape) > 4:   # (a 4x4 image is all mirror: every derivative vanishes)
        assert res["speed_functional"] != res["remodelling_functional"]


def test_batching_is_invisible(of):
    """Chunked batches (max_pairs_in_flight < pairs) give the same result as one batch."""
    movie = orc.make_texture_stack(48, 6, seed=21)
    a = of.variational_optical_flow(movie, remodelling_alpha=1e4, rtol=1e-10, max_pairs_in_flight=5)
    b = of.variational_optical_flow(movie, remodelling_alpha=1e4, rtol=1e-10, max_pairs_in_flight=2)
    for k in ("v_x", "v_y", "remodelling", "speed"):
        assert relerr(b[k], a[k]) < 1e-8
    # and runs are bit-reproducible (deterministic reductions)
    c = of.variational_optical_flow(movie, remodelling_alpha=1e4, rtol=1e-10, max_pairs_in_flight=5)
    for k in ("v_x", "v_y", "remodelling"):
        np.testing.assert_array_equal(a[k], c[k])


def test_initial_guess_does_not_change_the_answer(of):
    movie = orc.make_texture_stack(40, 2, seed=3)
    a = of.variational_optical_flow(movie, remodelling_alpha=1e3, rtol=1e-11)
    b = of.variational_optical_flow(movie, remodelling_alpha=1e3, rtol=1e-11, initial_v_x=0.4, initial_v_y=-0.3,
                                    initial_remodelling=0.02, delta_x=1.0)
    for k in ("v_x", "v_y", "remodelling"):
        assert relerr(b[k], a[k]) < 1e-8


def test_identical_frames_give_zero_flow(of):
    fr = orc.make_texture_stack(32, 1, seed=1)[0]
    res = of.variational_optical_flow(np.stack([fr, fr]), return_stats=True)
    assert abs(res["v_x"]).max() == 0 and abs(res["remodelling"]).max() == 0
    assert res["stats"]["iterations"][0] == 0 and res["converged"]


def test_nonconvergence_is_reported_not_raised(of):
    """max_iterations=1 cannot reach 1e-12: the reference only prints a warning (OF.py:1135-1138)."""
    movie = orc.make_texture_stack(64, 2, seed=2)
    res = of.variational_optical_flow(movie, remodelling_alpha=1e4, rtol=1e-13, max_iterations=1, return_stats=True,
                                      preconditioner="multigrid")
    assert res["converged"] is False
    assert res["stats"]["iterations"][0] == 1
    assert np.isfinite(res["v_x"]).all()


def test_coarse_stencil_precision_does_not_change_the_answer(of):
    """bfloat16 + row-sum-keeping float32 diagonal (default) vs float32 vs float64 storage of the Galerkin stencils:
    preconditioner only - same answer, and (the point of keeping the row sums) the same iteration counts."""
    movie = orc.make_texture_stack(96, 3, seed=4)
    a = of.variational_optical_flow(movie, remodelling_alpha=1e4, rtol=1e-10, coarse_precision="float64",
                                    return_stats=True)
    for fmt in ("float32", "bfloat16", "float8"):
        b = of.variational_optical_flow(movie, remodelling_alpha=1e4, rtol=1e-10, coarse_precision=fmt, return_stats=True)
        assert b["stats"]["converged"].all()
        for k in ("v_x", "v_y", "remodelling"):
            assert relerr(b[k], a[k]) < 1e-8, (fmt, k)
        assert np.abs(b["stats"]["iterations"].astype(int) - a["stats"]["iterations"]).max() <= (2 if fmt == "float8" else 1), fmt


def test_medium_size_properties_512(of):
    """512x512 (direct oracle impractical): independent residual, mirror structure, known flow."""
    movie = orc.make_texture_stack(512, 3, seed=0)
    res = of.variational_optical_flow(movie, remodelling_alpha=1e4, return_stats=True)
    st = res["stats"]
    assert st["converged"].all() and st["relative_residual"].max() <= 1.5e-6
    assert st["iterations"].max() <= 12
    for k in ("v_x", "v_y", "remodelling"):
        f = res[k]
        np.testing.assert_array_equal(f[:, 0, :], f[:, 2, :])
        np.testing.assert_array_equal(f[:, -1, :], f[:, -3, :])
        np.testing.assert_array_equal(f[:, :, 0], f[:, :, 2])
        np.testing.assert_array_equal(f[:, :, -1], f[:, :, -3])
    # exactly translating texture: true flow (0.3, 0.6) px/frame, gamma ~ 0
    assert np.mean(res["v_x"]) == pytest.approx(0.3, abs=0.03)
    assert np.mean(res["v_y"]) == pytest.approx(0.6, abs=0.03)
    # independent residual through the CPU matrix-free operator on pair 0
    xi = np.stack([res["v_x"][0], res["v_y"][0], res["remodelling"][0]])[:, 1:-1, 1:-1]
    r = orc.rhs_interior(movie[0], movie[1]) - orc.apply_operator_interior(movie[0], xi, 1.0, 1e4)
    assert np.linalg.norm(r) / np.linalg.norm(orc.rhs_interior(movie[0], movie[1])) <= 1.5e-6


@pytest.mark.parametrize("name", ["g1_avof_simple_50.npz", "g3_stack_32x48x4.npz", "g7_texture_64x3.npz", "g6_8bit_64.npz"])
def test_float32_vcycle_storage_same_answer(of, name):
    """Mixed precision: float32 storage inside the preconditioner, FP64 Krylov iteration and stopping rule.
    The converged answer must still match the reference's exact solution to the tight tolerance."""
    g = load_golden(name)
    kw = golden_kwargs(g)
    res = of.variational_optical_flow(g["movie"], rtol=1e-10, vcycle_precision="float32", return_stats=True, **kw)
    assert res["stats"]["converged"].all()
    assert res["stats"]["relative_residual"].max() < 1e-9
    check_fields(res, g, 1e-6 if "8bit" in name else TIGHT)


def test_vary_regularisation_against_reference_fixture(of, tmp_path):
    """The batch caller of the hot path (OF.py:1918-1998): same dictionary, same optional np.save."""
    g = load_golden("g9_vary_regularisation.npz")
    fn = str(tmp_path / "sweep.npy")
    r = of.vary_regularisation(g["movie"], speed_alpha_values=g["speed_alpha_values"],
                               remodelling_alpha_values=g["remodelling_alpha_values"], filename=fn,
                               delta_x=0.5, delta_t=1.0, rtol=1e-10)
    for k in ("speed_means", "speed_variances", "remodelling_means", "remodelling_variances", "functional"):
        np.testing.assert_allclose(r[k], g[k], rtol=2e-6, atol=1e-12, err_msg=k)
    assert r["converged"].all() and r["converged"].dtype == bool
    saved = np.load(fn, allow_pickle=True).item()          # our own file (the reference's scripts read it this way)
    np.testing.assert_array_equal(saved["speed_means"], r["speed_means"])


def test_full_size_properties_1024(of):
    """BASELINE full frame size (1024x1024; the direct oracle is impractical there): size-independent properties -
    stopping rule, independent CPU residual of one pair, mirror structure, known synthetic flow, determinism."""
    movie = orc.make_texture_stack(1024, 3, seed=1)
    res = of.variational_optical_flow(movie, remodelling_alpha=1e4, return_stats=True)
    st = res["stats"]
    assert st["converged"].all() and st["relative_residual"].max() <= 1.5e-6 and st["iterations"].max() <= 12
    xi = np.stack([res["v_x"][1], res["v_y"][1], res["remodelling"][1]])[:, 1:-1, 1:-1]
    b = orc.rhs_interior(movie[1], movie[2])
    r = b - orc.apply_operator_interior(movie[1], xi, 1.0, 1e4)
    assert np.linalg.norm(r) / np.linalg.norm(b) <= 1.5e-6
    assert np.linalg.norm(r) / np.linalg.norm(b) == pytest.approx(st["relative_residual"][1], rel=1e-6)
    for k in ("v_x", "v_y", "remodelling", "speed"):
        f = res[k]
        np.testing.assert_array_equal(f[:, 0, :], f[:, 2, :])
        np.testing.assert_array_equal(f[:, :, -1], f[:, :, -3])
    assert np.mean(res["v_x"]) == pytest.approx(0.3, abs=0.02) and np.mean(res["v_y"]) == pytest.approx(0.6, abs=0.02)
    assert abs(np.mean(res["remodelling"])) < 5e-3
    again = of.variational_optical_flow(movie, remodelling_alpha=1e4)
    np.testing.assert_array_equal(again["v_x"], res["v_x"])          # bit-reproducible


@pytest.mark.parametrize("name", ["g1_avof_simple_50.npz", "g3_stack_32x48x4.npz", "g6_8bit_64.npz", "g7_texture_64x3.npz"])
def test_gmres_reaches_the_reference_solution(of, name):
    """krylov_method='gmres' (the fallback method, used alone): same preconditioner and stopping rule, so the converged
    answer is the exact solution of the reference-assembled system to the tight tolerance."""
    g = load_golden(name)
    kw = golden_kwargs(g)
    res = of.variational_optical_flow(g["movie"], rtol=1e-10, krylov_method="gmres", return_stats=True, **kw)
    assert res["stats"]["converged"].all() and res["stats"]["relative_residual"].max() < 1e-9
    check_fields(res, g, 1e-6 if "8bit" in name else TIGHT)
    short = of.variational_optical_flow(g["movie"], rtol=1e-10, krylov_method="gmres", gmres_restart=5, return_stats=True, **kw)
    assert short["stats"]["converged"].all()                      # restarts: x_0 + M V y accumulated over several cycles
    check_fields(short, g, 1e-6 if "8bit" in name else TIGHT)


def test_gmres_fallback_finishes_what_bicgstab_starts(of):
    """8-bit data without blur, alpha / I^2 ~ 0.15 (the grad-div dominated regime): BiCGStab needs
    > 100 iterations at 258^2; the default hands the pairs to GMRES after 25 iterations, which then needs far
    fewer multigrid cycles.  Both satisfy the reference's stopping rule (checked on the CPU) and agree to
    the accuracy that rule implies."""
    movie = np.round(orc.make_texture_stack(258, 3, seed=1) * 255.0)
    kw = dict(speed_alpha=1e4, remodelling_alpha=1e2, return_stats=True)
    auto = of.variational_optical_flow(movie, **kw)
    bicg = of.variational_optical_flow(movie, krylov_method="bicgstab", **kw)
    for r in (auto, bicg):
        assert r["stats"]["converged"].all() and r["stats"]["relative_residual"].max() <= 1.5e-6
    cycles_auto = 2 * 25 + (auto["stats"]["iterations"] - 25)    # BiCGStab: two cycles per iteration, GMRES: one
    assert (cycles_auto < 2 * bicg["stats"]["iterations"]).all()
    assert auto["stats"]["iterations"].max() <= 140
    xi = np.stack([auto["v_x"][0], auto["v_y"][0], auto["remodelling"][0]])[:, 1:-1, 1:-1]
    b = orc.rhs_interior(movie[0], movie[1])
    rr = np.linalg.norm(b - orc.apply_operator_interior(movie[0], xi, 1e4, 1e2)) / np.linalg.norm(b)
    assert rr <= 1.5e-6
    check_fields(auto, bicg, LOOSE)


def test_easy_regimes_never_touch_the_fallback(of):
    """The benchmark regime converges in a handful of BiCGStab iterations: 'auto' and 'bicgstab' are bit-identical."""
    movie = orc.make_texture_stack(96, 4, seed=2)
    a = of.variational_optical_flow(movie, remodelling_alpha=1e4, return_stats=True)
    b = of.variational_optical_flow(movie, remodelling_alpha=1e4, krylov_method="bicgstab", return_stats=True)
    assert a["stats"]["iterations"].max() <= 8           # far below the switch (25)
    np.testing.assert_array_equal(a["v_x"], b["v_x"])
    np.testing.assert_array_equal(a["stats"]["iterations"], b["stats"]["iterations"])


@pytest.mark.parametrize("shape", [(4, 1500), (1500, 4), (121, 123), (122, 129), (137, 241), (257, 120), (6, 259)])
def test_strip_and_band_boundary_sizes(of, shape):
    """Image sizes around the strip widths (120 / 128 columns) and band heights of the streaming kernels, and extremely
    elongated images: the independent CPU residual of the GPU solution (oracle operator, OF.py:1150-1151) must meet the
    tolerance, i.e. every kernel of the pipeline handled the ragged edges."""
    rng = np.random.default_rng(shape[0] * 7 + shape[1])
    n = max(shape)
    base = orc.make_texture_stack(max(n, 16), 2, seed=shape[1])[:, :shape[0], :shape[1]]
    movie = base + 0.01 * rng.random((2,) + shape)
    res = of.variational_optical_flow(movie, speed_alpha=1.0, remodelling_alpha=200.0, rtol=1e-9, return_stats=True)
    assert res["stats"]["converged"].all()
    xi = np.stack([res["v_x"][0], res["v_y"][0], res["remodelling"][0]])[:, 1:-1, 1:-1]
    b = orc.rhs_interior(movie[0], movie[1])
    rr = np.linalg.norm(b - orc.apply_operator_interior(movie[0], xi, 1.0, 200.0)) / np.linalg.norm(b)
    assert rr <= 2e-9
    for k in ("v_x", "v_y", "remodelling", "speed"):        # mirror fix-up of the border (OF.py:1159-1166) is idempotent
        f = res[k][0].copy()
        of.apply_constant_boundary_condition(f)
        np.testing.assert_array_equal(f, res[k][0])


def test_input_forms_the_reference_accepts(of):
    """OF.py:769 casts whatever it gets with astype(float): lists, Fortran order, strided views, small integer and
    float32 dtypes all give the result of the equivalent float64 C-ordered array."""
    base = np.round(orc.make_texture_stack(40, 3, seed=6) * 200.0)
    ref = of.variational_optical_flow(base, speed_alpha=5e4, remodelling_alpha=1e3)
    big = np.zeros((3, 80, 80)); big[:, ::2, ::2] = base
    forms = {"list": base.tolist(), "fortran": np.asfortranarray(base), "strided": big[:, ::2, ::2],
             "int16": base.astype(np.int16), "uint8": base.astype(np.uint8), "float32": base.astype(np.float32)}
    for name, mv in forms.items():
        res = of.variational_optical_flow(mv, speed_alpha=5e4, remodelling_alpha=1e3)
        np.testing.assert_array_equal(res["v_x"], ref["v_x"], err_msg=name)
        assert res["original_data"].dtype == np.float64 and res["v_x"].flags["C_CONTIGUOUS"]
    with pytest.raises(ValueError):
        of.variational_optical_flow(base[0])
    with pytest.raises(ValueError):
        of.variational_optical_flow(base[:1])


def test_non_finite_input_is_reported_not_hung(of):
    """A NaN pixel poisons its pair only: that pair is reported as not converged, the others are solved."""
    movie = orc.make_texture_stack(48, 4, seed=12)
    movie[1, 20, 20] = np.nan                      # frame 1 belongs to pairs 0 and 1
    res = of.variational_optical_flow(movie, remodelling_alpha=1e4, max_iterations=50, return_stats=True)
    st = res["stats"]
    assert st["converged"].tolist() == [0, 0, 1]
    assert np.isfinite(res["v_x"][2]).all() and st["relative_residual"][2] <= 1.5e-6
    assert res["converged"] is True                # flag of the last pair, as in the reference


@pytest.mark.parametrize("case", range(int(os.environ.get("VOF_FUZZ_CASES", "24"))))   # more cases: VOF_FUZZ_CASES=400
def test_seeded_random_configurations_against_oracle(of, case):
    """Seeded sweep over shapes, parameter magnitudes, units, initial fields, quirk switch and solver options (Krylov
    method, storage precisions, cycle shape): the converged answer is the exact solution of the reference's system
    (oracle direct solve), whatever the options - they only change the path to it."""
    rng = np.random.default_rng(1000 + case)
    n_i, n_j = int(rng.integers(4, 72)), int(rng.integers(4, 72))
    T = int(rng.integers(2, 5))
    scale = [1.0, 1.0, 255.0][case % 3]
    movie = orc.make_texture_stack(max(n_i, n_j, 16), T, seed=case)[:, :n_i, :n_j] * scale
    if case % 4 == 0:
        movie = movie + 0.02 * scale * rng.random(movie.shape)
    alpha = float(10 ** rng.uniform(-0.5, 2.0)) * scale ** 2       # alpha / I^2 in the convergent envelope
    beta = float(10 ** rng.uniform(0.0, 4.0))
    kw = dict(speed_alpha=alpha, remodelling_alpha=beta, delta_x=float(rng.uniform(0.2, 2.0)), delta_t=float(rng.uniform(0.5, 2.0)),
              initial_v_x=float(rng.uniform(-0.5, 0.5)), initial_v_y=float(rng.uniform(-0.5, 0.5)),
              initial_remodelling=float(rng.uniform(-0.1, 0.1)), reference_quirks=bool(case % 5 != 0))
    opts = dict(krylov_method=["auto", "bicgstab", "gmres"][case % 3], coarse_precision=["bfloat16", "float32", "float64", "float8"][case % 4],
                vcycle_precision=["float64", "float32", "auto", "coarse_float32"][(case // 2) % 4], w_cycle_level=[None, -1, 0, (1, 2)][case % 4],
                multigrid_sweeps=[None, (1, 1), (2, 1, 2, 2), (3, 3)][(case // 3) % 4], max_pairs_in_flight=[None, 1, 2][case % 3])
    ref = orc.variational_optical_flow(movie, **kw)
    res = of.variational_optical_flow(movie, rtol=1e-10, return_stats=True, **kw, **opts)
    if opts["krylov_method"] == "bicgstab":
        # BiCGStab alone (the reference's KSP type) can stall a decade above rtol = 1e-10 (its recursively updated residual
        # drifts from the true one); it is reported, and 'auto' / 'gmres' finish such pairs
        assert res["stats"]["relative_residual"].max() < 1e-8, (res["stats"], opts)
    else:
        # (a pair may sit on its attainable accuracy ~ eps ||A|| ||x|| / ||b|| above 1e-10: tiny grids with a huge alpha and
        # a non-zero initial field; it is reported after one stagnating GMRES cycle, not iterated to max_iterations)
        st = res["stats"]
        assert ((st["converged"] == 1) | (st["relative_residual"] < 1e-9)).all(), (st, opts)
        assert st["relative_residual"][st["converged"] == 1].max() <= 1.6e-10, (st, opts)
        assert st["iterations"].max() <= 400, (st, opts)
    check_fields(res, ref, 1e-6, keys=("v_x", "v_y", "remodelling", "speed"))
    for key in ("L1_functional", "remodelling_functional", "speed_functional"):
        assert res[key] == pytest.approx(ref[key], rel=1e-5, abs=1e-10)


@pytest.mark.parametrize("case", range(int(os.environ.get("VOF_FUZZ_CASES_MEDIUM", "10"))))
def test_seeded_random_medium_sizes_by_independent_residual(of, case):
    """As above at sizes the direct oracle is too slow for (up to ~400 pixels a side, several strips / bands per image):
    the CPU-evaluated residual of the reference's operator (OF.py:1150-1151) must meet the requested tolerance for
    every pair, whatever the solver options."""
    rng = np.random.default_rng(5000 + case)
    n_i, n_j = int(rng.integers(60, 400)), int(rng.integers(60, 400))
    T = int(rng.integers(2, 5))
    movie = orc.make_texture_stack(max(n_i, n_j), T, seed=100 + case)[:, :n_i, :n_j]
    alpha, beta = float(10 ** rng.uniform(-0.3, 1.5)), float(10 ** rng.uniform(1.0, 4.0))
    quirks = bool(case % 4 != 0)
    opts = dict(krylov_method=["auto", "gmres"][case % 2], coarse_precision=["bfloat16", "float32", "float64", "float8"][(case // 2) % 4],
                vcycle_precision=["float64", "float32", "auto", "coarse_float32"][case % 4], w_cycle_level=[None, -1, (1, 2), 2][case % 4],
                max_pairs_in_flight=[None, 2][case % 2])
    res = of.variational_optical_flow(movie, speed_alpha=alpha, remodelling_alpha=beta, reference_quirks=quirks, rtol=1e-8,
                                      return_stats=True, **opts)
    assert res["stats"]["converged"].all(), (res["stats"], opts)
    for k in range(T - 1):
        xi = np.stack([res["v_x"][k], res["v_y"][k], res["remodelling"][k]])[:, 1:-1, 1:-1]
        b = orc.rhs_interior(movie[k], movie[k + 1], quirks)
        rr = np.linalg.norm(b - orc.apply_operator_interior(movie[k], xi, alpha, beta, quirks)) / np.linalg.norm(b)
        assert rr <= 1.6e-8, (k, rr, opts)
        assert rr == pytest.approx(res["stats"]["relative_residual"][k], rel=1e-3)


def test_reference_side_binding_runs_on_g1():
    """examples/reference_binding.py (the stub INTEGRATION.md prints) executed as is: the reference's own experiment
    AVOF.py:26-50 through vof_solve_stack_host, against the reference-produced fixture G1 at the reference's rtol."""
    import importlib.util
    from conftest import ROOT
    spec = importlib.util.spec_from_file_location("reference_binding", os.path.join(ROOT, "examples", "reference_binding.py"))
    rb = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(rb)
    g = load_golden("g1_avof_simple_50.npz")
    kw = golden_kwargs(g)
    out, stats = rb.solve_stack(g["movie"], kw["delta_x"], kw.get("delta_t", 1.0), kw["speed_alpha"], kw["remodelling_alpha"],
                                kw.get("initial_v_x", 0.0), kw.get("initial_v_y", 0.0), kw.get("initial_remodelling", 0.0))
    assert all(s.converged for s in stats) and max(s.relative_residual for s in stats) <= 1e-6
    for arr, key in zip(out[:3], ("v_x", "v_y", "remodelling")):
        assert relerr(arr, g[key]) < LOOSE, key
    np.testing.assert_allclose(out[3], np.sqrt(out[0] ** 2 + out[1] ** 2), rtol=1e-14)


# ---------------------------------------------------------------------------------------------------------
# Direct preconditioner (block-tridiagonal LU by image rows): use_direct_solver=True and the robust fallback
# ---------------------------------------------------------------------------------------------------------
@pytest.mark.parametrize("name", GOLDEN + ["g6_8bit_64.npz"])
def test_direct_preconditioner_against_reference_fixtures(of, name):
    """preconditioner="direct": the reference's own direct-solver outputs are reproduced in one or two Krylov steps."""
    g = load_golden(name)
    res = of.variational_optical_flow(g["movie"], rtol=1e-10, preconditioner="direct", return_stats=True, **golden_kwargs(g))
    st = res["stats"]
    assert st["converged"].all() and st["iterations"].max() <= 3, st
    check_fields(res, g, 2e-6 if "8bit" in name else TIGHT)


@pytest.mark.parametrize("n,alpha,beta,scale", [(66, 1e4, 1e2, 255.0), (130, 1e4, 1e2, 255.0), (98, 2e3, 1.0, 255.0),
                                                (130, 0.1, 1e2, 1.0), (66, 0.3, 0.1, 255.0)])
def test_hard_regimes_are_rescued_by_the_direct_preconditioner(of, n, alpha, beta, scale):
    """The grad-div dominated regimes (8-bit data with speed_alpha <~ 1e5), where the multigrid cycle
    needs 40-200+ iterations or stagnates: with the default preconditioner="auto" every pair ends converged (re-solved with
    the direct preconditioner where necessary), and the answer is the reference system's exact solution (oracle, SuperLU)."""
    movie = orc.make_texture_stack(n, 3, seed=5) * scale
    res = of.variational_optical_flow(movie, speed_alpha=alpha, remodelling_alpha=beta, rtol=1e-9, max_iterations=60,
                                      return_stats=True)
    st = res["stats"]
    assert st["converged"].all() and st["relative_residual"].max() <= 1e-9, st
    ref = orc.variational_optical_flow(movie, speed_alpha=alpha, remodelling_alpha=beta)
    check_fields(res, ref, 1e-5)
    # and the reference's own switch
    res2 = of.variational_optical_flow(movie, speed_alpha=alpha, remodelling_alpha=beta, use_direct_solver=True, return_stats=True)
    assert res2["stats"]["converged"].all() and res2["stats"]["iterations"].max() <= 4
    check_fields(res2, ref, 1e-6)


@pytest.mark.parametrize("name,alpha,beta", [("T", 1e4, 1e2), ("W", 2e3, 1.0)])
def test_real_data_regimes_at_258_with_default_arguments(of, name, alpha, beta):
    """The reference's own real-data regimes (T: analyse_variational_optical_flow.py:201-233, 8-bit data with
    speed_alpha 1e4; W: analyse_short_timeinterval_data.py:835, speed_alpha 2e3) at 258 x 258 with DEFAULT arguments only:
    the multigrid path does not converge there, the automatic re-solve with the direct preconditioner (m = 768 unknowns per
    image row: the in-house blocked inverse on the FP64 matrix cores, no rocSOLVER) must leave every pair converged by the
    rule on the independent residual, and the answer is the exact solution of the reference system (oracle, SuperLU)."""
    movie = np.round(orc.make_texture_stack(258, 2, seed=5) * 255.0)
    res = of.variational_optical_flow(movie, speed_alpha=alpha, remodelling_alpha=beta, return_stats=True)
    st = res["stats"]
    assert st["converged"].all() and st["relative_residual"].max() <= 1e-6, st
    assert res["converged"] is True
    ref = orc.variational_optical_flow(movie, speed_alpha=alpha, remodelling_alpha=beta)
    check_fields(res, ref, 1e-4)     # what the reference's own stopping rule (rtol 1e-6) implies in these regimes
    # the reference's switch: exact to the tight tolerance in a few Krylov steps
    res2 = of.variational_optical_flow(movie, speed_alpha=alpha, remodelling_alpha=beta, use_direct_solver=True, return_stats=True)
    assert res2["stats"]["converged"].all() and res2["stats"]["iterations"].max() <= 4, res2["stats"]
    check_fields(res2, ref, 1e-6)


def test_real_data_regime_at_514_by_the_cpu_residual(of):
    """Regime T at 514 x 514 (1536 unknowns per image row, 9.7 GB of inverse Schur blocks for the pair), default arguments:
    converged by the library's rule, and the same residual evaluated on the CPU from the returned fields (the oracle's
    direct solve would take minutes at this size)."""
    movie = np.round(orc.make_texture_stack(514, 2, seed=5) * 255.0)
    res = of.variational_optical_flow(movie, speed_alpha=1e4, remodelling_alpha=1e2, return_stats=True)
    st = res["stats"]
    assert st["converged"].all(), st
    xi = np.stack([res[f][0] for f in ("v_x", "v_y", "remodelling")])[:, 1:-1, 1:-1]
    b = orc.rhs_interior(movie[0], movie[1])
    r = b - orc.apply_operator_interior(movie[0], xi, 1e4, 1e2)
    rel = np.linalg.norm(r) / np.linalg.norm(b)
    assert rel <= 1e-6 * (1 + 1e-6), rel
    assert rel == pytest.approx(st["relative_residual"][0], rel=1e-4)


def test_real_data_regime_at_1026_by_the_cpu_residual(of):
    """Regime T at 1026 x 1026 - the largest frame size of the reference's scripts (3072 unknowns per image row: the blocked
    inverse on the FP64 matrix cores, 79 GB of inverse Schur blocks for the pair) -, default arguments: the multigrid attempt
    is handed over after 150 Krylov steps, the direct preconditioner settles the pair in one step (about 18 s)."""
    movie = np.round(orc.make_texture_stack(1026, 2, seed=5) * 255.0)
    res = of.variational_optical_flow(movie, speed_alpha=1e4, remodelling_alpha=1e2, return_stats=True)
    st = res["stats"]
    assert st["converged"].all(), st
    xi = np.stack([res[f][0] for f in ("v_x", "v_y", "remodelling")])[:, 1:-1, 1:-1]
    b = orc.rhs_interior(movie[0], movie[1])
    r = b - orc.apply_operator_interior(movie[0], xi, 1e4, 1e2)
    rel = np.linalg.norm(r) / np.linalg.norm(b)
    assert rel <= 1e-6 * (1 + 1e-6), rel
    assert rel == pytest.approx(st["relative_residual"][0], rel=1e-4)
    of.release_device_memory()


def test_parameter_sweep_of_the_reference_script_on_8bit_data(of):
    """AVOF.py:608-615: vary_regularisation over logspace(-1, 4) x logspace(-1, 4) on a down-sampled 8-bit stack with
    smoothing_sigma=1 and use_direct_solver=True - almost all of that grid lies in the regimes the multigrid cycle does not
    handle.  Every combination must come back converged and equal to the oracle's direct solve."""
    movie = (orc.make_texture_stack(56, 2, seed=9) * 255).astype(np.uint8)
    sa, ra = np.logspace(-1, 4, 6), np.logspace(-1, 4, 4)
    r = of.vary_regularisation(movie, speed_alpha_values=sa, remodelling_alpha_values=ra, smoothing_sigma=1, use_direct_solver=True,
                               delta_x=0.4, delta_t=10.0, return_stats=True)
    assert r["stats"]["converged_all"].all(), r["stats"]["max_relative_residual"]
    ref = orc.vary_regularisation(movie, sa, ra, smoothing_sigma=1, delta_x=0.4, delta_t=10.0)
    for k in ("speed_means", "speed_variances", "remodelling_means", "remodelling_variances", "functional"):
        np.testing.assert_allclose(r[k], ref[k], rtol=1e-6, atol=1e-12, err_msg=k)
    # default policy (multigrid first, direct for what it leaves unconverged) gives the same tables at the reference's rtol
    r2 = of.vary_regularisation(movie, speed_alpha_values=sa, remodelling_alpha_values=ra, smoothing_sigma=1, delta_x=0.4, delta_t=10.0,
                                max_iterations=80, return_stats=True)
    assert r2["stats"]["converged_all"].all()
    np.testing.assert_allclose(r2["speed_means"], ref["speed_means"], rtol=2e-3)
