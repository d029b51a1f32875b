"""CPU tests: the oracle restatement (oracle/vof_oracle.py) against the golden fixtures that
tests/golden/make_golden.py produced by running the reference itself (its direct-solver branch)."""
import numpy as np
import pytest
import scipy.sparse

from conftest import load_golden, golden_kwargs
from oracle import vof_oracle as orc


def _run(g):
    kw = golden_kwargs(g)
    return orc.variational_optical_flow(g["movie"], return_stats=True, **kw), kw


@pytest.mark.parametrize("name", ["g2_matrix_6x7.npz", "g2b_matrix_9x11.npz"])
def test_assembled_matrix_entry_for_entry(name):
    g = load_golden(name)
    kw = golden_kwargs(g)
    A, b = orc.assemble_system(g["movie"][0], g["movie"][1], kw["speed_alpha"], kw["remodelling_alpha"])
    A.sort_indices()
    n = b.size
    Aref = scipy.sparse.csr_matrix((g["data"], g["indices"], g["indptr"]), shape=(n, n))
    assert A.nnz == Aref.nnz
    np.testing.assert_array_equal(A.indptr, Aref.indptr)
    np.testing.assert_array_equal(A.indices, Aref.indices)
    np.testing.assert_allclose(A.data, Aref.data, rtol=0, atol=1e-15)
    np.testing.assert_allclose(b, g["b"], rtol=0, atol=1e-15)


def test_corner_rows_have_two_offdiagonals():
    g = load_golden("g2_matrix_6x7.npz")
    A, _ = orc.assemble_system(g["movie"][0], g["movie"][1], 2.0, 3.0)
    row0 = dict(zip(A[0].indices.tolist(), A[0].data.tolist()))
    assert row0 == {0: 1.0, 6: -1.0, 42: -1.0}          # SURVEY Appendix A.4 probe dump
    row18 = dict(zip(A[18].indices.tolist(), A[18].data.tolist()))
    assert row18 == {18: 1.0, 12: -1.0, 60: -1.0}
    row3 = dict(zip(A[3].indices.tolist(), A[3].data.tolist()))
    assert row3 == {3: 1.0, 45: -1.0}


@pytest.mark.parametrize("name,tol", [
    ("g1_avof_simple_50.npz", 1e-9), ("g2_matrix_6x7.npz", 1e-11), ("g2b_matrix_9x11.npz", 1e-11),
    ("g3_stack_32x48x4.npz", 1e-9), ("g4_blur_64.npz", 1e-8), ("g6_8bit_64.npz", 1e-8),
    ("g7_texture_64x3.npz", 1e-8)])
def test_fields_and_functionals_match_reference(name, tol):
    g = load_golden(name)
    res, kw = _run(g)
    for key in ("v_x", "v_y", "remodelling"):
        ref = g[key]
        err = np.linalg.norm(res[key] - ref) / np.linalg.norm(ref)
        assert err < tol, (key, err)
    for key in ("L1_functional", "remodelling_functional", "speed_functional"):
        assert res[key] == pytest.approx(float(g[key]), rel=1e-7, abs=1e-12), key
    # the reference assigns the remodelling sum to 'speed_functional' (OF.py:1205)
    assert res["speed_functional"] == res["remodelling_functional"]
    assert res["_relres"].max() < 1e-8


def test_g1_anchor_values():
    """Anchor values printed in SURVEY.md section 8(c) for the AVOF.py:26-50 case."""
    g = load_golden("g1_avof_simple_50.npz")
    assert g["v_x"].mean() == pytest.approx(1.049780578641995e-01, rel=1e-12)
    assert g["v_y"].mean() == pytest.approx(1.950243954755939e-01, rel=1e-12)
    assert g["remodelling"].mean() == pytest.approx(4.862204763997428e-02, rel=1e-12)
    assert float(g["L1_functional"]) == pytest.approx(5.757566910241558, rel=1e-10)
    f1, dx = orc.make_fake_data_frame(2.5, 2.5, sigma=3, width=5, dimension=50)
    np.testing.assert_allclose(f1, g["movie"][0], rtol=4e-15, atol=0)
    assert dx == float(g["kw_delta_x"])


def test_blur_matches_reference():
    g = load_golden("g4_blur_64.npz")
    np.testing.assert_allclose(orc.blur_movie(g["movie"], 2.0), g["blurred"], rtol=0, atol=1e-15)


def test_blur_pinned_on_the_real_skimage():
    """G4b: the reference's blur_movie (OF.py:282-306) run with the REAL skimage.filters.gaussian (conda python3.9,
    tests/golden/make_blur_golden.py).  The scipy stand-in that produced G4's blurred stack agrees with it to one ulp
    (2.2e-16 on [0, 1] data: scipy 1.7 vs 1.15 summation), and so does the oracle; an 8-bit stack as well."""
    g4, gb = load_golden("g4_blur_64.npz"), load_golden("g4b_blur_skimage.npz")
    np.testing.assert_array_equal(gb["movie"], g4["movie"])
    assert tuple(gb["skimage_version"]) == (0, 18, 3)
    np.testing.assert_allclose(g4["blurred"], gb["blurred_skimage"], rtol=0, atol=2.3e-16)
    np.testing.assert_allclose(orc.blur_movie(gb["movie"], float(gb["sigma"])), gb["blurred_skimage"], rtol=0, atol=2.3e-16)
    for key, sigma in (("blurred_u8_sigma_1p0", 1.0), ("blurred_u8_sigma_2p48", 2.48)):
        got = orc.blur_movie(gb["movie_u8"], sigma)
        assert got.dtype == np.float64
        np.testing.assert_allclose(got, gb[key], rtol=9e-16, atol=0)      # 2 ulp


def test_fake_frame_generator():
    g = load_golden("g8_fake_frame.npz")
    fr, dx = orc.make_fake_data_frame(1.3, 2.9, sigma=1.7, width=6.0, dimension=37)
    # vectorised exp vs the reference's scalar loop: <= 1 ulp
    np.testing.assert_allclose(fr, g["frame"], rtol=4e-15, atol=0)
    assert dx == float(g["delta_x"])


def test_g5_gaussian_config1_plumbing():
    """BASELINE config 1 (128x128x8): crops, strided samples and per-pair means."""
    g = load_golden("g5_gaussian_128x8.npz")
    movie, dx = orc.make_gaussian_stack(128, 8)
    assert dx == float(g["kw_delta_x"])
    res = orc.variational_optical_flow(movie[:3], delta_x=dx, speed_alpha=1.0, remodelling_alpha=10000.0)
    c = slice(56, 72)
    np.testing.assert_allclose(res["v_x"][:, c, c], g["v_x_crop"][:2], rtol=1e-8, atol=1e-12)
    np.testing.assert_allclose(res["v_y"][:, ::8, ::8], g["v_y_sub"][:2], rtol=1e-8, atol=1e-12)
    np.testing.assert_allclose(res["remodelling"].mean(axis=(1, 2)), g["remodelling_mean"][:2], rtol=1e-8)
    np.testing.assert_allclose(res["v_x"][:, :3, :], g["v_x_border"][:2], rtol=1e-8, atol=1e-12)


def test_matrix_free_interior_operator_equals_assembled():
    """The boundary-eliminated matrix-free operator reproduces the interior rows of A."""
    g = load_golden("g2b_matrix_9x11.npz")
    I, J = g["movie"][0], g["movie"][1]
    A, b = orc.assemble_system(I, J, 2.5, 7.0)
    rng = np.random.default_rng(0)
    xi = rng.standard_normal((3, 7, 9))
    full = orc.interior_to_full(xi)                      # satisfies the boundary rows exactly
    x = np.moveaxis(full, 0, -1).ravel()
    r = (A @ x).reshape(9, 11, 3)
    # boundary rows vanish
    assert abs(r[0]).max() < 1e-14 and abs(r[-1]).max() < 1e-14
    assert abs(r[:, 0]).max() < 1e-14 and abs(r[:, -1]).max() < 1e-14
    got = orc.apply_operator_interior(I, xi, 2.5, 7.0)
    np.testing.assert_allclose(got, np.moveaxis(r[1:-1, 1:-1], -1, 0), rtol=1e-13, atol=1e-13)
    np.testing.assert_allclose(orc.rhs_interior(I, J), np.moveaxis(b.reshape(9, 11, 3)[1:-1, 1:-1], -1, 0),
                               rtol=0, atol=1e-15)


def test_eliminated_system_reproduces_full_solution():
    g = load_golden("g2b_matrix_9x11.npz")
    I, J = g["movie"][0], g["movie"][1]
    vx, vy, gm, rr, _ = orc.solve_pair_direct(I, J, 2.5, 7.0)
    xi = np.stack([vx, vy, gm])[:, 1:-1, 1:-1]
    np.testing.assert_allclose(orc.interior_to_full(xi), np.stack([vx, vy, gm]), rtol=1e-10, atol=1e-12)
    r = orc.apply_operator_interior(I, xi, 2.5, 7.0) - orc.rhs_interior(I, J)
    assert np.linalg.norm(r) / np.linalg.norm(orc.rhs_interior(I, J)) < 1e-10


def test_vary_regularisation_matches_reference():
    """OF.py:1918-1998 on a 2x3 parameter grid (fixture produced by the reference itself)."""
    g = load_golden("g9_vary_regularisation.npz")
    r = orc.vary_regularisation(g["movie"], g["speed_alpha_values"], g["remodelling_alpha_values"], delta_x=0.5, delta_t=1.0)
    for k in ("speed_means", "speed_variances", "remodelling_means", "remodelling_variances", "functional"):
        np.testing.assert_allclose(r[k], g[k], rtol=1e-8, atol=1e-14, err_msg=k)
    assert r["converged"].all() and r["speed_means"].shape == (2, 3)


def _g10_result(g):
    return dict(original_data=np.zeros((int(g["n_frames"]), 17, 23)), v_x=g["v_x"], v_y=g["v_y"], delta_x=float(g["delta_x"]))


@pytest.mark.parametrize("box", [1, 2, 3, 4, 5, 7])
def test_subsample_matches_reference(box):
    """OF.py:1574-1646 (index work: bit-exact), incl. the half-to-even offsets round(2.5) = 2, round(3.5) = 4."""
    g = load_golden("g10_subsample.npz")
    out = orc.subsample_velocities_for_visualisation(_g10_result(g), arrow_boxsize=box)
    for got, key in zip(out, ("x_positions", "y_positions", "v_x", "v_y")):
        np.testing.assert_array_equal(got, g[f"{key}_{box}"], err_msg=key)
