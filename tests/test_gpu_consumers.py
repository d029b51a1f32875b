"""GPU tests of the rows next to the hot path: the native (speed_alpha, remodelling_alpha) sweep with its
device-side mean / variance reductions, the device-resident result mode, and the HIP gather behind
subsample_velocities_for_visualisation.  Everything goes through the C ABI (libvof.so).

Tolerances: index work (sub-sampling, device-resident vs host result) is bit-exact; the two-pass device reductions agree
with numpy's pairwise float64 mean / var to 1e-12 relative."""
import numpy as np
import pytest

from conftest import load_golden
from oracle import vof_oracle as orc

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module")
def of():
    from opticalflow_amd import optical_flow
    return optical_flow


def test_native_sweep_equals_per_combination_solves(of):
    """vof_vary_regularisation_host == looping variational_optical_flow + numpy statistics (OF.py:1974-1983)."""
    movie = orc.make_texture_stack(40, 4, seed=21)[:, :, :33]
    sa, ra = np.array([0.5, 2.0]), np.array([20.0, 3000.0])
    kw = dict(delta_x=0.5, delta_t=2.0, rtol=1e-9, initial_v_x=0.1)
    r = of.vary_regularisation(movie, sa, ra, return_stats=True, **kw)
    for i, a in enumerate(sa):
        for j, b in enumerate(ra):
            one = of.variational_optical_flow(movie, speed_alpha=a, remodelling_alpha=b, **kw)
            np.testing.assert_allclose(r["speed_means"][i, j], np.mean(one["speed"]), rtol=1e-12)
            np.testing.assert_allclose(r["speed_variances"][i, j], np.var(one["speed"]), rtol=1e-12)
            np.testing.assert_allclose(r["remodelling_means"][i, j], np.mean(one["remodelling"]), rtol=1e-12, atol=1e-18)
            np.testing.assert_allclose(r["remodelling_variances"][i, j], np.var(one["remodelling"]), rtol=1e-12)
            np.testing.assert_allclose(r["functional"][i, j], one["L1_functional"] + one["speed_functional"]
                                       + one["remodelling_functional"], rtol=1e-12)
            assert r["converged"][i, j] == one["converged"]
    assert r["stats"]["converged_all"].all() and r["stats"]["max_relative_residual"].max() < 1e-8


def test_native_sweep_batches_and_blur(of):
    """More pairs than fit in one batch (chunk statistics are merged with Chan's formula) and the blur applied once on the
    device give the same summaries as one batch / the host-side call."""
    movie = np.round(orc.make_texture_stack(36, 6, seed=22) * 255.0)
    sa, ra = np.array([1e4]), np.array([1e2, 1e3])
    kw = dict(smoothing_sigma=1.5, rtol=1e-9)
    a = of.vary_regularisation(movie, sa, ra, max_pairs_in_flight=2, **kw)
    b = of.vary_regularisation(movie, sa, ra, max_pairs_in_flight=5, **kw)
    one = of.variational_optical_flow(movie, speed_alpha=1e4, remodelling_alpha=1e3, **kw)
    for k in ("speed_means", "speed_variances", "remodelling_means", "remodelling_variances", "functional"):
        np.testing.assert_allclose(a[k], b[k], rtol=1e-12, err_msg=k)
    np.testing.assert_allclose(a["speed_variances"][0, 1], np.var(one["speed"]), rtol=1e-12)
    np.testing.assert_allclose(a["remodelling_means"][0, 1], np.mean(one["remodelling"]), rtol=1e-12)


def test_native_sweep_empty_grid_and_errors(of):
    movie = orc.make_texture_stack(16, 2, seed=3)
    r = of.vary_regularisation(movie, np.array([]), np.array([1.0, 2.0]))
    assert r["speed_means"].shape == (0, 2) and r["converged"].shape == (0, 2)
    with pytest.raises(TypeError):
        of.vary_regularisation(movie, np.array([1.0]), np.array([1.0]), no_such_argument=1)
    with pytest.raises(ValueError):
        of.vary_regularisation(movie[0], np.array([1.0]), np.array([1.0]))


def test_field_moments_are_accurate_with_a_large_offset():
    """Two-pass reduction: a field with mean 1e6 and unit variance keeps 1e-10 relative accuracy in the variance (a
    one-pass sum-of-squares formula would lose 4 digits)."""
    import torch
    from opticalflow_amd import _native
    rng = np.random.default_rng(0)
    x = 1e6 + rng.standard_normal(3 * 50 * 61)
    ref_mean = float(np.mean(x.astype(np.longdouble)))
    ref_var = float(np.var(x.astype(np.longdouble)))
    xd = torch.as_tensor(x, device="cuda:0")
    torch.cuda.synchronize()
    with _native.Solver(50, 61, 3) as s:
        m, v = s.field_moments_dev(xd, x.size)
    assert m == pytest.approx(ref_mean, rel=1e-15)
    assert v == pytest.approx(ref_var, rel=1e-10)


@pytest.mark.parametrize("box", [1, 2, 3, 4, 5, 7])
def test_device_subsample_matches_reference_fixture(of, box):
    """OF.py:1574-1646 on a device-resident result: HIP gather, bit-exact against the reference's output."""
    import torch
    g = load_golden("g10_subsample.npz")
    fr = dict(original_data=torch.zeros((int(g["n_frames"]), 17, 23), dtype=torch.float64, device="cuda:0"),
              v_x=torch.as_tensor(g["v_x"], device="cuda:0"), v_y=torch.as_tensor(g["v_y"], device="cuda:0"),
              delta_x=float(g["delta_x"]))
    out = of.subsample_velocities_for_visualisation(fr, arrow_boxsize=box)
    for got, key in zip(out, ("x_positions", "y_positions", "v_x", "v_y")):
        np.testing.assert_array_equal(got, g[f"{key}_{box}"], err_msg=key)


def test_device_resident_result_equals_host_result(of):
    """output="torch": same kernels, no PCIe in between -> bit-identical fields, functionals and flags; the input may be
    a host array or a device tensor of any real dtype (OF.py:769 casts to float64)."""
    import torch
    movie = np.round(orc.make_texture_stack(48, 4, seed=5)[:, :40, :] * 255.0).astype(np.uint8)
    kw = dict(speed_alpha=50.0, remodelling_alpha=1e3, smoothing_sigma=1.2, delta_x=0.4, delta_t=0.5, max_pairs_in_flight=2)
    host = of.variational_optical_flow(movie, **kw)
    for mv in (movie, torch.as_tensor(movie, device="cuda:0")):
        dev = of.variational_optical_flow(mv, output="torch", **kw)
        for k in ("v_x", "v_y", "speed", "remodelling", "blurred_data", "original_data"):
            assert dev[k].is_cuda and dev[k].dtype == torch.float64
            np.testing.assert_array_equal(dev[k].cpu().numpy(), host[k], err_msg=k)
        for k in ("L1_functional", "remodelling_functional", "speed_functional", "converged", "delta_x", "delta_t"):
            assert dev[k] == host[k], k
    xs, ys, vx, vy = of.subsample_velocities_for_visualisation(dev, arrow_boxsize=5)
    hx, hy, hvx, hvy = of.subsample_velocities_for_visualisation(host, arrow_boxsize=5)
    np.testing.assert_array_equal(vx, hvx)
    np.testing.assert_array_equal(vy, hvy)
    np.testing.assert_array_equal(xs, hx)


def test_result_dictionary_survives_np_save(of, tmp_path):
    """The reference's scripts store the result with np.save and read it back with np.load(...).item()
    (AVOF.py:235-238), then hand it to the overlay-movie writer."""
    import matplotlib
    matplotlib.use("Agg")
    g = load_golden("g1_avof_simple_50.npz")
    res = of.variational_optical_flow(g["movie"], delta_x=float(g["kw_delta_x"]), remodelling_alpha=1e4)
    fn = str(tmp_path / "result.npy")
    np.save(fn, res)
    back = np.load(fn, allow_pickle=True).item()          # our own file
    assert set(back) == {"v_x", "v_y", "speed", "remodelling", "original_data", "delta_x", "delta_t", "blurred_data",
                         "converged", "L1_functional", "remodelling_functional", "speed_functional"}
    np.testing.assert_array_equal(back["v_x"], res["v_x"])
    of.make_joint_overlay_movie(back, str(tmp_path / "joint.gif"), autoscale=True, arrow_scale=0.5, arrow_boxsize=4, dpi=30)


@pytest.mark.parametrize("case", range(int(__import__("os").environ.get("VOF_FUZZ_CASES_SWEEP", "6"))))
def test_seeded_random_sweeps_and_subsampling(of, case):
    """Seeded sweep over movie shapes, grid shapes, batch sizes (several combinations per batch / one / chunked movie),
    blur and units: the native parameter sweep equals per-combination solves + numpy statistics, and the device gather
    equals numpy slicing."""
    import torch
    rng = np.random.default_rng(9000 + case)
    n_i, n_j, T = int(rng.integers(8, 90)), int(rng.integers(8, 90)), int(rng.integers(2, 6))
    movie = orc.make_texture_stack(max(n_i, n_j, 16), T, seed=case)[:, :n_i, :n_j] * [1.0, 255.0][case % 2]
    scale2 = [1.0, 255.0 ** 2][case % 2]
    sa = scale2 * 10 ** rng.uniform(0.0, 1.5, int(rng.integers(1, 4)))
    ra = 10 ** rng.uniform(1.0, 4.0, int(rng.integers(1, 4)))
    kw = dict(delta_x=float(rng.uniform(0.3, 2.0)), delta_t=float(rng.uniform(0.5, 2.0)), rtol=1e-10,
              smoothing_sigma=[None, float(rng.uniform(0.6, 2.0))][case % 2])
    B = [None, T - 1, 1, 2 * (T - 1) + 1][case % 4]
    r = of.vary_regularisation(movie, sa, ra, max_pairs_in_flight=B, **kw)
    for i, a in enumerate(sa):
        for j, b in enumerate(ra):
            one = of.variational_optical_flow(movie, speed_alpha=a, remodelling_alpha=b, **kw)
            np.testing.assert_allclose(r["speed_means"][i, j], np.mean(one["speed"]), rtol=1e-8)
            np.testing.assert_allclose(r["speed_variances"][i, j], np.var(one["speed"]), rtol=1e-7)
            np.testing.assert_allclose(r["remodelling_variances"][i, j], np.var(one["remodelling"]), rtol=1e-7)
            np.testing.assert_allclose(r["functional"][i, j], one["L1_functional"] + one["speed_functional"]
                                       + one["remodelling_functional"], rtol=1e-8)
            assert r["converged"][i, j] == one["converged"]
    box = int(rng.integers(1, 9))
    dev = {k: (torch.as_tensor(v, device="cuda:0") if isinstance(v, np.ndarray) else v) for k, v in one.items()}
    for got, want in zip(of.subsample_velocities_for_visualisation(dev, arrow_boxsize=box),
                         orc.subsample_velocities_for_visualisation(one, arrow_boxsize=box)):
        np.testing.assert_array_equal(got, want)


def test_host_entry_point_optional_outputs_and_batch_schedule():
    """vof_solve_stack_host with stats == NULL and speed == NULL, one batch and several (uploads of the next batch and
    downloads of the previous one overlap the solves; uneven last batch): same fields as the plain call."""
    import ctypes as C
    from opticalflow_amd import _native
    movie = np.ascontiguousarray(orc.make_texture_stack(72, 60, seed=31)[:, :, :65])    # 59 pairs
    p = _native.default_params(speed_alpha=1.0, remodelling_alpha=1e3, rtol=1e-9)
    with _native.Solver(72, 65, 59) as s:
        ref = s.solve_host(movie, p)
    for B in (59, 50, 7, 1):                                                     # 1 / 2 (37 + 13 split) / many batches
        with _native.Solver(72, 65, B) as s:
            out = [np.full((59, 72, 65), np.nan) for _ in range(3)]
            rc = s.lib.vof_solve_stack_host(s.h, _native._ptr(movie), 60, C.byref(p), _native._ptr(out[0]), _native._ptr(out[1]),
                                            _native._ptr(out[2]), None, None)
            assert rc == 0
            for got, want in zip(out, ref[:3]):
                assert np.isfinite(got).all()
                np.testing.assert_allclose(got, want, rtol=1e-6, atol=1e-9)


@pytest.mark.parametrize("case", range(int(__import__("os").environ.get("VOF_FUZZ_CASES_BLUR", "8"))))
def test_seeded_random_blur_and_device_mode(of, case):
    """Seeded sweep over shapes, dtypes and sigmas: the device blur equals scipy's filter (the reference's skimage call)
    to 4 ulp of the data range, and the device-resident mode equals the host mode bit for bit."""
    import scipy.ndimage
    import torch
    rng = np.random.default_rng(12000 + case)
    T, n_i, n_j = int(rng.integers(1, 6)), int(rng.integers(4, 150)), int(rng.integers(4, 150))
    dtype = [np.float64, np.uint8, np.uint16, np.float32][case % 4]
    top = {np.float64: 1.0, np.uint8: 255, np.uint16: 4095, np.float32: 100.0}[dtype]
    movie = (rng.random((T, n_i, n_j)) * top).astype(dtype)
    sigma = float(rng.uniform(0.3, 6.0))
    got = of.blur_movie(movie, sigma)
    want = np.stack([scipy.ndimage.gaussian_filter(f.astype(np.float64), sigma, mode="nearest", truncate=4.0) for f in movie])
    assert got.shape == movie.shape and got.dtype == np.float64
    np.testing.assert_allclose(got, want, rtol=0, atol=4 * np.finfo(np.float64).eps * float(top))
    if T >= 2:
        kw = dict(speed_alpha=float(top) ** 2 * 2.0, remodelling_alpha=float(10 ** rng.uniform(1, 3)), smoothing_sigma=[None, sigma][case % 2],
                  max_pairs_in_flight=[None, 1][case % 2])
        host = of.variational_optical_flow(movie, **kw)
        dev = of.variational_optical_flow(torch.as_tensor(movie.astype(np.float64) if dtype == np.uint16 else movie, device="cuda:0"),
                                          output="torch", **kw)
        for k in ("v_x", "v_y", "speed", "remodelling", "blurred_data"):
            np.testing.assert_array_equal(dev[k].cpu().numpy(), host[k], err_msg=k)
        assert dev["converged"] == host["converged"] and dev["L1_functional"] == host["L1_functional"]


def test_warm_started_stack_solve_equals_cold_solve():
    """vof_solve_stack_dev solves every 3rd pair first and starts the others from their solved neighbour (the reference
    warm-starts pair k from pair k-1, OF.py:803-806): same stopping rule, same answer to the accuracy that rule implies,
    fewer iterations; outputs and statistics land in the natural pair order, also when a phase needs several batches."""
    import torch
    from opticalflow_amd import _native
    from opticalflow_amd.synthetic import texture_stack_torch
    n, T = 768, 90          # first phase 29 pairs x 0.59 Mpixel = 17 Mpixel: above the threshold of the two-phase solve
    dev = torch.device("cuda", 0)
    movie = texture_stack_torch(n, T, 4, dev)
    outs = {}
    for label, stride, B in (("cold", 0, 89), ("warm", 3, 89), ("warm, small batches", 3, 13), ("warm, stride 2", 2, 89)):
        p = _native.default_params(speed_alpha=1.0, remodelling_alpha=1e4, rtol=1e-9, warm_start_stride=stride)
        f = [torch.empty((T - 1, n, n), dtype=torch.float64, device=dev) for _ in range(4)]
        torch.cuda.synchronize()
        with _native.Solver(n, n, B) as s:
            st = s.solve_dev(movie, T, p, *f)
        assert st["converged"].all() and st["relative_residual"].max() <= 1.5e-9
        outs[label] = ([t.cpu().numpy() for t in f], st)
    cold_fields, cold_st = outs["cold"]
    for label in ("warm", "warm, small batches", "warm, stride 2"):
        fields, st = outs[label]
        for a, b in zip(fields, cold_fields):
            assert np.linalg.norm(a - b) / np.linalg.norm(b) < 1e-6
        assert st["iterations"].sum() < cold_st["iterations"].sum()
        np.testing.assert_allclose(st["L1_functional"], cold_st["L1_functional"], rtol=1e-5)
    # the pairs solved first (every 3rd) do not see the warm start: same iterations, same fields up to what the summation
    # order of the reductions can do (the band height of the streaming kernels follows the batch size: the Krylov scalars
    # differ in their last bits, which the float32 storage of the coarse-level cycle vectors turns into differences of the
    # preconditioner's output at the 1e-7 level - both runs then stop at rtol 1e-9, i.e. agree to about that)
    np.testing.assert_allclose(outs["warm"][0][0][::3], cold_fields[0][::3], rtol=2e-8, atol=1e-11)
    np.testing.assert_array_equal(outs["warm"][1]["iterations"][::3], cold_st["iterations"][::3])


def test_host_entry_point_warm_start_per_batch():
    """vof_solve_stack_host applies the same two-phase warm start inside each of its batches: equal to the cold solve to the
    accuracy of the stopping rule, fewer iterations, statistics and fields in natural order across the batch schedule."""
    from opticalflow_amd import _native
    from opticalflow_amd.synthetic import texture_stack_numpy
    n, T = 768, 100
    movie = texture_stack_numpy(n, T, seed=6)
    res = {}
    for label, stride, B in (("cold", 0, 99), ("warm", 3, 99), ("warm, two batches", 3, 90)):   # 90 + 9 pairs: the small batch stays cold
        p = _native.default_params(speed_alpha=1.0, remodelling_alpha=1e4, rtol=1e-9, warm_start_stride=stride)
        with _native.Solver(n, n, B) as s:
            res[label] = s.solve_host(movie, p)
        st = res[label][4]
        assert st["converged"].all() and st["relative_residual"].max() <= 1.5e-9
    for label in ("warm", "warm, two batches"):
        for a, b in zip(res[label][:4], res["cold"][:4]):
            assert np.linalg.norm(a - b) / np.linalg.norm(b) < 1e-6
        assert res[label][4]["iterations"].sum() < res["cold"][4]["iterations"].sum()
        np.testing.assert_allclose(res[label][4]["L1_functional"], res["cold"][4]["L1_functional"], rtol=1e-5)
