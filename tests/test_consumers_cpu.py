"""Host logic of the result consumers; no GPU, no native compute calls."""
import os
import sys

import numpy as np
import pytest

sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
from tests.conftest import load_golden  # noqa: E402
from opticalflow_amd import optical_flow as of  # noqa: E402


def fake_result(T=3, n=(17, 23), seed=0, delta_x=0.25):
    rng = np.random.default_rng(seed)
    r = dict(original_data=rng.random((T,) + n) * 255, delta_x=delta_x, delta_t=1.0)
    r["blurred_data"] = r["original_data"]
    for k in ("v_x", "v_y", "remodelling"):
        r[k] = rng.standard_normal((T - 1,) + n)
    r["speed"] = np.sqrt(r["v_x"] ** 2 + r["v_y"] ** 2)
    return r


@pytest.mark.parametrize("box", [1, 2, 3, 4, 5, 7])
def test_subsample_host_matches_reference_fixture(box):
    """Host-resident results are sampled by numpy index arithmetic: bit-exact against the reference's output."""
    g = load_golden("g10_subsample.npz")
    fr = dict(original_data=np.zeros((int(g["n_frames"]), 17, 23)), v_x=g["v_x"], v_y=g["v_y"], delta_x=float(g["delta_x"]))
    out = of.subsample_velocities_for_visualisation(fr, arrow_boxsize=box)
    for got, key in zip(out, ("x_positions", "y_positions", "v_x", "v_y")):
        np.testing.assert_array_equal(got, g[f"{key}_{box}"], err_msg=key)
        assert got.dtype == np.float64


def test_subsample_box_larger_than_image_and_bad_box():
    fr = fake_result()
    xs, ys, vx, vy = of.subsample_velocities_for_visualisation(fr, arrow_boxsize=40)
    assert xs.shape == (0,) and ys.shape == (0,) and vx.shape == (2, 0, 0) and vy.shape == (2, 0, 0)
    with pytest.raises(ValueError):
        of.subsample_velocities_for_visualisation(fr, arrow_boxsize=0)


def test_subsample_iteration_branch():
    """OF.py:1621-1625: 'v_x_steps' / 'v_y_steps' (frames, iterations, x, y) entries of the legacy iterative solvers."""
    fr = fake_result()
    rng = np.random.default_rng(1)
    fr["v_x_steps"] = rng.standard_normal((2, 4, 17, 23))
    fr["v_y_steps"] = rng.standard_normal((2, 4, 17, 23))
    _, _, vx, vy = of.subsample_velocities_for_visualisation(fr, iteration=2, arrow_boxsize=5)
    np.testing.assert_array_equal(vx, fr["v_x_steps"][:, 2, 2::5, 2::5][:, :3, :4])
    np.testing.assert_array_equal(vy, fr["v_y_steps"][:, 2, 2::5, 2::5][:, :3, :4])


def test_overlay_movies_render(tmp_path):
    """The two movie writers the reference's scripts call after the solve (AVOF.py:52, 240): they must accept the
    result dict and write a file (gif through pillow here; the scripts use mp4 where ffmpeg is installed)."""
    import matplotlib
    matplotlib.use("Agg")
    fr = fake_result()
    f1, f2 = str(tmp_path / "velocity.gif"), str(tmp_path / "joint.gif")
    of.make_velocity_overlay_movie(fr, f1, arrow_boxsize=4, autoscale=True, dpi=40)
    of.make_joint_overlay_movie(fr, f2, arrow_boxsize=4, autoscale=True, arrow_scale=0.5, dpi=40)
    assert os.path.getsize(f1) > 0 and os.path.getsize(f2) > 0


def test_costum_imshow_extent_and_labels():
    import matplotlib
    matplotlib.use("Agg")
    import matplotlib.pyplot as plt
    fig = plt.figure()
    of.costum_imshow(np.zeros((10, 20)), delta_x=0.5, v_min=0, v_max=1)
    ax = plt.gca()
    assert ax.images[0].get_extent() == [0, 10.0, 5.0, 0]
    assert ax.get_xlabel().startswith("y-position") and ax.get_ylabel().startswith("x-position")
    assert ax.images[0].get_clim() == (0, 1)
    plt.close(fig)


def test_shim_module_exports_what_the_scripts_use():
    """analysis/analyse_variational_optical_flow.py:22-66 does `import optical_flow` from source/ and calls these."""
    import importlib.util
    path = os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", "source", "optical_flow.py")
    spec = importlib.util.spec_from_file_location("optical_flow_shim", path)
    m = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(m)
    for name in ("variational_optical_flow", "vary_regularisation", "make_fake_data_frame", "costum_imshow",
                 "make_joint_overlay_movie", "make_velocity_overlay_movie", "subsample_velocities_for_visualisation",
                 "blur_movie", "apply_constant_boundary_condition", "format_elapsed_time"):
        assert callable(getattr(m, name)), name
