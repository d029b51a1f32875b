"""GPU property tests on the BASELINE.json configurations at FULL depth (C2 512x512x64 seed 0,
C3 1024x1024x256 seed 1, one GPU's share of C4 1024x1024x1024 seed 2 and of C5 2048x2048x128 seed 3, and C5 on one GPU).

A direct CPU solve is impractical at these sizes, so parity is checked through size-independent properties against the
oracle's matrix-free operator (oracle/vof_oracle.py, pinned on the reference's own matrices): the reference's stopping
rule ||b - A x|| <= 1e-6 ||b|| (OF.py:1120,1126) met by EVERY pair on the independent residual (OF.py:1150-1151), the
same residual re-evaluated on the CPU for sampled pairs (including pairs of the warm-started second phase), the mirror
structure of OF.py:1159-1166, and the known flow (0.3, 0.6) px/frame of the exactly translating texture.  The same tests
check the device-side workload generator (the bench's input) against the oracle's restatement of the recipe.
"""
import numpy as np
import pytest

from oracle import vof_oracle as orc

pytestmark = pytest.mark.gpu

RTOL = 1e-6


@pytest.fixture(scope="module")
def of():
    from opticalflow_amd import optical_flow
    yield optical_flow
    optical_flow.release_device_memory()


def _solve_and_check(of, n, T, seed, first_frame, stride, pairs_in_flight=None, sample=(0, 1, 2), shift_tol=0.02):
    import torch
    from opticalflow_amd.synthetic import texture_stack_torch
    dev = torch.device("cuda", 0)
    movie = texture_stack_torch(n, T, seed, dev, first_frame=first_frame)
    res = of.variational_optical_flow(movie, speed_alpha=1.0, remodelling_alpha=1e4, output="torch", return_stats=True,
                                      warm_start_stride=stride, max_pairs_in_flight=pairs_in_flight)
    st = res["stats"]
    P = T - 1
    assert st.shape == (P,)
    assert st["converged"].all(), np.flatnonzero(st["converged"] == 0)
    assert st["relative_residual"].max() <= RTOL * (1 + 1e-9)
    assert st["iterations"].max() <= 12
    assert res["converged"] is True
    # mirror structure (rows first, then columns: OF.py:1304-1316) on every pair, evaluated on the device
    for k in ("v_x", "v_y", "remodelling", "speed"):
        f = res[k]
        assert tuple(f.shape) == (P, n, n) and f.dtype == torch.float64
        assert torch.equal(f[:, 0, :], f[:, 2, :]) and torch.equal(f[:, -1, :], f[:, -3, :])
        assert torch.equal(f[:, :, 0], f[:, :, 2]) and torch.equal(f[:, :, -1], f[:, :, -3])
        assert bool(torch.isfinite(f).all())
    assert float(res["v_x"].mean()) == pytest.approx(0.3, abs=shift_tol)
    assert float(res["v_y"].mean()) == pytest.approx(0.6, abs=shift_tol)
    assert abs(float(res["remodelling"].mean())) < 5e-3
    torch.testing.assert_close(res["speed"], torch.sqrt(res["v_x"] ** 2 + res["v_y"] ** 2), rtol=1e-15, atol=0)   # OF.py:1191
    # sampled pairs: generator vs the oracle's recipe, and the independent residual on the CPU
    for k in sample:
        k = min(k, P - 1)
        frames = movie[k:k + 2].cpu().numpy()
        ref = orc.make_texture_stack(n, 2, seed=seed, first_frame=first_frame + k)
        np.testing.assert_allclose(frames, ref, rtol=0, atol=1e-12)
        xi = np.stack([res[f][k].cpu().numpy() for f in ("v_x", "v_y", "remodelling")])[:, 1:-1, 1:-1]
        b = orc.rhs_interior(frames[0], frames[1])
        r = b - orc.apply_operator_interior(frames[0], xi, 1.0, 1e4)
        rel = np.linalg.norm(r) / np.linalg.norm(b)
        assert rel <= RTOL * (1 + 1e-6), (k, rel)
        assert rel == pytest.approx(st["relative_residual"][k], rel=1e-5), k
    return res, st


def test_config2_512x512x64_full_depth(of):
    """C2: 63 pairs of 512x512, seed 0; default warm start (below the two-phase threshold: every pair cold) and stride 0."""
    _res, st = _solve_and_check(of, 512, 64, seed=0, first_frame=0, stride=None, sample=(0, 31, 62))
    _res, st0 = _solve_and_check(of, 512, 64, seed=0, first_frame=0, stride=0, sample=(17,))
    assert st0["iterations"].sum() >= st["iterations"].sum() - 2


def test_config3_1024x1024x256_full_depth_warm_and_cold(of):
    """C3, the configuration the metric is quoted on: all 255 pairs in one batch (as bench.py runs it), two-phase warm
    start (stride 3: pairs 0, 3, 6, ... first) and the cold solve.  Samples 1 and 200 are phase-2 pairs."""
    res_w, st_w = _solve_and_check(of, 1024, 256, seed=1, first_frame=0, stride=None, pairs_in_flight=255,
                                   sample=(0, 1, 200, 254))
    res_c, st_c = _solve_and_check(of, 1024, 256, seed=1, first_frame=0, stride=0, pairs_in_flight=255, sample=(1, 128))
    assert st_w["iterations"].mean() < st_c["iterations"].mean()          # the warm start pays on this stack
    # same answer to the accuracy the stopping rule implies
    import torch
    for k in ("v_x", "v_y", "remodelling"):
        num = float(torch.linalg.vector_norm(res_w[k] - res_c[k])); den = float(torch.linalg.vector_norm(res_c[k]))
        assert num / den < 1e-3, k


def test_config4_share_of_one_gpu_1024x1024x129(of):
    """C4 = 1024x1024x1024 over 8 GPUs: rank 3's share (128 pairs, frames 384 .. 512 of the seed-2 stack)."""
    _solve_and_check(of, 1024, 129, seed=2, first_frame=3 * 128, stride=None, sample=(0, 64, 127))


def test_config5_share_of_one_gpu_2048x2048x17(of):
    """C5 = 2048x2048x128 over 8 GPUs: rank 5's share (16 pairs), full-depth hierarchy (9 levels)."""
    _solve_and_check(of, 2048, 17, seed=3, first_frame=5 * 16, stride=None, sample=(0, 7, 15))


def test_config5_2048x2048x128_on_one_gpu(of):
    """C5 on a single GPU: all 127 pairs of 2048x2048 in ONE batch (the stencil storage is sized by the 8-bit format:
    1.3 GB per pair in flight; round 2 needed two batches), two-phase warm start inside the stack."""
    _solve_and_check(of, 2048, 128, seed=3, first_frame=0, stride=None, sample=(0, 1, 100, 126))
    assert of._cache["solver"].max_pairs == 127


@pytest.mark.parametrize("n", [64, 512, 1024])
def test_device_texture_generator_against_the_recipe(n):
    """f-2: the HIP generator that feeds bench.py vs the oracle's restatement of SURVEY.md section 8(d) (draw order,
    clip, shift, first_frame offset), and vs the package's own numpy version."""
    import torch
    from opticalflow_amd.synthetic import texture_stack_torch, texture_stack_numpy
    T, seed, first = 5, {64: 4, 512: 0, 1024: 1}[n], 11
    dev = torch.device("cuda", 0)
    got = texture_stack_torch(n, T, seed, dev, first_frame=first).cpu().numpy()
    ref = orc.make_texture_stack(n, T, seed=seed, first_frame=first)
    np.testing.assert_allclose(got, ref, rtol=0, atol=1e-12)
    assert got.min() >= 0.0 and got.max() <= 1.0 and got.std() > 0.05
    np.testing.assert_allclose(texture_stack_numpy(n, T, seed, first_frame=first), ref, rtol=0, atol=1e-13)
    # the time-varying variant of bench.py: same generator, other per-frame offsets
    w = texture_stack_torch(n, T, seed, dev, first_frame=first, wobble=0.3).cpu().numpy()
    np.testing.assert_allclose(w, texture_stack_numpy(n, T, seed, first_frame=first, wobble=0.3), rtol=0, atol=1e-12)
    assert np.abs(w - got).max() > 1e-3


def test_nan_frame_above_the_two_phase_threshold_poisons_its_pairs_only(of):
    """A NaN pixel in frame 4 of a 768x768x91 stack (first phase = 30 pairs = 17.7 Mpixel: the two-phase warm start is
    on).  Pairs 3 and 4 hold the frame and are reported unconverged; pair 2 would take its guess from pair 3 and must
    fall back to the constant initial fields instead of inheriting the NaN."""
    import torch
    from opticalflow_amd.synthetic import texture_stack_torch
    movie = texture_stack_torch(768, 91, 5, torch.device("cuda", 0))
    movie[4, 300, 301] = float("nan")
    res = of.variational_optical_flow(movie, remodelling_alpha=1e4, output="torch", return_stats=True, max_iterations=60)
    st = res["stats"]
    bad = np.flatnonzero(st["converged"] == 0).tolist()
    assert bad == [3, 4]
    good = [k for k in range(90) if k not in (3, 4)]
    assert st["relative_residual"][good].max() <= RTOL * (1 + 1e-9)
    assert bool(torch.isfinite(res["v_x"][good]).all())
    assert st["iterations"][2] <= 12


def test_params_of_another_abi_are_rejected_by_the_solver():
    """The ABI guard at the solve entry points: a vof_params whose size / version fields are wrong is refused."""
    from opticalflow_amd import _native
    movie = orc.make_texture_stack(32, 2, seed=1)
    with _native.Solver(32, 32, 1) as s:
        p = _native.default_params()
        p.struct_size = 104
        with pytest.raises(_native.VofError, match="ABI mismatch"):
            s.solve_host(movie, p)
        p = _native.default_params()
        p.abi_version = 104
        with pytest.raises(_native.VofError, match="ABI mismatch"):
            s.solve_host(movie, p)
        out = s.solve_host(movie, _native.default_params(remodelling_alpha=1e4))
        assert out[-1]["converged"].all()
