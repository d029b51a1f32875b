"""GPU tests of the solver's building blocks, each called through the C ABI (vof_debug_*) and
compared with the independent numpy implementation in oracle/mg_prototype.py / oracle/vof_oracle.py.
FP64 everywhere: tolerances are rounding-level (relative 1e-11 .. 1e-9)."""
import numpy as np
import pytest

from oracle import vof_oracle as orc, mg_prototype as mg

pytestmark = pytest.mark.gpu


def relerr(a, b):
    return float(np.linalg.norm(np.asarray(a) - np.asarray(b)) / max(np.linalg.norm(b), 1e-300))


@pytest.fixture(scope="module")
def native():
    import os
    from opticalflow_amd import _native
    _native.load_library()
    # the register-resident level-0 pass k_sweep0r is the default from 512 one-wave blocks per launch on; the small grids of this
    # file must run it too (reference_quirks=0 cases run the LDS pass k_sweep0m, so both are covered)
    old = os.environ.get("VOF_SWEEP0R_MIN_BLOCKS")
    os.environ["VOF_SWEEP0R_MIN_BLOCKS"] = "0"
    yield _native
    if old is None:
        os.environ.pop("VOF_SWEEP0R_MIN_BLOCKS", None)
    else:
        os.environ["VOF_SWEEP0R_MIN_BLOCKS"] = old


def make_case(kind, shape, npairs, seed):
    rng = np.random.default_rng(seed)
    if kind == "texture":
        n = max(shape)
        mv = orc.make_texture_stack(n, npairs + 1, seed=seed)[:, :shape[0], :shape[1]]
    else:
        mv = rng.random((npairs + 1,) + shape) * 2.0
    return np.ascontiguousarray(mv)


CASES = [("random", (9, 11), 1, 2.5, 7.0, 1), ("random", (20, 37), 2, 1.0, 50.0, 2),
         ("texture", (66, 66), 2, 1.0, 1e4, 3), ("texture", (50, 83), 1, 1.0, 1e4, 4)]


@pytest.mark.parametrize("kind,shape,npairs,alpha,beta,seed", CASES)
@pytest.mark.parametrize("quirks", [1, 0])
def test_rhs_operator_smoother(native, kind, shape, npairs, alpha, beta, seed, quirks):
    mv = make_case(kind, shape, npairs, seed)
    p = native.default_params(speed_alpha=alpha, remodelling_alpha=beta, reference_quirks=quirks)
    rng = np.random.default_rng(seed + 100)
    with native.Solver(shape[0], shape[1], npairs) as s:
        s.debug_setup(mv, p)
        assert s.level_shape(0) == (shape[0] - 2, shape[1] - 2)
        b = s.debug_rhs()
        x = rng.standard_normal(b.shape)
        y = s.debug_apply(0, x)
        for k in range(npairs):
            assert relerr(b[k], orc.rhs_interior(mv[k], mv[k + 1], bool(quirks))) < 1e-14
            assert relerr(y[k], orc.apply_operator_interior(mv[k], x[k], alpha, beta, bool(quirks))) < 1e-13
        if s.num_levels > 1:
            # one colour at a time, against the serial numpy sweep in the same colour order
            xg = x.copy()
            xr = x.copy()
            C = [mg.fine_stencil(mv[k], alpha, beta, bool(quirks)) for k in range(npairs)]
            for colour in (0, 1, 2, 3, 3, 2, 1, 0):
                xg = s.debug_gs(0, xg, b, colour)
                for k in range(npairs):
                    mg.gs_colour(C[k], xr[k], b[k], colour)
                assert relerr(xg, xr) < 1e-11, colour


def test_stencil_storage_follows_the_format_in_use(native):
    """The stencil storage of the stored levels is sized by the format (120 / 180 / 324 / 648 bytes per coarse point): a
    context starts with the default format's, grows when a call asks for a wider one (never shrinks), reports it through
    vof_workspace_bytes == vof_query_workspace_for, and the hierarchy is right after every re-allocation."""
    kind, shape, npairs, alpha, beta, seed = CASES[2]
    mv = make_case(kind, shape, npairs, seed)
    H = [mg.Hierarchy(mv[k], alpha, beta) for k in range(npairs)]
    tol = {0: 1e-12, 1: 2e-6, 2: 4e-3, 3: 6e-2}
    with native.Solver(shape[0], shape[1], npairs) as s:
        widest = 3
        assert s.workspace_bytes == pytest.approx(native.query_workspace(shape[0], shape[1], npairs), rel=1e-2)   # small grids: the fixed-size buffers count
        for fmt in (3, 2, 3, 0, 1, 3):
            s.debug_setup(mv, native.default_params(speed_alpha=alpha, remodelling_alpha=beta, coarse_precision=fmt))
            widest = min(widest, fmt)
            staging = (npairs + 1 + 4 * npairs) * shape[0] * shape[1] * 8          # vof_debug_setup uploads through the host-API staging
            assert s.workspace_bytes - staging == pytest.approx(native.query_workspace(shape[0], shape[1], npairs, widest, 3), rel=1e-2)
            for lvl in range(1, s.num_levels):
                Cg = s.debug_stencil(lvl)
                for k in range(npairs):
                    assert relerr(Cg[k], H[k].levels[lvl]) < tol[fmt], (fmt, lvl)


@pytest.mark.parametrize("kind,shape,npairs,alpha,beta,seed", CASES[1:])
@pytest.mark.parametrize("coarse_precision", [0, 1, 2, 3])
def test_galerkin_hierarchy_and_transfers(native, kind, shape, npairs, alpha, beta, seed, coarse_precision):
    mv = make_case(kind, shape, npairs, seed)
    p = native.default_params(speed_alpha=alpha, remodelling_alpha=beta, coarse_precision=coarse_precision)
    # 0: float64 stencils; 1: float32 (float32 accumulation of the Galerkin products); 2: bfloat16 off-diagonal blocks
    # (8-bit mantissa: 2^-9 relative) + float32 diagonal block that absorbs their rounding (block row sums kept)
    # 3: 8-bit floats (4 significant bits: 2^-5 relative) in units of a power of two per equation, same diagonal block
    tol = {0: 1e-12, 1: 2e-6, 2: 4e-3, 3: 6e-2}[coarse_precision]
    rng = np.random.default_rng(seed)
    with native.Solver(shape[0], shape[1], npairs) as s:
        s.debug_setup(mv, p)
        H = [mg.Hierarchy(mv[k], alpha, beta) for k in range(npairs)]
        assert s.num_levels == len(H[0].levels)
        for lvl in range(s.num_levels):
            assert s.level_shape(lvl) == tuple(H[0].levels[lvl].shape[-2:])
        for lvl in range(1, s.num_levels):
            Cg = s.debug_stencil(lvl)
            for k in range(npairs):
                assert relerr(Cg[k], H[k].levels[lvl]) < tol, lvl
                if coarse_precision >= 2:
                    # what the format is about: every block row sum (sum over the 9 neighbours of the 3x3 blocks) is the
                    # unrounded one - the screening term of the gamma rows (-1 next to 4 beta) lives there
                    Cr = H[k].levels[lvl]
                    rs_g, rs_r = Cg[k].sum(axis=(0, 1)), Cr.sum(axis=(0, 1))
                    assert np.abs(rs_g - rs_r).max() <= 2e-5 * max(1.0, np.abs(Cr).max()), lvl
                    off = np.ones((3, 3), bool); off[1, 1] = False
                    q = Cg[k][off]                       # off-diagonal blocks are exactly representable in the format
                    if coarse_precision == 2:
                        assert np.array_equal(q.astype(np.float32).view(np.uint32) & 0xFFFF, np.zeros(q.shape, np.uint32))
                    else:   # 4 significant bits
                        m, _ = np.frexp(q)
                        assert np.array_equal(m * 16, np.round(m * 16))
            # stored-stencil operator and smoother
            x = rng.standard_normal((npairs, 3) + s.level_shape(lvl))
            b = rng.standard_normal(x.shape)
            y = s.debug_apply(lvl, x)
            for k in range(npairs):
                assert relerr(y[k], mg.apply_stencil(H[k].levels[lvl], x[k])) < max(tol, 1e-12)
                # the stencil as the host unpacks it is the operator the device kernels apply (packed formats: the decoders agree)
                assert relerr(y[k], mg.apply_stencil(Cg[k], x[k])) < 1e-13
            if coarse_precision == 0:
                xg, xr = x.copy(), x.copy()
                for colour in (0, 1, 2, 3):
                    xg = s.debug_gs(lvl, xg, b, colour)
                    for k in range(npairs):
                        mg.gs_colour(H[k].levels[lvl], xr[k], b[k], colour)
                assert relerr(xg, xr) < 1e-9, lvl
        # transfer operators on every level
        for lvl in range(s.num_levels - 1):
            fshape = (npairs, 3) + s.level_shape(lvl)
            f = rng.standard_normal(fshape)
            cg = s.debug_restrict(lvl, f)
            cshape = cg.shape
            for k in range(npairs):
                assert relerr(cg[k], mg.restrict(f[k])) < 1e-13
            e = rng.standard_normal(cshape)
            fg = s.debug_prolong_add(lvl, f, e)
            for k in range(npairs):
                assert relerr(fg[k], f[k] + mg.prolong(e[k], *fshape[-2:])) < 1e-13
        # coarsest dense solve
        last = s.num_levels - 1
        r = rng.standard_normal((npairs, 3) + s.level_shape(last))
        eg = s.debug_coarse_solve(r)
        for k in range(npairs):
            er = (H[k].coarse_inv @ r[k].ravel()).reshape(r[k].shape)
            assert relerr(eg[k], er) < {0: 1e-8, 1: 1e-3, 2: 5e-2, 3: 0.6}[coarse_precision]


@pytest.mark.parametrize("kind,shape,npairs,alpha,beta,seed", CASES[1:3])
def test_vcycle_matches_prototype(native, kind, shape, npairs, alpha, beta, seed):
    mv = make_case(kind, shape, npairs, seed)
    p = native.default_params(speed_alpha=alpha, remodelling_alpha=beta, coarse_precision=0, vcycle_precision=0, nu_pre=2, nu_post=2,
                              nu_pre_coarse=2, nu_post_coarse=2, w_cycle_level=-1)
    with native.Solver(shape[0], shape[1], npairs) as s:
        s.debug_setup(mv, p)
        r = s.debug_rhs()
        e = s.debug_vcycle(r)
        for k in range(npairs):
            H = mg.Hierarchy(mv[k], alpha, beta)
            assert relerr(e[k], H.vcycle(r[k], 2, 2)) < 1e-9


def test_tiny_grid_single_level(native):
    """6x7 image: the interior 4x5 grid is below the coarsening threshold -> dense solve only."""
    from conftest import load_golden
    g = load_golden("g2_matrix_6x7.npz")
    p = native.default_params(speed_alpha=2.0, remodelling_alpha=3.0)
    with native.Solver(6, 7, 1) as s:
        s.debug_setup(g["movie"], p)
        assert s.num_levels == 1
        C = s.debug_stencil(0)
        assert relerr(C[0], mg.fine_stencil(g["movie"][0], 2.0, 3.0)) < 1e-14


def test_create_rejects_bad_arguments(native):
    with pytest.raises(native.VofError, match="4x4"):
        native.Solver(3, 10, 1)
    with pytest.raises(native.VofError):
        native.Solver(16, 16, 0)
    with pytest.raises(native.VofError, match="device"):
        native.Solver(16, 16, 1, device=99)


SWEEP_CASES = [("random", (20, 37), 2, 1.0, 50.0, 2), ("texture", (66, 66), 2, 1.0, 1e4, 3),
               ("texture", (50, 83), 1, 1.0, 1e4, 4), ("texture", (140, 270), 1, 1.0, 1e4, 5),
               ("random", (12, 300), 1, 2.0, 5.0, 6), ("random", (263, 13), 1, 2.0, 5.0, 7)]


@pytest.mark.parametrize("kind,shape,npairs,alpha,beta,seed", SWEEP_CASES)
@pytest.mark.parametrize("quirks", [1, 0])
def test_fused_sweep_equals_colour_by_colour_order(native, kind, shape, npairs, alpha, beta, seed, quirks):
    """The streaming fused sweep must reproduce the global colour order 0,1,2,3 (reverse: 3,2,1,0)
    exactly - it is compared with the serial numpy sweep and, bit for bit, with the per-colour kernels."""
    mv = make_case(kind, shape, npairs, seed)
    p = native.default_params(speed_alpha=alpha, remodelling_alpha=beta, reference_quirks=quirks)
    rng = np.random.default_rng(seed)
    with native.Solver(shape[0], shape[1], npairs) as s:
        s.debug_setup(mv, p)
        b = s.debug_rhs()
        x = rng.standard_normal(b.shape)
        C = [mg.fine_stencil(mv[k], alpha, beta, bool(quirks)) for k in range(npairs)]
        for reverse in (False, True):
            for from_zero in (False, True):
                xg = s.debug_sweep(0, x, b, reverse=reverse, from_zero=from_zero)
                xr = np.zeros_like(x) if from_zero else x.copy()
                xc = xr.copy()
                for k in range(npairs):
                    mg.smooth(C[k], xr[k], b[k], 1, reverse=reverse)
                assert relerr(xg, xr) < 1e-11, (reverse, from_zero)
                for colour in ((3, 2, 1, 0) if reverse else (0, 1, 2, 3)):
                    xc = s.debug_gs(0, xc, b, colour)
                np.testing.assert_array_equal(xg, xc)


M_CASES = SWEEP_CASES + [("texture", (130, 258), 2, 1.0, 1e4, 8), ("random", (301, 250), 1, 0.7, 30.0, 9),
                        ("texture", (6, 130), 1, 1.0, 1e4, 10), ("random", (77, 6), 2, 3.0, 2.0, 11)]


@pytest.mark.parametrize("kind,shape,npairs,alpha,beta,seed", M_CASES)
@pytest.mark.parametrize("quirks", [1, 0])
def test_level0_smoothing_passes_equal_colour_by_colour_order(native, kind, shape, npairs, alpha, beta, seed, quirks):
    """The smoother as the cycle runs it on level 0 (k_sweep0m: merged colours, two sweeps per pass when the row length is
    even; k_sweep0 otherwise) against nu x 4 launches of the per-colour kernel: bit for bit, forward and reverse order,
    from zero and from a given x, one to three sweeps (three = one double pass + one single pass)."""
    mv = make_case(kind, shape, npairs, seed)
    p = native.default_params(speed_alpha=alpha, remodelling_alpha=beta, reference_quirks=quirks)
    rng = np.random.default_rng(seed)
    with native.Solver(shape[0], shape[1], npairs) as s:
        s.debug_setup(mv, p)
        if s.num_levels < 2:
            pytest.skip("single-level grid")
        b = s.debug_rhs()
        x = rng.standard_normal(b.shape)
        for nu in (1, 2, 3):
            for reverse in (False, True):
                for from_zero in (False, True):
                    xg = s.debug_smooth(0, x, b, nu, reverse=reverse, from_zero=from_zero)
                    xc = np.zeros_like(x) if from_zero else x.copy()
                    for _ in range(nu):
                        for colour in ((3, 2, 1, 0) if reverse else (0, 1, 2, 3)):
                            xc = s.debug_gs(0, xc, b, colour)
                    np.testing.assert_array_equal(xg, xc, err_msg=f"nu={nu} reverse={reverse} from_zero={from_zero}")


@pytest.mark.parametrize("kind,shape,npairs,alpha,beta,seed", SWEEP_CASES[1:4])
@pytest.mark.parametrize("coarse_precision", [0, 1, 2, 3])
def test_fused_sweep_on_stored_levels(native, kind, shape, npairs, alpha, beta, seed, coarse_precision):
    mv = make_case(kind, shape, npairs, seed)
    p = native.default_params(speed_alpha=alpha, remodelling_alpha=beta, coarse_precision=coarse_precision)
    rng = np.random.default_rng(seed)
    with native.Solver(shape[0], shape[1], npairs) as s:
        s.debug_setup(mv, p)
        for lvl in range(1, s.num_levels - 1):
            x = rng.standard_normal((npairs, 3) + s.level_shape(lvl))
            b = rng.standard_normal(x.shape)
            for reverse in (False, True):
                xg = s.debug_sweep(lvl, x, b, reverse=reverse)
                xc = x.copy()
                for colour in ((3, 2, 1, 0) if reverse else (0, 1, 2, 3)):
                    xc = s.debug_gs(lvl, xc, b, colour)
                assert relerr(xg, xc) < 1e-13, (lvl, reverse)


@pytest.mark.parametrize("kind,shape,npairs,alpha,beta,seed", SWEEP_CASES[1:4] + [("texture", (300, 530), 1, 1.0, 1e4, 12)])
@pytest.mark.parametrize("coarse_precision,vcycle_precision", [(0, 0), (1, 0), (2, 0), (2, 1), (3, 0), (3, 1)])
def test_coarse_rhs_from_the_sweep_update(native, kind, shape, npairs, alpha, beta, seed, coarse_precision, vcycle_precision):
    """k_resrestrict_u: after ONE forward sweep x_old -> x_new the residual is -U (x_new - x_old) (U = couplings to the
    neighbours updated later in the colour order); its restriction must equal R (b - A x_new) from the stand-alone operator
    and restriction kernels, from zero and from a given x_old, on every stored level (odd / even sizes, chunk edges)."""
    mv = make_case(kind, shape, npairs, seed)
    p = native.default_params(speed_alpha=alpha, remodelling_alpha=beta, coarse_precision=coarse_precision,
                              vcycle_precision=vcycle_precision)
    rng = np.random.default_rng(seed)
    # float32 vectors: the stand-alone residual also sees the rounding of x_new to float32 (D_i x_i = ... no longer holds to
    # float64 accuracy), the update formula does not: agreement to float32 rounding x the size of the diagonal blocks
    tol = 1e-11 if vcycle_precision == 0 else 5e-4
    with native.Solver(shape[0], shape[1], npairs) as s:
        s.debug_setup(mv, p)
        for lvl in range(1, s.num_levels - 1):
            b = rng.standard_normal((npairs, 3) + s.level_shape(lvl))
            for x_old in (None, rng.standard_normal(b.shape)):
                x_new = s.debug_sweep(lvl, np.zeros_like(b) if x_old is None else x_old, b, from_zero=x_old is None)
                want = s.debug_restrict(lvl, b - s.debug_apply(lvl, x_new))
                got = s.debug_resrestrict_u(lvl, x_new, x_old)
                scale = np.linalg.norm(s.debug_restrict(lvl, b))
                assert np.linalg.norm(got - want) < tol * scale, (lvl, x_old is None, np.linalg.norm(got - want) / scale)


def test_coarse_correction_folded_into_the_stored_sweep(native, monkeypatch):
    """VOF_FOLD_STORED=1: k_sweep_st interpolates the coarse-grid correction while it loads its rows (the stand-alone
    prolongation kernel disappears) - same cycle, float64 and float32 vectors below level 0."""
    mv = make_case("texture", (270, 530), 2, 5)
    rng = np.random.default_rng(5)
    for vp in (0, 3):
        p = native.default_params(speed_alpha=1.0, remodelling_alpha=1e4, vcycle_precision=vp)
        res = []
        for fold in ("0", "1"):
            monkeypatch.setenv("VOF_FOLD_STORED", fold)
            with native.Solver(270, 530, 2) as s:
                s.debug_setup(mv, p)
                if not res:
                    r = rng.standard_normal((2, 3) + s.level_shape(0))
                res.append(s.debug_vcycle(r))
        assert relerr(res[1], res[0]) < 1e-12, vp


@pytest.mark.parametrize("shape", [(130, 258), (131, 257)])
def test_storage_modes_of_the_cycle_vectors_agree(native, shape):
    """vcycle_precision 0 / 1 / 2 / 3: the cycle is the same operator up to float32 rounding of the vectors it stores as
    float32 (mode 3: the levels below 0; odd row lengths fall back to float64 everywhere)."""
    mv = make_case("texture", shape, 2, 7)
    rng = np.random.default_rng(7)
    out = {}
    for vp in (0, 1, 2, 3):
        p = native.default_params(speed_alpha=1.0, remodelling_alpha=1e4, vcycle_precision=vp)
        with native.Solver(shape[0], shape[1], 2) as s:
            s.debug_setup(mv, p)
            if not out:
                r = rng.standard_normal((2, 3) + s.level_shape(0))
            out[vp] = s.debug_vcycle(r)
    for vp in (1, 2, 3):
        assert relerr(out[vp], out[0]) < 2e-5, vp


def test_vcycle_fused_equals_per_colour(native):
    mv = make_case("texture", (130, 130), 2, 3)
    p = native.default_params(speed_alpha=1.0, remodelling_alpha=1e4, vcycle_precision=0)
    with native.Solver(130, 130, 2) as s:
        s.debug_setup(mv, p)
        r = s.debug_rhs()
        e1 = s.debug_vcycle(r)
        s.set_fused_sweeps(False)
        e0 = s.debug_vcycle(r)
        assert relerr(e1, e0) < 1e-12


@pytest.mark.parametrize("kind,shape,npairs,alpha,beta,seed", SWEEP_CASES[1:4])
def test_float32_vcycle_vectors_building_blocks(native, kind, shape, npairs, alpha, beta, seed):
    """vcycle_precision=1: V-cycle vectors stored as float32, arithmetic FP64 -> float32-rounding agreement."""
    mv = make_case(kind, shape, npairs, seed)
    p = native.default_params(speed_alpha=alpha, remodelling_alpha=beta, vcycle_precision=1, coarse_precision=1,
                              nu_pre=2, nu_post=2, nu_pre_coarse=2, nu_post_coarse=2, w_cycle_level=-1)
    rng = np.random.default_rng(seed)
    with native.Solver(shape[0], shape[1], npairs) as s:
        s.debug_setup(mv, p)
        b = s.debug_rhs()
        x = rng.standard_normal(b.shape)
        # Krylov product: float32 input vector, FP64 operator and result
        y = s.debug_apply(0, x)
        x32 = x.astype(np.float32).astype(np.float64)
        for k in range(npairs):
            assert relerr(y[k], orc.apply_operator_interior(mv[k], x32[k], alpha, beta)) < 1e-13
        C = [mg.fine_stencil(mv[k], alpha, beta) for k in range(npairs)]
        for reverse in (False, True):
            xg = s.debug_sweep(0, x, b, reverse=reverse)
            xr = x.copy()
            for k in range(npairs):
                mg.smooth(C[k], xr[k], b[k], 1, reverse=reverse)
            assert relerr(xg, xr) < 5e-6
        H = [mg.Hierarchy(mv[k], alpha, beta) for k in range(npairs)]
        for lvl in range(s.num_levels - 1):
            f = rng.standard_normal((npairs, 3) + s.level_shape(lvl))
            cg = s.debug_restrict(lvl, f)
            for k in range(npairs):
                assert relerr(cg[k], mg.restrict(f[k])) < 1e-6
            e = rng.standard_normal(cg.shape)
            fg = s.debug_prolong_add(lvl, f, e)
            for k in range(npairs):
                assert relerr(fg[k], f[k] + mg.prolong(e[k], *f.shape[-2:])) < 1e-6
            if lvl >= 1:
                xs = rng.standard_normal(f.shape)
                bs = rng.standard_normal(f.shape)
                xg = s.debug_sweep(lvl, xs, bs)
                xr = xs.copy()
                for k in range(npairs):
                    mg.smooth(H[k].levels[lvl], xr[k], bs[k], 1)
                assert relerr(xg, xr) < 1e-4
        r = s.debug_rhs()
        e = s.debug_vcycle(r)
        for k in range(npairs):
            assert relerr(e[k], H[k].vcycle(r[k], 2, 2)) < 1e-3
        with pytest.raises(native.VofError, match="float64"):
            s.debug_gs(0, x, b, 0)


def test_w_cycle_matches_prototype(native):
    """One-level W-cycle (level 1 visits level 2 three times, each visit continuing from the previous result) against
    the numpy prototype of the same recursion."""
    shape, npairs, alpha, beta = (130, 130), 1, 1.0, 1e4
    mv = make_case("texture", shape, npairs, 3)
    p = native.default_params(speed_alpha=alpha, remodelling_alpha=beta, coarse_precision=0, vcycle_precision=0, nu_pre=2, nu_post=2,
                              nu_pre_coarse=1, nu_post_coarse=1, w_cycle_level=1, w_cycle_visits=3)
    H = mg.Hierarchy(mv[0], alpha, beta)

    def cyc(b, level, x0=None):
        C = H.levels[level]
        if level == len(H.levels) - 1:
            return (H.coarse_inv @ b.ravel()).reshape(b.shape)
        nu = 2 if level == 0 else 1
        x = np.zeros_like(b) if x0 is None else x0.copy()
        mg.smooth(C, x, b, nu)
        rc = mg.restrict(b - mg.apply_stencil(C, x))
        ec = cyc(rc, level + 1)
        if level == 1 and level + 1 < len(H.levels) - 1:
            for _ in range(2):
                ec = cyc(rc, level + 1, x0=ec)
        x += mg.prolong(ec, *x.shape[-2:])
        mg.smooth(C, x, b, nu, reverse=True)
        return x

    with native.Solver(shape[0], shape[1], npairs) as s:
        s.debug_setup(mv, p)
        r = s.debug_rhs()
        e = s.debug_vcycle(r)
        assert relerr(e[0], cyc(r[0], 0)) < 1e-9


@pytest.mark.parametrize("shape,npairs,sweeps", [((130, 258), 2, None), ((301, 250), 1, (2, 1, 1, 1)), ((66, 66), 2, (2, 3, 1, 1)),
                                                 ((70, 1030), 1, None), ((50, 83), 1, None)])
def test_krylov_product_fused_into_the_last_smoothing_pass(native, shape, npairs, sweeps):
    """The trailing stage of k_sweep0m: v = A y and the dot products (v, r), (v, v) out of the cycle's last pass must be
    what the separate operator kernel gives for the same y (odd row lengths take the separate kernel: fused is False)."""
    mv = make_case("texture", shape, npairs, 21)
    kw = dict(speed_alpha=1.0, remodelling_alpha=1e4)
    if sweeps:
        kw.update(nu_pre=sweeps[0], nu_post=sweeps[1], nu_pre_coarse=sweeps[2], nu_post_coarse=sweeps[3])
    p = native.default_params(**kw)
    rng = np.random.default_rng(3)
    with native.Solver(shape[0], shape[1], npairs) as s:
        s.debug_setup(mv, p)
        r = rng.standard_normal((npairs, 3) + s.level_shape(0))
        y, v, dots, fused = s.debug_vcycle_apply(r)
        assert fused == (s.level_shape(0)[1] % 2 == 0)
        np.testing.assert_array_equal(y, s.debug_vcycle(r))                 # the cycle itself is unchanged, bit for bit
        v_ref = s.debug_apply(0, y)
        assert relerr(v, v_ref) < 1e-12      # same products, another order of summation
        for k in range(npairs):
            assert dots[k, 0] == pytest.approx(float(np.vdot(v_ref[k], r[k])), rel=1e-11, abs=1e-9 * np.linalg.norm(v_ref[k]) * np.linalg.norm(r[k]))
            assert dots[k, 1] == pytest.approx(float(np.vdot(v_ref[k], v_ref[k])), rel=1e-12)


def test_debug_switches_guard_regions_poison_and_batch_time(native, monkeypatch):
    """Fault-attribution switches (include/vof.h): with guard regions around every device buffer, poisoned (0xFF) allocations and
    a synchronisation + error check after every launch scope, a solve gives the same answer as without them, no guard region is
    touched (vof_debug_check_canaries) and nothing reads uninitialised workspace (the poison is NaN: the result stays finite).
    Also: vof_pair_stats carries the batch time."""
    mv = make_case("texture", (66, 66), 3, 3)
    p = native.default_params(speed_alpha=1.0, remodelling_alpha=1e4, rtol=1e-9)
    with native.Solver(66, 66, 3) as s:
        ref = s.solve_host(mv, p)
    for k in ("VOF_DEBUG_SYNC", "VOF_DEBUG_CANARY", "VOF_DEBUG_POISON"):
        monkeypatch.setenv(k, "1")
    with native.Solver(66, 66, 3) as s:
        got = s.solve_host(mv, p)
        s.check_canaries()
        s.debug_setup(mv, p)                      # the debug entry points run under the same checks
        x = np.random.default_rng(0).standard_normal((3, 3, 64, 64))
        assert np.isfinite(s.debug_apply(0, x)).all()
        s.check_canaries()
    for a, b in zip(got[:4], ref[:4]):
        np.testing.assert_array_equal(a, b)
    st = got[4]
    assert st["converged"].all() and np.isfinite(st["relative_residual"]).all()
    assert (st["batch_pairs"] == 3).all() and (st["batch_ms"] > 0).all() and np.ptp(st["batch_ms"]) == 0


@pytest.mark.gpu
def test_vector_updates_folded_into_the_first_pass_of_the_cycle(native, monkeypatch):
    """The BiCGStab updates s = r - alpha v and p = r + beta (p - omega v) are formed inside the first pre-smoothing pass of the
    cycle that consumes them (k_sweep0r, BF = 1 / 2) by the operations of k_update_s / k_update_p (VOF_FUSE_B=0: those
    kernels).  Same iteration counts, the same solution up to the summation order of (s, s), fewer launches of the vector
    class - with and without the residual + restriction stage in the same pass."""
    mv = make_case("texture", (200, 264), 4, 5)
    p = native.default_params(speed_alpha=1.0, remodelling_alpha=1e4, rtol=1e-9)

    def run():
        with native.Solver(200, 264, 4) as s:
            s.profile_enable(True)
            out = s.solve_host(mv, p)
            return out, s.profile_get("vector", 0)[0], s.profile_get("gs0", 0)[0]

    for rr in ("1", "0"):
        monkeypatch.setenv("VOF_FUSE_RR", rr)
        monkeypatch.setenv("VOF_FUSE_B", "0")
        ref, nvec_ref, ngs_ref = run()
        monkeypatch.delenv("VOF_FUSE_B")
        got, nvec, ngs = run()
        assert ngs == ngs_ref and nvec < nvec_ref - 2
        assert (got[4]["iterations"] == ref[4]["iterations"]).all() and got[4]["converged"].all()
        for a, b in zip(got[:4], ref[:4]):
            np.testing.assert_allclose(a, b, rtol=1e-9, atol=1e-12)


@pytest.mark.parametrize("shape,npairs", [((66, 66), 2), ((140, 270), 1), ((12, 300), 1), ((300, 402), 2), ((258, 130), 1),
                                          ((67, 66), 1), ((141, 270), 2), ((263, 14), 1)])   # the last three: odd row counts
@pytest.mark.parametrize("vcycle_precision", [0, 3])
def test_coarse_rhs_from_the_pre_smoothing_pass(native, monkeypatch, shape, npairs, vcycle_precision):
    """Level 0: the coarse right-hand side R (b - A x) is the trailing stage of the pre-smoothing pass (k_sweep0r, TRAIL = 2)
    instead of a pass of k_stream_resrestrict0 over x, b and the image (VOF_FUSE_RR=0).  Same cycle up to the summation order
    of the restriction (float32 coarse right-hand sides: up to their rounding); several strips / bands, short and narrow grids."""
    mv = make_case("texture", shape, npairs, 11)
    p = native.default_params(speed_alpha=1.0, remodelling_alpha=1e4, vcycle_precision=vcycle_precision)

    def cycle():
        with native.Solver(shape[0], shape[1], npairs) as s:
            s.debug_setup(mv, p)
            s.profile_enable(True)
            e = s.debug_vcycle(s.debug_rhs())
            return e, s.profile_get("apply0", 0)[0]

    fused, n_apply = cycle()
    monkeypatch.setenv("VOF_FUSE_RR", "0")
    plain, n_apply_plain = cycle()
    assert n_apply == n_apply_plain - 1          # the stand-alone residual + restriction launch is gone
    assert relerr(fused, plain) < (1e-11 if vcycle_precision == 0 else 2e-6)
