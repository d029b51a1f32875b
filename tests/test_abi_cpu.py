"""CPU tests of the boundary: the C-ABI library loads and exports every symbol include/vof.h
declares, struct layouts agree, and the product path fails loudly without a GPU."""
import ctypes as C
import os
import re

import numpy as np
import pytest

from conftest import ROOT


@pytest.fixture(scope="module")
def lib():
    from opticalflow_amd import build, _native
    build.build_native(verbose=False)
    return _native.load_library()


def declared_symbols():
    text = open(os.path.join(ROOT, "include", "vof.h")).read()
    text = re.sub(r"/\*.*?\*/", "", text, flags=re.S)
    return sorted(set(re.findall(r"\b(vof_[a-z0-9_]+)\s*\(", text)))


def test_every_declared_symbol_is_exported(lib):
    from opticalflow_amd import _native
    syms = declared_symbols()
    assert len(syms) >= 25
    for s in syms:
        assert hasattr(lib, s), f"{s} declared in include/vof.h but not exported by libvof.so"
        assert s in _native.SIGNATURES, f"{s} has no ctypes prototype"
    assert set(_native.SIGNATURES) == set(syms)


def test_version_and_default_params(lib):
    from opticalflow_amd import _native
    assert lib.vof_version() == 202
    p = _native.default_params()
    # the reference's solver settings: OF.py:718-719, 1120
    assert (p.speed_alpha, p.remodelling_alpha, p.rtol, p.max_iterations) == (1.0, 1000.0, 1e-6, 1000)
    assert (p.nu_pre, p.nu_post, p.nu_pre_coarse, p.nu_post_coarse, p.w_cycle_level, p.w_cycle_visits) == (2, 2, 1, 1, 1, 3)
    assert (p.reference_quirks, p.coarse_precision, p.vcycle_precision) == (1, 3, 3)
    assert C.sizeof(_native.VofParams) == 2 * 4 + 8 * 8 + 16 * 4 == lib.vof_params_size()
    assert C.sizeof(_native.VofPairStats) == 56
    with pytest.raises(TypeError):
        _native.default_params(no_such_field=1)


def test_workspace_query(lib):
    from opticalflow_amd import _native
    one = _native.query_workspace(1024, 1024, 1)
    assert 2.5e8 < one < 3.4e8                   # 0.32 GB per 1024^2 pair with the default formats (round 2: 0.50)
    assert abs(_native.query_workspace(1024, 1024, 8) / one - 8) < 0.01
    assert _native.query_workspace(3, 3, 1) == 0  # invalid size
    # the stencil storage follows the format: 120 / 180 / 324 / 648 bytes per coarse point (1/3 of a fine point's share)
    by_fmt = [_native.query_workspace(1024, 1024, 1, f, 3) for f in (3, 2, 1, 0)]
    assert by_fmt[0] == one and by_fmt == sorted(by_fmt)
    third = 1022 * 1022 / 3.0
    for got, want in zip([b - by_fmt[0] for b in by_fmt[1:]], [60, 204, 528]):
        assert abs(got / third - want) < 0.05 * want
    # float32 cycle vectors on every level (vcycle_precision 1 / 2) keep one more level-0 vector
    assert _native.query_workspace(1024, 1024, 1, 3, 1) - one == pytest.approx(3 * 1022 * 1022 * 8, rel=1e-6)


def test_kernel_names(lib):
    from opticalflow_amd import _native
    for i, n in enumerate(_native.K_NAMES):
        assert lib.vof_kernel_name(i).decode() == n


def test_no_cpu_fallback_without_gpu():
    """Without a HIP device the product path must raise, never silently compute on the CPU."""
    import torch
    if torch.cuda.is_available():
        pytest.skip("GPU present")
    from opticalflow_amd import optical_flow, _native
    movie = np.random.default_rng(0).random((2, 16, 16))
    with pytest.raises(_native.VofError):
        optical_flow.variational_optical_flow(movie)


def test_missing_library_fails_loudly(tmp_path):
    from opticalflow_amd import _native
    with pytest.raises(_native.VofError, match="no CPU fallback"):
        _native.load_library(str(tmp_path / "libvof_missing.so"))


def test_product_path_never_imports_oracle():
    """The oracle is test infrastructure: nothing under opticalflow_amd/ or source/ may reference it."""
    for base in ("opticalflow_amd", "source"):
        for dirpath, _d, files in os.walk(os.path.join(ROOT, base)):
            for f in files:
                if f.endswith((".py", ".hip", ".hpp", ".h")):
                    txt = open(os.path.join(dirpath, f)).read()
                    assert "oracle" not in txt.replace("no oracle", ""), os.path.join(dirpath, f)


def test_host_helpers_match_reference_semantics():
    from opticalflow_amd import optical_flow as of
    from conftest import load_golden
    g = load_golden("g8_fake_frame.npz")
    fr, dx = of.make_fake_data_frame(1.3, 2.9, sigma=1.7, width=6.0, dimension=37)
    np.testing.assert_allclose(fr, g["frame"], rtol=4e-15)
    assert dx == float(g["delta_x"])
    # the Gaussian taps handed to the device blur are scipy's
    import scipy.ndimage
    imp = np.zeros(41); imp[20] = 1.0
    for sigma in (0.7, 2.0, 2.48):
        taps = of.gaussian_taps(sigma)
        r = taps.size // 2
        np.testing.assert_allclose(taps, scipy.ndimage.gaussian_filter1d(imp, sigma, mode="constant")[20 - r:21 + r],
                                   rtol=0, atol=1e-17)
    assert of.format_elapsed_time(125.25) == (2, 5, 250)
    a = np.arange(25.0).reshape(5, 5)
    of.apply_constant_boundary_condition(a)
    assert a[0, 0] == a[2, 2] and a[-1, -1] == a[2, 2] and a[0, 3] == a[2, 3]


def test_reference_binding_matches_the_library_and_the_document(lib):
    """examples/reference_binding.py is the binding INTEGRATION.md prints: same text, same struct layout as the library
    and as opticalflow_amd/_native.py (field names, order and types), and it refuses another ABI."""
    import importlib.util
    from opticalflow_amd import _native
    path = os.path.join(ROOT, "examples", "reference_binding.py")
    text = open(path).read()
    assert text in open(os.path.join(ROOT, "INTEGRATION.md")).read(), "INTEGRATION.md does not print the binding file verbatim"
    spec = importlib.util.spec_from_file_location("reference_binding", path)
    rb = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(rb)
    assert [(n, t) for n, t in rb.vof_params._fields_] == [(n, t) for n, t in _native.VofParams._fields_]
    assert [(n, t) for n, t in rb.vof_pair_stats._fields_] == [(n, t) for n, t in _native.VofPairStats._fields_]
    blib = rb.load(_native.LIB_PATH)
    assert C.sizeof(rb.vof_params) == blib.vof_params_size() == lib.vof_params_size()
    assert rb.VOF_VERSION == lib.vof_version()
    # header <-> binding: every field of struct vof_params in include/vof.h, in order
    hdr = open(os.path.join(ROOT, "include", "vof.h")).read()
    body = re.search(r"typedef struct vof_params \{(.*?)\} vof_params;", hdr, flags=re.S).group(1)
    body = re.sub(r"/\*.*?\*/", "", body, flags=re.S)
    names = re.findall(r"\b(?:double|int32_t|uint32_t)\s+([a-z_0-9]+)\s*;", body)
    assert names == [n for n, _ in rb.vof_params._fields_]


def test_abi_guard_rejects_a_stale_struct(lib):
    """A binding built against a shorter (older) vof_params: vof_default_params writes nothing and says so."""
    class OldParams(C.Structure):      # the round-1 layout INTEGRATION.md once printed (104 bytes)
        _fields_ = [(f"d{i}", C.c_double) for i in range(8)] + [(f"i{i}", C.c_int32) for i in range(10)]
    guard = (C.c_char * 64)()
    C.memset(guard, 0x5A, 64)
    old = OldParams()
    fn = lib.vof_default_params
    rc = C.CFUNCTYPE(C.c_int, C.c_void_p, C.c_size_t)(C.cast(fn, C.c_void_p).value)(C.addressof(old), C.sizeof(old))
    assert rc != 0
    assert bytes(old) == b"\0" * C.sizeof(old)          # untouched
    assert bytes(guard) == b"\x5a" * 64


def test_every_environment_switch_the_library_reads_is_documented_in_the_header():
    """include/vof.h lists the experiment / debug switches; round 2's review found five the library read and the header did not
    name.  Every getenv("VOF_...") of the native sources must appear in the header."""
    import re
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    header = open(os.path.join(root, "include", "vof.h")).read()
    read = set()
    csrc = os.path.join(root, "opticalflow_amd", "csrc")
    for name in os.listdir(csrc):
        if name.endswith((".hip", ".hpp")):
            read |= set(re.findall(r'getenv\("(VOF_[A-Z0-9_]+)"\)', open(os.path.join(csrc, name)).read()))
    assert read, "no switches found: the pattern is stale"
    missing = sorted(v for v in read if v not in header)
    assert not missing, missing
