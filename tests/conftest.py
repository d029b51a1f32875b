import os
import sys

import numpy as np
import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)
GOLDEN = os.path.join(ROOT, "tests", "golden")


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu on the GPU box)")
    # Fault diagnosis (VOF_TEST_STDERR_FILE=path): a GPU memory fault ends the process inside the HIP runtime, and the runtime's
    # own message (faulting address) goes to file descriptor 2, which pytest has pointed at a capture file that dies with the
    # process.  With the variable set, fd 2 is pointed at a file that survives; the current test id is appended per test.
    path = os.environ.get("VOF_TEST_STDERR_FILE")
    if path:
        fd = os.open(path, os.O_WRONLY | os.O_CREAT | os.O_APPEND, 0o644)
        os.dup2(fd, 2)
        os.close(fd)


def pytest_runtest_logstart(nodeid, location):
    if os.environ.get("VOF_TEST_STDERR_FILE"):
        os.write(2, f"[test] {nodeid}\n".encode())


def load_golden(name):
    """Golden fixtures are plain numeric .npz files (no pickles)."""
    with np.load(os.path.join(GOLDEN, name), allow_pickle=False) as z:
        return {k: z[k] for k in z.files}


def golden_kwargs(g):
    kw = {k[3:]: float(v) for k, v in g.items() if k.startswith("kw_")}
    return kw


@pytest.fixture(scope="session")
def golden():
    return load_golden


@pytest.fixture(scope="session", autouse=True)
def _native_library_is_built():
    """The in-tree libvof.so normally travels with the tree; if it is missing or stale, (re)build it with hipcc (the GPU
    box has the same toolchain).  A failed build is not hidden: the tests that need the library fail on load."""
    try:
        from opticalflow_amd import build
        build.build_native(verbose=False)
    except Exception as exc:      # noqa: BLE001 - reported, the loader raises for the tests that need it
        print(f"[conftest] could not build libvof.so: {exc}", file=sys.stderr)
    yield
