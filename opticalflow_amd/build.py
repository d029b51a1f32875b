"""Build the native library (libvof.so) in-tree with hipcc for gfx950."""
import os
import subprocess
import sys

HERE = os.path.dirname(os.path.abspath(__file__))
CSRC = os.path.join(HERE, "csrc")
LIB = os.path.join(CSRC, "libvof.so")
SOURCES = ["vof.hip"]
# every source and header of csrc/ (a header missing from this list once left a stale library in the tree) + the public header
DEPS = sorted(f for f in os.listdir(CSRC) if f.endswith((".hip", ".hpp", ".h"))) + [os.path.join("..", "..", "include", "vof.h")]


def needs_build():
    if not os.path.exists(LIB):
        return True
    t = os.path.getmtime(LIB)
    return any(os.path.getmtime(os.path.join(CSRC, d)) > t for d in DEPS)


def build_native(force=False, verbose=True):
    """Compile every HIP source for gfx950 into opticalflow_amd/csrc/libvof.so."""
    if not force and not needs_build():
        return LIB
    hipcc = os.environ.get("HIPCC", "/opt/rocm/bin/hipcc")
    cmd = [hipcc, "-O3", "-std=c++17", "--offload-arch=gfx950", "-Wno-unused-value", "-shared", "-fPIC", "-pthread",
           "-o", LIB] + SOURCES
    if verbose:
        print("[opticalflow_amd] " + " ".join(cmd), file=sys.stderr)
    subprocess.run(cmd, cwd=CSRC, check=True)
    return LIB


if __name__ == "__main__":
    build_native(force="--force" in sys.argv)
