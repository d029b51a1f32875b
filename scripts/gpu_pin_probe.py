"""Probe of the HIP runtime's pageable-copy path (no kernel of this repository involved).

hipMemcpy(D2H) into PAGEABLE host memory larger than 1 MiB pins the destination range on the fly.  Question: is such a
pinning reused for a later copy into a NEW host mapping that happens to sit at the same address after the first one was
unmapped (numpy / std::vector buffers above glibc's mmap threshold are unmapped on free)?  Sequence:
  A = mmap(size); hipMemcpy(A <- dev); munmap(A); wait; B = mmap(size) [same address]; hipMemcpy(B <- dev).
Prints the HIP error code of every step; run under `timeout`."""
import ctypes as C
import mmap
import sys
import time

from opticalflow_amd import _native

_native.load_library()          # binds the HIP runtime the way the library does (the copy torch bundles, if torch is installed)
import importlib.util
import os
_spec = importlib.util.find_spec("torch")
_p = os.path.join(os.path.dirname(_spec.origin), "lib", "libamdhip64.so") if _spec and _spec.origin else "libamdhip64.so"
hip = C.CDLL(_p if os.path.exists(_p) else "libamdhip64.so", mode=C.RTLD_GLOBAL)  # already resident: same handle
hip.hipGetErrorString.restype = C.c_char_p


def chk(rc, what):
    print(f"{what}: rc={rc} ({hip.hipGetErrorString(rc).decode()})", flush=True)
    return rc


def addr_of(m):
    return C.addressof(C.c_char.from_buffer(m))


size = int(sys.argv[1]) if len(sys.argv) > 1 else 1643328
wait = float(sys.argv[2]) if len(sys.argv) > 2 else 0.5
rounds = int(sys.argv[3]) if len(sys.argv) > 3 else 3
dev = C.c_void_p()
chk(hip.hipMalloc(C.byref(dev), C.c_size_t(size)), "hipMalloc")
chk(hip.hipMemset(dev, 0x5A, C.c_size_t(size)), "hipMemset")
chk(hip.hipDeviceSynchronize(), "sync")
seen = set()
for r in range(rounds):
    m = mmap.mmap(-1, size)
    m[:] = b"\0" * size                      # touch every page
    a = addr_of(m)
    again = a in seen
    seen.add(a)
    print(f"round {r}: host mapping at {a:#x}{' (address seen before)' if again else ''}", flush=True)
    rc = chk(hip.hipMemcpy(C.c_void_p(a), dev, C.c_size_t(size), 2), "  hipMemcpy D2H pageable")
    ok = rc == 0 and m[0] == 0x5A and m[size - 1] == 0x5A and m[size // 2] == 0x5A
    print(f"  data {'OK' if ok else 'WRONG'}", flush=True)
    if rc != 0:
        sys.exit(3)
    m.close()                                 # munmap
    time.sleep(wait)                          # the driver's user-pointer worker runs while the range is unmapped
print("no failure", flush=True)
