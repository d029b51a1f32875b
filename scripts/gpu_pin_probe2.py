"""Second probe of the runtime's host-memory paths, shaped like what the GPU suite does before the call that failed:
  1. a large malloc/free raises glibc's dynamic mmap threshold (32 MiB cap), so that later buffers of a few MiB come from
     the brk heap: not page aligned, neighbours share pages, free() does not unmap them;
  2. three such neighbours are registered with hipHostRegister (as vof_solve_stack_host pins its output arrays in place),
     written by device-to-host copies, unregistered and freed;
  3. a new malloc of 1.6 MiB lands on the same heap bytes and is the destination of a plain pageable hipMemcpy (as in
     vof_debug_stencil).
Prints every HIP return code and where the buffers sit."""
import ctypes as C
import importlib.util
import os
import sys
import threading

from opticalflow_amd import _native

_native.load_library()
_spec = importlib.util.find_spec("torch")
_p = os.path.join(os.path.dirname(_spec.origin), "lib", "libamdhip64.so") if _spec and _spec.origin else "libamdhip64.so"
hip = C.CDLL(_p if os.path.exists(_p) else "libamdhip64.so", mode=C.RTLD_GLOBAL)
hip.hipGetErrorString.restype = C.c_char_p
libc = C.CDLL("libc.so.6")
libc.malloc.restype = C.c_void_p
libc.malloc.argtypes = [C.c_size_t]
libc.free.argtypes = [C.c_void_p]
libc.memset.argtypes = [C.c_void_p, C.c_int, C.c_size_t]


def chk(rc, what):
    print(f"{what}: rc={rc} ({hip.hipGetErrorString(rc).decode()})", flush=True)
    return rc


threads = len(sys.argv) > 1 and sys.argv[1] == "threads"
big = libc.malloc(40 << 20)
libc.memset(big, 1, 40 << 20)
libc.free(big)                                   # mmap threshold is now 32 MiB (dynamic threshold)
n_out = 59 * 72 * 65 * 8                         # the arrays of test_host_entry_point_optional_outputs_and_batch_schedule
dev = C.c_void_p()
chk(hip.hipMalloc(C.byref(dev), C.c_size_t(n_out)), "hipMalloc")
chk(hip.hipMemset(dev, 0x5A, C.c_size_t(n_out)), "hipMemset")
chk(hip.hipDeviceSynchronize(), "sync")
stream = C.c_void_p()
chk(hip.hipStreamCreate(C.byref(stream)), "hipStreamCreate")
for rnd in range(3):
    outs = [libc.malloc(n_out) for _ in range(3)]
    print("round", rnd, "outputs at", [hex(o) for o in outs], "page offsets", [o & 4095 for o in outs], flush=True)

    def reg(o):
        hip.hipSetDevice(0)
        chk(hip.hipHostRegister(C.c_void_p(o), C.c_size_t(n_out), 0), f"  hipHostRegister {o:#x}")
    if threads:
        ts = [threading.Thread(target=reg, args=(o,)) for o in outs]
        [t.start() for t in ts]
        [t.join() for t in ts]
    else:
        for o in outs:
            reg(o)
    for o in outs:
        chk(hip.hipMemcpyAsync(C.c_void_p(o), dev, C.c_size_t(n_out), 2, stream), "  D2H into registered")
    chk(hip.hipStreamSynchronize(stream), "  stream sync")
    for o in outs:
        chk(hip.hipHostUnregister(C.c_void_p(o)), f"  hipHostUnregister {o:#x}")
    for o in outs:
        libc.free(o)
    tmp = libc.malloc(1643328)
    libc.memset(tmp, 0, 1643328)
    print("  tmp at", hex(tmp), flush=True)
    rc = chk(hip.hipMemcpy(C.c_void_p(tmp), dev, C.c_size_t(1643328), 2), "  pageable hipMemcpy D2H into recycled heap bytes")
    b = (C.c_ubyte * 1643328).from_address(tmp)
    print("  data", "OK" if rc == 0 and b[0] == 0x5A and b[1643327] == 0x5A else "WRONG", flush=True)
    libc.free(tmp)
    if rc != 0:
        sys.exit(3)
print("no failure", flush=True)
