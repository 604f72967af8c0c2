"""GPU experiment: device -> PAGEABLE host copy time by size (hipMemcpy into a numpy array), and -> pinned."""
import ctypes as C, time, numpy as np
hip = C.CDLL("libamdhip64.so")
hip.hipMalloc.argtypes = [C.POINTER(C.c_void_p), C.c_size_t]
hip.hipMemcpy.argtypes = [C.c_void_p, C.c_void_p, C.c_size_t, C.c_int]
hip.hipHostMalloc.argtypes = [C.POINTER(C.c_void_p), C.c_size_t, C.c_uint]
d = C.c_void_p(); assert hip.hipMalloc(C.byref(d), 256 << 20) == 0
p = C.c_void_p(); assert hip.hipHostMalloc(C.byref(p), 256 << 20, 0) == 0
host = np.zeros(256 << 20, np.uint8); host[:] = 1
for mb in (0.25, 1, 2, 4, 8, 12, 16, 24, 32, 64, 128):
    n = int(mb * (1 << 20))
    for name, dst in (("pageable", host.ctypes.data), ("pinned", p.value)):
        hip.hipMemcpy(dst, d, n, 2)
        t0 = time.perf_counter()
        for _ in range(10): hip.hipMemcpy(dst, d, n, 2)
        dt = (time.perf_counter() - t0) / 10
        print("%8.2f MB -> %-8s %8.1f us  %6.1f GB/s" % (mb, name, dt * 1e6, n / dt / 1e9), flush=True)
