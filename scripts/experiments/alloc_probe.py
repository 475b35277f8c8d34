"""What does the first render of a context cost before any kernel runs?  hipMalloc of the path pool (14 queue arrays + the result ring)."""
import time, ctypes as C
hip = C.CDLL("libamdhip64.so")
hip.hipMalloc.argtypes = [C.POINTER(C.c_void_p), C.c_size_t]
hip.hipFree.argtypes = [C.c_void_p]
hip.hipDeviceSynchronize.argtypes = []
p = C.c_void_p()
hip.hipMalloc(C.byref(p), 1 << 20); hip.hipFree(p)
for gb in (0.25, 1, 3, 16):
    n = int(gb * (1 << 30))
    ts = []
    for rep in range(3):
        t = time.time(); rc = hip.hipMalloc(C.byref(p), n); hip.hipDeviceSynchronize(); t1 = time.time() - t
        t = time.time(); hip.hipFree(p); hip.hipDeviceSynchronize(); t2 = time.time() - t
        ts.append((t1, t2))
    print("hipMalloc %5.2f GB: %s ms (free %s ms)" % (gb, " ".join("%.1f" % (a * 1e3) for a, _ in ts), " ".join("%.1f" % (b * 1e3) for _, b in ts)), flush=True)
t = time.time()
ps = []
for k in range(14):
    q = C.c_void_p(); hip.hipMalloc(C.byref(q), int(3.05 * (1 << 30))); ps.append(q)
q = C.c_void_p(); hip.hipMalloc(C.byref(q), 16 << 30); ps.append(q)
hip.hipDeviceSynchronize()
print("14 x 3.05 GB + 16 GB: %.1f ms" % ((time.time() - t) * 1e3))
t = time.time()
for q in ps: hip.hipFree(q)
hip.hipDeviceSynchronize()
print("freeing them: %.1f ms" % ((time.time() - t) * 1e3))
# first touch: memset of a fresh allocation against a second memset of the same buffer
hip.hipMemset.argtypes = [C.c_void_p, C.c_int, C.c_size_t]
for gb in (4, 16, 48):
    n = int(gb * (1 << 30))
    q = C.c_void_p(); hip.hipMalloc(C.byref(q), n); hip.hipDeviceSynchronize()
    t = time.time(); hip.hipMemset(q, 0, n); hip.hipDeviceSynchronize(); t1 = time.time() - t
    t = time.time(); hip.hipMemset(q, 1, n); hip.hipDeviceSynchronize(); t2 = time.time() - t
    hip.hipFree(q); hip.hipDeviceSynchronize()
    q = C.c_void_p(); hip.hipMalloc(C.byref(q), n); hip.hipDeviceSynchronize()
    t = time.time(); hip.hipMemset(q, 0, n); hip.hipDeviceSynchronize(); t3 = time.time() - t
    hip.hipFree(q)
    print("%2d GB: first memset %.1f ms, second %.1f ms, first memset of a re-allocation %.1f ms" % (gb, t1 * 1e3, t2 * 1e3, t3 * 1e3), flush=True)
