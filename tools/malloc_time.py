import ctypes as C, time
hip = C.CDLL("libamdhip64.so")
hip.hipMalloc.argtypes = [C.POINTER(C.c_void_p), C.c_size_t]
hip.hipFree.argtypes = [C.c_void_p]
hip.hipSetDevice(0)
p = C.c_void_p(); hip.hipMalloc(C.byref(p), 1 << 20); hip.hipFree(p)
for chunk in (8, 16, 32, 64, 100):
    ps = []
    t0 = time.time()
    n = int(160 // chunk)
    for i in range(n):
        p = C.c_void_p(); rc = hip.hipMalloc(C.byref(p), int(chunk * 1e9)); ps.append(p)
    t1 = time.time()
    for p in ps: hip.hipFree(p)
    print(f"{n} x {chunk} GB rc={rc}: {t1-t0:.3f} s", flush=True)
