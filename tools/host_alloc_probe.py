"""How long does it take to get N bytes of host memory that a table fetch can fill at the link's rate?
pinned (crp_host_alloc = hipHostMalloc), fresh numpy pages touched by k threads, mmap(MAP_POPULATE)."""
import ctypes, json, mmap, sys, threading, time
import numpy as np
sys.path.insert(0, ".")
from cropsr_amd import Engine, _native as nat

def main():
    nbytes = int(float(sys.argv[1])) if len(sys.argv) > 1 else 700_000_000
    out = {"bytes": nbytes}
    with Engine(0) as eng:
        L = nat.lib()
        for rep in range(2):
            p = ctypes.c_void_p()
            t0 = time.perf_counter(); st = L.crp_host_alloc(nbytes, ctypes.byref(p)); t1 = time.perf_counter()
            assert st == 0
            L.crp_host_free(p); t2 = time.perf_counter()
            out["pinned_alloc_s_%d" % rep] = t1 - t0
            out["pinned_free_s_%d" % rep] = t2 - t1
        for threads in (1, 4, 8, 16):
            a = np.empty(nbytes, np.uint8)
            step = (nbytes + threads - 1) // threads
            def touch(k):
                a[k * step:(k + 1) * step:4096] = 0
            t0 = time.perf_counter()
            ts = [threading.Thread(target=touch, args=(k,)) for k in range(threads)]
            [t.start() for t in ts]; [t.join() for t in ts]
            out["numpy_touch_%d_threads_s" % threads] = time.perf_counter() - t0
            del a
        t0 = time.perf_counter()
        m = mmap.mmap(-1, nbytes, flags=mmap.MAP_PRIVATE | mmap.MAP_ANONYMOUS | getattr(mmap, "MAP_POPULATE", 0))
        out["mmap_populate_s"] = time.perf_counter() - t0
        m.close()
        try:
            out["thp"] = open("/sys/kernel/mm/transparent_hugepage/enabled").read().strip()
        except Exception as e:
            out["thp"] = str(e)
    print(json.dumps(out))

main()
