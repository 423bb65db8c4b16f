import sys, time
sys.path.insert(0, '.')
import bench_workload as bw
from cropsr_amd import Engine
eng = Engine(0)
wl = bw.switchgrass_like()
strings = [wl.contig_string(k) for k in range(len(wl.specs))]
for rep in range(3):
    b = eng.arena_builder([s.size for s in strings])
    t_big = t_small = 0.0
    for s in strings:
        t0 = time.perf_counter()
        b.add(s)
        dt = time.perf_counter() - t0
        if s.size > 8 << 20: t_big += dt
        else: t_small += dt
    t0 = time.perf_counter()
    a = b.seal()
    t_seal = time.perf_counter() - t0
    n = a.scan_score_device(20)
    t0 = time.perf_counter(); cols = a.fetch(*n); t_f1 = time.perf_counter() - t0
    t0 = time.perf_counter(); cols = a.fetch(*n); t_f2 = time.perf_counter() - t0
    print("rep %d: big %.4f s (%d contigs, %.0f MB)  small %.4f s (%d contigs, %.0f MB)  seal %.4f  fetch %.4f / %.4f" % (
        rep, t_big, sum(s.size > 8 << 20 for s in strings), sum(s.size for s in strings if s.size > 8 << 20) / 1e6,
        t_small, sum(s.size <= 8 << 20 for s in strings), sum(s.size for s in strings if s.size <= 8 << 20) / 1e6, t_seal, t_f1, t_f2), flush=True)
    a.close()
