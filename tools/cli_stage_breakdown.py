"""What the CLI's "upload + scan + fetch" stage is made of on the bench genome: library load, Engine() (HIP start-up,
pinned staging buffers), upload, scan + fetch, slicing per contig.  GPU box, repo root: python tools/cli_stage_breakdown.py"""
import sys, time
sys.path.insert(0, '.')
t0=time.perf_counter()
import numpy as np
import bench_workload as bw
t1=time.perf_counter()
from cropsr_amd import Engine
from cropsr_amd import _native as nat
nat.lib()
t2=time.perf_counter()
eng = Engine(0)
t3=time.perf_counter()
print("import numpy+workload %.3f  load lib %.3f  Engine(0) %.3f" % (t1-t0, t2-t1, t3-t2))
wl = bw.switchgrass_like()
strings = [wl.contig_string(k) for k in range(len(wl.specs))]
for rep in range(2):
    a=time.perf_counter()
    g = eng.genome(strings)
    b=time.perf_counter()
    hits = g.scan_score(20, want_pre=False)
    c=time.perf_counter()
    out = [hits.contig(k) for k in range(len(strings))]
    d=time.perf_counter()
    g.close()
    e=time.perf_counter()
    print("rep %d: genome(upload) %.3f  scan_score(+fetch) %.3f  slicing %.3f  close %.3f" % (rep, b-a, c-b, d-c, e-d))
