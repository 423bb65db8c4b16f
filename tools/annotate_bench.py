"""The annotation join's kernels on a resident genome, for profiling: python tools/annotate_bench.py {tair10|sorghum} N_GENES
Uploads the synthetic genome, scans it once, builds the synthetic GFF's track and runs crp_annotate_lookup 20 times (ids stay
in HBM).  Prints one JSON line: hits, track points, kernel ms per look-up (HIP events in the library), algorithmic bytes."""
import json
import os
import sys
import tempfile
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)


def main():
    import bench_workload as bw
    from cropsr_amd import Engine, annotate
    name, n_genes = sys.argv[1], int(sys.argv[2])
    wl = {"tair10": bw.tair10_like, "sorghum": bw.sorghum_like}[name]()
    tmp = tempfile.mkdtemp()
    gff, info = os.path.join(tmp, "g.gff3"), os.path.join(tmp, "info.txt")
    bw.synthetic_annotation(wl, gff, info, n_genes=n_genes)
    t0 = time.perf_counter()
    ann = annotate.Annotation(gff, info)
    t_build = time.perf_counter() - t0
    eng = Engine(0)
    b = eng.arena_builder([s.length + 4 for s in wl.specs])
    for k in range(len(wl.specs)):
        b.add(wl.contig_string(k))
    arena = b.seal()
    n_plus, n_minus = arena.scan_score_device(20)
    req = annotate.Request(ann, [s.name for s in wl.specs], 1)
    t0 = time.perf_counter()
    points, ids = req.track([(k, int(arena.offsets[k]), int(arena.lengths[k])) for k in range(len(wl.specs))])
    arena.annotate_set_track(points, ids)
    t_track = time.perf_counter() - t0
    for _ in range(5):
        arena.annotate_lookup(n_plus, n_minus, fetch=False)
    eng.profile(2)
    eng.profile_read(reset=True)
    for _ in range(20):
        arena.annotate_lookup(n_plus, n_minus, fetch=False)
    p = eng.profile_read()["annotate"]
    hits = n_plus + n_minus
    ms = p["ms"] / max(1, p["launches"])
    algo = 16 * hits  # 4 B position + 8 B score in, 4 B id out per hit
    print(json.dumps({"workload": wl.name, "kept_hits": hits, "track_points": int(points.size), "strings": len(ann.strings),
                      "host_build_s": t_build, "track_layout_upload_s": t_track, "lookup_ms": ms,
                      "algorithmic_bytes": algo, "GBs": algo / (ms * 1e-3) / 1e9, "roofline_frac_of_8TBs": algo / (ms * 1e-3) / 8e12}))
    arena.close()
    eng.close()


if __name__ == "__main__":
    main()
