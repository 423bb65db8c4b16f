"""crp_scan_stream on the switchgrass-like genome under different slice sizes / lane counts / copy threads (GPU box)."""
import json, os, sys, time
import numpy as np
sys.path.insert(0, ".")
import bench_workload as bw

def main():
    wl = bw.switchgrass_like()
    strings = [wl.contig_string(k) for k in range(len(wl.specs))]
    rows = []
    for threads in (8, 12, 16):
        os.environ["CRP_COPY_THREADS"] = str(threads)
        from cropsr_amd import Engine
        with Engine(0) as eng:
            out_pinned = None
            for lanes in (3, 4, 6):
                os.environ["CRP_STREAM_LANES"] = str(lanes)
                for slice_mi in (32, 64, 128):
                    for pinned in (False, True):
                        if pinned and out_pinned is None:
                            h = eng.scan_stream(strings, 20, slice_chars=slice_mi << 20)
                            out_pinned = eng.empty_tables(h.n_plus, h.n_minus)
                            del h
                        walls, st = [], None
                        for rep in range(5):
                            h = eng.scan_stream(strings, 20, slice_chars=slice_mi << 20, out=out_pinned if pinned else None)
                            if rep:
                                walls.append(h.stream_stats["wall_s"] * 1e3)
                                st = h.stream_stats
                            del h
                        rows.append(dict(threads=threads, lanes=lanes, slice_Mi=slice_mi, pinned=pinned, best_ms=min(walls), median_ms=sorted(walls)[len(walls) // 2],
                                         up_busy_ms=st["uploader_busy_s"] * 1e3, up_wait_ms=st["uploader_waiting_s"] * 1e3,
                                         copier_busy_ms=st["copier_busy_s"] * 1e3, copier_wait_ms=st["copier_waiting_s"] * 1e3))
                        print(json.dumps(rows[-1]), flush=True)
    for pinned in (False, True):
        best = sorted([r for r in rows if r["pinned"] == pinned], key=lambda r: r["median_ms"])[:4]
        print("best pinned=%s:" % pinned, json.dumps(best))

main()
