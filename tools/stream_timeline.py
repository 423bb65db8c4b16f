"""Three crp_scan_stream calls on the switchgrass-like genome (pinned tables unless --pageable), for a rocprofv3
--kernel-trace --memory-copy-trace run: the copy timeline of the LAST call is what tools/stream_timeline_report.py reads."""
import json, os, sys, time
sys.path.insert(0, ".")
import bench_workload as bw
from cropsr_amd import Engine

def main():
    pageable = "--pageable" in sys.argv
    wl = bw.switchgrass_like()
    strings = [wl.contig_string(k) for k in range(len(wl.specs))]
    with Engine(0) as eng:
        h = eng.scan_stream(strings, 20)
        out = None if pageable else eng.empty_tables(h.n_plus, h.n_minus)
        del h
        for rep in range(3):
            t0 = time.time_ns()
            h = eng.scan_stream(strings, 20, out=out)
            t1 = time.time_ns()
            st = h.stream_stats
            del h
        print(json.dumps({"last_call_ns": [t0, t1], "stats": st}))

main()
