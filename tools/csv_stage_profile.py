#!/usr/bin/env python3
"""Where the CLI's format + write stage spent its time BEFORE round 6's crp_write_segments (GPU box): the per-contig loop of cropsr_amd/cli.py run(), one formatter call per 1 M-row chunk of a pass, CROPSR.py:409-474
with --each-contig-once, on the bench genome, with a timer around every step of a pass.

usage: python tools/csv_stage_profile.py [switchgrass|sorghum|tair10] [--out /tmp/x.csv] [--threads N]
Prints one JSON line: seconds per step summed over the passes, split into the passes of large (>= 1e5 rows) and small contigs.
"""
import argparse
import json
import os
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)

import numpy as np  # noqa: E402


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("workload", nargs="?", default="switchgrass")
    ap.add_argument("--out", default="/tmp/csv_stage_profile.csv")
    ap.add_argument("--threads", type=int, default=0)
    ap.add_argument("--ids-ahead", action="store_true", help="draw every pass's ids before the loop (takes the id stream out of the picture)")
    args = ap.parse_args()
    import bench_workload as bw
    from cropsr_amd import cli, rows
    wl = {"switchgrass": bw.switchgrass_like, "sorghum": bw.sorghum_like, "tair10": bw.tair10_like}[args.workload]()
    strings = [wl.contig_string(k) for k in range(len(wl.specs))]
    names = [">c%05d" % k for k in range(len(strings))]
    backend = cli.EngineBackend(0)
    all_hits = backend.scan(strings, 20)
    np.random.seed(7)
    rows.write_header(args.out, offtarget=False)
    per_contig = [int(h["pos_plus"].size + h["pos_minus"].size) for h in all_hits]
    t_ids0 = time.perf_counter()
    if args.ids_ahead:
        drawn = [rows.draw_ids(n, reverse=True) for n in per_contig]
        ids = None
    else:
        ids = rows.IdStream(per_contig, reverse=True)
    t_ids0 = time.perf_counter() - t_ids0
    acc = {"large": {}, "small": {}}

    def add(kind, key, dt):
        acc[kind][key] = acc[kind].get(key, 0.0) + dt

    sink = open(os.devnull, "w")
    t_all = time.perf_counter()
    for k, (name, s, hits) in enumerate(zip(names, strings, all_hits)):
        kind = "large" if per_contig[k] >= 100_000 else "small"
        t = time.perf_counter()
        print("Searching on Chromosome: ", name[:25], file=sink)
        print("With start of sequence: ", bytes(s[:25]).decode("latin-1"), file=sink)
        add(kind, "prints", time.perf_counter() - t)
        t = time.perf_counter()
        block = rows.ContigTable(name, s, hits, 20)
        add(kind, "contig_table", time.perf_counter() - t)
        t = time.perf_counter()
        dataset = rows.NativeDataset(args.threads or None)
        dataset.append(block)
        size = len(dataset)
        add(kind, "dataset", time.perf_counter() - t)
        t = time.perf_counter()
        ids_rev = drawn[k] if ids is None else ids.next(size)
        add(kind, "ids_wait", time.perf_counter() - t)
        t = time.perf_counter()
        with open(args.out, "ab") as f:
            fd = f.fileno()
            add(kind, "open", time.perf_counter() - t)
            for index_range, count in rows.flush_plan(size):
                t = time.perf_counter()
                dataset.chunk_to_fd(fd, index_range, count, None, index_range, backend.rescore, ids_rev=ids_rev)
                add(kind, "chunk_to_fd", time.perf_counter() - t)
            t = time.perf_counter()
        add(kind, "close", time.perf_counter() - t)
        add(kind, "passes", 1)
        add(kind, "rows", size)
    t_all = time.perf_counter() - t_all
    if ids is not None:
        ids.close()
    n_bytes = os.path.getsize(args.out)
    os.unlink(args.out)
    backend.close()
    print(json.dumps({"workload": args.workload, "csv_bytes": n_bytes, "loop_s": round(t_all, 4), "GB_per_s": round(n_bytes / t_all / 1e9, 2),
                      "ids_before_loop_s": round(t_ids0, 4), "threads": dataset.n_threads,
                      **{kind: {k: round(v, 4) for k, v in d.items()} for kind, d in acc.items()}}))


if __name__ == "__main__":
    main()
