#!/usr/bin/env python3
"""Calibrate bench.py's CPU baseline against the REAL reference (development container only).

BASELINE.md section 3 / SURVEY.md 8d: the in-run CPU baseline on the GPU box is the build's
reference-faithful numpy port (oracle/faithful_port.py, "kind": "port"), which must first land within
+-20 % of the real reference's speed where the real reference can run.  This script runs both, in child
processes, on BASELINE.md section 2's inputs -- a 2 Mb and a 9 Mb seeded iid contig -- with one BLAS
thread, records wall / user / sys seconds and page faults of each, and writes
profiles/cpu_calibration.json; it exits non-zero if a ratio leaves the band.

The reference (unmodified, /root/reference/CROPSR.py) is timed by its own timer (CROPSR.py:334,476-477:
whole run without its 5 s sleep) -- that includes its id draws, row tuples and csv writing, so the port is
run in its whole-run mode (faithful_port.full_run) for the comparison, and once more in the hot-path-only
mode that bench.py times.  Both programs are page-fault bound here (12 KB of fresh float64 temporaries
per gRNA): the kernel's share (sys) is 60-75 % of the wall time and drifts by 2x between identical runs
of the same program, so the band is asserted on user-CPU seconds (the algorithmic work) and the wall
ratio is recorded beside it.

    python tools/calibrate_cpu_baseline.py [--sizes 2000000,9000000]
"""
import argparse
import json
import os
import subprocess
import sys
import tempfile
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
REF = "/root/reference"

REF_CHILD = r"""
import json, resource, sys, time
fa, gff, out = sys.argv[1:4]
sys.argv = ['CROPSR.py', '-f', fa, '-g', gff, '-o', out, '--cas9']
sys.path.insert(0, %r)
time.sleep = lambda s: None
import numpy as np
import CROPSR
np.random.seed(1)
ru0 = resource.getrusage(resource.RUSAGE_SELF)   # after the imports (pandas alone costs ~1 s of CPU)
t0 = time.time()
CROPSR.main()
wall = time.time() - t0
ru = resource.getrusage(resource.RUSAGE_SELF)
own = open('time.txt').read().replace('Total runtime of the program is ', '')
rows = sum(1 for _ in open(out)) - 1
print(json.dumps(dict(wall_s=wall, own_timer_s=float(own), user_s=ru.ru_utime - ru0.ru_utime, sys_s=ru.ru_stime - ru0.ru_stime,
                      minor_faults=ru.ru_minflt - ru0.ru_minflt, max_rss_mb=ru.ru_maxrss / 1024.0, rows=rows)))
""" % REF

PORT_CHILD = r"""
import json, resource, sys, time
sys.path.insert(0, %r)
fa = sys.argv[1]
from oracle import faithful_port as fp
text = open(fa).read().split('\n', 1)[1].replace('\n', '')
s = "'" + text + "')]"          # the string the reference scans for a one-contig FASTA (SURVEY.md A.1)
mode = sys.argv[2]
ru0 = resource.getrusage(resource.RUSAGE_SELF)
t0 = time.time()
if mode == "full":      # hot path + ids + row tuples + csv, like the reference's whole run
    n_rows = fp.full_run(s, sys.argv[3])
else:                   # hot path only: what bench.py's cpu_baseline leg times
    rows, scores = fp.scan_score(s)
    n_rows = len(rows)
wall = time.time() - t0
ru = resource.getrusage(resource.RUSAGE_SELF)
print(json.dumps(dict(wall_s=wall, user_s=ru.ru_utime - ru0.ru_utime, sys_s=ru.ru_stime - ru0.ru_stime,
                      minor_faults=ru.ru_minflt - ru0.ru_minflt, max_rss_mb=ru.ru_maxrss / 1024.0, rows=n_rows)))
""" % ROOT


def make_fasta(path, n, seed):
    import numpy as np
    rng = np.random.default_rng(seed)
    a = np.frombuffer(b"ACGT", dtype=np.uint8)[rng.integers(0, 4, n)]
    with open(path, "wb") as f:
        f.write(b">chrCal\n")
        for i in range(0, n, 80):
            f.write(a[i:i + 80].tobytes() + b"\n")


def run(code, args, cwd):
    env = dict(os.environ, OPENBLAS_NUM_THREADS="1")
    p = subprocess.run([sys.executable, "-c", code] + args, cwd=cwd, env=env, capture_output=True, text=True)
    if p.returncode != 0:
        raise SystemExit(p.stderr[-3000:])
    return json.loads([l for l in p.stdout.splitlines() if l.startswith("{")][-1])


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--sizes", default="2000000,9000000")
    ap.add_argument("--reps", type=int, default=2)
    ap.add_argument("--out", default=os.path.join(ROOT, "profiles", "cpu_calibration.json"))
    a = ap.parse_args()
    if not os.path.isdir(REF):
        raise SystemExit("the real reference is only present in the development container")
    result = {"host": {"cpus": os.cpu_count(), "note": "development container (no GPU); one BLAS thread"},
              "band": "user-CPU seconds of the port's whole run (hot path + ids + rows + csv, like the reference) within "
                      "[0.8, 1.2] x the reference's; best of the repetitions each.  Wall time is reported too but not "
                      "asserted: in this VM both programs spend 60-75 % of their wall time in the kernel's page-fault "
                      "path (sys), and that share moves by 2x between identical runs", "cases": []}
    ok = True
    with tempfile.TemporaryDirectory() as d:
        gff = os.path.join(d, "e.gff")
        open(gff, "w").write("##gff-version 3\n")
        for n in [int(x) for x in a.sizes.split(",")]:
            fa = os.path.join(d, "cal_%d.fa" % n)
            make_fasta(fa, n, 12345)
            t0 = time.time()
            # alternate the two programs (the container's page-fault cost drifts with what ran before)
            refs, fulls = [], []
            for rep in range(a.reps):
                refs.append(run(REF_CHILD, [fa, gff, os.path.join(d, "ref_%d_%d.csv" % (n, rep))], d))
                fulls.append(run(PORT_CHILD, [fa, "full", os.path.join(d, "port_%d_%d.csv" % (n, rep))], d))
            hot = run(PORT_CHILD, [fa, "hot"], d)
            best = lambda runs, key: min(r[key] for r in runs)
            hits = hot["rows"]
            ref_rate = hits / best(refs, "own_timer_s")
            full_rate = hits / best(fulls, "wall_s")
            ratio = full_rate / ref_rate
            user_ratio = best(refs, "user_s") / best(fulls, "user_s")
            case = {"bases": n, "kept_hits": hits, "reference_runs": refs, "port_whole_run": fulls, "port_hot_path_only": hot,
                    "reference_gRNAs_per_s": ref_rate, "port_whole_run_gRNAs_per_s": full_rate,
                    "port_hot_path_gRNAs_per_s": hits / hot["wall_s"],
                    "port_over_reference_wall": ratio, "port_over_reference_user_cpu": user_ratio}
            result["cases"].append(case)
            ok = ok and 0.8 <= user_ratio <= 1.2
            print("%d bases, %d hits: reference %.0f gRNAs/s (best of %d: %.1f s; user %.1f, sys %.1f) | port whole run %.0f gRNAs/s "
                  "(%.1f s; user %.1f, sys %.1f) wall ratio %.2f, user-CPU ratio %.2f | port hot path only %.0f gRNAs/s  [%.0f s]"
                  % (n, hits, ref_rate, a.reps, best(refs, "own_timer_s"), best(refs, "user_s"), best(refs, "sys_s"), full_rate,
                     best(fulls, "wall_s"), best(fulls, "user_s"), best(fulls, "sys_s"), ratio, user_ratio,
                     hits / hot["wall_s"], time.time() - t0), flush=True)
    result["within_band"] = ok
    with open(a.out, "w") as f:
        json.dump(result, f, indent=1)
        f.write("\n")
    sys.exit(0 if ok else 1)


if __name__ == "__main__":
    main()
