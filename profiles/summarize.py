#!/usr/bin/env python3
"""Summarise the rocprofv3 passes of profiles/collect.sh into small files:

  pmc_summary.csv   per kernel, per counter: mean value per launch
  traffic.json      HBM bytes per launch of the dominant kernel, corrected as
                    MI355X_MICROARCH.md prescribes for gfx950:
                      FETCH_SIZE is reported in KiB and counts exactly 1/2 of a wide
                      coalesced (16 B/lane) streaming read  -> bytes = FETCH_SIZE*1024*2
                      WRITE_SIZE is reported in KiB and reads exactly -> bytes = WRITE_SIZE*1024
                    (the emit kernel's reads are 16-byte-per-lane streaming loads of
                    the planes; its writes are 4- and 8-byte-per-lane coalesced
                    stores, for which WRITE_SIZE is not separately calibrated)
"""
import collections
import csv
import glob
import json
import os
import sys


def load(raw, tag):
    files = glob.glob(os.path.join(raw, tag, "*", "*_counter_collection.csv"))
    agg = collections.defaultdict(lambda: collections.defaultdict(list))
    for f in files:
        for r in csv.DictReader(open(f)):
            name = r["Kernel_Name"]
            if "crp::" not in name:
                continue
            short = name.split("(")[0].replace("void ", "")
            agg[short][r["Counter_Name"]].append(float(r["Counter_Value"]))
    return agg


def main():
    raw, out = sys.argv[1], sys.argv[2]
    round_tag = sys.argv[3] if len(sys.argv) > 3 else "r03"
    rows = []
    merged = collections.defaultdict(dict)
    for tag in ("fetch", "write", "sq1", "sq2"):
        for k, counters in load(raw, tag).items():
            for c, vals in counters.items():
                merged[k][c] = (sum(vals) / len(vals), len(vals))
    with open(os.path.join(out, "pmc_summary.csv"), "w") as f:
        w = csv.writer(f)
        w.writerow(["kernel", "counter", "mean_per_launch", "launches"])
        for k in sorted(merged):
            for c in sorted(merged[k]):
                w.writerow([k, c, "%.6g" % merged[k][c][0], merged[k][c][1]])
    bench = {}
    try:
        bench = json.loads(open(os.path.join(out, "bench_under_trace.json")).read().strip().splitlines()[-1])
    except Exception:
        pass
    # the bench's timed steps run ONE emit_kernel variant (the others appear with a few launches: the off-target
    # block's scans with seed words, the first sizing scan): the dominant one is the one with the most launches
    emit = sorted((k for k in merged if "emit_kernel" in k),
                  key=lambda k: -max((n for _, n in merged[k].values()), default=0))
    if emit:
        k = emit[0]
        fetch_kib = merged[k].get("FETCH_SIZE", (0, 0))[0]
        write_kib = merged[k].get("WRITE_SIZE", (0, 0))[0]
        t = {"workload": bench.get("config", {}).get("workload"), "kernel": "emit_kernel", "kernel_variant": k,
             "fetch_size_kib_raw": fetch_kib, "write_size_kib_raw": write_kib,
             "read_bytes_corrected": fetch_kib * 1024 * 2, "write_bytes": write_kib * 1024,
             "hbm_bytes_per_launch": fetch_kib * 1024 * 2 + write_kib * 1024,
             "algorithmic_bytes_per_launch": bench.get("roofline", {}).get("algorithmic_bytes_per_launch"),
             # ties the numbers to the library they were measured on: bench.py refuses another build
             "build_id": bench.get("config", {}).get("library_build"),
             "source": "profiles/%s (rocprofv3 --pmc passes of profiles/collect.sh)" % round_tag,
             # the second ceiling: VALU instructions issued per launch and the kernel's length in shader
             # cycles (GRBM_GUI_ACTIVE is summed over the 8 XCDs)
             "valu_insts_per_launch": merged[k].get("SQ_INSTS_VALU", (None, 0))[0],
             "kernel_cycles_per_launch": (merged[k].get("GRBM_GUI_ACTIVE", (0, 0))[0] / 8.0) or None,
             "note": "FETCH_SIZE doubled per MI355X_MICROARCH.md (gfx950 counts 64 B per 128-B request)"}
        # the off-target block's kernel: the three Hamming-ball passes together (FETCH_SIZE doubled as above: 16-byte
        # per-lane streaming reads; WRITE_SIZE exact)
        ball = [kk for kk in merged if "ot_ball_kernel" in kk]
        if len(ball) == 3 and all("FETCH_SIZE" in merged[kk] and "WRITE_SIZE" in merged[kk] for kk in ball):
            t["offtarget_ball_hbm_bytes_per_step"] = sum(merged[kk]["FETCH_SIZE"][0] * 2048 + merged[kk]["WRITE_SIZE"][0] * 1024
                                                         for kk in ball)
        with open(os.path.join(out, "traffic.json"), "w") as f:
            json.dump(t, f, indent=1)
        print(json.dumps(t))


if __name__ == "__main__":
    main()
