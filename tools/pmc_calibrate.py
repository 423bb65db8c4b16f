"""Summarise tools/pmc_calibrate.sh: python tools/pmc_calibrate.py RAW_DIR OUT_DIR

  fetch_calibration.json   per access pattern of profiles/microbench/fetch_calibration.hip: time, known requests / lines /
                           useful bytes, FETCH_SIZE and WRITE_SIZE as reported (KiB -> bytes), bytes tallied per request
  stage_traffic.json       FETCH_SIZE / WRITE_SIZE per launch of the off-target kernels and the annotation look-up on the bench
                           genome, raw and corrected with the factors the calibration gives for their access pattern
"""
import collections
import csv
import glob
import json
import os
import sys


def counters(raw, tag):
    agg = collections.defaultdict(lambda: collections.defaultdict(list))
    for f in glob.glob(os.path.join(raw, tag, "*", "*_counter_collection.csv")):
        for r in csv.DictReader(open(f)):
            name = r["Kernel_Name"].split("(")[0].replace("void ", "")
            agg[name][r["Counter_Name"]].append(float(r["Counter_Value"]))
    return {k: {c: sum(v) / len(v) for c, v in d.items()} for k, d in agg.items()}


def main():
    raw, out = sys.argv[1], sys.argv[2]
    os.makedirs(out, exist_ok=True)
    pats = [json.loads(l) for l in open(os.path.join(raw, "patterns.jsonl")) if l.startswith("{")]
    cf, cw = counters(raw, "cal_fetch"), counters(raw, "cal_write")

    def find(d, pattern):
        key = pattern.split("_w")[0] if pattern.startswith("scatter_stride") else pattern
        key = {"gather_random_256MiB": "gather_random", "scatter_random4_0.2GB": "scatter_random4"}.get(key, key)
        for k, v in d.items():
            base = k.replace("<", "_").replace(">", "").replace(", ", "_w")
            if base == pattern or k == key or base == key:
                return v
        return {}
    cal = []
    for p in pats:
        f = find(cf, p["pattern"]).get("FETCH_SIZE")
        w = find(cw, p["pattern"]).get("WRITE_SIZE")
        row = dict(p)
        if f is not None:
            row["FETCH_SIZE_bytes_raw"] = f * 1024
            row["fetch_bytes_per_request_raw"] = f * 1024 / p["requests"]
        if w is not None:
            row["WRITE_SIZE_bytes_raw"] = w * 1024
            row["write_bytes_per_request_raw"] = w * 1024 / p["requests"]
        cal.append(row)
    with open(os.path.join(out, "fetch_calibration.json"), "w") as fjson:
        json.dump(cal, fjson, indent=1)
    by = {r["pattern"]: r for r in cal}
    # factors: what one tallied byte stands for, per pattern class
    stream = by.get("stream_read16", {})
    f_stream = stream.get("useful_bytes", 0) / stream["FETCH_SIZE_bytes_raw"] if stream.get("FETCH_SIZE_bytes_raw") else None
    bf, bw = counters(raw, "bench_fetch"), counters(raw, "bench_write")
    af, aw = counters(raw, "ann_fetch"), counters(raw, "ann_write")
    stages = {}
    for src_f, src_w in ((bf, bw), (af, aw)):
        for k in sorted(set(src_f) | set(src_w)):
            if "crp::" not in k or "emit_kernel" in k or "pack" in k:
                continue
            stages[k] = {"FETCH_SIZE_bytes_raw": src_f.get(k, {}).get("FETCH_SIZE", 0) * 1024,
                         "WRITE_SIZE_bytes_raw": src_w.get(k, {}).get("WRITE_SIZE", 0) * 1024}
    with open(os.path.join(out, "stage_traffic.json"), "w") as fjson:
        json.dump({"streaming_read_factor": f_stream, "kernels": stages,
                   "note": "raw = counter x 1024; see fetch_calibration.json for what a tallied byte stands for per access pattern"},
                  fjson, indent=1)
    # ---- the off-target block, per STEP of bench.py (both strands' launches), corrected as the calibration says:
    #   reads : FETCH_SIZE tallies every 128-byte line that reaches the fabric at 64 bytes, for streams, strided and random
    #           gathers alike (fetch_calibration.json: 8 B raw per 16-byte streaming request, 64 B raw per line touched by a
    #           gather; the gathers' timings show that whole lines move) -> bytes = raw x 2
    #   writes: WRITE_SIZE is exact for 16-byte streaming stores and tallies 32 bytes per scattered 4- or 16-byte store
    #           (the partition kernels' scatters) -> bytes = raw, in 32-byte sectors for scatters
    bench = {}
    try:
        bench = json.loads(open(os.path.join(raw, "bench_under_trace.json")).read().strip().splitlines()[-1])
    except Exception:
        pass

    def step_bytes(name, launches_per_step):
        v = stages.get(name)
        if not v:
            return None
        return launches_per_step * (2 * v["FETCH_SIZE_bytes_raw"] + v["WRITE_SIZE_bytes_raw"])
    groups = {"seed_partition_histogram": [("crp::ot_seed_from_raw_kernel", 2), ("crp::ot_bucket_scan_kernel", 2),
                                           # (round 6: level 1 staged through LDS, level 2 with its chunk in registers)
                                           ("crp::ot_partition1_staged_kernel", 2), ("crp::ot_partition_reg_kernel", 2),
                                           ("crp::ot_bucket_hist_kernel", 2)],
              "ball_passes": [("crp::ot_ball_kernel<0, true>", 1), ("crp::ot_ball_kernel<8, false>", 1), ("crp::ot_ball_kernel<16, false>", 1)],
              "lookup_gather": [("crp::ot_lookup_kernel", 2)]}
    ot = {}
    for g, ks in groups.items():
        vals = [step_bytes(k, n) for k, n in ks]
        ot[g] = None if any(v is None for v in vals) else sum(vals)
    algo = (bench.get("offtarget", {}).get("roofline", {}) or {}).get("stage_bytes", {})
    ot_json = {"workload": bench.get("config", {}).get("workload"), "build_id": bench.get("config", {}).get("library_build"),
               "stage_traffic_bytes_per_step": ot,
               "stage_traffic_over_algorithmic": {g: (ot[g] / algo[g] if ot.get(g) and algo.get(g) else None) for g in ot},
               "traffic_bytes_per_step": None if any(v is None for v in ot.values()) else sum(ot.values()),
               "lookup_gather_bytes_per_request": (ot["lookup_gather"] / bench["config"]["kept_hits_total"]
                                                   if ot.get("lookup_gather") and bench.get("config") else None),
               "correction": "reads = FETCH_SIZE x 1024 x 2 (every line reaching the fabric is tallied at 64 B: calibrated on "
                             "streams, strided and random 16-byte gathers, profiles/r06/fetch_calibration.json); writes = "
                             "WRITE_SIZE x 1024 (exact for streaming stores; 32-byte sectors per scattered store)",
               "source": "tools/pmc_calibrate.sh (rocprofv3 --pmc FETCH_SIZE / WRITE_SIZE, separate passes)"}
    # the annotation look-up: bench.py's own annotate block runs it on the BENCH workload's tables (that is the figure the
    # bench line's annotate.roofline.traffic quotes); tools/annotate_bench.py runs it on the sorghum-like genome
    name = "crp::annot_lookup_kernel"
    if bf.get(name, {}).get("FETCH_SIZE") is not None and bw.get(name, {}).get("WRITE_SIZE") is not None:
        ot_json["annotate_lookup_traffic_bytes_per_launch"] = 1024 * (2 * bf[name]["FETCH_SIZE"] + bw[name]["WRITE_SIZE"])
        ot_json["annotate_lookup_algorithmic_bytes_per_launch"] = (bench.get("annotate", {}).get("roofline", {}) or {}).get("algorithmic_bytes_per_launch")
    if af.get(name, {}).get("FETCH_SIZE") is not None and aw.get(name, {}).get("WRITE_SIZE") is not None:
        ot_json["annotate_lookup_traffic_bytes_per_launch_sorghum_like"] = 1024 * (2 * af[name]["FETCH_SIZE"] + aw[name]["WRITE_SIZE"])
    with open(os.path.join(out, "offtarget_traffic.json"), "w") as fjson:
        json.dump(ot_json, fjson, indent=1)
    print(json.dumps(ot_json, indent=1))
    for r in cal:
        print("%-26s %8.3f ms  req %11.0f  fetch/req %7s  write/req %7s  GB/s if whole lines %8.1f" % (
            r["pattern"], r["ms"], r["requests"],
            "%.1f" % r["fetch_bytes_per_request_raw"] if "fetch_bytes_per_request_raw" in r else "-",
            "%.1f" % r["write_bytes_per_request_raw"] if "write_bytes_per_request_raw" in r else "-", r["GBs_if_whole_lines"]))
    for k, v in stages.items():
        print("%-60s fetch %.4g  write %.4g" % (k[:60], v["FETCH_SIZE_bytes_raw"], v["WRITE_SIZE_bytes_raw"]))


if __name__ == "__main__":
    main()
