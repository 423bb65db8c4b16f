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
    for r in cal:
        print("%-26s %8.3f ms  req %11.0f  fetch/req %7s  write/req %7s  GB/s if whole lines %8.1f" % (
            r["pattern"], r["ms"], r["requests"],
            "%.1f" % r["fetch_bytes_per_request_raw"] if "fetch_bytes_per_request_raw" in r else "-",
            "%.1f" % r["write_bytes_per_request_raw"] if "write_bytes_per_request_raw" in r else "-", r["GBs_if_whole_lines"]))
    for k, v in stages.items():
        print("%-60s fetch %.4g  write %.4g" % (k[:60], v["FETCH_SIZE_bytes_raw"], v["WRITE_SIZE_bytes_raw"]))


if __name__ == "__main__":
    main()
