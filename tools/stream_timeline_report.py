"""Reads rocprofv3's memory_copy_trace.csv of `tools/stream_timeline.py` (three timed crp_scan_stream calls after a sizing one):
for the last call, how long each direction of the host link was busy, how long both were busy at once, the gaps.
usage: python3 tools/stream_timeline_report.py <rocprof output dir>"""
import csv, glob, json, sys


def union(iv):
    out = []
    for a, b in sorted(iv):
        if out and a <= out[-1][1]:
            out[-1][1] = max(out[-1][1], b)
        else:
            out.append([a, b])
    return out


def length(iv):
    return sum(b - a for a, b in iv)


def intersect(x, y):
    out, i, j = [], 0, 0
    while i < len(x) and j < len(y):
        a, b = max(x[i][0], y[j][0]), min(x[i][1], y[j][1])
        if a < b:
            out.append([a, b])
        if x[i][1] < y[j][1]:
            i += 1
        else:
            j += 1
    return out


def main():
    copies = []
    for f in glob.glob(sys.argv[1] + "/**/*memory_copy_trace.csv", recursive=True):
        for r in csv.DictReader(open(f)):
            copies.append((r["Direction"].replace("MEMORY_COPY_", ""), int(r["Start_Timestamp"]), int(r["End_Timestamp"])))
    # the big copies only (the link's work; this rocprofv3 has no size column: a 32 MiB chunk takes ~0.6 ms, a table column
    # 0.1-0.25 ms, the pipeline's control copies microseconds), clustered into calls by gaps of more than 8 ms
    big = sorted([c for c in copies if c[2] - c[1] > 40_000], key=lambda c: c[1])
    calls = [[big[0]]]
    for c in big[1:]:
        if c[1] - max(x[2] for x in calls[-1]) > 8_000_000:
            calls.append([c])
        else:
            calls[-1].append(c)
    last = calls[-1]
    t0, t1 = last[0][1], max(c[2] for c in last)
    up = union([(c[1], c[2]) for c in last if c[0] == "HOST_TO_DEVICE"])
    down = union([(c[1], c[2]) for c in last if c[0] == "DEVICE_TO_HOST"])
    both = intersect(up, down)
    out = {"calls_seen": len(calls), "copies_in_the_last_call": len(last), "span_ms": (t1 - t0) / 1e6,
           "h2d": {"copies": sum(1 for c in last if c[0] == "HOST_TO_DEVICE"), "busy_ms": length(up) / 1e6},
           "d2h": {"copies": sum(1 for c in last if c[0] == "DEVICE_TO_HOST"), "busy_ms": length(down) / 1e6},
           "both_directions_busy_ms": length(both) / 1e6,
           "neither_busy_ms": ((t1 - t0) - length(union(up + down))) / 1e6,
           "d2h_busy_while_h2d_busy_fraction": length(both) / max(1, length(down)),
           "first_d2h_after_start_ms": (down[0][0] - t0) / 1e6 if down else None,
           "last_h2d_end_ms": (up[-1][1] - t0) / 1e6 if up else None}
    print(json.dumps(out, indent=1))


main()
