"""Reads rocprofv3's memory_copy_trace.csv (+ kernel_trace.csv) and the JSON line of tools/stream_timeline.py: for the last
crp_scan_stream call, how long each direction of the host link was busy, how long both were busy at once, the gaps."""
import csv, glob, json, sys

def union(iv):
    iv = sorted(iv)
    out = []
    for a, b in iv:
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
    d, line = sys.argv[1], json.loads(open(sys.argv[2]).read().strip().splitlines()[-1])
    copies = []
    for f in glob.glob(d + "/**/*memory_copy_trace.csv", recursive=True):
        for r in csv.DictReader(open(f)):
            copies.append((r["Direction"], int(r["Start_Timestamp"]), int(r["End_Timestamp"]), int(r.get("Size", 0) or 0)))
    kernels = []
    for f in glob.glob(d + "/**/*kernel_trace.csv", recursive=True):
        for r in csv.DictReader(open(f)):
            kernels.append((r["Kernel_Name"], int(r["Start_Timestamp"]), int(r["End_Timestamp"])))
    # the last call = the last third of the big copies: take copies after the start of the last run of H2D chunks
    big = sorted([c for c in copies if c[3] >= (1 << 20)], key=lambda c: c[1])
    h2d = [c for c in big if "HOST_TO_DEVICE" in c[0].upper() or c[0].upper().startswith("H2D")]
    total_up = sum(c[3] for c in h2d)
    per_call = total_up / 4.0  # four calls in the script (one sizing call + three)
    acc, t_start = 0, None
    for c in reversed(h2d):
        acc += c[3]
        t_start = c[1]
        if acc >= per_call * 0.999:
            break
    last = [c for c in big if c[1] >= t_start]
    up = union([(c[1], c[2]) for c in last if c in h2d])
    down = union([(c[1], c[2]) for c in last if c not in h2d])
    t0, t1 = min(c[1] for c in last), max(c[2] for c in last)
    both = intersect(up, down)
    ks = [k for k in kernels if t0 <= k[1] <= t1]
    emit = [k for k in ks if "emit_kernel" in k[0]]
    out = {
        "span_ms": (t1 - t0) / 1e6, "wall_ms_by_the_call": line["stats"]["wall_s"] * 1e3,
        "h2d": {"bytes": sum(c[3] for c in last if c in h2d), "busy_ms": length(up) / 1e6, "copies": sum(1 for c in last if c in h2d)},
        "d2h": {"bytes": sum(c[3] for c in last if c not in h2d), "busy_ms": length(down) / 1e6, "copies": sum(1 for c in last if c not in h2d)},
        "both_directions_busy_ms": length(both) / 1e6,
        "neither_busy_ms": ((t1 - t0) - length(union(up + down))) / 1e6,
        "emit_kernels": len(emit), "emit_kernel_ms_sum": sum(k[2] - k[1] for k in emit) / 1e6,
        "first_d2h_after_start_ms": (min(c[1] for c in last if c not in h2d) - t0) / 1e6 if down else None,
        "last_h2d_end_ms": (max(c[2] for c in last if c in h2d) - t0) / 1e6,
    }
    out["h2d"]["GBs_while_busy"] = out["h2d"]["bytes"] / out["h2d"]["busy_ms"] / 1e6
    out["d2h"]["GBs_while_busy"] = out["d2h"]["bytes"] / out["d2h"]["busy_ms"] / 1e6 if down else None
    print(json.dumps(out, indent=1))

main()
