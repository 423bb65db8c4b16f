#!/bin/bash
# LARGE against MEDIUM (512 threads on 896 words; apply profiles/r05/medium_geometry_experiment.patch first -- the shape did not ship)
# GPU box, repo root:  bash tools/ab_medium.sh > gpurun_out/ab_medium.jsonl
set -e
cd "${GRAFT_REPO_ROOT:-.}"
run() {  # label, bench arguments
    local label=$1; shift
    for geo in large medium auto; do
        python3 bench.py "$@" --geometry $geo --steps 200 --warmup 3 --cpu-sample-bases 0 --offtarget-steps 0 --annotate-steps 0 2>/dev/null |
            python3 -c "import json,sys; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print(json.dumps({'arena': '$label', 'asked': '$geo', 'geometry': d['config']['tile_geometry'], 'tiles': d['config']['tiles_per_launch'], 'kernel_ms': round(d['roofline']['kernel_ms'], 5), 'ms_per_step': round(d['ms_per_step'], 5), 'frac': round(d['roofline']['frac'], 4)}))"
    done
}
run "switchgrass x0.04" --scale 0.04
run "switchgrass x0.08" --scale 0.08
run "tair10-like" --workload tair10
run "switchgrass x0.125 (the share of one of 8 ranks)" --scale 0.125
run "switchgrass x0.16" --scale 0.16
run "switchgrass x0.25" --scale 0.25
run "sorghum-like" --workload sorghum
run "switchgrass 1.13 Gb" --scale 1.0
