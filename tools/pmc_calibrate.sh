#!/bin/bash
# FETCH_SIZE / WRITE_SIZE on known access patterns (profiles/microbench/fetch_calibration.hip) and on the off-target block's
# and the annotation join's kernels.  GPU box, repo root:  bash tools/pmc_calibrate.sh [tag]
# Separate --pmc passes, --kernel-trace only beside them (MI355X_MICROARCH.md; the pool refuses other combinations).
set -e
TAG=${1:-r06}
cd "${GRAFT_REPO_ROOT:-.}"
export TMPDIR=/tmp
RAW=gpurun_out/pmc_cal_$TAG
rm -rf $RAW
mkdir -p $RAW gpurun_out/profiles_$TAG
hipcc -O3 --offload-arch=gfx950 profiles/microbench/fetch_calibration.hip -o $RAW/fetch_cal
$RAW/fetch_cal > $RAW/patterns.jsonl
rocprofv3 --pmc FETCH_SIZE --kernel-trace --output-format csv -d $RAW/cal_fetch -- $RAW/fetch_cal > /dev/null 2> $RAW/cal_fetch.err
rocprofv3 --pmc WRITE_SIZE --kernel-trace --output-format csv -d $RAW/cal_write -- $RAW/fetch_cal > /dev/null 2> $RAW/cal_write.err
# the product's kernels on the bench genome: scan, off-target block (5 steps), annotation join (tools/annotate_bench.py)
BENCH="python3 bench.py --steps 20 --warmup 2 --cpu-sample-bases 0 --offtarget-steps 5 --no-pipelined"
rocprofv3 --pmc FETCH_SIZE --kernel-trace --output-format csv -d $RAW/bench_fetch -- $BENCH > /dev/null 2> $RAW/bench_fetch.err
rocprofv3 --pmc WRITE_SIZE --kernel-trace --output-format csv -d $RAW/bench_write -- $BENCH > /dev/null 2> $RAW/bench_write.err
rocprofv3 --kernel-trace --stats --output-format csv -d $RAW/bench_trace -- $BENCH > $RAW/bench_under_trace.json 2> $RAW/bench_trace.err
ANN="python3 tools/annotate_bench.py sorghum 34000"
rocprofv3 --pmc FETCH_SIZE --kernel-trace --output-format csv -d $RAW/ann_fetch -- $ANN > /dev/null 2> $RAW/ann_fetch.err
rocprofv3 --pmc WRITE_SIZE --kernel-trace --output-format csv -d $RAW/ann_write -- $ANN > /dev/null 2> $RAW/ann_write.err
rocprofv3 --kernel-trace --stats --output-format csv -d $RAW/ann_trace -- $ANN > $RAW/annotate_bench.json 2> $RAW/ann_trace.err
python3 tools/pmc_calibrate.py $RAW gpurun_out/profiles_$TAG
