#!/bin/bash
# Collect the rocprofv3 evidence bench.py's roofline numbers are judged against.
# Run on the GPU box from the repo root:   bash profiles/collect.sh r06 [tair10|ecoli|sorghum]
# Writes raw rocprof output under gpurun_out/prof_<tag>/ (scratch) and the
# summaries under gpurun_out/profiles_<tag>/ -- copy those into profiles/.
#   1. --kernel-trace --stats      : per-kernel average durations of the bench command
#   2. --pmc FETCH_SIZE            : HBM read traffic   (separate pass, see MI355X_MICROARCH.md)
#   3. --pmc WRITE_SIZE            : HBM write traffic  (separate pass)
#   4. --pmc SQ_* (two passes)     : where the emit kernel's wave time goes
set -e
TAG=${1:-r06}
WL=${2:-}   # optional: tair10 | ecoli | sorghum -- the same passes on one of the smaller configs of BASELINE.json (-> *_$WL directories)
cd "${GRAFT_REPO_ROOT:-.}"
export TMPDIR=/tmp
RAW=gpurun_out/prof_$TAG${WL:+_$WL}
OUT=gpurun_out/profiles_$TAG${WL:+_$WL}
rm -rf "$RAW" "$OUT"
mkdir -p "$RAW" "$OUT"
# 300 timed steps: rocprofv3's per-kernel AVERAGE is over every launch of the process, and the ~100 untimed launches
# in front of the timed region include the ones that bring the clocks up (bench.py --preheat-ms)
BENCH="python3 bench.py --steps 300 --warmup 2 --cpu-sample-bases 0 --offtarget-steps 3 --no-pipelined${WL:+ --workload $WL}"

rocprofv3 --kernel-trace --stats --output-format csv -d $RAW/trace -- $BENCH > $OUT/bench_under_trace.json 2> $RAW/trace.err
rocprofv3 --pmc FETCH_SIZE --kernel-trace --output-format csv -d $RAW/fetch -- $BENCH > /dev/null 2> $RAW/fetch.err
rocprofv3 --pmc WRITE_SIZE --kernel-trace --output-format csv -d $RAW/write -- $BENCH > /dev/null 2> $RAW/write.err
rocprofv3 --pmc SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_SCA SQ_ACTIVE_INST_LDS SQ_BUSY_CYCLES \
    --kernel-trace --output-format csv -d $RAW/sq1 -- $BENCH > /dev/null 2> $RAW/sq1.err
rocprofv3 --pmc SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_INSTS_VALU_FMA_F64 SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_WAVES GRBM_GUI_ACTIVE \
    --kernel-trace --output-format csv -d $RAW/sq2 -- $BENCH > /dev/null 2> $RAW/sq2.err
python3 bench.py --steps 20 --warmup 3${WL:+ --workload $WL --cpu-sample-bases 0} > $OUT/bench_unprofiled.json 2> $RAW/bench.err

cp $RAW/trace/*/*_kernel_stats.csv $OUT/kernel_stats.csv
python3 profiles/summarize.py $RAW $OUT $TAG
ls -la $OUT
