#!/bin/bash
# A/B of kernel builds on ONE GPU box in ONE call: every library under build/ab/ runs the bench
# workload (kernel time from HIP events), interleaved over ROUNDS rounds; prints name, kernel_ms, ms_per_step.
#   make -C cropsr_amd/csrc OUT=../../build/ab/lib_X.so EXTRA="-DCRP_..."   (see cropsr_amd/csrc/Makefile)
#   bash tools/ab_kernels.sh [rounds] [extra bench args]
ROUNDS=${1:-2}
shift
cd "${GRAFT_REPO_ROOT:-.}"
mkdir -p gpurun_out
OUT=gpurun_out/ab_kernels.txt
: > $OUT
for r in $(seq 1 $ROUNDS); do
  for lib in build/ab/lib_*.so; do
    name=$(basename $lib .so)
    CROPSR_HIP_LIB=$PWD/$lib python3 bench.py --steps 30 --warmup 3 --cpu-sample-bases 0 --offtarget-steps 0 "$@" 2> gpurun_out/ab_err.txt | \
      python3 -c "import sys,json; d=json.loads(sys.stdin.readline()); print('$name round $r kernel_ms %.4f ms_per_step %.4f timeouts %d' % (d['roofline']['kernel_ms'], d['ms_per_step'], d['config']['chain_timeouts']))" >> $OUT || echo "$name round $r FAILED" >> $OUT
  done
done
cat $OUT
