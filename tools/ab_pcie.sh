#!/bin/bash
# A/B of the host <-> device leg on ONE box: every library under build/ab/ runs the bench workload's upload and fetch
# (pcie_inclusive of bench.py), interleaved over ROUNDS rounds.
ROUNDS=${1:-3}
cd "${GRAFT_REPO_ROOT:-.}"
mkdir -p gpurun_out
: > gpurun_out/ab_pcie.txt
for r in $(seq 1 $ROUNDS); do
  for lib in build/ab/lib_*.so; do
    name=$(basename $lib .so)
    CROPSR_HIP_LIB=$PWD/$lib python3 bench.py --steps 5 --warmup 1 --cpu-sample-bases 0 --offtarget-steps 0 2> gpurun_out/ab_err.txt | \
      python3 -c "import sys,json; d=json.loads(sys.stdin.readline()); p=d['pcie_inclusive']; print('$name round $r upload_pack_s %.4f fetch_tables_s %.4f' % (p['upload_pack_s'], p['fetch_tables_s']))" >> gpurun_out/ab_pcie.txt || echo "$name round $r FAILED" >> gpurun_out/ab_pcie.txt
  done
done
cat gpurun_out/ab_pcie.txt
