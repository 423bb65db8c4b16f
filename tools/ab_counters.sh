#!/bin/bash
# VALU / LDS / SALU instruction counts and wave cycles of the emit kernel for every library under build/ab/
# (one rocprofv3 --pmc pass each; the counters are exact dynamic counts, the timing of a profiled run is not comparable)
cd /tmp && export TMPDIR=/tmp && cd "${GRAFT_REPO_ROOT:-.}"
mkdir -p gpurun_out
: > gpurun_out/ab_counters.txt
for lib in build/ab/lib_*.so; do
  name=$(basename $lib .so)
  rm -rf gpurun_out/abc_$name
  CROPSR_HIP_LIB=$PWD/$lib rocprofv3 --pmc SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_VALU GRBM_GUI_ACTIVE \
     --kernel-trace --output-format csv -d gpurun_out/abc_$name -- python3 bench.py --steps 3 --warmup 1 --cpu-sample-bases 0 --offtarget-steps 0 > /dev/null 2> gpurun_out/abc_err.txt
  python3 - "$name" gpurun_out/abc_$name >> gpurun_out/ab_counters.txt <<'PY'
import csv, glob, sys, collections
name, d = sys.argv[1], sys.argv[2]
agg = collections.defaultdict(list)
for f in glob.glob(d + "/*/*_counter_collection.csv"):
    for r in csv.DictReader(open(f)):
        if "emit_kernel" in r["Kernel_Name"]:
            agg[r["Counter_Name"]].append(float(r["Counter_Value"]))
print(name, " ".join("%s=%.4g" % (k, sum(v) / len(v)) for k, v in sorted(agg.items())))
PY
done
cat gpurun_out/ab_counters.txt
