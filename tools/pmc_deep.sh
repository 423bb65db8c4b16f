#!/bin/bash
# A wider look at where the emit kernel's cycles go than profiles/collect.sh takes: one rocprofv3 --pmc pass per counter
# group (counters only: no trace domains besides --kernel-trace), means per launch of the dominant emit_kernel variant.
cd /tmp && export TMPDIR=/tmp && cd "${GRAFT_REPO_ROOT:-.}"
mkdir -p gpurun_out/pmc_deep
OUT=gpurun_out/pmc_deep/summary.txt
: > $OUT
BENCH="python3 bench.py --steps 30 --warmup 2 --cpu-sample-bases 0 --offtarget-steps 0"
i=0
while read -r group; do
  [ -z "$group" ] && continue
  i=$((i+1))
  rm -rf gpurun_out/pmc_deep/g$i
  rocprofv3 --pmc $group --kernel-trace --output-format csv -d gpurun_out/pmc_deep/g$i -- $BENCH > /dev/null 2> gpurun_out/pmc_deep/g$i.err
  python3 - gpurun_out/pmc_deep/g$i >> $OUT <<'PY'
import csv, glob, sys, collections
agg = collections.defaultdict(lambda: collections.defaultdict(list))
for f in glob.glob(sys.argv[1] + "/*/*_counter_collection.csv"):
    for r in csv.DictReader(open(f)):
        if "emit_kernel" in r["Kernel_Name"]:
            agg[r["Kernel_Name"].split("(")[0]][r["Counter_Name"]].append(float(r["Counter_Value"]))
if agg:
    k = max(agg, key=lambda k: max(len(v) for v in agg[k].values()))
    for c, v in sorted(agg[k].items()):
        print("%-36s %14.6g   (%d launches)" % (c, sum(v) / len(v), len(v)))
PY
done <<'GROUPS'
SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_BUSY_CU_CYCLES SQ_LEVEL_WAVES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_WAIT_INST_LDS GRBM_GUI_ACTIVE
SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_SCA SQ_ACTIVE_INST_LDS SQ_ACTIVE_INST_VMEM SQ_ACTIVE_INST_MISC SQ_ACTIVE_INST_FLAT SQ_THREAD_CYCLES_VALU
SQ_INST_LEVEL_LDS SQ_INST_LEVEL_VMEM SQ_INST_CYCLES_VMEM_WR SQ_INST_CYCLES_VMEM_RD SQ_INST_CYCLES_SALU SQ_INSTS_VMEM_WR SQ_INSTS_VMEM_RD SQ_INSTS_BRANCH
SQ_LDS_ADDR_CONFLICT SQ_LDS_BANK_CONFLICT SQ_LDS_UNALIGNED_STALL SQ_LDS_CMD_FIFO_FULL SQ_LDS_DATA_FIFO_FULL SQ_LDS_IDX_ACTIVE SQ_INSTS_LDS SQ_LDS_ATOMIC_RETURN
SQ_VMEM_TA_ADDR_FIFO_FULL SQ_VMEM_TA_CMD_FIFO_FULL SQ_VMEM_WR_TA_DATA_FIFO_FULL SQ_IFETCH SQ_IFETCH_LEVEL SQC_ICACHE_REQ SQC_ICACHE_MISSES SQC_TC_STALL
GROUPS
cat $OUT
