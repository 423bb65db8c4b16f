# A/B of the "chain tables from global memory" experiment (profiles/EXPERIMENTS.md, round 4).  The three libraries come from a
# scratch copy of cropsr_amd/csrc with profiles/r04/tables_from_global_experiment.patch applied:
#   make OUT=../lib_x1.so EXTRA=-DCRP_EXP_TABLES_GLOBAL                                  (same tables, read from global memory)
#   CRP_TABLE_BITS="fC=11,fG=10" make OUT=../lib_x2.so EXTRA=-DCRP_EXP_TABLES_GLOBAL CRP_TABLE_BITS="fC=11,fG=10"
#   CRP_TABLE_BITS="fC=11,fG=10,sG=11" make OUT=../lib_x3.so ...                          (after removing the generated .inc files)
# GPU box, repo root: bash tools/ab_exp_tables.sh -- smoke() on each build (bit-exactness), then three interleaved bench rounds.
set -e
out=gpurun_out/r04/exp_tables
mkdir -p $out
libs="shipped:$PWD/cropsr_amd/libcropsr_hip.so x1:$PWD/build/exp/lib_x1.so x2:$PWD/build/exp/lib_x2.so x3:$PWD/build/exp/lib_x3.so"
for kv in $libs; do t=${kv%%:*}; l=${kv#*:}; CROPSR_HIP_LIB=$l python -c "import __graft_entry__ as g; g.smoke()" > $out/smoke_$t.log 2>&1 && echo "$t smoke ok" || echo "$t smoke FAILED"; done
for r in 1 2 3; do for kv in $libs; do t=${kv%%:*}; l=${kv#*:};
  CROPSR_HIP_LIB=$l python bench.py --steps 200 --warmup 20 --offtarget-steps 0 --cpu-sample-bases 0 > $out/${t}_$r.json 2>> $out/err.log; done; done
python - <<'PY'
import json,glob,collections
d=collections.defaultdict(list)
for f in sorted(glob.glob("gpurun_out/r04/exp_tables/*_[0-9].json")):
    j=json.loads(open(f).read().strip().splitlines()[-1]); d[f.split("/")[-1].rsplit("_",1)[0]].append(j["roofline"]["kernel_ms"])
for k,v in d.items(): print(k, ["%.4f"%x for x in v])
PY
