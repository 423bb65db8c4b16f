# A/B of "two 768-thread workgroups per CU with 80 KB of LDS each" (profiles/EXPERIMENTS.md, round 4): the same 24 waves per
# CU as the shipped three 512-thread workgroups, but room for bigger chain tables IN LDS (fewer gated tail FMAs).
# Libraries from a scratch copy of cropsr_amd/csrc with the generator half of profiles/r04/tables_from_global_experiment.patch
# (split gather) and GeoLarge made overridable:
#   y0 shipped shape and tables            y1 -DCRP_GEO_LARGE=768,1024,2,5016,81920
#   y2 y1 + CRP_TABLE_BITS="fC=11,fG=10"   y3 y2 with a 4 000-entry list   y4 + sG=10 and a 3 200-entry list
# GPU box, repo root: bash tools/ab_exp_geometry768.sh
set -e
out=gpurun_out/r04/exp_geo768
mkdir -p $out
libs="y0:$PWD/build/exp/lib_y0.so y1:$PWD/build/exp/lib_y1.so y2:$PWD/build/exp/lib_y2.so y3:$PWD/build/exp/lib_y3.so y4:$PWD/build/exp/lib_y4.so"
for kv in $libs; do t=${kv%%:*}; l=${kv#*:}; CROPSR_HIP_LIB=$l python -c "import __graft_entry__ as g; g.smoke()" > $out/smoke_$t.log 2>&1 && echo "$t smoke ok" || echo "$t smoke FAILED"; done
for r in 1 2 3; do for kv in $libs; do t=${kv%%:*}; l=${kv#*:};
  CROPSR_HIP_LIB=$l python bench.py --geometry large --steps 200 --warmup 20 --offtarget-steps 0 --annotate-steps 0 --cpu-sample-bases 0 > $out/${t}_$r.json 2>> $out/err.log; done; done
python - <<'PY'
import json,glob,collections
d=collections.defaultdict(list)
for f in sorted(glob.glob("gpurun_out/r04/exp_geo768/*_[0-9].json")):
    j=json.loads(open(f).read().strip().splitlines()[-1]); d[f.split("/")[-1].rsplit("_",1)[0]].append(j["roofline"]["kernel_ms"])
for k,v in d.items(): print(k, ["%.4f"%x for x in v])
PY
