# A/B of tile geometries on small arenas (GPU box, repo root): bash tools/ab_geometry.sh
# build/ab/lib_small_<tag>.so = builds with another SMALL shape (make EXTRA='-DCRP_GEO_SMALL=...' OUT=...)
set -e
out=gpurun_out/r04/geo2
mkdir -p $out
run() { # tag lib geometry workload-args...
  tag=$1; lib=$2; g=$3; shift 3
  CROPSR_HIP_LIB=$lib python bench.py "$@" --geometry $g --steps 200 --warmup 20 --offtarget-steps 0 --cpu-sample-bases 0 > $out/$tag.json 2>> $out/err.log
}
for g in large small; do
  run ecoli_$g "" $g --workload ecoli
  for s in 0.01 0.02 0.04 0.08; do run sg${s}_$g "" $g --scale $s; done
done
for t in a b c d e f; do
  run ecoli_small_$t $PWD/build/ab/lib_small_$t.so small --workload ecoli
  for s in 0.01 0.02 0.04 0.08; do run sg${s}_small_$t $PWD/build/ab/lib_small_$t.so small --scale $s; done
done
CROPSR_HIP_LIB="" python bench.py --workload ecoli --two-pass --geometry small --steps 200 --warmup 20 --offtarget-steps 0 --cpu-sample-bases 0 > $out/ecoli_small_twopass.json 2>> $out/err.log
python - <<'PY'
import json,glob
for f in sorted(glob.glob("gpurun_out/r04/geo2/*.json")):
    d=json.load(open(f)); r=d["roofline"]; c=d["config"]
    print("%-28s %-6s words %5d tiles %6d kernel %.4f ms (count %.4f scan %.4f) step %.4f ms" % (f.split("/")[-1][:-5], c["tile_geometry"], c["tile_words"], c["tiles_per_launch"], r["kernel_ms"], r["count_kernel_ms"], r["tile_scan_ms"], d["ms_per_step"]))
PY
