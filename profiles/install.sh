#!/bin/bash
# Copy what profiles/collect.sh, tools/pmc_calibrate.sh and profiles/extras.sh left under gpurun_out/profiles_<tag>*/ into
# profiles/<tag>/ and refresh the copies bench.py reads (profiles/traffic*.json, profiles/offtarget_traffic.json).
#   bash profiles/install.sh r06      (development container, repo root, after the gpurun calls have merged their output back)
set -e
TAG=${1:-r06}
mkdir -p profiles/$TAG
cp gpurun_out/profiles_$TAG/* profiles/$TAG/
[ -f gpurun_out/profiles_$TAG/traffic.json ] && cp gpurun_out/profiles_$TAG/traffic.json profiles/traffic.json
[ -f gpurun_out/profiles_$TAG/offtarget_traffic.json ] && cp gpurun_out/profiles_$TAG/offtarget_traffic.json profiles/offtarget_traffic.json
for wl in tair10 ecoli sorghum; do
  s=gpurun_out/profiles_${TAG}_$wl
  [ -d $s ] || continue
  cp $s/bench_under_trace.json profiles/$TAG/bench_${wl}_like_under_trace.json
  cp $s/kernel_stats.csv profiles/$TAG/kernel_stats_${wl}_like.csv
  cp $s/pmc_summary.csv profiles/$TAG/pmc_summary_${wl}_like.csv
  cp $s/traffic.json profiles/$TAG/traffic_${wl}_like.json
  cp $s/traffic.json profiles/traffic_$wl.json
done
grep -h build_id profiles/traffic*.json profiles/offtarget_traffic.json
