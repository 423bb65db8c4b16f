#!/bin/bash
# The files of profiles/rNN/ that profiles/collect.sh does not write, for the SAME library build:
#   bash profiles/extras.sh r06 [soak]     (GPU box, repo root; `soak` adds the 900-trial fuzz in all four scan-mode x geometry combinations and of the node handle, ~9 minutes)
# Everything lands under gpurun_out/profiles_<tag>/; copy it into profiles/<tag>/ beside collect.sh's files.
set -e
TAG=${1:-r06}
cd "${GRAFT_REPO_ROOT:-.}"
export TMPDIR=/tmp
RAW=gpurun_out/prof_$TAG
OUT=gpurun_out/profiles_$TAG
mkdir -p "$RAW" "$OUT"

# the bench with no flags (CPU baseline legs included) and with the driver's flags, plain and under the kernel trace
python3 bench.py > $OUT/bench_default_run.json 2> $RAW/default.err
python3 bench.py --gpus 1 --steps 20 --warmup 5 > $OUT/bench_driver_flags_run.json 2> $RAW/driver.err
rocprofv3 --kernel-trace --stats --output-format csv -d $RAW/driver_trace -- python3 bench.py --gpus 1 --steps 20 --warmup 5 \
    > $OUT/bench_driver_flags_under_trace.json 2> $RAW/driver_trace.err
cp $RAW/driver_trace/*/*_kernel_stats.csv $OUT/kernel_stats_driver_flags.csv
# the smaller configs of BASELINE.json (parity cases, not the headline)
python3 bench.py --workload tair10 --cpu-sample-bases 0 > $OUT/bench_tair10_like.json 2> $RAW/tair10.err
python3 bench.py --workload ecoli --cpu-sample-bases 0 > $OUT/bench_ecoli_like.json 2> $RAW/ecoli.err
# N = 2 started by the program itself, both ranks on this one GPU (host transport for fences and the final gatherv)
python3 bench.py --gpus 2 --share-gpu0 --scale 0.25 --steps 10 --warmup 3 --cpu-sample-bases 0 \
    > $OUT/bench_2ranks_self_launched_one_gpu.json 2> $RAW/2ranks.err
# six ranks on the one GPU (the most processes the pool lets one job keep on a card: no room for rank 0's node-block child),
# weak headline + strong-scaling block
python3 bench.py --gpus 6 --share-gpu0 --scale 0.25 --steps 10 --warmup 3 --cpu-sample-bases 0 --offtarget-steps 0 --no-node-block \
    > $OUT/bench_6ranks_self_launched_one_gpu.json 2> $RAW/6ranks.err
# ONE process over four logical devices (the library's node handle; on this one-GPU box they are GPU 0 four times, the exchange
# runs as device-to-device copies): weak headline of 4 x 0.25 genomes, gatherv packed / raw, strong block with digest check
python3 bench.py --gpus 4 --single-process --share-gpu0 --scale 0.25 --steps 10 --warmup 3 --cpu-sample-bases 0 \
    > $OUT/bench_single_process_4_logical_devices.json 2> $RAW/sp4.err
python3 bench.py --gpus 4 --single-process --share-gpu0 --steps 10 --warmup 3 --cpu-sample-bases 0 \
    > $OUT/bench_single_process_4_logical_devices_full_size.json 2> $RAW/sp4full.err
# the whole CLI end to end (FASTA in, CSV out)
python3 tools/e2e_cli.py switchgrass > $OUT/e2e_cli_switchgrass_first_process_on_the_box.json 2> $RAW/e2e0.err
python3 tools/e2e_cli.py switchgrass > $OUT/e2e_cli_switchgrass.json 2> $RAW/e2e1.err
python3 tools/e2e_cli.py switchgrass --cli-flag=--offtarget > $OUT/e2e_cli_switchgrass_offtarget.json 2> $RAW/e2e2.err
python3 tools/e2e_cli.py switchgrass --cli-flag=--score-finalize --cli-flag=host > $OUT/e2e_cli_switchgrass_finalize_host.json 2> $RAW/e2e3.err
python3 tools/e2e_cli.py tair10 --reference-behaviour > $OUT/e2e_cli_tair10_reference_behaviour.json 2> $RAW/e2e4.err
# BASELINE.json configs[2], [3]: the genomes that come with a GFF (+ annotation_info): --annotate, one process and two
python3 tools/e2e_cli.py sorghum > $OUT/e2e_cli_sorghum.json 2> $RAW/e2e5.err
python3 tools/e2e_cli.py sorghum --annotate 34000 > $OUT/e2e_cli_sorghum_annotate.json 2> $RAW/e2e6.err
python3 tools/e2e_cli.py tair10 --annotate 27000 > $OUT/e2e_cli_tair10_annotate.json 2> $RAW/e2e7.err
python3 tools/e2e_cli.py sorghum --annotate 34000 --procs 2 > $OUT/e2e_cli_sorghum_annotate_2ranks_one_gpu.json 2> $RAW/e2e8.err
python3 tools/e2e_cli.py sorghum --annotate 34000 --devices 0,0,0,0 > $OUT/e2e_cli_sorghum_annotate_4_logical_devices_one_process.json 2> $RAW/e2e9.err
python3 tools/annotate_bench.py sorghum 34000 > $OUT/annotate_lookup_sorghum.json 2> $RAW/ann1.err
python3 tools/annotate_bench.py tair10 27000 > $OUT/annotate_lookup_tair10.json 2> $RAW/ann2.err
{ for p in 1 4; do python3 tools/e2e_cli.py tair10 --procs $p --md5 --cli-flag=--offtarget; done; } > $OUT/multi_process_cli_md5.txt 2> $RAW/md5.err
# round 6: seam 1 as a pipeline on the default genome (the C-side wall clock of crp_scan_stream under the knobs) and the link / file
# microbenchmarks its design rests on
python3 tools/stream_tune.py > $OUT/stream_tune.jsonl 2> $RAW/stream_tune.err
hipcc -O2 --offload-arch=gfx950 profiles/microbench/duplex_copy.hip -o $RAW/duplex_copy && $RAW/duplex_copy 1024 > $OUT/duplex_copy.json
g++ -O2 -pthread profiles/microbench/file_write.cpp -o $RAW/file_write && $RAW/file_write /tmp/crp_file_write.tmp 6 16 > $OUT/file_write.txt
if [ "$2" = soak ]; then
  CROPSR_FUZZ_TRIALS=900 CROPSR_FUZZ_PROGRESS=$RAW/fuzz_progress.txt python3 -m pytest -q tests/test_gpu_parity.py::test_randomised_arenas_vs_oracle \
      tests/test_offtarget.py::test_gpu_randomised_genomes_vs_oracle tests/test_node.py::test_node_randomised_genomes_vs_oracle \
      tests/test_stream.py::test_scan_stream_randomised_genomes_vs_oracle tests/test_fake_rccl.py::test_node_randomised_genomes_on_rccl_double \
      > $OUT/fuzz_soak_900_trials_x4_modes.log 2>&1
fi
ls -la $OUT
