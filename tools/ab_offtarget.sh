cd $GRAFT_REPO_ROOT
for lib in build/ab/lib_*.so; do
  CROPSR_HIP_LIB=$PWD/$lib python3 bench.py --steps 2 --warmup 1 --cpu-sample-bases 0 --offtarget-steps 5 2>/dev/null | python3 -c "import sys,json; d=json.loads(sys.stdin.readline()); print('$lib', d['offtarget']['ms_per_step'], d['offtarget']['kernels_ms'])"
done
