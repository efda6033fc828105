# One GPU round trip: parity suite, smoke, bench, rocprof kernel stats. Every step has its own timeout.
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out
tag=${1:-x}
timeout 420 python -m pytest tests -m gpu -x -q 2>&1 | tail -4
timeout 120 python __graft_entry__.py smoke 2>&1 | tail -2
timeout 240 python bench.py --steps 3 --warmup 1 > gpurun_out/bench_$tag.json 2> gpurun_out/bench_$tag.err; tail -c 3000 gpurun_out/bench_$tag.json
if [ "$2" = "prof" ]; then
  cd /tmp && export TMPDIR=/tmp
  timeout 300 rocprofv3 --kernel-trace --stats -d $GRAFT_REPO_ROOT/gpurun_out/prof_$tag -o bench -- python $GRAFT_REPO_ROOT/bench.py --steps 2 --warmup 1 --no-cpu-baseline > $GRAFT_REPO_ROOT/gpurun_out/prof_$tag.json 2> $GRAFT_REPO_ROOT/gpurun_out/prof_$tag.err
  ls $GRAFT_REPO_ROOT/gpurun_out/prof_$tag | head
fi
