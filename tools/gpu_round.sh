# One GPU round trip: parity suite, smoke, bench, rocprof kernel stats, PMC passes. Every step has its own timeout.
# usage: gpu_round.sh <tag> [prof] [pmc]
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out
tag=${1:-x}
R=$GRAFT_REPO_ROOT
timeout 420 python -m pytest tests -m gpu -x -q 2>&1 | tail -4
timeout 120 python __graft_entry__.py smoke 2>&1 | tail -2
timeout 240 python bench.py --steps 3 --warmup 1 > gpurun_out/bench_$tag.json 2> gpurun_out/bench_$tag.err; tail -c 3000 gpurun_out/bench_$tag.json
cd /tmp && export TMPDIR=/tmp
if [[ " $* " == *" prof "* ]]; then
  timeout 300 rocprofv3 --kernel-trace --stats -d $R/gpurun_out/prof_$tag -o bench -- python $R/bench.py --steps 2 --warmup 1 --no-cpu-baseline > $R/gpurun_out/prof_$tag.json 2> $R/gpurun_out/prof_$tag.err
  ls $R/gpurun_out/prof_$tag | head
fi
if [[ " $* " == *" pmc "* ]]; then
  # counters in their own passes (FETCH_SIZE and WRITE_SIZE do not fit one pass), no tracing domains besides kernel-trace
  timeout 300 rocprofv3 --kernel-trace --pmc FETCH_SIZE -d $R/gpurun_out/pmc_fetch_$tag -o bench -- python $R/bench.py --steps 1 --warmup 1 --no-cpu-baseline > /dev/null 2> $R/gpurun_out/pmc_fetch_$tag.err
  timeout 300 rocprofv3 --kernel-trace --pmc WRITE_SIZE -d $R/gpurun_out/pmc_write_$tag -o bench -- python $R/bench.py --steps 1 --warmup 1 --no-cpu-baseline > /dev/null 2> $R/gpurun_out/pmc_write_$tag.err
  ls $R/gpurun_out/pmc_fetch_$tag $R/gpurun_out/pmc_write_$tag | head
fi
