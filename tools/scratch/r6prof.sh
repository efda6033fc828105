cd $GRAFT_REPO_ROOT; mkdir -p gpurun_out; R=$GRAFT_REPO_ROOT; tag=r6w
cd /tmp && export TMPDIR=/tmp
db() { ls /tmp/$1/*/*.db /tmp/$1/*.db 2>/dev/null | head -1; }
timeout 200 rocprofv3 --kernel-trace --stats -d /tmp/prof_$tag -o bench -- python $R/bench.py --steps 10 --warmup 2 --no-cpu-baseline --no-pipelined --no-big --no-fast --no-uastc > $R/gpurun_out/prof_$tag.json 2> $R/gpurun_out/prof_$tag.err
python $R/tools/rocprof_summary.py stats $(db prof_$tag) > $R/gpurun_out/${tag}_kernel_stats.csv; wc -l $R/gpurun_out/${tag}_kernel_stats.csv
python $R/tools/rocprof_timeline.py $(db prof_$tag) 3 > $R/gpurun_out/${tag}_step_timeline_t0.txt 2>&1; tail -1 $R/gpurun_out/${tag}_step_timeline_t0.txt
