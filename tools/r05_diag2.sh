# round 5, second GPU call: the frontend pipeline (one driver thread, cooperative tasks): parity tests, then throughput against the thread-per-image form
cd $GRAFT_REPO_ROOT; mkdir -p gpurun_out; R=$GRAFT_REPO_ROOT
timeout 600 python -m pytest tests/test_gpu_frontend_pipeline.py -x -q --durations=5 2>&1 | tail -12 > gpurun_out/r05b_tests.txt; tail -4 gpurun_out/r05b_tests.txt
timeout 300 python tools/inflight_probe.py --pipeline --streams 1,2,3,4,6 > gpurun_out/r05b_pipe.txt 2>&1; cat gpurun_out/r05b_pipe.txt | cut -c1-400
BU_PIPELINE_SLEEP_US=0 timeout 200 python tools/inflight_probe.py --pipeline --streams 3,4 > gpurun_out/r05b_pipe_spin.txt 2>&1; cat gpurun_out/r05b_pipe_spin.txt | cut -c1-400
BU_PIPELINE_SPIN_US=50 timeout 200 python tools/inflight_probe.py --pipeline --streams 3,4 > gpurun_out/r05b_pipe_spin50.txt 2>&1; cat gpurun_out/r05b_pipe_spin50.txt | cut -c1-400
timeout 200 python tools/inflight_probe.py --pipeline --streams 4 --threads 8 > gpurun_out/r05b_pipe_t8.txt 2>&1; cat gpurun_out/r05b_pipe_t8.txt | cut -c1-400
timeout 200 python tools/inflight_probe.py --streams 4 > gpurun_out/r05b_threads4.txt 2>&1; cat gpurun_out/r05b_threads4.txt | cut -c1-300
cd /tmp && export TMPDIR=/tmp
db() { ls /tmp/$1/*/*.db /tmp/$1/*.db 2>/dev/null | head -1; }
timeout 200 rocprofv3 --kernel-trace -d /tmp/trp -o t -- python $R/tools/inflight_probe.py --pipeline --streams 4 --per-stream 3 --no-check > $R/gpurun_out/r05b_trp.txt 2>&1
python $R/tools/rocprof_concurrency.py $(db trp) -120 -5 > $R/gpurun_out/r05b_concurrency_pipeline4.txt 2>&1
grep "executing at once" -A 8 $R/gpurun_out/r05b_concurrency_pipeline4.txt
