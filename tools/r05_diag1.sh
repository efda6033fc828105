# round 5, first GPU call: parity suite (with the round's new sharded / codec-grid tests), smoke, and what limits N images in flight
cd $GRAFT_REPO_ROOT; mkdir -p gpurun_out; R=$GRAFT_REPO_ROOT
timeout 1500 python -m pytest tests -m gpu -x -q --durations=12 2>&1 | tail -22 > gpurun_out/r05a_tests.txt; tail -3 gpurun_out/r05a_tests.txt
timeout 120 python __graft_entry__.py smoke 2>&1 | tail -1
timeout 240 python tools/inflight_probe.py --streams 1,2,3,4,6 > gpurun_out/r05a_probe_default.txt 2>&1
GPU_MAX_HW_QUEUES=8 timeout 240 python tools/inflight_probe.py --streams 1,2,3,4,6 > gpurun_out/r05a_probe_q8.txt 2>&1
GPU_MAX_HW_QUEUES=8 BU_TSVQ_POLL=spin timeout 200 python tools/inflight_probe.py --streams 3,4 > gpurun_out/r05a_probe_q8_spin.txt 2>&1
timeout 200 python tools/inflight_probe.py --streams 3 --null-stream > gpurun_out/r05a_probe_null.txt 2>&1
cat gpurun_out/r05a_probe_*.txt
cd /tmp && export TMPDIR=/tmp
db() { ls /tmp/$1/*/*.db /tmp/$1/*.db 2>/dev/null | head -1; }
GPU_MAX_HW_QUEUES=8 timeout 200 rocprofv3 --kernel-trace -d /tmp/tr3 -o t -- python $R/tools/inflight_probe.py --streams 3 --per-stream 4 --no-check > $R/gpurun_out/r05a_tr3.txt 2>&1
python $R/tools/rocprof_concurrency.py $(db tr3) -150 -5 > $R/gpurun_out/r05a_concurrency_3_q8.txt 2>&1
timeout 200 rocprofv3 --kernel-trace -d /tmp/tr3d -o t -- python $R/tools/inflight_probe.py --streams 3 --per-stream 4 --no-check > $R/gpurun_out/r05a_tr3d.txt 2>&1
python $R/tools/rocprof_concurrency.py $(db tr3d) -150 -5 > $R/gpurun_out/r05a_concurrency_3_default.txt 2>&1
GPU_MAX_HW_QUEUES=8 timeout 240 rocprofv3 --kernel-trace --hip-trace -d /tmp/tr3h -o t -- python $R/tools/inflight_probe.py --streams 3 --per-stream 4 --no-check > $R/gpurun_out/r05a_tr3h.txt 2>&1
python $R/tools/rocprof_concurrency.py $(db tr3h) -150 -5 > $R/gpurun_out/r05a_concurrency_3_hip.txt 2>&1
head -30 $R/gpurun_out/r05a_concurrency_3_q8.txt
