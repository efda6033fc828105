# round 6: everything profiles/ needs at one commit -- parity suite, smoke, the bench line, rocprofv3 kernel stats + one step's time line, PMC passes, TSVQ round log,
# and the concurrency of the pipelined leg.   usage (on the GPU box): bash tools/scratch/r06_final.sh [tag]
cd $GRAFT_REPO_ROOT; mkdir -p gpurun_out; R=$GRAFT_REPO_ROOT; tag=${1:-r06}
if [[ " $* " != *" notests "* ]]; then
  timeout 1500 python -m pytest tests -m gpu -x -q --durations=6 2>&1 | tail -12 > gpurun_out/${tag}_tests.txt; tail -2 gpurun_out/${tag}_tests.txt
  timeout 120 python __graft_entry__.py smoke 2>&1 | tail -1
fi
timeout 600 python bench.py --steps 20 --warmup 3 > gpurun_out/bench_$tag.json 2> gpurun_out/bench_$tag.err; tail -c 300 gpurun_out/bench_$tag.err
python - <<PY
import json
d=json.load(open("gpurun_out/bench_$tag.json"))
print({k:d.get(k) for k in ("value","ms_per_step","identical_to_reference","host_gap_ms","host_cpu_s_per_step")}, d["roofline"]["frac"], d["roofline"]["avg_launch_ms"])
p=d.get("pipelined") or {}; print({k:p.get(k) for k in ("value","host_cpu_s_per_image","driver_thread_cpu_s_per_image","identical_to_reference","thread_per_image")})
print("with backend", (d.get("pipelined_with_backend") or {}).get("value"))
e=d.get("end_to_end",{}); print({k:(e.get(k) or {}).get("seconds", (e.get(k) or {}).get("mpix_s")) for k in ("stock_1_thread","stock_all_cores","resident","resident_1_thread")}, {k:(e.get(k) or {}).get("mpix_s") for k in ("resident_parallel","resident_parallel_glibc_hugetlb")})
print({k:(d.get(k) or {}).get("value") for k in ("uastc","uastc_rdo","etc1s_8192_q255","reference_default_threads","fast_codebooks")}, (d.get("uastc_rdo") or {}).get("one_batch_start_to_finish"), (d.get("uastc_rdo") or {}).get("images_identical_to_reference"))
PY
cd /tmp && export TMPDIR=/tmp
db() { ls /tmp/$1/*/*.db /tmp/$1/*.db 2>/dev/null | head -1; }
timeout 200 rocprofv3 --kernel-trace --stats -d /tmp/prof_$tag -o bench -- python $R/bench.py --steps 10 --warmup 2 --no-cpu-baseline --no-pipelined --no-big --no-fast --no-uastc > $R/gpurun_out/prof_$tag.json 2> $R/gpurun_out/prof_$tag.err
python $R/tools/rocprof_summary.py stats $(db prof_$tag) > $R/gpurun_out/${tag}_kernel_stats.csv; wc -l $R/gpurun_out/${tag}_kernel_stats.csv
python $R/tools/rocprof_timeline.py $(db prof_$tag) 3 > $R/gpurun_out/${tag}_step_timeline_t0.txt 2>&1   # the 4th step of the run: a timed headline step; tail -1 $R/gpurun_out/${tag}_step_timeline_t0.txt
for c in FETCH_SIZE WRITE_SIZE; do
  l=$(echo $c | tr A-Z a-z)
  timeout 150 rocprofv3 --kernel-trace --pmc $c -d /tmp/pmc_${l}_$tag -o bench -- python $R/bench.py --steps 1 --warmup 1 --no-cpu-baseline --no-pipelined --no-big > /dev/null 2> $R/gpurun_out/pmc_${l}_$tag.err
  python $R/tools/rocprof_summary.py pmc $(db pmc_${l}_$tag) > $R/gpurun_out/${tag}_pmc_${l}.csv 2>/dev/null; wc -l $R/gpurun_out/${tag}_pmc_${l}.csv
done
timeout 150 rocprofv3 --kernel-trace --pmc SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_ACTIVE_INST_VALU SQ_INSTS_VALU SQ_WAIT_INST_ANY SQ_INSTS_SALU SQ_WAIT_ANY SQ_ACTIVE_INST_ANY -d /tmp/pmc_sq_$tag -o bench -- python $R/bench.py --steps 1 --warmup 1 --no-cpu-baseline --no-pipelined --no-big > /dev/null 2> $R/gpurun_out/pmc_sq_$tag.err
python $R/tools/rocprof_summary.py pmc $(db pmc_sq_$tag) > $R/gpurun_out/${tag}_pmc_sq.csv 2>/dev/null; wc -l $R/gpurun_out/${tag}_pmc_sq.csv
timeout 200 rocprofv3 --kernel-trace -d /tmp/trp_$tag -o t -- python $R/tools/inflight_probe.py --pipeline --streams 4 --per-stream 4 --no-check > $R/gpurun_out/${tag}_pipelined_probe.txt 2>&1
python $R/tools/rocprof_concurrency.py $(db trp_$tag) -150 -5 > $R/gpurun_out/${tag}_pipelined_concurrency.txt 2>&1; grep "executing at once" -A 7 $R/gpurun_out/${tag}_pipelined_concurrency.txt
cd $R && BU_TSVQ_ROUNDS=1 timeout 120 python bench.py --steps 1 --warmup 1 --no-cpu-baseline --no-pipelined --no-big --no-uastc --no-fast > /dev/null 2> gpurun_out/rounds_$tag.log
grep "tsvq round" gpurun_out/rounds_$tag.log | sed -n 30,58p > gpurun_out/${tag}_tsvq_rounds_t0.txt; wc -l gpurun_out/${tag}_tsvq_rounds_t0.txt   # the timed step (the warm-up step's 29 rounds come first)
# round 6 additions: the reference's own seam benchmark through the HIP seam build, the UASTC + RDO lanes x reserved-CU table, the 6-float stress up to the endpoint builder's ceiling
if [ -x $R/oracle/_ref/basisu_hip ]; then (cd /tmp && timeout 300 $R/oracle/_ref/basisu_hip -clbench > $R/gpurun_out/${tag}_clbench_hip_seam.txt 2>&1; tail -4 $R/gpurun_out/${tag}_clbench_hip_seam.txt); fi
cd $R && timeout 300 python tools/rdo_lanes.py 12 3,4 0,32,64 > gpurun_out/${tag}_rdo_lanes.txt 2>&1; tail -6 gpurun_out/${tag}_rdo_lanes.txt
timeout 200 python tools/wide6_stress.py 90 > gpurun_out/${tag}_wide6_stress.txt 2>&1; tail -2 gpurun_out/${tag}_wide6_stress.txt
timeout 200 python tools/wide16_stress.py 60 > gpurun_out/${tag}_wide16_stress.txt 2>&1; tail -2 gpurun_out/${tag}_wide16_stress.txt
