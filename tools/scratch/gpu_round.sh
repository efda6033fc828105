# One GPU round trip: parity suite, smoke, bench, rocprof kernel stats, PMC passes. Every step has its own timeout. The rocpd
# databases stay on the box (they exceed what gpurun merges back); what comes home are the CSV / JSON summaries under gpurun_out/.
# usage: gpu_round.sh <tag> [tests] [bench] [prof] [pmc] [calib] [rounds]
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out
tag=${1:-x}
R=$GRAFT_REPO_ROOT
if [[ " $* " == *" tests "* ]]; then
  timeout 900 python -m pytest tests -m gpu -x -q 2>&1 | tail -4
  timeout 120 python __graft_entry__.py smoke 2>&1 | tail -2
fi
if [[ " $* " == *" bench "* ]]; then
  timeout 420 python bench.py --steps 10 --warmup 2 > gpurun_out/bench_$tag.json 2> gpurun_out/bench_$tag.err; tail -c 1500 gpurun_out/bench_$tag.json
fi
cd /tmp && export TMPDIR=/tmp
db() { ls /tmp/$1/*/*.db /tmp/$1/*.db 2>/dev/null | head -1; }
if [[ " $* " == *" prof "* ]]; then
  timeout 120 rocprofv3 --kernel-trace --stats -d /tmp/prof_$tag -o bench -- python $R/bench.py --steps 10 --warmup 2 --no-cpu-baseline --no-pipelined --no-big > $R/gpurun_out/prof_$tag.json 2> $R/gpurun_out/prof_$tag.err
  python $R/tools/rocprof_summary.py stats $(db prof_$tag) > $R/gpurun_out/${tag}_kernel_stats.csv; wc -l $R/gpurun_out/${tag}_kernel_stats.csv
fi
if [[ " $* " == *" pmc "* ]]; then
  # counters in their own passes (FETCH_SIZE and WRITE_SIZE do not fit one pass), no tracing domains besides kernel-trace
  for c in FETCH_SIZE WRITE_SIZE; do
    l=$(echo $c | tr A-Z a-z)
    timeout 120 rocprofv3 --kernel-trace --pmc $c -d /tmp/pmc_${l}_$tag -o bench -- python $R/bench.py --steps 1 --warmup 1 --no-cpu-baseline --no-pipelined --no-big > /dev/null 2> $R/gpurun_out/pmc_${l}_$tag.err
    python $R/tools/rocprof_summary.py pmc $(db pmc_${l}_$tag) > $R/gpurun_out/${tag}_pmc_${l}.csv; wc -l $R/gpurun_out/${tag}_pmc_${l}.csv
  done
  timeout 120 rocprofv3 --kernel-trace --pmc SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_ACTIVE_INST_VALU SQ_INSTS_VALU SQ_WAIT_INST_ANY SQ_INSTS_SALU SQ_WAIT_ANY SQ_ACTIVE_INST_ANY -d /tmp/pmc_sq_$tag -o bench -- python $R/bench.py --steps 1 --warmup 1 --no-cpu-baseline --no-pipelined --no-big > /dev/null 2> $R/gpurun_out/pmc_sq_$tag.err
  python $R/tools/rocprof_summary.py pmc $(db pmc_sq_$tag) > $R/gpurun_out/${tag}_pmc_sq.csv; wc -l $R/gpurun_out/${tag}_pmc_sq.csv
fi

if [[ " $* " == *" rounds "* ]]; then
  # one step with the TSVQ round time line on stderr (BU_TSVQ_ROUNDS)
  cd $R && BU_TSVQ_ROUNDS=1 timeout 120 python bench.py --steps 1 --warmup 1 --no-cpu-baseline --no-pipelined --no-big --no-uastc --no-fast --no-big > /dev/null 2> gpurun_out/rounds_$tag.log; grep -c "tsvq round" gpurun_out/rounds_$tag.log
  cd /tmp
fi
if [[ " $* " == *" calib "* ]]; then
  # VALU calibration (tools/valu_calib.hip): measured cycles per wave-instruction, then the SQ counters of the same launches
  timeout 300 $R/tools/bin/valu_calib > $R/gpurun_out/valu_calibration_$tag.json 2> $R/gpurun_out/valu_calib_$tag.err; wc -c $R/gpurun_out/valu_calibration_$tag.json
  timeout 300 rocprofv3 --kernel-trace --pmc SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_ACTIVE_INST_VALU SQ_INSTS_VALU SQ_WAIT_INST_ANY SQ_INSTS_SALU SQ_WAIT_ANY SQ_ACTIVE_INST_ANY -d /tmp/pmc_calib_$tag -o calib -- $R/tools/bin/valu_calib > /dev/null 2> $R/gpurun_out/pmc_calib_$tag.err
  python $R/tools/rocprof_summary.py pmc $(db pmc_calib_$tag) > $R/gpurun_out/${tag}_pmc_calib.csv; wc -l $R/gpurun_out/${tag}_pmc_calib.csv
fi
if [[ " $* " == *" rdoprof "* ]]; then
  # where the serial RDO step goes (instrumented library, tools/build_rdo_profile.sh)
  cd $R && timeout 300 python tools/rdo_step_profile.py > gpurun_out/rdo_step_$tag.txt 2> gpurun_out/rdo_step_$tag.err; cat gpurun_out/rdo_step_$tag.txt | tail -14
  cd /tmp
fi
