cd $GRAFT_REPO_ROOT; mkdir -p gpurun_out; R=$GRAFT_REPO_ROOT
cd /tmp && export TMPDIR=/tmp
for pf in 0 1; do
BU_RESULT_PREFETCH=$pf timeout 150 rocprofv3 --kernel-trace --pmc FETCH_SIZE -d /tmp/pmc_t$pf -o bench -- python $R/bench.py --steps 1 --warmup 1 --no-cpu-baseline --no-pipelined --no-big --no-uastc --no-fast > /dev/null 2> $R/gpurun_out/r6s_pmc$pf.err
echo "prefetch $pf rc=$?"; tail -3 $R/gpurun_out/r6s_pmc$pf.err
done
