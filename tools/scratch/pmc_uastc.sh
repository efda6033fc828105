# SQ counters of the UASTC kernels alone (one encode of the bench image and of the Kodak batch, tools/uastc_time.py), summary -> gpurun_out/pmc_uastc.csv.   gpurun -- bash tools/scratch/pmc_uastc.sh
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
timeout 200 rocprofv3 --kernel-trace --pmc SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_INSTS_VALU SQ_WAIT_INST_ANY SQ_INSTS_SALU SQ_WAIT_ANY SQ_ACTIVE_INST_ANY SQ_INSTS_VMEM -d /tmp/pmc_u -o u -- python $R/tools/uastc_time.py 1 $R/basis_universal_amd/lib/libbasisu_hip.so level2 > /dev/null 2> $R/gpurun_out/pmc_u.err
python $R/tools/rocprof_summary.py pmc $(ls /tmp/pmc_u/*/*.db /tmp/pmc_u/*.db 2>/dev/null | head -1) > $R/gpurun_out/pmc_uastc.csv; wc -l $R/gpurun_out/pmc_uastc.csv
