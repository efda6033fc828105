cd $GRAFT_REPO_ROOT; mkdir -p gpurun_out
cp basis_universal_amd/lib/libbasisu_hip.so /tmp/keep.so
cp tools/bin/libbasisu_hip_tqprof.so basis_universal_amd/lib/libbasisu_hip.so
timeout 300 python tools/tsvq_split_profile.py > gpurun_out/r6t_split_profile.txt 2>&1
cp /tmp/keep.so basis_universal_amd/lib/libbasisu_hip.so
cat gpurun_out/r6t_split_profile.txt
