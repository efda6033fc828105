cd $GRAFT_REPO_ROOT; mkdir -p gpurun_out
echo NEW; python tools/encode_time.py 2>&1 | grep -v amdgpu.ids | tail -20
cp basis_universal_amd/lib/libbasisu_hip.so /tmp/new.so; cp tools/scratch/libbasisu_hip_old.so basis_universal_amd/lib/libbasisu_hip.so
echo OLD; python tools/encode_time.py 2>&1 | grep -v amdgpu.ids | tail -20
cp /tmp/new.so basis_universal_amd/lib/libbasisu_hip.so
