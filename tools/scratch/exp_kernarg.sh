cd $GRAFT_REPO_ROOT
for env in "" "HIP_FORCE_DEV_KERNARG=0"; do
  echo "=== env [$env]"
  env $env timeout 90 python tools/tsvq_root_repeat.py 120000 30 > gpurun_out/rr_$$.log 2>&1; echo "root_repeat rc=$? good=$(grep -c 'root 1502a0baef' gpurun_out/rr_$$.log) bad=$(grep -vc 'root 1502a0baef' gpurun_out/rr_$$.log)"; grep -v 'root 1502a0baef' gpurun_out/rr_$$.log | cut -c1-200 | head -5
  for i in 1 2 3; do env $env timeout 60 python tools/tsvq_verify.py 1 2>/dev/null | grep -v "^\[tsvq" | cut -c1-150; done
done
