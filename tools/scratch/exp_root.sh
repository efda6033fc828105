cd $GRAFT_REPO_ROOT
for d in p8 p8fz p8o1; do
  BU_HIP_LIB_DIR=$GRAFT_REPO_ROOT/build/lib_$d timeout 90 python tools/tsvq_root_repeat.py 120000 40 packed > gpurun_out/rr_$d.log 2>&1
  echo "$d: good=$(grep -c 'packed.*root 1502a0baef' gpurun_out/rr_$d.log) bad=$(grep 'packed' gpurun_out/rr_$d.log | grep -vc 'root 1502a0baef')"
done
