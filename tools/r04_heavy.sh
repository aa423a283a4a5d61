cd $GRAFT_REPO_ROOT
for v in "$@"; do
  echo "== $v"
  if [ "$v" = shipped ]; then python tools/heavy_bench.py bench 2>&1 | grep -v amdgpu
  elif [ "$v" = prof ]; then MOOG_HIP_LIB=$GRAFT_REPO_ROOT/tools/ubench/build/libmoog_prof.so python tools/heavy_bench.py bench --sections 2>&1 | grep -v amdgpu
  else MOOG_HIP_LIB=$GRAFT_REPO_ROOT/tools/ubench/build/libmoog_$v.so python tools/heavy_bench.py bench 2>&1 | grep -v amdgpu; fi
done
