# A/B of a variant library (tools/build_variant.sh NAME) against the shipped one on the two BASELINE workloads that collide:
#   bash tools/ab_bench.sh NAME [pytest -k expression]
cd $GRAFT_REPO_ROOT
V=$GRAFT_REPO_ROOT/tools/ubench/build/libmoog_$1.so
for lib in shipped $V; do
  echo "== $lib"; if [ $lib = shipped ]; then unset MOOG_HIP_LIB; else export MOOG_HIP_LIB=$lib; fi
  for wl in "--workload falling_balls_64 --envs-per-gpu 8192 --steps 100 --warmup 20" "--steps 200 --warmup 20"; do
    python bench.py $wl --no-cpu-baseline --no-extras 2>/dev/null | tail -1 | python -c "
import json,sys
d=json.loads(sys.stdin.read()); print('  %-28s %.3f M  %.3f ms  step %.0f us raster %.0f us' % (d['config']['workload'].split(':')[0], d['value']/1e6, d['ms_per_step'], d['kernels_avg_us']['step'], d['kernels_avg_us']['raster']))"
  done
done
if [ -n "$2" ]; then export MOOG_HIP_LIB=$V; timeout 900 python -m pytest tests/test_gpu_parity.py tests/test_gpu_fuzz.py -x -q -k "$2" 2>&1 | tail -4; fi
