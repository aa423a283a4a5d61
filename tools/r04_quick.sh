# quick check of an A/B build of the plain step kernel: the tests of the headline workload's path, heavy-env replay, bench
cd $GRAFT_REPO_ROOT
export MOOG_HIP_LIB=$GRAFT_REPO_ROOT/tools/ubench/build/libmoog_${1:-cur}.so
python -m pytest tests -m gpu -x -q -k "colliding or falling or lockstep or full_size or known_answers or pile or window" 2>&1 | tail -3
python tools/heavy_bench.py bench 2>&1 | grep -v amdgpu
RANDOM_SAMPLE=1 HEAVY_ONLY=256 python tools/heavy_bench.py bench 2>&1 | grep -v amdgpu
python tools/r04_exp.py --args "--no-fused --no-cpu-baseline --no-extras --steps 100 --warmup 10" ${1:-cur}=
