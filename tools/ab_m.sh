cd $GRAFT_REPO_ROOT
for rep in 1 2 3; do
echo "== old"
MOOG_SPEC_PREBUILT=1 MOOG_HIP_LIB=$PWD/tools/ubench/build/libmoog_old.so MOOG_SPEC_DIR=$PWD/tools/ubench/build/specA python bench.py --no-cpu-baseline --steps 200 2>/dev/null | tail -1
echo "== new"
python bench.py --no-cpu-baseline --steps 200 2>/dev/null | tail -1
done
echo "== old torus"; MOOG_SPEC_PREBUILT=1 MOOG_HIP_LIB=$PWD/tools/ubench/build/libmoog_old.so MOOG_SPEC_DIR=$PWD/tools/ubench/build/specA python bench.py --no-cpu-baseline --workload chase_avoid_torus 2>/dev/null | tail -1
echo "== new torus"; python bench.py --no-cpu-baseline --workload chase_avoid_torus 2>/dev/null | tail -1
echo "== old maze"; MOOG_SPEC_PREBUILT=1 MOOG_HIP_LIB=$PWD/tools/ubench/build/libmoog_old.so MOOG_SPEC_DIR=$PWD/tools/ubench/build/specA python bench.py --no-cpu-baseline --workload functional_maze@128 --envs-per-gpu 8192 2>/dev/null | tail -1
echo "== new maze"; python bench.py --no-cpu-baseline --workload functional_maze@128 --envs-per-gpu 8192 2>/dev/null | tail -1
echo "== old balls"; MOOG_SPEC_PREBUILT=1 MOOG_HIP_LIB=$PWD/tools/ubench/build/libmoog_old.so MOOG_SPEC_DIR=$PWD/tools/ubench/build/specA python bench.py --no-cpu-baseline --workload falling_balls_64 --envs-per-gpu 8192 --steps 60 2>/dev/null | tail -1
echo "== new balls"; python bench.py --no-cpu-baseline --workload falling_balls_64 --envs-per-gpu 8192 --steps 60 2>/dev/null | tail -1
echo "== old"; MOOG_HIP_LIB=$PWD/tools/ubench/build/libmoog_old.so MOOG_SPEC_DIR=$PWD/tools/ubench/build/specA python tools/bench_configs.py pacman multi_tracking_with_feature_l3 2>&1 | grep -v "amdgpu\|dynamic"
echo "== new"; python tools/bench_configs.py pacman multi_tracking_with_feature_l3 2>&1 | grep -v "amdgpu\|dynamic"
python -m pytest tests -m gpu -x -q -n 4 -k "specialised_step_kernel_is_result_neutral or full_size" 2>&1 | tail -2
