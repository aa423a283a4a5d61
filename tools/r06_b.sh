#!/bin/bash
# round 6: draw records -- the GPU suite, then bench lines with the records written by the step kernel and by the derive kernel
cd "$GRAFT_REPO_ROOT" || exit 1
O=gpurun_out/r06b; rm -rf $O; mkdir -p $O
python -m pytest tests -m gpu -x -q -n 4 > $O/pytest.log 2>&1; echo "pytest rc=$?" >> $O/pytest.log
tail -5 $O/pytest.log
line() { echo "== $*" >> $O/bench.txt; "$@" 2>>$O/bench.err | tail -1 >> $O/bench.txt; }
line python bench.py --no-cpu-baseline
line env MOOG_DRAW_IN_STEP=0 python bench.py --no-cpu-baseline
line python bench.py --no-cpu-baseline --phase render
line python bench.py --no-cpu-baseline --workload chase_avoid_torus
line python bench.py --no-cpu-baseline --workload functional_maze@128 --envs-per-gpu 8192
line python bench.py --no-cpu-baseline --workload falling_balls_64 --envs-per-gpu 8192 --steps 60
line python bench.py --no-cpu-baseline --envs-per-gpu 8192
python tools/bench_configs.py > $O/bench_configs.txt 2>&1
