#!/bin/bash
# round 6: emission before the record's stores, row records assigned with the load, 4-byte edge records
cd "$GRAFT_REPO_ROOT" || exit 1
O=gpurun_out/r06e; rm -rf $O; mkdir -p $O
python -m pytest tests -m gpu -x -q -n 4 -k "raster or frames or torus or polygon or prefix or full_size or smoke or color or first_person or recordings or teacher or free_running or sub_batch or reset_pool or late_reset" > $O/pytest.log 2>&1; echo "pytest rc=$?" >> $O/pytest.log
tail -4 $O/pytest.log
line() { echo "== $*" >> $O/bench.txt; "$@" 2>>$O/bench.err | tail -1 >> $O/bench.txt; }
line python bench.py --no-cpu-baseline
line env MOOG_DRAW_IN_STEP=0 python bench.py --no-cpu-baseline
line python bench.py --no-cpu-baseline
line env MOOG_DRAW_IN_STEP=0 python bench.py --no-cpu-baseline
line env MOOG_RASTER_COMPACT=1 python bench.py --no-cpu-baseline
line python bench.py --no-cpu-baseline --workload chase_avoid_torus
line env MOOG_DRAW_IN_STEP=0 python bench.py --no-cpu-baseline --workload chase_avoid_torus
line python bench.py --no-cpu-baseline --workload functional_maze@128 --envs-per-gpu 8192
line python bench.py --no-cpu-baseline --workload falling_balls_64 --envs-per-gpu 8192 --steps 40
line env MOOG_RASTER_COMPACT=0 python bench.py --no-cpu-baseline --workload falling_balls_64 --envs-per-gpu 8192 --steps 40
export MOOG_RASTER_COMPACT=1
STEPS=100 bash tools/r05_ab.sh w6 w7 > $O/ab.txt 2>&1
unset MOOG_RASTER_COMPACT
python tools/bench_configs.py first_person_predators_prey cleanup maze_zoo match_to_sample_l3 parallelogram_catch pong > $O/bench_configs.txt 2>&1
