#!/bin/bash
# round 6: rolled emitter loops (code size), late-reset programs emit in step; A/B of the load-time row assignment on one box
cd "$GRAFT_REPO_ROOT" || exit 1
O=gpurun_out/r06f; rm -rf $O; mkdir -p $O
python -m pytest tests -m gpu -x -q -n 4 -k "raster or frames or torus or polygon or prefix or full_size or smoke or color or first_person or recordings or teacher or free_running or sub_batch or reset_pool or late_reset" > $O/pytest.log 2>&1; echo "pytest rc=$?" >> $O/pytest.log
tail -4 $O/pytest.log
line() { echo "== $*" >> $O/bench.txt; "$@" 2>>$O/bench.err | tail -1 >> $O/bench.txt; }
line python bench.py --no-cpu-baseline
line env MOOG_DRAW_IN_STEP=0 python bench.py --no-cpu-baseline
line python bench.py --no-cpu-baseline
line env MOOG_DRAW_IN_STEP=0 python bench.py --no-cpu-baseline
line python bench.py --no-cpu-baseline --workload chase_avoid_torus
line env MOOG_DRAW_IN_STEP=0 python bench.py --no-cpu-baseline --workload chase_avoid_torus
STEPS=100 bash tools/r05_ab.sh base oldp2 > $O/ab.txt 2>&1
BENCH_ARGS="--workload chase_avoid_torus" STEPS=100 bash tools/r05_ab.sh base oldp2 >> $O/ab.txt 2>&1
python tools/bench_configs.py predators_arena_l2 parallelogram_catch match_to_sample_l3 > $O/bench_configs.txt 2>&1
MOOG_DRAW_IN_STEP=0 python tools/bench_configs.py predators_arena_l2 parallelogram_catch match_to_sample_l3 >> $O/bench_configs.txt 2>&1
