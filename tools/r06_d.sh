#!/bin/bash
# round 6: emitter with prefetched slot numbers / extremes for copies -- raster-related GPU tests, bench lines, per-phase counters
cd "$GRAFT_REPO_ROOT" || exit 1
O=gpurun_out/r06d; rm -rf $O; mkdir -p $O
python -m pytest tests -m gpu -x -q -n 4 -k "raster or frames or torus or polygon or prefix or full_size or smoke or color or first_person or recordings or teacher or free_running or sub_batch or reset_pool or late_reset" > $O/pytest.log 2>&1; echo "pytest rc=$?" >> $O/pytest.log
tail -4 $O/pytest.log
line() { echo "== $*" >> $O/bench.txt; "$@" 2>>$O/bench.err | tail -1 >> $O/bench.txt; }
line python bench.py --no-cpu-baseline
line env MOOG_DRAW_IN_STEP=0 python bench.py --no-cpu-baseline
line python bench.py --no-cpu-baseline
line env MOOG_DRAW_IN_STEP=0 python bench.py --no-cpu-baseline
line python bench.py --no-cpu-baseline --phase render
line python bench.py --no-cpu-baseline --workload chase_avoid_torus
line env MOOG_DRAW_IN_STEP=0 python bench.py --no-cpu-baseline --workload chase_avoid_torus
line python bench.py --no-cpu-baseline --workload functional_maze@128 --envs-per-gpu 8192
line env MOOG_DRAW_IN_STEP=0 python bench.py --no-cpu-baseline --workload functional_maze@128 --envs-per-gpu 8192
bash tools/r05_mask_pmc.sh colliding_predators_32 2 3 4 5 0 > $O/mask_pmc.txt 2>&1
