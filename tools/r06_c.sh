#!/bin/bash
# round 6: the vertex-parallel emitter -- raster-related GPU tests, then A/B of raster kernel variants on one box
cd "$GRAFT_REPO_ROOT" || exit 1
O=gpurun_out/r06c; rm -rf $O; mkdir -p $O
python -m pytest tests -m gpu -x -q -n 4 -k "raster or frames or torus or polygon or prefix or full_size or smoke or color or first_person or recordings or teacher or free_running or sub_batch or reset_pool or late_reset" > $O/pytest.log 2>&1; echo "pytest rc=$?" >> $O/pytest.log
tail -4 $O/pytest.log
line() { echo "== $*" >> $O/bench.txt; "$@" 2>>$O/bench.err | tail -1 >> $O/bench.txt; }
line python bench.py --no-cpu-baseline
line env MOOG_DRAW_IN_STEP=0 python bench.py --no-cpu-baseline
line python bench.py --no-cpu-baseline --phase render
line python bench.py --no-cpu-baseline --workload chase_avoid_torus
line env MOOG_DRAW_IN_STEP=0 python bench.py --no-cpu-baseline --workload chase_avoid_torus
line python bench.py --no-cpu-baseline --workload falling_balls_64 --envs-per-gpu 8192 --steps 40
STEPS=100 bash tools/r05_ab.sh base w6 w7 t64w4 t64w5 > $O/ab.txt 2>&1
for p in 10 8 6; do echo "persist $p" >> $O/ab.txt; MOOG_RASTER_PERSIST=$p STEPS=100 bash tools/r05_ab.sh persist >> $O/ab.txt 2>&1; done
for p in 12 10; do echo "persist6 $p" >> $O/ab.txt; MOOG_RASTER_PERSIST=$p STEPS=100 bash tools/r05_ab.sh persist6 >> $O/ab.txt 2>&1; done
