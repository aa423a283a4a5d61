#!/bin/bash
# round 6: full GPU suite on the build with the vinfo emitter, emitter cycles, bench lines in-step / derived
cd "$GRAFT_REPO_ROOT" || exit 1
O=gpurun_out/r06g; rm -rf $O; mkdir -p $O
python -m pytest tests -m gpu -x -q -n 4 > $O/pytest.log 2>&1; echo "pytest rc=$?" >> $O/pytest.log
tail -4 $O/pytest.log
python tools/emit_cycles.py colliding_predators_32 4096 2>&1 | grep -v amdgpu > $O/emit_cycles.txt
python tools/emit_cycles.py chase_avoid_torus 4096 2>&1 | grep -v amdgpu >> $O/emit_cycles.txt
line() { echo "== $*" >> $O/bench.txt; "$@" 2>>$O/bench.err | tail -1 >> $O/bench.txt; }
line python bench.py --no-cpu-baseline
line env MOOG_DRAW_IN_STEP=0 python bench.py --no-cpu-baseline
line python bench.py --no-cpu-baseline --phase render
line python bench.py --no-cpu-baseline --workload chase_avoid_torus
line env MOOG_DRAW_IN_STEP=0 python bench.py --no-cpu-baseline --workload chase_avoid_torus
