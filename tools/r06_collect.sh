# Round-6 evidence, one gpurun call on the final build: bench lines (headline, phases, BASELINE configs 2-5, batch sizes, generic
# kernels, derived draw records), per-config table, rank hook, step tails, emitter cycles, runtime benchmark phases, rocprofv3 trace +
# PMC passes of the bench, the mask rasteriser phase by phase.  Output: gpurun_out/r06/ ; tools/r06_profiles.py -> profiles/r06_*.txt.
cd $GRAFT_REPO_ROOT
O=gpurun_out/r06; rm -rf $O; mkdir -p $O
export TMPDIR=/tmp
line() { echo "== $*" >> $O/bench.txt; "$@" 2>/dev/null | tail -1 >> $O/bench.txt; }
echo -n > $O/bench.txt
line python bench.py
line python bench.py --no-cpu-baseline
line env MOOG_DRAW_IN_STEP=0 python bench.py --no-cpu-baseline
line python bench.py --no-cpu-baseline --no-spec
line python bench.py --no-cpu-baseline --no-schedule
line env GPU_MAX_HW_QUEUES=8 python bench.py --sub-batches 2 --no-cpu-baseline
line python bench.py --envs-per-gpu 8192 --no-cpu-baseline
line python bench.py --envs-per-gpu 16384 --no-cpu-baseline
line python bench.py --no-cpu-baseline --phase step
line python bench.py --no-cpu-baseline --phase physics
line python bench.py --no-cpu-baseline --phase render
line python bench.py --workload chase_avoid_torus --phase step
line python bench.py --workload chase_avoid_torus --no-cpu-baseline
line python bench.py --workload functional_maze@128 --envs-per-gpu 8192
line python bench.py --workload falling_balls_64 --envs-per-gpu 8192
line env MOOG_RASTER_COMPACT=0 python bench.py --workload falling_balls_64 --envs-per-gpu 8192 --no-cpu-baseline
python tools/bench_configs.py > $O/bench_configs.txt 2>&1
bash tools/bench_ranks.sh > $O/bench_ranks.txt 2>&1
python tools/step_tail.py colliding_predators_32 4096 60 2>&1 | grep -v amdgpu > $O/step_tail.txt
python tools/step_tail.py falling_balls_64 8192 130 2>&1 | grep -v amdgpu | tail -8 >> $O/step_tail.txt
python tools/emit_cycles.py colliding_predators_32 4096 2>&1 | grep -v amdgpu > $O/emit_cycles.txt
python tools/emit_cycles.py chase_avoid_torus 4096 2>&1 | grep -v amdgpu >> $O/emit_cycles.txt
python tools/emit_cycles.py functional_maze@128 8192 2>&1 | grep -v amdgpu >> $O/emit_cycles.txt
( cd moog.github.io_amd && python -m moog_demos.runtime_benchmark --config pong --num_envs 1 --reps 50 --render_sizes --render_envs 1 2>&1 | grep -v amdgpu
  python -m moog_demos.runtime_benchmark --config pong --num_envs 4096 --reps 50 --render_sizes 2>&1 | grep -v amdgpu
  python -m moog_demos.runtime_benchmark --config colliding_predators_32 --num_envs 4096 --reps 50 --render_sizes 2>&1 | grep -v amdgpu ) > $O/runtime_benchmark.txt
bash tools/r05_mask_pmc.sh colliding_predators_32 2 3 4 5 0 > $O/mask_pmc.txt 2>&1
bash tools/r05_mask_pmc.sh falling_balls_64 0 > $O/mask_pmc_balls.txt 2>&1
bash tools/r05_mask_pmc.sh chase_avoid_torus 0 > $O/mask_pmc_torus.txt 2>&1
bash tools/prof.sh
cp gpurun_out/prof_summary.txt $O/prof_summary.txt
cp gpurun_out/prof/bench_trace.log $O/bench_trace.log 2>/dev/null
ls -la $O
