# Round-5 evidence, one gpurun call on the final build (no extra builds needed): bench lines, per-config table with the specialised
# and the generic step kernels, rank hook, step tails, rocprofv3 trace + PMC passes of the bench, the mask rasteriser phase by phase.
# Output: gpurun_out/r05/ ; tools/r05_profiles.py turns it into profiles/r05_*.txt.
cd $GRAFT_REPO_ROOT
O=gpurun_out/r05; rm -rf $O; mkdir -p $O
export TMPDIR=/tmp
line() { echo "== $*" >> $O/bench.txt; "$@" 2>/dev/null | tail -1 >> $O/bench.txt; }
echo -n > $O/bench.txt
line python bench.py
line python bench.py --no-cpu-baseline
line python bench.py --no-cpu-baseline --no-spec
line python bench.py --no-cpu-baseline --no-schedule
line env GPU_MAX_HW_QUEUES=8 python bench.py --sub-batches 2 --no-cpu-baseline
line python bench.py --envs-per-gpu 8192 --no-cpu-baseline
line python bench.py --envs-per-gpu 16384 --no-cpu-baseline
line python bench.py --workload falling_balls_64 --envs-per-gpu 8192 --no-cpu-baseline
line python bench.py --workload falling_balls_64 --envs-per-gpu 8192 --no-cpu-baseline --no-spec
line python bench.py --workload chase_avoid_torus --no-cpu-baseline
python tools/bench_configs.py > $O/bench_configs.txt 2>&1
MOOG_STEP_SPEC=0 python tools/bench_configs.py > $O/bench_configs_generic.txt 2>&1
bash tools/bench_ranks.sh > $O/bench_ranks.txt 2>&1
python tools/pool_bench.py 1024 4096 > $O/reset_pool.txt 2>&1
python tools/step_tail.py colliding_predators_32 4096 60 2>&1 | grep -v amdgpu > $O/step_tail.txt
python tools/step_tail.py falling_balls_64 8192 130 2>&1 | grep -v amdgpu | tail -8 >> $O/step_tail.txt
bash tools/r05_mask_pmc.sh colliding_predators_32 > $O/mask_pmc.txt 2>&1
bash tools/r05_mask_pmc.sh chase_avoid_torus 0 > $O/mask_pmc_torus.txt 2>&1
bash tools/prof.sh
cp gpurun_out/prof_summary.txt $O/prof_summary.txt
cp gpurun_out/prof/bench_trace.log $O/bench_trace.log 2>/dev/null
ls -la $O
