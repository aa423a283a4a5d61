# Round-4 evidence refreshed on the final build (reset pool, per-env prefix): the bench lines, the per-config table, the rank
# hook, the pool and prefix A/B runs and the rocprofv3 passes.  (The step kernel's section / function / latency profiles of
# tools/r04_collect.sh are those of the same step-kernel code and are not repeated.)
cd $GRAFT_REPO_ROOT
O=gpurun_out/r04; mkdir -p $O
echo "== default bench (python bench.py)" > $O/bench.txt
python bench.py 2>/dev/null | tail -1 >> $O/bench.txt
echo "== python bench.py --no-fused --no-cpu-baseline" >> $O/bench.txt
python bench.py --no-fused --no-cpu-baseline 2>/dev/null | tail -1 >> $O/bench.txt
echo "== GPU_MAX_HW_QUEUES=8 python bench.py --sub-batches 2 --no-cpu-baseline" >> $O/bench.txt
GPU_MAX_HW_QUEUES=8 python bench.py --sub-batches 2 --no-cpu-baseline 2>/dev/null | tail -1 >> $O/bench.txt
echo "== python bench.py --workload falling_balls_64 --envs-per-gpu 8192 --no-cpu-baseline" >> $O/bench.txt
python bench.py --workload falling_balls_64 --envs-per-gpu 8192 --no-cpu-baseline 2>/dev/null | tail -1 >> $O/bench.txt
python tools/bench_configs.py 2>&1 | grep -v amdgpu > $O/bench_configs.txt
python tools/pool_bench.py 1024 4096 2>&1 | grep -v amdgpu > $O/reset_pool.txt
python tools/dbg/pacman_bench.py 2>&1 | grep -v amdgpu > $O/env_prefix.txt
bash tools/bench_ranks.sh > $O/bench_ranks.txt 2>&1
bash tools/prof.sh
cp gpurun_out/prof_summary.txt $O/prof_summary.txt
ls -la $O
