# round 4: the whole GPU suite, the default bench line, BASELINE config 5, the per-config table
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out
python -m pytest tests -m gpu -x -q 2>&1 | tail -4
python bench.py 2>/dev/null | tee gpurun_out/r04_bench_default.json | cut -c1-600
python bench.py --workload falling_balls_64 --envs-per-gpu 8192 --no-cpu-baseline --no-extras 2>/dev/null | tee gpurun_out/r04_bench_config5.json | cut -c1-400
python tools/bench_configs.py 2>&1 | grep -v amdgpu | tee gpurun_out/r04_bench_configs.txt
