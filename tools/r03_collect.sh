# Round-3 evidence, one gpurun call: bench lines, per-config table, rank hook, harness phases, tails, the wave rasteriser's stages.
cd $GRAFT_REPO_ROOT
O=gpurun_out/r03; rm -rf $O; mkdir -p $O
echo "== default bench (python bench.py)" > $O/bench.txt
python bench.py 2>/dev/null | tail -1 >> $O/bench.txt
echo "== python bench.py --no-fused" >> $O/bench.txt
python bench.py --no-fused --no-cpu-baseline 2>/dev/null | tail -1 >> $O/bench.txt
echo "== MOOG_RASTER_DL=1 python bench.py --no-fused" >> $O/bench.txt
MOOG_RASTER_DL=1 python bench.py --no-fused --no-cpu-baseline --no-extras 2>/dev/null | tail -1 >> $O/bench.txt
echo "== MOOG_RASTER_WAVE=1 python bench.py --no-fused" >> $O/bench.txt
MOOG_RASTER_WAVE=1 python bench.py --no-fused --no-cpu-baseline --no-extras 2>/dev/null | tail -1 >> $O/bench.txt
echo "== python bench.py --workload falling_balls_64 --envs-per-gpu 8192 --steps 60 --warmup 10" >> $O/bench.txt
python bench.py --workload falling_balls_64 --envs-per-gpu 8192 --steps 60 --warmup 10 --no-cpu-baseline 2>/dev/null | tail -1 >> $O/bench.txt
python tools/bench_configs.py > $O/bench_configs.txt 2>&1
bash tools/bench_ranks.sh > $O/bench_ranks.txt 2>&1
( cd moog.github.io_amd && python -m moog_demos.runtime_benchmark --config pong --num_envs 1 --reps 200; python -m moog_demos.runtime_benchmark --config pong --num_envs 4096; python -m moog_demos.runtime_benchmark --config colliding_predators_32 --num_envs 4096 --render_sizes ) > $O/runtime_benchmark.txt 2>&1
python tools/step_tail.py colliding_predators_32 4096 60 2>&1 | grep -v amdgpu > $O/step_tail.txt
python tools/step_tail.py falling_balls_64 8192 130 2>&1 | grep -v amdgpu | tail -8 >> $O/step_tail.txt
MOOG_RASTER_WAVE=1 bash tools/wave_pmc.sh > $O/wave_pmc.txt 2>&1
MOOG_RASTER_WAVE=1 bash tools/wave_occupancy.sh > $O/wave_occupancy.txt 2>&1
bash tools/prof.sh
cp gpurun_out/prof_summary.txt $O/prof_summary.txt
ls -la $O
