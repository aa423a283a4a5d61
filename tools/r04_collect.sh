# Round-4 evidence, one gpurun call on the final build: bench lines, launch-structure and batch-size sweeps, per-config table,
# rank hook, tails, rocprofv3 trace + PMC passes, watcher section profiles, function costs round 3 vs round 4, unit latencies.
# (The -DMOOG_WATCH library and the fn_bench libraries under tools/ubench/build are built per ABI version -- tools/build_variant.sh,
#  tools/fn_bench.sh: rebuild them before re-running this after include/moog_engine.h changed; tools/r04_collect2.sh is the part that
#  needs no extra builds and was re-run on the final build of the round.)
cd $GRAFT_REPO_ROOT
O=gpurun_out/r04; rm -rf $O; mkdir -p $O
W=$GRAFT_REPO_ROOT/tools/ubench/build
echo "== default bench (python bench.py)" > $O/bench.txt
python bench.py 2>/dev/null | tail -1 >> $O/bench.txt
echo "== python bench.py --no-fused --no-cpu-baseline" >> $O/bench.txt
python bench.py --no-fused --no-cpu-baseline 2>/dev/null | tail -1 >> $O/bench.txt
echo "== GPU_MAX_HW_QUEUES=8 python bench.py --sub-batches 2 --no-cpu-baseline" >> $O/bench.txt
GPU_MAX_HW_QUEUES=8 python bench.py --sub-batches 2 --no-cpu-baseline 2>/dev/null | tail -1 >> $O/bench.txt
echo "== python bench.py --workload falling_balls_64 --envs-per-gpu 8192 --no-cpu-baseline" >> $O/bench.txt
python bench.py --workload falling_balls_64 --envs-per-gpu 8192 --no-cpu-baseline 2>/dev/null | tail -1 >> $O/bench.txt
python tools/r04_exp.py --args "--no-fused --no-cpu-baseline --no-extras --steps 100 --warmup 10" \
  'n1024=@--envs-per-gpu 1024' 'n2048=@--envs-per-gpu 2048' 'n4096=' 'n8192=@--envs-per-gpu 8192' 'n16384=@--envs-per-gpu 16384' \
  'sub2=GPU_MAX_HW_QUEUES=8;@--sub-batches 2' 'sub4=GPU_MAX_HW_QUEUES=16;@--sub-batches 4' 'sub8=GPU_MAX_HW_QUEUES=16;@--sub-batches 8' \
  'sub2_8192=GPU_MAX_HW_QUEUES=8;@--sub-batches 2 --envs-per-gpu 8192' 'sub2_16384=GPU_MAX_HW_QUEUES=8;@--sub-batches 2 --envs-per-gpu 16384' \
  'wps4=MOOG_STEP_WPS=4' 'wps2=MOOG_STEP_WPS=2' 'prio=MOOG_STEP_PRIO=250,500,750' > $O/sweeps.txt 2>&1
python tools/bench_configs.py > $O/bench_configs.txt 2>&1
bash tools/bench_ranks.sh > $O/bench_ranks.txt 2>&1
python tools/step_tail.py colliding_predators_32 4096 60 2>&1 | grep -v amdgpu > $O/step_tail.txt
python tools/step_tail.py falling_balls_64 8192 130 2>&1 | grep -v amdgpu | tail -8 >> $O/step_tail.txt
for wl in colliding_predators_32 falling_balls_64; do
  echo "== heavy envs of $wl, shipped library" >> $O/sections.txt
  python tools/heavy_bench.py bench $wl 2>&1 | grep -v amdgpu >> $O/sections.txt
  echo "== heavy envs of $wl, -DMOOG_WATCH build (section samples)" >> $O/sections.txt
  MOOG_WATCH=1 HEAVY_ONLY=256 MOOG_HIP_LIB=$W/libmoog_watch.so python tools/heavy_bench.py bench $wl 2>&1 | grep -v amdgpu >> $O/sections.txt
done
echo "== a random sample of colliding_predators_32 envs, -DMOOG_WATCH build" >> $O/sections.txt
RANDOM_SAMPLE=1 HEAVY_ONLY=256 python tools/heavy_bench.py bench 2>&1 | grep -v amdgpu >> $O/sections.txt
MOOG_WATCH=1 RANDOM_SAMPLE=1 HEAVY_ONLY=256 MOOG_HIP_LIB=$W/libmoog_watch.so python tools/heavy_bench.py bench 2>&1 | grep -v amdgpu >> $O/sections.txt
for lib in libfn_bench_r03 libfn_bench; do
  echo "== $lib: heavy envs" >> $O/fn_bench.txt
  FN_BENCH_LIB=$W/$lib.so FN_ENVS=256 FN_ONLY=0,1,2,4,5,13,10,11 python tools/fn_bench.py 2>&1 | grep -v amdgpu >> $O/fn_bench.txt
  echo "== $lib: random sample of envs" >> $O/fn_bench.txt
  FN_BENCH_LIB=$W/$lib.so RANDOM_SAMPLE=1 FN_ENVS=256 FN_ONLY=5,13,10,11 python tools/fn_bench.py 2>&1 | grep -v amdgpu >> $O/fn_bench.txt
done
$W/latency > $O/latency.txt 2>&1
bash tools/r04_heavy_pmc.sh > $O/heavy_pmc.txt 2>&1
( cd moog.github.io_amd && python -m moog_demos.runtime_benchmark --config pong --num_envs 1 --reps 200; python -m moog_demos.runtime_benchmark --config pong --num_envs 4096; python -m moog_demos.runtime_benchmark --config colliding_predators_32 --num_envs 4096 --render_sizes ) > $O/runtime_benchmark.txt 2>&1
bash tools/prof.sh
cp gpurun_out/prof_summary.txt $O/prof_summary.txt
ls -la $O
