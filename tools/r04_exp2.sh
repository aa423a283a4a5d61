# round 4, experiment 2: latency floor (small batches), HW queues for the sub-batch streams, section profile
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out
python tools/r04_exp.py --args "--no-fused --no-cpu-baseline --no-extras --steps 100 --warmup 10" \
  'n256=@--envs-per-gpu 256' 'n512=@--envs-per-gpu 512' 'n1024=@--envs-per-gpu 1024' \
  'sub4_q8=GPU_MAX_HW_QUEUES=8;@--sub-batches 4' 'sub4_q16=GPU_MAX_HW_QUEUES=16;@--sub-batches 4' 'sub8_q16=GPU_MAX_HW_QUEUES=16;@--sub-batches 8' \
  'sub2_q8=GPU_MAX_HW_QUEUES=8;@--sub-batches 2' 'sub2_8192=GPU_MAX_HW_QUEUES=8;@--sub-batches 2 --envs-per-gpu 8192' 'sub4_8192=GPU_MAX_HW_QUEUES=16;@--sub-batches 4 --envs-per-gpu 8192' \
  2>&1 | tee gpurun_out/r04_exp2.txt
bash tools/step_profile.sh 2>&1 | tee gpurun_out/r04_step_sections_a.txt
