# round 4, experiment 1: wave priorities by launch rank, the two-waves-per-SIMD register variant, batch-size sweep, sub-batches
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out
python -m pytest tests/test_gpu_parity.py -k "sub_batches" -x -q 2>&1 | tail -5
python tools/r04_exp.py --args "--no-fused --no-cpu-baseline --no-extras --steps 100 --warmup 10" \
  base= \
  'prio_q=MOOG_STEP_PRIO=250,500,750' 'prio_8=MOOG_STEP_PRIO=125,250,500' 'prio_top=MOOG_STEP_PRIO=250,250,250' 'prio_16=MOOG_STEP_PRIO=60,125,250' \
  'wps2=MOOG_STEP_WPS=2' 'wps2_prio=MOOG_STEP_WPS=2;MOOG_STEP_PRIO=250,500,750' 'wps4_prio=MOOG_STEP_WPS=4;MOOG_STEP_PRIO=250,500,750' \
  'n2048=@--envs-per-gpu 2048' 'n8192=@--envs-per-gpu 8192' 'n16384=@--envs-per-gpu 16384' \
  'sub2=@--sub-batches 2' 'sub4=@--sub-batches 4' 'sub8=@--sub-batches 8' 'sub4_prio=MOOG_STEP_PRIO=250,500,750;@--sub-batches 4' \
  base2= 2>&1 | tee gpurun_out/r04_exp1.txt
