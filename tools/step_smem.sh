# Scalar-memory behaviour of the step kernel (uniform loads of the lowered config go through the scalar data cache)
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
rm -rf $R/gpurun_out/prof_m && mkdir -p $R/gpurun_out/prof_m
rocprofv3 --pmc SQ_INSTS_SMEM SQ_WAIT_INST_ANY SQC_DCACHE_REQ SQC_DCACHE_MISSES SQC_DCACHE_HITS SQ_INST_CYCLES_SMEM -d $R/gpurun_out/prof_m/pmc_sq -o r1 -- python3 $R/bench.py --steps 6 --warmup 2 --no-cpu-baseline > $R/gpurun_out/prof_m/log1 2>&1
python3 $R/tools/prof_summary.py $R/gpurun_out/prof_m | grep -E "step_kernel|raster"
tail -3 $R/gpurun_out/prof_m/log1 | cut -c1-200
