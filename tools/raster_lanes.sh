# VALU lane utilisation of the raster kernel: thread-cycles vs instruction-cycles
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
rm -rf $R/gpurun_out/prof_l && mkdir -p $R/gpurun_out/prof_l
rocprofv3 --pmc SQ_INSTS_VALU SQ_ACTIVE_INST_VALU SQ_THREAD_CYCLES_VALU SQ_INST_CYCLES_SALU SQ_ACTIVE_INST_SCA SQ_ACTIVE_INST_LDS SQ_BUSY_CYCLES -d $R/gpurun_out/prof_l/pmc_sq -o r1 -- python3 $R/bench.py --steps 6 --warmup 2 --no-cpu-baseline > $R/gpurun_out/prof_l/log1 2>&1
python3 $R/tools/prof_summary.py $R/gpurun_out/prof_l | grep -E "raster|step_kernel"
tail -3 $R/gpurun_out/prof_l/log1 | cut -c1-300
