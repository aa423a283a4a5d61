cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
rm -rf $R/gpurun_out/prof && mkdir -p $R/gpurun_out/prof
rocprofv3 --kernel-trace --stats -d $R/gpurun_out/prof/trace -o r1 -- python3 $R/bench.py --steps 30 --warmup 5 --no-cpu-baseline > $R/gpurun_out/prof/bench_trace.log 2>&1
rocprofv3 --pmc FETCH_SIZE -d $R/gpurun_out/prof/pmc_fetch -o r1 -- python3 $R/bench.py --steps 10 --warmup 2 --no-cpu-baseline > $R/gpurun_out/prof/bench_pmc1.log 2>&1
rocprofv3 --pmc WRITE_SIZE -d $R/gpurun_out/prof/pmc_write -o r1 -- python3 $R/bench.py --steps 10 --warmup 2 --no-cpu-baseline > $R/gpurun_out/prof/bench_pmc2.log 2>&1
rocprofv3 --pmc SQ_WAVES SQ_INSTS_VALU SQ_INSTS_LDS SQ_INSTS_SALU SQ_WAVE_CYCLES SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_WAIT_ANY -d $R/gpurun_out/prof/pmc_sq -o r1 -- python3 $R/bench.py --steps 10 --warmup 2 --no-cpu-baseline > $R/gpurun_out/prof/bench_pmc3.log 2>&1
find $R/gpurun_out/prof -type f | head -50
du -sh $R/gpurun_out/prof
