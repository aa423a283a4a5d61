# rocprofv3 evidence for the bench workload: kernel trace + stats, then separate PMC passes (FETCH_SIZE | WRITE_SIZE | SQ_*).
# usage (GPU box): bash tools/prof.sh ; python3 tools/prof_summary.py gpurun_out/prof > gpurun_out/prof_summary.txt
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
B="$R/bench.py --no-cpu-baseline --no-extras"
rm -rf $R/gpurun_out/prof && mkdir -p $R/gpurun_out/prof
rocprofv3 --kernel-trace --stats -d $R/gpurun_out/prof/trace -o r1 -- python3 $B --steps 60 --warmup 5 > $R/gpurun_out/prof/bench_trace.log 2>&1
rocprofv3 --pmc FETCH_SIZE -d $R/gpurun_out/prof/pmc_fetch -o r1 -- python3 $B --steps 10 --warmup 2 > $R/gpurun_out/prof/bench_pmc1.log 2>&1
rocprofv3 --pmc WRITE_SIZE -d $R/gpurun_out/prof/pmc_write -o r1 -- python3 $B --steps 10 --warmup 2 > $R/gpurun_out/prof/bench_pmc2.log 2>&1
rocprofv3 --pmc SQ_WAVES SQ_INSTS_VALU SQ_INSTS_LDS SQ_INSTS_SALU SQ_WAVE_CYCLES SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_WAIT_ANY -d $R/gpurun_out/prof/pmc_sq -o r1 -- python3 $B --steps 10 --warmup 2 > $R/gpurun_out/prof/bench_pmc3.log 2>&1
rocprofv3 --pmc SQ_INSTS_VALU SQ_ACTIVE_INST_VALU SQ_THREAD_CYCLES_VALU SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_BUSY_CYCLES -d $R/gpurun_out/prof/pmc_lanes -o r1 -- python3 $B --steps 10 --warmup 2 > $R/gpurun_out/prof/bench_pmc4.log 2>&1
python3 $R/tools/prof_summary.py $R/gpurun_out/prof > $R/gpurun_out/prof_summary.txt
find $R/gpurun_out/prof -name '*.db' -delete   # (the databases are tens of MB each: only the summary travels back)
