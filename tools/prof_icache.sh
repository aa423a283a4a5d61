cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
rm -rf $R/gpurun_out/prof_i && mkdir -p $R/gpurun_out/prof_i
rocprofv3 --pmc SQC_ICACHE_REQ SQC_ICACHE_HITS SQC_ICACHE_MISSES SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_IFETCH -d $R/gpurun_out/prof_i/pmc -o r1 -- python3 $R/bench.py --steps 10 --warmup 2 --no-cpu-baseline > $R/gpurun_out/prof_i/log1 2>&1
python3 $R/tools/prof_summary.py $R/gpurun_out/prof_i | grep -E "moog"
tail -3 $R/gpurun_out/prof_i/log1 | cut -c1-200
