cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
for st in 0 16 8; do
rm -rf $R/gpurun_out/prof_s && mkdir -p $R/gpurun_out/prof_s
MOOG_STEP_DEBUG=$st rocprofv3 --pmc SQ_WAVES SQ_INSTS_VALU SQ_INSTS_LDS SQ_INSTS_SALU SQ_WAVE_CYCLES SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_WAIT_ANY -d $R/gpurun_out/prof_s/pmc_sq -o r1 -- python3 $R/bench.py --steps 10 --warmup 2 --no-cpu-baseline > $R/gpurun_out/prof_s/log1 2>&1
echo "== dbg $st"; python3 $R/tools/prof_summary.py $R/gpurun_out/prof_s | grep -E "step_kernel" | awk '{print $2, $3, $4/4096}'
done
