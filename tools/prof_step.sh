cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
rm -rf $R/gpurun_out/prof_s && mkdir -p $R/gpurun_out/prof_s
rocprofv3 --pmc SQ_WAVES SQ_INSTS_VALU SQ_INSTS_LDS SQ_INSTS_SALU SQ_WAVE_CYCLES SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_WAIT_ANY -d $R/gpurun_out/prof_s/pmc_sq -o r1 -- python3 $R/bench.py --steps 10 --warmup 2 --no-cpu-baseline > $R/gpurun_out/prof_s/log1 2>&1
rocprofv3 --pmc SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_INSTS_SMEM SQ_INSTS_VMEM_RD SQ_INSTS_BRANCH SQ_INSTS_VMEM_WR SQ_WAIT_INST_LDS SQ_INSTS_VALU_TRANS -d $R/gpurun_out/prof_s/pmc_sq2 -o r1 -- python3 $R/bench.py --steps 10 --warmup 2 --no-cpu-baseline > $R/gpurun_out/prof_s/log2 2>&1
python3 $R/tools/prof_summary.py $R/gpurun_out/prof_s | grep -E "step|==" 
