# instruction mix of the heavy envs replayed alone (one wavefront per CU): what a lone wave executes per env-step
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
rm -rf $R/gpurun_out/prof_h && mkdir -p $R/gpurun_out/prof_h
export HEAVY_ONLY=${HEAVY_ONLY:-256}
rocprofv3 --pmc SQ_WAVES SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_INSTS_SMEM SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_INSTS_BRANCH -d $R/gpurun_out/prof_h/pmc1 -o r1 -- python3 $R/tools/heavy_bench.py bench > $R/gpurun_out/prof_h/log1 2>&1
rocprofv3 --pmc SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_INST_ANY SQ_WAIT_INST_LDS SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_ACTIVE_INST_SCA -d $R/gpurun_out/prof_h/pmc2 -o r1 -- python3 $R/tools/heavy_bench.py bench > $R/gpurun_out/prof_h/log2 2>&1
rocprofv3 --pmc SQ_INST_CYCLES_VMEM_RD SQ_INST_CYCLES_VMEM_WR SQ_INST_CYCLES_SMEM SQ_INST_CYCLES_SALU SQ_THREAD_CYCLES_VALU SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_ACTIVE_INST_MISC -d $R/gpurun_out/prof_h/pmc3 -o r1 -- python3 $R/tools/heavy_bench.py bench > $R/gpurun_out/prof_h/log3 2>&1
python3 $R/tools/prof_summary.py $R/gpurun_out/prof_h | grep -E "step_kernel|=="
tail -2 $R/gpurun_out/prof_h/log3 | cut -c1-300
