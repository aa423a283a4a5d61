# Instruction / cycle counters of the wave rasteriser, with the stages behind a debug stop left out (where the issue slots go).
# usage: bash tools/wave_pmc.sh [stops...]   (default 3 4 5 0; 3 = no pushes / rows / compose, 4 = no rows / compose, 5 = no compose)
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
for stop in ${@:-2 3 4 5 0}; do
  rm -rf $R/gpurun_out/prof_w && mkdir -p $R/gpurun_out/prof_w
  MOOG_RASTER_STOP=$stop rocprofv3 --pmc SQ_WAVES SQ_INSTS_VALU SQ_INSTS_LDS SQ_INSTS_SALU SQ_WAVE_CYCLES SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_WAIT_ANY -d $R/gpurun_out/prof_w/pmc_sq -o r1 -- python3 $R/tools/raster_only.py > $R/gpurun_out/prof_w/log1 2>&1
  echo "== stop $stop (per frame)"
  python3 $R/tools/prof_summary.py $R/gpurun_out/prof_w | grep -E "raster_wave" | awk '{printf "%s=%.0f ", $(NF-2), $NF/4096} END {print ""}'
done
rm -rf $R/gpurun_out/prof_w && mkdir -p $R/gpurun_out/prof_w
rocprofv3 --pmc SQ_INSTS_VALU SQ_ACTIVE_INST_VALU SQ_THREAD_CYCLES_VALU SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_INSTS_SMEM SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR -d $R/gpurun_out/prof_w/pmc_sq -o r1 -- python3 $R/tools/raster_only.py > $R/gpurun_out/prof_w/log2 2>&1
echo "== lanes / LDS (per launch)"
python3 $R/tools/prof_summary.py $R/gpurun_out/prof_w | grep -E "raster_wave|drawlist"
rm -rf $R/gpurun_out/prof_w && mkdir -p $R/gpurun_out/prof_w
rocprofv3 --kernel-trace --stats -d $R/gpurun_out/prof_w/trace -o r1 -- python3 $R/tools/raster_only.py > $R/gpurun_out/prof_w/log3 2>&1
python3 $R/tools/prof_summary.py $R/gpurun_out/prof_w | head -12
