# Instruction counters of the raster kernel truncated after each phase (debug stop): where the issue slots go.
# usage: bash tools/raster_pmc.sh [stops...]   (default 1 2 3 4 5 0)
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
for stop in ${@:-1 2 3 4 5 0}; do
  rm -rf $R/gpurun_out/prof_r && mkdir -p $R/gpurun_out/prof_r
  MOOG_RASTER_STOP=$stop rocprofv3 --pmc SQ_WAVES SQ_INSTS_VALU SQ_INSTS_LDS SQ_INSTS_SALU SQ_WAVE_CYCLES SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_WAIT_ANY -d $R/gpurun_out/prof_r/pmc_sq -o r1 -- python3 $R/tools/raster_only.py > $R/gpurun_out/prof_r/log1 2>&1
  echo "== stop $stop"
  python3 $R/tools/prof_summary.py $R/gpurun_out/prof_r | grep -E "raster" | awk '{printf "%s=%.0f ", $(NF-2), $NF/4096} END {print ""}'
done
