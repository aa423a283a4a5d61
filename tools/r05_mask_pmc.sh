# Round 5: instruction counters of the mask rasteriser truncated after each phase (debug stop), and its traced duration.
# usage (on a gpurun box): bash tools/r05_mask_pmc.sh [workload] [stops...]   (stops default 1 2 3 4 5 0)
# stops: 1 tables + colours | 2 + vertices | 3 + item scan | 4 + edges and census | 5 + rows | 0 = the whole kernel
R=$GRAFT_REPO_ROOT
WL=${1:-colliding_predators_32}; shift
cd /tmp && export TMPDIR=/tmp
O=$R/gpurun_out/r05_mask_pmc; mkdir -p $O
for stop in ${@:-1 2 3 4 5 0}; do
  rm -rf $O/p && mkdir -p $O/p
  MOOG_RASTER_STOP=$stop rocprofv3 --pmc SQ_WAVES SQ_INSTS_VALU SQ_INSTS_LDS SQ_INSTS_SALU SQ_WAVE_CYCLES SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_INSTS_SMEM -d $O/p/pmc_a -o r1 -- python3 $R/tools/raster_only.py $WL > $O/p/log1 2>&1
  MOOG_RASTER_STOP=$stop rocprofv3 --pmc SQ_ACTIVE_INST_VALU SQ_THREAD_CYCLES_VALU SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_INST_CYCLES_SALU SQ_ACTIVE_INST_LDS -d $O/p/pmc_b -o r1 -- python3 $R/tools/raster_only.py $WL > $O/p/log2 2>&1
  MOOG_RASTER_STOP=$stop rocprofv3 --kernel-trace -d $O/p/trace -o r1 -- python3 $R/tools/raster_only.py $WL > $O/p/log3 2>&1
  echo "== $WL stop $stop (per frame; trace avg in ns per launch)"
  python3 $R/tools/prof_summary.py $O/p | grep -E "raster_mask" | awk '{ if ($(NF-2) ~ /^[A-Z_]+$/) printf "%s=%.0f ", $(NF-2), $NF/4096; else printf "\ntrace: calls=%s avg_ns=%s ", $(NF-5), $(NF-3) } END {print ""}'
done
