# Wave rasteriser time vs resident workgroups per CU (LDS padding) and vs row-record capacity.
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
run() {
  rm -rf $R/gpurun_out/prof_o && mkdir -p $R/gpurun_out/prof_o
  env "$@" rocprofv3 --kernel-trace --stats -d $R/gpurun_out/prof_o/trace -o r1 -- python3 $R/tools/raster_only.py > $R/gpurun_out/prof_o/log 2>&1
  echo "$@ :" $(python3 $R/tools/prof_summary.py $R/gpurun_out/prof_o | grep -E "raster_wave|moog_raster_kernel" | awk '{print $(NF-3)}')
}
run X=0
for pad in 3000 6000 10000 15000 24000 36000; do run MOOG_WAVE_LDS_PAD=$pad; done
for rows in 192 160 128; do run MOOG_RASTER_ROWS=$rows; done
run MOOG_WAVE_EDGE_ROUNDS=3
