cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
rm -rf $R/gpurun_out/pcs && mkdir -p $R/gpurun_out/pcs
export HEAVY_ONLY=${HEAVY_ONLY:-256}
export MOOG_HIP_LIB=$R/tools/ubench/build/libmoog_dbg.so
rocprofv3 -L > $R/gpurun_out/pcs/avail.txt 2>&1
grep -i -A12 "pc.sampling" $R/gpurun_out/pcs/avail.txt | head -40
rocprofv3 --pc-sampling-beta-enabled --pc-sampling-unit cycles --pc-sampling-method stochastic --pc-sampling-interval 65536 --kernel-trace --output-format csv json -d $R/gpurun_out/pcs/st -o r1 -- python3 $R/tools/heavy_bench.py bench > $R/gpurun_out/pcs/log_st 2>&1
tail -3 $R/gpurun_out/pcs/log_st | cut -c1-300
ls -la $R/gpurun_out/pcs/st/* | head
rocprofv3 --pc-sampling-beta-enabled --pc-sampling-unit time --pc-sampling-method host_trap --pc-sampling-interval 1 --kernel-trace --output-format csv -d $R/gpurun_out/pcs/ht -o r1 -- python3 $R/tools/heavy_bench.py bench > $R/gpurun_out/pcs/log_ht 2>&1
tail -3 $R/gpurun_out/pcs/log_ht | cut -c1-300
ls -la $R/gpurun_out/pcs/ht/* | head
