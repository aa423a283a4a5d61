# A/B builds of the plain step kernel (the f3 unit: 3 waves per SIMD, what the headline workload runs) next to the shipped
# library: bash tools/build_variant.sh NAME [extra hipcc flags]  ->  tools/ubench/build/libmoog_NAME.so
# (run with MOOG_HIP_LIB=<that path>; CPU only, ~2 minutes)
cd "$(dirname "$0")/.."
NAME=$1; shift
L=moog.github.io_amd/lib
B=tools/ubench/build
mkdir -p $B
/opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -std=c++17 -ffp-contract=off -fPIC -Wno-unused-value "$@" \
  -DMOOG_STEP_DYN=0 -DMOOG_STEP_WPS=3 -DMOOG_STEP_TAG=f3 -c moog.github.io_amd/csrc/moog_step_inst.hip -o $B/step_f3_$NAME.o 2>&1 | grep -E "error|warning: v" 
/opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -fPIC $B/step_f3_$NAME.o $L/moog_step_f2.o $L/moog_step_f4.o $L/moog_step_t3.o $L/moog_step_t4.o \
  $L/moog_step_m3.o $L/moog_step_m4.o $L/moog_reset_r0.o $L/moog_reset_r1.o $L/moog_engine.o $L/moog_raster.o -o $B/libmoog_$NAME.so && echo built $B/libmoog_$NAME.so
