# Builds the plain step kernel with -DMOOG_PROFILE (per-section cycle counters) next to the shipped
# library and prints the breakdown for the mean and the slowest envs.
cd $GRAFT_REPO_ROOT
L=moog.github.io_amd/lib
/opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -std=c++17 -ffp-contract=off -fPIC -Wno-unused-value -DMOOG_PROFILE \
  -DMOOG_STEP_DYN=0 -DMOOG_STEP_WPS=3 -DMOOG_STEP_TAG=f3 -c moog.github.io_amd/csrc/moog_step_inst.hip -o gpurun_out/moog_step_f3_prof.o 2>&1 | grep -E "error"
/opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -fPIC gpurun_out/moog_step_f3_prof.o $L/moog_step_f4.o $L/moog_step_t3.o $L/moog_step_t4.o $L/moog_step_m3.o $L/moog_step_m4.o \
  $L/moog_reset_r0.o $L/moog_reset_r1.o $L/moog_engine.o $L/moog_raster.o -o gpurun_out/libmoog_hip_prof.so
MOOG_HIP_LIB=$GRAFT_REPO_ROOT/gpurun_out/libmoog_hip_prof.so python tools/step_profile.py "$@" 2>&1 | grep -v amdgpu
