# Builds the engine with -DMOOG_PROFILE (per-section cycle counters in the step kernel) next to the
# shipped library and prints the breakdown for the slowest envs.
cd $GRAFT_REPO_ROOT
/opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -std=c++17 -ffp-contract=off -fPIC -Wno-unused-value -DMOOG_PROFILE \
  -Iinclude -c moog.github.io_amd/csrc/moog_engine.hip -o gpurun_out/moog_engine_prof.o 2>&1 | grep -E "error"
/opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -fPIC gpurun_out/moog_engine_prof.o moog.github.io_amd/lib/moog_raster.o -o gpurun_out/libmoog_hip_prof.so
MOOG_HIP_LIB=$GRAFT_REPO_ROOT/gpurun_out/libmoog_hip_prof.so python tools/step_profile.py 2>&1 | grep -v amdgpu
