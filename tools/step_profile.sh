# Builds the engine with -DMOOG_PROFILE (per-section cycle counters in the step kernel) next to the
# shipped library and prints the breakdown for the slowest envs.
cd $GRAFT_REPO_ROOT
/opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -std=c++17 -ffp-contract=off -fPIC -shared -DMOOG_PROFILE \
  -Iinclude moog.github.io_amd/csrc/moog_engine.hip -o gpurun_out/libmoog_hip_prof.so 2>&1 | grep -E "error" 
MOOG_HIP_LIB=$GRAFT_REPO_ROOT/gpurun_out/libmoog_hip_prof.so python tools/step_profile.py 2>&1 | grep -v amdgpu
