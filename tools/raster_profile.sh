# Builds the library with -DMOOG_RASTER_PROFILE (phase clocks / work counters of the raster kernel, written
# instead of the frame) next to the shipped one and prints both for a workload: bash tools/raster_profile.sh [config [size]]
cd $GRAFT_REPO_ROOT
F="--offload-arch=gfx950 -O3 -std=c++17 -ffp-contract=off -fPIC -Wno-unused-value -DMOOG_RASTER_PROFILE"
/opt/rocm/bin/hipcc $F -c moog.github.io_amd/csrc/moog_raster.hip -o gpurun_out/moog_raster_prof.o 2>&1 | grep -E "error"
/opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -fPIC moog.github.io_amd/lib/moog_engine.o gpurun_out/moog_raster_prof.o -o gpurun_out/libmoog_hip_rprof.so
export MOOG_HIP_LIB=$GRAFT_REPO_ROOT/gpurun_out/libmoog_hip_rprof.so
python tools/raster_clocks.py "$@" 2>&1 | tail -2
python tools/raster_counts.py "$@" 2>&1 | tail -6
