# As tools/step_profile.sh, for another kernel variant: bash tools/step_profile_variant.sh m3 2 3 pacman
# (tag, MOOG_STEP_DYN, MOOG_STEP_WPS, config).  The profiled object replaces that variant in a copy of the library.
cd $GRAFT_REPO_ROOT
L=moog.github.io_amd/lib
TAG=$1; DYN=$2; WPS=$3; shift 3
/opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -std=c++17 -ffp-contract=off -fPIC -Wno-unused-value -DMOOG_PROFILE \
  -DMOOG_STEP_DYN=$DYN -DMOOG_STEP_WPS=$WPS -DMOOG_STEP_TAG=$TAG -c moog.github.io_amd/csrc/moog_step_inst.hip -o gpurun_out/moog_step_${TAG}_prof.o 2>&1 | grep -E "error"
OBJS=""
for t in f3 f4 t3 t4 m3 m4; do
  if [ $t = $TAG ]; then OBJS="$OBJS gpurun_out/moog_step_${TAG}_prof.o"; else OBJS="$OBJS $L/moog_step_$t.o"; fi
done
/opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -fPIC $OBJS $L/moog_reset_r0.o $L/moog_reset_r1.o $L/moog_engine.o $L/moog_raster.o -o gpurun_out/libmoog_hip_prof.so
MOOG_HIP_LIB=$GRAFT_REPO_ROOT/gpurun_out/libmoog_hip_prof.so MOOG_PROFILE_ENVS=${MOOG_PROFILE_ENVS:-1024} python ${PROF_SCRIPT:-tools/step_profile.py} "$@" 2>&1 | grep -v amdgpu
