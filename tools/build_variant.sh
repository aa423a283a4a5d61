# A/B builds of ONE step-kernel unit next to the shipped library:
#   bash tools/build_variant.sh NAME [extra hipcc flags]      -> tools/ubench/build/libmoog_NAME.so
# The unit is f3 (plain programs, 3 waves per SIMD: what the headline workload runs) unless UNIT=t3|m3|... DYN=1|2 WPS=3|4 are set.
# Run with MOOG_HIP_LIB=<that path>; CPU only, ~1 minute for f3, ~4 for t3 / m3.
cd "$(dirname "$0")/.."
NAME=$1; shift
UNIT=${UNIT:-f3}; DYN=${DYN:-0}; WPS=${WPS:-3}
L=moog.github.io_amd/lib
B=tools/ubench/build
mkdir -p $B
/opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -std=c++17 -ffp-contract=off -fPIC -Wno-unused-value "$@" \
  -DMOOG_STEP_DYN=$DYN -DMOOG_STEP_WPS=$WPS -DMOOG_STEP_TAG=$UNIT -c moog.github.io_amd/csrc/moog_step_inst.hip -o $B/step_${UNIT}_$NAME.o 2>&1 | grep -E "error|warning: v"
OBJS=""
for u in f2 f3 f4 t3 t4 m3 m4; do
  if [ $u = $UNIT ]; then OBJS="$OBJS $B/step_${UNIT}_$NAME.o"; else OBJS="$OBJS $L/moog_step_$u.o"; fi
done
/opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -fPIC $OBJS $L/moog_reset_r0.o $L/moog_reset_r1.o $L/moog_engine.o $L/moog_raster.o \
  -o $B/libmoog_$NAME.so && echo built $B/libmoog_$NAME.so
