# A/B builds of the raster + engine units next to the shipped library (round 5):
#   bash tools/build_raster_variant.sh NAME [extra hipcc flags, e.g. -DRM_THREADS=256 -DRM_WAVES_PER_SIMD=5]  -> tools/ubench/build/libmoog_NAME.so
# Run with MOOG_HIP_LIB=<that path>.  CPU only, about half a minute.
cd "$(dirname "$0")/.."
NAME=$1; shift
L=moog.github.io_amd/lib
B=tools/ubench/build
mkdir -p $B
F="--offload-arch=gfx950 -O3 -std=c++17 -ffp-contract=off -fPIC -Wno-unused-value"
/opt/rocm/bin/hipcc $F "$@" -c moog.github.io_amd/csrc/moog_raster.hip -o $B/raster_$NAME.o &
/opt/rocm/bin/hipcc $F "$@" -c moog.github.io_amd/csrc/moog_engine.hip -o $B/engine_$NAME.o &
wait
OBJS=""
for u in f2 f3 f4 t3 t4 m3 m4; do OBJS="$OBJS $L/moog_step_$u.o"; done
/opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -fPIC $OBJS $L/moog_reset_r0.o $L/moog_reset_r1.o $B/engine_$NAME.o $B/raster_$NAME.o \
  -o $B/libmoog_$NAME.so && echo built $B/libmoog_$NAME.so
