#!/bin/bash
# Round 5: resident workgroups per CU (LDS padding) x register budget (library variants) for the mask rasteriser.
export TMPDIR=/tmp
for v in "$@"; do
  lib=""; [ "$v" != base ] && lib=$PWD/tools/ubench/build/libmoog_$v.so
  for pad in ${PADS:-0 1024 2048 3072 4096}; do
    echo -n "$v pad $pad: "
    env ${lib:+MOOG_HIP_LIB=$lib} MOOG_RASTER_LDS_PAD=$pad python bench.py --steps 100 --warmup 10 --no-cpu-baseline --no-extras --no-fused 2>/dev/null | tail -1 | python -c "import sys,json; d=json.loads(sys.stdin.read()); print('raster us', round(d['kernels_avg_us']['raster'],2))"
  done
done
