#!/bin/bash
# Round 5: A/B of library variants (tools/build_raster_variant.sh) on one box: raster kernel us of the headline bench.
# usage: bash tools/r05_ab.sh NAME...   ("base" = the shipped library)
export TMPDIR=/tmp
for rep in 1 2; do
for v in "$@"; do
  lib=""; [ "$v" != base ] && lib=$PWD/tools/ubench/build/libmoog_$v.so
  echo -n "$v: "
  env ${lib:+MOOG_HIP_LIB=$lib} python bench.py --steps ${STEPS:-100} --warmup 10 --no-cpu-baseline --no-extras --no-fused ${BENCH_ARGS} 2>/dev/null | tail -1 | python -c "import sys,json; d=json.loads(sys.stdin.read()); print('raster us', round(d['kernels_avg_us']['raster'],2), 'step us', round(d['kernels_avg_us']['step'],1), 'value', int(d['value']))"
done
done
