#!/bin/bash
# Round 5: the mask rasteriser's duration against resident workgroups per CU (LDS padding) and batch size.
export TMPDIR=/tmp
for pad in ${PADS:-0 5000 7000 9500 13000 17000 25000}; do
  for n in 4096 ${EXTRA_N}; do
    echo -n "pad $pad envs $n: "
    MOOG_RASTER_LDS_PAD=$pad python bench.py --steps 80 --warmup 10 --no-cpu-baseline --no-extras --no-fused --envs-per-gpu $n 2>/dev/null | tail -1 | python -c "import sys,json; d=json.loads(sys.stdin.read()); print('raster us', round(d['kernels_avg_us']['raster'],2), 'step us', round(d['kernels_avg_us']['step'],1), 'value', int(d['value']))"
  done
done
