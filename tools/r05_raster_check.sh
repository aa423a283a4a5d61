#!/bin/bash
# Round 5: the raster tests + a bench line + the raster kernel's trace, on one gpurun box.  Output: gpurun_out/r05_raster_check/
out=gpurun_out/r05_raster_check
mkdir -p $out
export TMPDIR=/tmp
python -m pytest tests/test_gpu_parity.py -x -q -m gpu -k "raster or frames_many_states or static_prefix or smoke or full_size_vs_oracle" > $out/pytest.log 2>&1
echo "pytest rc=$?" >> $out/pytest.log
tail -5 $out/pytest.log
python bench.py --steps 200 --warmup 20 > $out/bench.log 2>&1
tail -2 $out/bench.log
MOOG_RASTER_MASK=0 python bench.py --steps 200 --warmup 20 > $out/bench_old.log 2>&1
tail -1 $out/bench_old.log
