#!/bin/bash
# Round 5 iteration loop on one gpurun box: the raster tests, then the mask kernel's per-phase counters.
out=gpurun_out/r05_iter
mkdir -p $out
export TMPDIR=/tmp
python -m pytest tests/test_gpu_parity.py -x -q -m gpu -k "raster or frames_many_states or static_prefix" > $out/pytest.log 2>&1
echo "pytest rc=$?" >> $out/pytest.log
tail -4 $out/pytest.log
bash tools/r05_mask_pmc.sh ${1:-colliding_predators_32} ${2:-1 2 3 4 5 0} > $out/pmc.txt 2>&1
cat $out/pmc.txt
