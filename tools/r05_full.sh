#!/bin/bash
# Round 5: the whole GPU suite, smoke, and the default bench line on one gpurun box.  Output: gpurun_out/r05_full/
out=gpurun_out/r05_full
mkdir -p $out
export TMPDIR=/tmp
python -m pytest tests -x -q -m gpu -n 3 > $out/pytest.log 2>&1
echo "pytest rc=$?" >> $out/pytest.log
tail -6 $out/pytest.log
python -c "import __graft_entry__ as g; g.smoke()" > $out/smoke.log 2>&1; tail -1 $out/smoke.log
python bench.py > $out/bench.log 2>&1
tail -1 $out/bench.log
