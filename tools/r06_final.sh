#!/bin/bash
# final build of round 6: the whole GPU suite, the headline bench, a kernel trace (the sched kernel's duration), a schedule sanity check
cd "$GRAFT_REPO_ROOT" || exit 1
O=gpurun_out/r06final; rm -rf $O; mkdir -p $O
python -m pytest tests -m gpu -x -q -n 4 > $O/pytest.log 2>&1; echo "pytest rc=$?" >> $O/pytest.log
tail -4 $O/pytest.log
line() { echo "== $*" >> $O/bench.txt; "$@" 2>/dev/null | tail -1 >> $O/bench.txt; }
line python bench.py
line python bench.py --no-cpu-baseline --no-schedule
line python bench.py --no-cpu-baseline --workload pong
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
rocprofv3 --kernel-trace --stats -d $R/gpurun_out/prof/trace -o r1 -- python3 $R/bench.py --no-cpu-baseline --no-extras --steps 60 --warmup 5 > $R/$O/trace.log 2>&1
rocprofv3 --kernel-trace --stats -d $R/gpurun_out/prof/trace_pong -o r1 -- python3 $R/bench.py --no-cpu-baseline --no-extras --steps 60 --warmup 5 --workload pong > $R/$O/trace_pong.log 2>&1
python3 $R/tools/prof_summary.py $R/gpurun_out/prof > $R/$O/prof_summary.txt
find $R/gpurun_out/prof -name '*.db' -delete
