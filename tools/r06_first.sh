#!/bin/bash
# round 6, first GPU call: the whole GPU suite on the build with source digests, then the bench lines of the BASELINE configs
# by phase and the sched kernel's time (item 9)
cd "$GRAFT_REPO_ROOT" || exit 1
mkdir -p gpurun_out/r06a
python -m pytest tests -m gpu -x -q -n 4 > gpurun_out/r06a/pytest.log 2>&1; echo "pytest rc=$?" >> gpurun_out/r06a/pytest.log
tail -5 gpurun_out/r06a/pytest.log
{
  for args in "" "--workload chase_avoid_torus --phase step" "--workload chase_avoid_torus" "--workload functional_maze@128 --envs-per-gpu 8192" "--workload falling_balls_64 --envs-per-gpu 8192" "--phase render" "--phase physics"; do
    echo "== python bench.py --no-cpu-baseline $args"
    python bench.py --no-cpu-baseline $args 2>&1 | tail -1
  done
} > gpurun_out/r06a/bench.log 2>&1
cd /tmp && export TMPDIR=/tmp
rocprofv3 --kernel-trace --stats -d "$GRAFT_REPO_ROOT/gpurun_out/r06a/prof" -o trace -- python3 "$GRAFT_REPO_ROOT/bench.py" --no-cpu-baseline --no-extras --steps 100 > "$GRAFT_REPO_ROOT/gpurun_out/r06a/prof.log" 2>&1
cd "$GRAFT_REPO_ROOT"; find gpurun_out/r06a/prof -name "*kernel_stats*" | head -3 | xargs -I{} sh -c 'head -12 {}' > gpurun_out/r06a/kernel_stats.txt 2>&1
find gpurun_out/r06a/prof -type f -size +2M -delete
