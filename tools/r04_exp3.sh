cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
mkdir -p $R/gpurun_out
rocprofv3 -L 2>/dev/null | grep -o -E "Name:\s*[A-Za-z0-9_]+" | sed 's/Name:\s*//' | sort -u | grep -E "^SQ_|^SQC_" | tr '\n' ' ' > $R/gpurun_out/r04_counters.txt
bash $R/tools/prof_icache.sh 2>&1 | tee $R/gpurun_out/r04_icache.txt
