# dynamic instruction mix of ONE device function (tools/fn_bench.py, FN_ONLY=<id>), lone waves: counters / (256 envs x iters)
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
rm -rf $R/gpurun_out/prof_f && mkdir -p $R/gpurun_out/prof_f
export FN_ENVS=256
for fn in "$@"; do
export FN_ONLY=$fn
rocprofv3 --pmc SQ_WAVES SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_INSTS_SMEM SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_INSTS_BRANCH -d $R/gpurun_out/prof_f/pmc$fn -o r1 -- python3 $R/tools/fn_bench.py > $R/gpurun_out/prof_f/log$fn 2>&1
grep "cycles per call" $R/gpurun_out/prof_f/log$fn
python3 - <<PY
import sqlite3
c = sqlite3.connect('$R/gpurun_out/prof_f/pmc$fn/r1_results.db')
cols = [r[1] for r in c.execute("pragma table_info(counters_collection)")]
rows = c.execute("select counter_name, value from counters_collection where kernel_name like '%fn_bench%' order by rowid").fetchall()
# launches in order: find, warm (2 iters), timed (iters): keep the last value of each counter
last = {}
for k, v in rows: last[k] = v
iters = {0: 200, 1: 200, 2: 200, 4: 50, 5: 10, 9: 2000}[$fn]
print('   per call:', ', '.join('%s %.0f' % (k.replace('SQ_INSTS_', ''), v / 256 / iters) for k, v in sorted(last.items()) if k != 'SQ_WAVES'))
PY
done
