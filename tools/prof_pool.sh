# rocprofv3 kernel trace of the reset pool at work: bounce_box_contact_prediction, 1024 envs, 700 calls (tools/pool_bench.py's pooled run).
# usage (GPU box): bash tools/prof_pool.sh   -> gpurun_out/prof_pool_summary.txt
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
rm -rf $R/gpurun_out/prof_pool && mkdir -p $R/gpurun_out/prof_pool
rocprofv3 --kernel-trace --stats -d $R/gpurun_out/prof_pool/trace -o r1 -- python3 $R/tools/pool_bench.py only bounce_box_contact_prediction 1024 > $R/gpurun_out/prof_pool/run.log 2>&1
python3 $R/tools/prof_summary.py $R/gpurun_out/prof_pool > $R/gpurun_out/prof_pool_summary.txt
python3 - >> $R/gpurun_out/prof_pool_summary.txt <<PY
import glob, sqlite3
db = glob.glob('$R/gpurun_out/prof_pool/trace/*_results.db')[0]
c = sqlite3.connect(db)
rows = c.execute("select name, start, end from kernels where name like '%moog_reset_kernel%' or name like '%moog_step_kernel%' order by start").fetchall()
fills = [(s, e) for n, s, e in rows if 'reset' in n]
steps = [(s, e) for n, s, e in rows if 'step' in n]
# how many fills overlap each step kernel; the fills that did work (longer than 1 ms)
long_f = [(s, e) for s, e in fills if e - s > 1_000_000]
import bisect
print('== overlap (trace)')
print('fill launches %d, of which %d longer than 1 ms (avg %.2f ms); step launches %d (avg %.1f us, p50 %.1f us, max %.1f us)' % (
    len(fills), len(long_f), sum(e - s for s, e in long_f) / max(1, len(long_f)) / 1e6, len(steps),
    sum(e - s for s, e in steps) / max(1, len(steps)) / 1e3, sorted(e - s for s, e in steps)[len(steps) // 2] / 1e3, max(e - s for s, e in steps) / 1e3))
ov = [sum(1 for fs, fe in long_f if fs < e and fe > s) for s, e in steps[len(steps) // 2:]]
print('fills (> 1 ms) running during a step kernel of the second half of the run: mean %.1f, max %d' % (sum(ov) / max(1, len(ov)), max(ov)))
PY
cat $R/gpurun_out/prof_pool/run.log | grep -v amdgpu >> $R/gpurun_out/prof_pool_summary.txt
find $R/gpurun_out/prof_pool -name '*.db' -delete
