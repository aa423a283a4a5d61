# Timeline of the kernels of the last steps of a short bench run (rocprofv3 kernel trace): start / end of every kernel
# relative to its step kernel's start, with the queue it ran on.   usage: bash tools/fused_timeline.sh [bench args]
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
rm -rf /tmp/ptl && mkdir -p /tmp/ptl
rocprofv3 --kernel-trace --output-format csv -d /tmp/ptl -o g -- python3 $R/bench.py --no-cpu-baseline --no-extras --steps 40 --warmup 5 "$@" > /tmp/ptl/log.txt 2>&1
python3 - <<'PY'
import csv, glob
f = glob.glob('/tmp/ptl/**/*kernel_trace.csv', recursive=True)[0]
rows = list(csv.DictReader(open(f)))
rows = [r for r in rows if 'moog' in r['Kernel_Name']]
rows.sort(key=lambda r: int(r['Start_Timestamp']))
rows = rows[-16:]
t0 = None
for r in rows:
    s, e = int(r['Start_Timestamp']), int(r['End_Timestamp'])
    name = r['Kernel_Name'][:34]
    if 'step_kernel' in name:
        if t0 is not None:
            print('-- next step kernel starts %.1f us after the previous one' % ((s - t0) / 1e3))
        t0 = s
    if t0 is None:
        continue
    print('%-36s queue %s  start %8.1f  end %8.1f  (%.1f us)' % (name, r.get('Queue_Id', '?'), (s - t0) / 1e3, (e - t0) / 1e3, (e - s) / 1e3))
PY
tail -1 /tmp/ptl/log.txt | cut -c1-200
