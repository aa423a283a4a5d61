# Where the time between kernels goes: rocprofv3 kernel trace of a short bench run, then the idle gaps of the GPU
# between consecutive kernels of the steady state, by (previous kernel -> next kernel).
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
rm -rf /tmp/gaps && mkdir -p /tmp/gaps
rocprofv3 --kernel-trace --output-format csv -d /tmp/gaps -o g -- python3 $R/bench.py --no-cpu-baseline --no-extras --steps 60 --warmup 5 > /tmp/gaps/log.txt 2>&1
python3 - <<'PY'
import csv, glob, collections
f = glob.glob('/tmp/gaps/**/*kernel_trace.csv', recursive=True)[0]
rows = list(csv.DictReader(open(f)))
rows = [r for r in rows if 'sched' not in r['Kernel_Name']]   # (the sort kernel runs on a side stream, under the rasteriser)
rows.sort(key=lambda r: int(r['Start_Timestamp']))
rows = rows[-60 * 3:]          # the steady state at the end
gaps = collections.defaultdict(list)
busy = 0
for a, b in zip(rows, rows[1:]):
    g = int(b['Start_Timestamp']) - int(a['End_Timestamp'])
    gaps[(a['Kernel_Name'][:28], b['Kernel_Name'][:28])].append(g)
span = int(rows[-1]['End_Timestamp']) - int(rows[0]['Start_Timestamp'])
print('span %.1f us for %d kernels' % (span / 1e3, len(rows)))
for k, v in sorted(gaps.items(), key=lambda kv: -sum(kv[1])):
    print('%-30s -> %-30s n %4d  mean gap %8.2f us  total %9.1f us' % (k[0], k[1], len(v), sum(v) / len(v) / 1e3, sum(v) / 1e3))
PY
