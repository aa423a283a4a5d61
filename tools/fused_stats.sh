# Step-kernel durations and step-to-step periods of the timed steps of a short bench run, split by whether the call took
# the follow grid or the separate launches (rocprofv3 kernel trace).   usage: bash tools/fused_stats.sh [bench args]
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
rm -rf /tmp/ptl && mkdir -p /tmp/ptl
rocprofv3 --kernel-trace --output-format csv -d /tmp/ptl -o g -- python3 $R/bench.py --no-cpu-baseline --no-extras --steps 80 --warmup 5 "$@" > /tmp/ptl/log.txt 2>&1
python3 - <<'PY'
import csv, glob, statistics as st
f = glob.glob('/tmp/ptl/**/*kernel_trace.csv', recursive=True)[0]
rows = [r for r in csv.DictReader(open(f)) if 'moog' in r['Kernel_Name']]
rows.sort(key=lambda r: int(r['Start_Timestamp']))
steps = []
for r in rows:
    s, e = int(r['Start_Timestamp']), int(r['End_Timestamp'])
    n = r['Kernel_Name']
    if 'step_kernel' in n: steps.append({'s': s, 'e': e, 'kind': '?', 'last': e})
    elif steps:
        if 'follow' in n: steps[-1]['kind'] = 'follow'
        elif 'raster_kernel' in n: steps[-1]['kind'] = 'separate'
        steps[-1]['last'] = max(steps[-1]['last'], e)
steps = steps[-80:]
for kind in ('follow', 'separate'):
    d = [(x['e'] - x['s']) / 1e3 for x in steps if x['kind'] == kind]
    p = [(b['s'] - a['s']) / 1e3 for a, b in zip(steps, steps[1:]) if a['kind'] == kind]
    t = [(a['last'] - a['e']) / 1e3 for a in steps if a['kind'] == kind]
    if d:
        print('%-8s n %3d  step kernel mean %.1f median %.1f max %.1f us | period to the next step mean %.1f us | last kernel ends %.1f us after the step kernel' % (
            kind, len(d), st.mean(d), st.median(d), max(d), st.mean(p) if p else 0, st.mean(t)))
PY
tail -1 /tmp/ptl/log.txt | grep -o '"ms_per_step": [0-9.]*'
