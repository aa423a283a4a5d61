"""gpurun_out/r06 (tools/r06_collect.sh) -> profiles/r06_current.txt, r06_bench_configs.txt, r06_bench_ranks.txt, r06_step_tail.txt,
r06_runtime_benchmark_phases.txt, r06_emit_cycles.txt, profiles/raster_traffic.json; prints the per-phase table of the mask rasteriser
(pasted into profiles/r06_raster.txt)."""
import json, os, re, sys
R = os.path.join(os.path.dirname(os.path.abspath(__file__)), '..')
O = os.path.join(R, 'gpurun_out', 'r06')
P = os.path.join(R, 'profiles')
KERNEL = 'moog_raster_mask_kernel<1, false, false>'


def rd(name):
    return open(os.path.join(O, name)).read()


def clean(txt):
    return '\n'.join(l for l in txt.splitlines() if 'amdgpu.ids' not in l)


out = ['# Round 6, final build: bench lines, rocprofv3 kernel trace and PMC passes of `python bench.py --no-cpu-baseline --no-extras`',
       '# (tools/r06_collect.sh -> tools/prof.sh, one MI355X box).  Counters are per launch, summed over the device; SQ_* cycle',
       '# counters are in units of four cycles.  moog_step_kernel<false, 3, 0> is the program-specialised kernel here (same mangled name',
       '# as the generic one: lib/spec/step_417c47560f31861d_d0w3.so) and now also writes the frames\' draw records;',
       '# %s is the roofline-graded kernel (it reads those records).' % KERNEL, '']
for l in rd('bench.txt').splitlines():
    if l.startswith('=='):
        out.append(l)
    elif l.startswith('{'):
        j = json.loads(l)
        j.pop('cpu_baseline_note', None)
        out.append(json.dumps(j))
ps = rd('prof_summary.txt')
out += ['', '## rocprofv3 summaries (tools/prof_summary.py)', clean(ps)]


def counter(kernel, name):
    for l in ps.splitlines():
        if kernel in l and (' ' + name + ' ') in l:
            return float(l.split()[-1]), int(l.split()[-2])
    raise KeyError((kernel, name))


fetch, nl = counter(KERNEL, 'FETCH_SIZE')
write, _ = counter(KERNEL, 'WRITE_SIZE')
traffic = int(round((2 * fetch + write) * 1024))
head = json.loads([l for l in rd('bench.txt').splitlines() if l.startswith('{')][0])
alg = head['roofline']['algorithmic_bytes_per_launch']
trace_avg = [float(l.split()[-4]) for l in ps.splitlines() if KERNEL in l and '%' in l][0]
out += ['', '## the roofline line of the headline bench',
        'algorithmic bytes per launch %d; HIP-event average %.2f us (bench.py, inside the timed region) -> %.1f GB/s = %.4f of 8 TB/s' % (
            alg, head['roofline']['avg_kernel_us'], head['roofline']['achieved'], head['roofline']['frac']),
        'rocprofv3 kernel-trace average %.2f us (the events bracket the launch on the stream)' % (trace_avg / 1e3),
        'HBM traffic (guide: 2 x FETCH_SIZE + WRITE_SIZE, KB): 2 x %.1f + %.1f = %.1f MB per launch = %.2f x algorithmic' % (
            fetch, write, traffic / 1e6, traffic / alg)]
open(os.path.join(P, 'r06_current.txt'), 'w').write('\n'.join(out) + '\n')
json.dump({'_comment': 'HBM traffic of the roofline-graded raster kernel per launch from rocprofv3 PMC passes: 2*FETCH_SIZE (gfx950 correction '
                       'for wide coalesced reads) + WRITE_SIZE, KB -> bytes',
           'colliding_predators_32': {'n_envs': 4096, 'traffic_bytes': traffic, 'kernel': KERNEL,
                                      'source': 'profiles/r06_current.txt (FETCH_SIZE x 2 + WRITE_SIZE, %d launches, end-of-round-6 build)' % nl,
                                      'fetch_kb': fetch, 'write_kb': write}},
          open(os.path.join(P, 'raster_traffic.json'), 'w'), indent=1)

t = ['# Round 6: tools/bench_configs.py on one MI355X (reset, 5 warm-up calls, `steps` timed calls with random actions, no cost schedule;',
     '# kernel times from HIP events; [specialised]: the four BASELINE workloads\' step kernels built by __graft_entry__.build(), the rest generic).', '',
     clean(rd('bench_configs.txt'))]
open(os.path.join(P, 'r06_bench_configs.txt'), 'w').write('\n'.join(t) + '\n')
open(os.path.join(P, 'r06_bench_ranks.txt'), 'w').write('# Round 6: tools/bench_ranks.sh (bench.py\'s multi-rank path on a 1-GPU box)\n' + clean(rd('bench_ranks.txt')) + '\n')
open(os.path.join(P, 'r06_step_tail.txt'), 'w').write(
    '# Round 6: per-env cycle counts of the step kernel (tools/step_tail.py: MOOG_STEP_DEBUG=128, the GENERIC kernel -- the debug hooks are\n'
    '# not in the specialised build), every 10th call.  Part 1: colliding_predators_32, 4096 envs; part 2: falling_balls_64, 8192 envs.\n'
    + clean(rd('step_tail.txt')) + '\n')
open(os.path.join(P, 'r06_runtime_benchmark_phases.txt'), 'w').write(
    '# Round 6: python -m moog_demos.runtime_benchmark (the batched counterpart of the reference\'s tests/runtime_benchmark.py:64-157) on one MI355X,\n'
    '# final build: the five phases and the six renderer settings (runtime_benchmark.py:31-38), pong with 1 env and 4096 envs, then the headline\n'
    '# workload.  tests/test_gpu_parity.py::test_runtime_benchmark_reports_every_phase runs the same entry point in the GPU suite.\n'
    + clean(rd('runtime_benchmark.txt')) + '\n')
open(os.path.join(P, 'r06_emit_cycles.txt'), 'w').write(
    '# Round 6: shader-clock cycles of the draw-record emitter inside the step kernel (tools/emit_cycles.py, step debug bit 256), per env-step.\n'
    + clean(rd('emit_cycles.txt')) + '\n')

rows = []
cur = None
for l in rd('mask_pmc.txt').splitlines():
    m = re.match(r'== (\S+) stop (\d)', l)
    if m:
        cur = {'stop': int(m.group(2))}
        rows.append(cur)
        continue
    if l.startswith('SQ_'):
        for kv in l.split():
            k, v = kv.split('=')
            cur[k] = float(v)
    m = re.match(r'trace: calls=\d+ avg_ns=(\d+)', l)
    if m:
        cur['ns'] = float(m.group(1))
names = {2: 'load: tables, draw record -> LDS', 3: '(p2: only frames of several passes)', 4: 'p3 edges + census',
         5: 'p4 row sort + rows -> masks', 0: 'p5 compose + store'}
prev = {k: 0.0 for k in rows[0]}
print('%-36s %8s %7s %7s %6s %6s %8s %9s' % ('phase (increment)', 'us', 'VALU', 'SALU', 'LDS', 'VMEM', 'wait q', 'conflict q'))
for r in rows:
    d = {k: r[k] - prev.get(k, 0.0) for k in r if k != 'stop'}
    print('%-36s %8.1f %7.0f %7.0f %6.0f %6.0f %8.0f %9.0f' % (names[r['stop']], d['ns'] / 1e3, d['SQ_INSTS_VALU'], d['SQ_INSTS_SALU'], d['SQ_INSTS_LDS'],
                                                       d['SQ_INSTS_VMEM_RD'] + d['SQ_INSTS_VMEM_WR'], d['SQ_WAIT_INST_ANY'], d['SQ_LDS_BANK_CONFLICT']))
    prev = r
r = rows[-1]
print('%-36s %8.1f %7.0f %7.0f %6.0f %6.0f %8.0f %9.0f' % ('whole kernel', r['ns'] / 1e3, r['SQ_INSTS_VALU'], r['SQ_INSTS_SALU'], r['SQ_INSTS_LDS'],
                                                   r['SQ_INSTS_VMEM_RD'] + r['SQ_INSTS_VMEM_WR'], r['SQ_WAIT_INST_ANY'], r['SQ_LDS_BANK_CONFLICT']))
print('active lanes per VALU instruction: %.1f of 64; wave lifetime %.0f cycles x 2 waves' % (r['SQ_THREAD_CYCLES_VALU'] / r['SQ_ACTIVE_INST_VALU'], 4 * r['SQ_WAVE_CYCLES'] / 2))
for f in ('mask_pmc_balls.txt', 'mask_pmc_torus.txt'):
    print('\n# ' + f)
    print(clean(rd(f)))
