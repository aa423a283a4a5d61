"""A/B runs of bench.py under environment-variable knobs, one compact line per run.

    python tools/r04_exp.py [--args "<bench.py arguments>"] NAME=VAR=val,VAR=val ... [-- NAME=...]

Each positional argument is one run: a label, then comma-separated VAR=value settings (label alone = no settings).
A setting `@--flag value` appends bench.py arguments to that run instead.  Prints value, ms_per_step and the
HIP-event kernel averages of each run; the full JSON lines go to gpurun_out/r04_exp.jsonl."""
import json
import os
import subprocess
import sys

REPO = os.path.join(os.path.dirname(os.path.abspath(__file__)), '..')


def main():
    argv = sys.argv[1:]
    base = ['--no-cpu-baseline', '--no-extras', '--steps', '100', '--warmup', '10']
    if argv and argv[0] == '--args':
        base = argv[1].split()
        argv = argv[2:]
    os.makedirs(os.path.join(REPO, 'gpurun_out'), exist_ok=True)
    log = open(os.path.join(REPO, 'gpurun_out', 'r04_exp.jsonl'), 'a')
    for spec in argv:
        label, _, rest = spec.partition('=')
        env = dict(os.environ)
        extra = []
        for kv in [t for t in rest.split(';') if t]:
            if kv.startswith('@'):
                extra += kv[1:].split()
            else:
                k, _, v = kv.partition('=')
                env[k] = v
        cmd = [sys.executable, os.path.join(REPO, 'bench.py')] + base + extra
        p = subprocess.run(cmd, env=env, stdout=subprocess.PIPE, stderr=subprocess.PIPE, text=True)
        line = None
        for ln in p.stdout.splitlines():
            if ln.startswith('{'):
                line = json.loads(ln)
        if line is None:
            print('%-28s FAILED rc=%d %s' % (label, p.returncode, p.stderr.strip().splitlines()[-3:]))
            continue
        line['label'] = label
        line['settings'] = rest
        log.write(json.dumps(line) + '\n')
        log.flush()
        k = line.get('kernels_avg_us', {})
        fused = 'follow' in line['config']['launch'][:20]
        print('%-28s %8.3f M env-steps/s  %7.4f ms/step  step %7.1f us  raster %6.1f us  %s  [%s]' % (
            label, line['value'] / 1e6, line['ms_per_step'], k.get('step', float('nan')), k.get('raster', float('nan')),
            'fused' if fused else 'separate', rest))
        sys.stdout.flush()


if __name__ == '__main__':
    main()
