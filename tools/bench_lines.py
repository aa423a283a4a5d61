"""Prints the bench lines of a log (== command / JSON line pairs) in one row each."""
import json
import sys
for line in open(sys.argv[1]):
    line = line.strip()
    if line.startswith('=='):
        print(line)
        continue
    try:
        d = json.loads(line)
    except ValueError:
        print('   ' + line[:200])
        continue
    print('   value %.3f M  ms/step %.3f  kernels %s  roofline.frac %.4f (%.1f us, %s)' % (
        d['value'] / 1e6, d['ms_per_step'], {k: round(v, 1) for k, v in d['kernels_avg_us'].items()}, d['roofline']['frac'],
        d['roofline']['avg_kernel_us'], d['config'].get('step_kernel', '')[:11]))
