#!/bin/bash
# Round 5: specialised step kernels (moog/_spec.py) against the generic ones, per BASELINE config.  Output: gpurun_out/r05_spec/
out=gpurun_out/r05_spec
mkdir -p $out
export TMPDIR=/tmp
python -m pytest tests/test_gpu_parity.py -x -q -m gpu -k "specialised" > $out/pytest.log 2>&1
echo "pytest rc=$?" >> $out/pytest.log
tail -4 $out/pytest.log
for rep in 1 2; do
  for mode in spec generic; do
    flag=""; [ $mode = generic ] && flag="--no-spec"
    python bench.py --no-extras $flag > $out/bench_${mode}_$rep.log 2>&1
    python - $out/bench_${mode}_$rep.log $mode <<'PY'
import json, sys
line = [l for l in open(sys.argv[1]) if l.startswith('{')][-1]
j = json.loads(line)
print('%-8s value %.3f M  ms/step %.4f  step_kernel %s' % (sys.argv[2], j['value'] / 1e6, j['ms_per_step'], j['config'].get('step_kernel', '?')[:12]),
      {k: v for k, v in j.get('kernels', {}).items()} if 'kernels' in j else '')
PY
  done
done
python tools/bench_configs.py > $out/configs_spec.txt 2>&1; tail -12 $out/configs_spec.txt
MOOG_STEP_SPEC=0 python tools/bench_configs.py > $out/configs_generic.txt 2>&1; tail -12 $out/configs_generic.txt
