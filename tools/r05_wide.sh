#!/bin/bash
# Round 5 experiment: 32-lane narrow batches for 17..32-gons (moog_device.h narrow_reject_prefix_g<32>), A/B against spec kernels
# built with -DMOOG_NO_WIDE_BATCH (gpurun_in/spec_nowide), then parity of the kernels that changed.
out=gpurun_out/r05_wide
mkdir -p $out
export TMPDIR=/tmp
show() { python - "$1" "$2" <<'PY'
import json, sys
try:
    j = json.loads([l for l in open(sys.argv[1]) if l.startswith('{')][-1])
    print('%-24s value %.3f M  ms/step %.4f  kernels %s' % (sys.argv[2], j['value'] / 1e6, j['ms_per_step'], {k: round(v) for k, v in j.get('kernels_avg_us', {}).items()}))
except Exception as exc:
    print(sys.argv[2], 'failed', exc)
PY
}
for rep in 1 2; do
for v in wide nowide; do
  d=""; [ $v != wide ] && d=$PWD/gpurun_in/spec_$v
  env ${d:+MOOG_SPEC_DIR=$d} ${d:+MOOG_SPEC_PREBUILT=1} python bench.py --no-extras --no-cpu-baseline > $out/${v}_$rep.log 2>&1; show $out/${v}_$rep.log $v
  env ${d:+MOOG_SPEC_DIR=$d} ${d:+MOOG_SPEC_PREBUILT=1} python bench.py --no-extras --no-cpu-baseline --workload falling_balls_64 --envs-per-gpu 8192 --steps 30 --warmup 5 > $out/${v}_c5_$rep.log 2>&1; show $out/${v}_c5_$rep.log ${v}_c5
done
done
python -m pytest tests/test_gpu_parity.py -x -q -m gpu -n 3 -k "teacher_forced or free_running or full_size or pile or specialised or own_rng" > $out/pytest.log 2>&1
echo "pytest rc=$?" >> $out/pytest.log; tail -5 $out/pytest.log
