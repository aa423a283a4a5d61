#!/bin/bash
# Round 5 experiment: specialised step kernels compiled with other scheduler options (gpurun_in/spec_<name>, built by hand with
# MOOG_SPEC_DIR / MOOG_SPEC_FLAGS), headline workload and config 5.
out=gpurun_out/r05_sched
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
for v in base maxilp maxmem ifcvt iterilp trackers; do
  d=""; [ $v != base ] && d=$PWD/gpurun_in/spec_$v
  env ${d:+MOOG_SPEC_DIR=$d} ${d:+MOOG_SPEC_PREBUILT=1} python bench.py --no-extras --no-cpu-baseline > $out/${v}_$rep.log 2>&1; show $out/${v}_$rep.log $v
  [ $rep = 1 ] && { env ${d:+MOOG_SPEC_DIR=$d} ${d:+MOOG_SPEC_PREBUILT=1} python bench.py --no-extras --no-cpu-baseline --workload falling_balls_64 --envs-per-gpu 8192 --steps 30 --warmup 5 > $out/${v}_c5.log 2>&1; show $out/${v}_c5.log ${v}_c5; }
done
done
