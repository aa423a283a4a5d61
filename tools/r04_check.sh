# correctness + speed check of a build: the lock-step / parity tests that exercise the collision path, then the heavy-env replay and a bench line
cd $GRAFT_REPO_ROOT
python -m pytest tests -m gpu -x -q 2>&1 | tail -4
python tools/heavy_bench.py bench 2>&1 | grep -v amdgpu
RANDOM_SAMPLE=1 HEAVY_ONLY=256 python tools/heavy_bench.py bench 2>&1 | grep -v amdgpu
python tools/r04_exp.py --args "--no-fused --no-cpu-baseline --no-extras --steps 100 --warmup 10" base=
