cd $GRAFT_REPO_ROOT
for w in 2 3 4; do echo "== WPS $w"; MOOG_STEP_WPS=$w python tools/heavy_bench.py bench 2>&1 | grep -v amdgpu; done
