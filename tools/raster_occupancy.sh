# Raster kernel time vs resident workgroups per CU (LDS padding limits residency): separates
# per-workgroup latency from contention on shared CU resources (instruction cache, LDS, issue).
cd $GRAFT_REPO_ROOT
for pad in 0 6000 12000 25000 40000 70000; do
  echo -n "pad $pad: "; MOOG_RASTER_LDS_PAD=$pad python tools/raster_phases.py 2>&1 | tail -1
done
