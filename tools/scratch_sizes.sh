# Scratch frame (bytes per lane) and spills of every step-kernel variant, from the compiler's resource-usage remarks.
# The step kernel falls off a cliff when its frame grows (120 B / lane: free; 724 B: 780 -> 1327 us), so run this after
# touching moog_device.h.  Usage: bash tools/scratch_sizes.sh   (CPU only, ~1 minute per variant, run in parallel)
cd "$(dirname "$0")/.."
for v in "f3 0 3" "f4 0 4" "t3 1 3" "t4 1 4" "m3 2 3" "m4 2 4"; do
  set -- $v
  ( /opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -std=c++17 -ffp-contract=off -fPIC -Wno-unused-value \
      -DMOOG_STEP_DYN=$2 -DMOOG_STEP_WPS=$3 -DMOOG_STEP_TAG=$1 -Rpass-analysis=kernel-resource-usage \
      -c moog.github.io_amd/csrc/moog_step_inst.hip -o /tmp/moog_scratch_$1.o 2>&1 |
      grep -E "ScratchSize|VGPRs:|VGPRs Spill|SGPRs Spill" | sed 's/^.*remark: *//; s/ \[-Rpass.*//' | tr '\n' ' ' |
      sed "s/^/$1: /"; echo ) &
done
wait
