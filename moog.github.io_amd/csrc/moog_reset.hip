// moog_reset.hip -- the reset kernel (explicit resets) and the launch-order sort, their own translation unit.
// Compiled twice: -DMOOG_RESET_FULL=0 (tag r0: every component but the maze generator / reset-time expressions, whose
// code would slow the resets of all other programs down; also holds the sort kernel) and =1 (tag r1: everything).
#include <hip/hip_runtime.h>

#define MOOG_DEFINE_RESET_KERNELS
#define MOOG_WITH_MAZE MOOG_RESET_FULL
#include "moog_kernels.h"

#if MOOG_RESET_FULL
#define RESET_FN(name) name##_full
#else
#define RESET_FN(name) name##_plain
#endif

void RESET_FN(moog_launch_reset)(int n_envs, size_t lds, hipStream_t s, const KArgs& a) {
  hipLaunchKernelGGL(moog_reset_kernel<MOOG_RESET_FULL>, dim3(n_envs), dim3(64), lds, s, a);
}

int RESET_FN(moog_configure_reset)(size_t lds) {
  return (int)hipFuncSetAttribute(reinterpret_cast<const void*>(moog_reset_kernel<MOOG_RESET_FULL>),
                                  hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
}

#if !MOOG_RESET_FULL
void moog_launch_sched(hipStream_t s, const float* cost, int32_t* perm, int n, const int32_t* reset_next, int stride) {
  hipLaunchKernelGGL(moog_sched_kernel, dim3(1), dim3(SCHED_THREADS), 0, s, cost, perm, n, reset_next, stride);
}
#endif
