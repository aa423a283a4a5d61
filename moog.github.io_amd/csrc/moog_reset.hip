// moog_reset.hip -- the reset kernel (explicit resets) and the launch-order sort, their own translation unit.
#include <hip/hip_runtime.h>

#define MOOG_DEFINE_RESET_KERNELS
#include "moog_kernels.h"

void moog_launch_reset(int n_envs, size_t lds, hipStream_t s, const KArgs& a) {
  hipLaunchKernelGGL(moog_reset_kernel, dim3(n_envs), dim3(64), lds, s, a);
}

int moog_configure_reset(size_t lds) {
  return (int)hipFuncSetAttribute(reinterpret_cast<const void*>(moog_reset_kernel),
                                  hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
}

void moog_launch_sched(hipStream_t s, const float* cost, int32_t* perm, int n, const int32_t* reset_next, int stride) {
  hipLaunchKernelGGL(moog_sched_kernel, dim3(1), dim3(1024), 0, s, cost, perm, n, reset_next, stride);
}
