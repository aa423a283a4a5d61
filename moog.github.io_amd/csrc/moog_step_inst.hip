// moog_step_inst.hip -- one instantiation of the step kernel per compilation
// (-DMOOG_STEP_DYN=0|1|2 -DMOOG_STEP_WPS=3|4 -DMOOG_STEP_TAG=f3|f4|t3|t4|m3|m4): the six variants build in parallel.
// DYN 0: the BASELINE components; 1: + run-time sampler, expression VM, dynamic layers; 2: + the maze components.
#include <hip/hip_runtime.h>

#define MOOG_WITH_MAZE (MOOG_STEP_DYN == 2)
#include "moog_kernels.h"

#define MOOG_CAT_(a, b) a##b
#define MOOG_CAT(a, b) MOOG_CAT_(a, b)

void MOOG_CAT(moog_launch_step_, MOOG_STEP_TAG)(int n_envs, size_t lds, hipStream_t s, const KArgs& a) {
  hipLaunchKernelGGL((moog_step_kernel<MOOG_STEP_DYN != 0, MOOG_STEP_WPS, MOOG_STEP_DYN>), dim3(n_envs), dim3(MOOG_STEP_THREADS), lds, s, a);
}

int MOOG_CAT(moog_configure_step_, MOOG_STEP_TAG)(size_t lds) {
  return (int)hipFuncSetAttribute(reinterpret_cast<const void*>(moog_step_kernel<MOOG_STEP_DYN != 0, MOOG_STEP_WPS, MOOG_STEP_DYN>),
                                  hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
}
