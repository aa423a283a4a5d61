// moog_step_spec.hip -- the step kernel specialised for ONE program (built by moog/_spec.py, loaded by moog_engine.hip).
//   hipcc ... -DMOOG_SPEC_PROGRAM_INC='"<generated>.inc"' -DMOOG_STEP_DYN=0|1|2 -DMOOG_STEP_WPS=2|3|4 -shared -o step_<hash>_d<dyn>w<wps>.so
// The generated include defines `static const moog_program_t MOOG_SPEC_PROGRAM = {...};` and MOOG_SPEC_HASH (FNV-1a 64 of
// the program's bytes).  Same source, same arithmetic as the generic kernels (moog_step_inst.hip): results are bit-identical
// (tests/test_gpu_parity.py::test_specialised_step_kernel_is_result_neutral); only what the program never uses is gone.
#include <hip/hip_runtime.h>

#define MOOG_WITH_MAZE (MOOG_STEP_DYN == 2)
#include "moog_kernels.h"

#ifndef MOOG_SRC_DIGEST
#define MOOG_SRC_DIGEST 0ull
#endif
#define MOOG_STR2(x) #x
#define MOOG_STR(x) MOOG_STR2(x)
extern "C" const char moog_src_digest_marker[] = "MOOG_SRC_DIGEST=" MOOG_STR(MOOG_SRC_DIGEST);   // (read as text by moog/_digest.py)

extern "C" {

// (moog/_digest.py: the kernel sources and flags this object was built from; the engine library refuses another build's)
unsigned long long moog_spec_source_digest(void) { return MOOG_SRC_DIGEST; }
int moog_spec_abi(void) { return MOOG_ABI_VERSION; }
unsigned long long moog_spec_hash(void) { return MOOG_SPEC_HASH; }
int moog_spec_variant(void) { return MOOG_STEP_DYN | (MOOG_STEP_WPS << 8); }
unsigned long long moog_spec_kargs_size(void) { return sizeof(KArgs); }
// (the program the kernel was compiled for: the engine compares it with its own before using the kernel)
const void* moog_spec_program(void) { return &MOOG_SPEC_PROGRAM; }

int moog_spec_configure(size_t lds) {
  return (int)hipFuncSetAttribute(reinterpret_cast<const void*>(moog_step_kernel<MOOG_STEP_DYN != 0, MOOG_STEP_WPS, MOOG_STEP_DYN>),
                                  hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
}

void moog_spec_launch(int n_envs, size_t lds, hipStream_t s, const KArgs* a) {
  hipLaunchKernelGGL((moog_step_kernel<MOOG_STEP_DYN != 0, MOOG_STEP_WPS, MOOG_STEP_DYN>), dim3(n_envs), dim3(MOOG_STEP_THREADS), lds, s, *a);
}

}  // extern "C"
