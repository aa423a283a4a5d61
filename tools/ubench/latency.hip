// latency.hip -- dependent-chain latencies seen by ONE wavefront on gfx950 (s_memtime cycles per operation):
// the step kernel is a lone dependent chain per env, so these are its unit costs.
//   hipcc --offload-arch=gfx950 -O3 tools/ubench/latency.hip -o gpurun_out/latency && gpurun_out/latency
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>
#define CONST_AS __attribute__((address_space(4)))

__global__ void k_sload(const int* tab, int n, int iters, long long* out) {   // pointer chase through the scalar cache
  const CONST_AS int* t = (const CONST_AS int*)(unsigned long long)tab;
  int i = 0;
  long long t0 = clock64();
  for (int k = 0; k < iters; ++k) i = __builtin_amdgcn_readfirstlane(t[i]);
  long long t1 = clock64();
  if (threadIdx.x == 0) { out[0] = t1 - t0; out[1] = i; }
}
__global__ void k_vload(const int* tab, int n, int iters, long long* out) {   // the same through the vector L1 (uniform address)
  int i = 0;
  long long t0 = clock64();
  for (int k = 0; k < iters; ++k) i = tab[i + (threadIdx.x & 0)];
  long long t1 = clock64();
  if (threadIdx.x == 0) { out[0] = t1 - t0; out[1] = i; }
}
__global__ void k_lds(int iters, long long* out) {   // ds_read -> address of the next ds_read
  __shared__ int s[1024];
  for (int k = threadIdx.x; k < 1024; k += 64) s[k] = (k * 17 + 5) & 1023;
  __syncthreads();
  int i = threadIdx.x;
  long long t0 = clock64();
  for (int k = 0; k < iters; ++k) i = s[i];
  long long t1 = clock64();
  if (threadIdx.x == 0) { out[0] = t1 - t0; out[1] = i; }
}
__global__ void k_bperm(int iters, long long* out) {   // ds_bpermute chain
  int v = threadIdx.x;
  long long t0 = clock64();
  for (int k = 0; k < iters; ++k) v = __builtin_amdgcn_ds_bpermute(((v + 1) & 63) << 2, v);
  long long t1 = clock64();
  if (threadIdx.x == 0) { out[0] = t1 - t0; out[1] = v; }
}
__global__ void k_readlane(int iters, long long* out) {   // v_readlane -> SALU -> VALU chain
  int v = threadIdx.x;
  long long t0 = clock64();
  for (int k = 0; k < iters; ++k) { int s = __builtin_amdgcn_readlane(v, 5); v = v + s; }
  long long t1 = clock64();
  if (threadIdx.x == 0) { out[0] = t1 - t0; out[1] = v; }
}
__global__ void k_fma64(int iters, long long* out) {   // dependent v_fma_f64 chain
  double v = threadIdx.x * 1e-3;
  long long t0 = clock64();
  for (int k = 0; k < iters; ++k) v = __builtin_fma(v, 1.0000001, 1e-9);
  long long t1 = clock64();
  if (threadIdx.x == 0) { out[0] = t1 - t0; out[1] = (long long)v; }
}
__global__ void k_fma64u(int iters, long long* out) {   // the same, 16 per loop iteration (the loop's own branch amortised)
  double v = threadIdx.x * 1e-3;
  long long t0 = clock64();
  for (int k = 0; k < iters; k += 16) {
#pragma unroll
    for (int u = 0; u < 16; ++u) v = __builtin_fma(v, 1.0000001, 1e-9);
  }
  long long t1 = clock64();
  if (threadIdx.x == 0) { out[0] = t1 - t0; out[1] = (long long)v; }
}
__global__ void k_fma32u(int iters, long long* out) {
  float v = threadIdx.x * 1e-3f;
  long long t0 = clock64();
  for (int k = 0; k < iters; k += 16) {
#pragma unroll
    for (int u = 0; u < 16; ++u) v = __builtin_fmaf(v, 1.0000001f, 1e-9f);
  }
  long long t1 = clock64();
  if (threadIdx.x == 0) { out[0] = t1 - t0; out[1] = (long long)v; }
}
__global__ void k_indep64u(int iters, long long* out) {   // four independent fp64 chains interleaved: issue rate of one wave
  double a = threadIdx.x * 1e-3, b = a + 1, c = a + 2, d = a + 3;
  long long t0 = clock64();
  for (int k = 0; k < iters; k += 16) {
#pragma unroll
    for (int u = 0; u < 4; ++u) {
      a = __builtin_fma(a, 1.0000001, 1e-9); b = __builtin_fma(b, 1.0000001, 1e-9);
      c = __builtin_fma(c, 1.0000001, 1e-9); d = __builtin_fma(d, 1.0000001, 1e-9);
    }
  }
  long long t1 = clock64();
  if (threadIdx.x == 0) { out[0] = t1 - t0; out[1] = (long long)(a + b + c + d); }
}
__global__ void k_salu_u(int iters, long long* out) {   // dependent SALU chain, 16 per iteration
  int v = __builtin_amdgcn_readfirstlane(threadIdx.x) + iters;
  long long t0 = clock64();
  for (int k = 0; k < iters; k += 16) {
#pragma unroll
    for (int u = 0; u < 16; ++u) v = (v * 3 + 1) ^ (v >> 3);
  }
  long long t1 = clock64();
  if (threadIdx.x == 0) { out[0] = t1 - t0; out[1] = v; }
}
__global__ void k_ldsu(int iters, long long* out) {   // ds_read chain, 8 per iteration
  __shared__ int s[1024];
  for (int k = threadIdx.x; k < 1024; k += 64) s[k] = (k * 17 + 5) & 1023;
  __syncthreads();
  int i = threadIdx.x;
  long long t0 = clock64();
  for (int k = 0; k < iters; k += 8) {
#pragma unroll
    for (int u = 0; u < 8; ++u) i = s[i];
  }
  long long t1 = clock64();
  if (threadIdx.x == 0) { out[0] = t1 - t0; out[1] = i; }
}
__global__ void k_bpermu(int iters, long long* out) {
  int v = threadIdx.x;
  long long t0 = clock64();
  for (int k = 0; k < iters; k += 8) {
#pragma unroll
    for (int u = 0; u < 8; ++u) v = __builtin_amdgcn_ds_bpermute(((v + 1) & 63) << 2, v);
  }
  long long t1 = clock64();
  if (threadIdx.x == 0) { out[0] = t1 - t0; out[1] = v; }
}
__global__ void k_ntbranch(int iters, long long* out, int z) {   // 8 not-taken uniform branches per iteration
  int v = __builtin_amdgcn_readfirstlane(threadIdx.x) | 1;
  long long t0 = clock64();
  for (int k = 0; k < iters; k += 8) {
#pragma unroll
    for (int u = 0; u < 8; ++u) { if (__builtin_expect(v == z + u, 0)) { v = v * 7; asm volatile("s_nop 0"); } v += 2; }
  }
  long long t1 = clock64();
  if (threadIdx.x == 0) { out[0] = t1 - t0; out[1] = v; }
}
__global__ void k_div64(int iters, long long* out) {   // dependent fp64 division chain
  double v = 1.0 + threadIdx.x * 1e-3;
  long long t0 = clock64();
  for (int k = 0; k < iters; ++k) v = 3.0 / (v + 0.5);
  long long t1 = clock64();
  if (threadIdx.x == 0) { out[0] = t1 - t0; out[1] = (long long)(v * 1e6); }
}
__global__ void k_sqrt64(int iters, long long* out) {
  double v = 1.0 + threadIdx.x * 1e-3;
  long long t0 = clock64();
  for (int k = 0; k < iters; ++k) v = sqrt(v + 2.0);
  long long t1 = clock64();
  if (threadIdx.x == 0) { out[0] = t1 - t0; out[1] = (long long)(v * 1e6); }
}
__global__ void k_ballot(int iters, long long* out) {   // v_cmp -> ballot -> SALU popcount -> VALU
  int v = threadIdx.x;
  long long t0 = clock64();
  for (int k = 0; k < iters; ++k) { unsigned long long m = __ballot(v & 1); v += __popcll(m); }
  long long t1 = clock64();
  if (threadIdx.x == 0) { out[0] = t1 - t0; out[1] = v; }
}
__global__ void k_branch(int iters, long long* out) {   // a taken uniform branch per iteration (besides the loop's own)
  int v = __builtin_amdgcn_readfirstlane(threadIdx.x);
  long long t0 = clock64();
  for (int k = 0; k < iters; ++k) { if (v & 1) v = v * 3 + 1; else v = v >> 1; v = __builtin_amdgcn_readfirstlane(v) | 1; }
  long long t1 = clock64();
  if (threadIdx.x == 0) { out[0] = t1 - t0; out[1] = v; }
}
__global__ void k_scratch(int iters, long long* out, int idx) {   // store + dependent load through the scratch / vector path
  volatile int a[64];
  for (int k = 0; k < 64; ++k) a[k] = k + 1;
  int i = idx;
  long long t0 = clock64();
  for (int k = 0; k < iters; ++k) { a[i & 63] = i + 1; i = a[(i + 1) & 63] + i; }
  long long t1 = clock64();
  if (threadIdx.x == 0) { out[0] = t1 - t0; out[1] = i; }
}

int main() {
  const int iters = 4096;
  long long* out; hipMalloc(&out, 16);
  long long h[2];
  auto report = [&](const char* name) {
    hipDeviceSynchronize(); hipMemcpy(h, out, 16, hipMemcpyDeviceToHost);
    printf("  %-64s %7.1f cycles\n", name, (double)h[0] / iters);
  };
  for (int kb : {1, 16, 64, 512}) {
    const int n = kb * 256;
    std::vector<int> t(n);
    for (int i = 0; i < n; ++i) t[i] = (int)(((long long)i * 4099 + 61) % n);   // a stride that leaves the cache line every time
    int* d; hipMalloc(&d, n * 4); hipMemcpy(d, t.data(), n * 4, hipMemcpyHostToDevice);
    char nm[128];
    for (int rep = 0; rep < 2; ++rep) hipLaunchKernelGGL(k_sload, 1, 64, 0, 0, d, n, iters, out);
    snprintf(nm, sizeof nm, "s_load -> address of the next s_load, %d KB table", kb); report(nm);
    for (int rep = 0; rep < 2; ++rep) hipLaunchKernelGGL(k_vload, 1, 64, 0, 0, d, n, iters, out);
    snprintf(nm, sizeof nm, "global_load (uniform address) chain, %d KB table", kb); report(nm);
    hipFree(d);
  }
  hipLaunchKernelGGL(k_lds, 1, 64, 0, 0, iters, out); report("ds_read_b32 -> address of the next ds_read");
  hipLaunchKernelGGL(k_bperm, 1, 64, 0, 0, iters, out); report("ds_bpermute_b32 chain");
  hipLaunchKernelGGL(k_readlane, 1, 64, 0, 0, iters, out); report("v_readlane -> v_add chain");
  hipLaunchKernelGGL(k_fma64, 1, 64, 0, 0, iters, out); report("dependent v_fma_f64");
  hipLaunchKernelGGL(k_fma64u, 1, 64, 0, 0, iters, out); report("dependent v_fma_f64, unrolled x16");
  hipLaunchKernelGGL(k_fma32u, 1, 64, 0, 0, iters, out); report("dependent v_fma_f32, unrolled x16");
  hipLaunchKernelGGL(k_indep64u, 1, 64, 0, 0, iters, out); report("four independent v_fma_f64 chains, per instruction");
  hipLaunchKernelGGL(k_salu_u, 1, 64, 0, 0, iters, out); report("dependent SALU (3 ops per step), unrolled x16, per step");
  hipLaunchKernelGGL(k_ldsu, 1, 64, 0, 0, iters, out); report("ds_read_b32 -> address chain, unrolled x8");
  hipLaunchKernelGGL(k_bpermu, 1, 64, 0, 0, iters, out); report("ds_bpermute_b32 chain (v_add, v_and, v_lshl between), unrolled x8");
  hipLaunchKernelGGL(k_ntbranch, 1, 64, 0, 0, iters, out, -100); report("not-taken uniform branch + s_add, unrolled x8");
  hipLaunchKernelGGL(k_div64, 1, 64, 0, 0, iters, out); report("dependent fp64 add + division");
  hipLaunchKernelGGL(k_sqrt64, 1, 64, 0, 0, iters, out); report("dependent fp64 add + sqrt");
  hipLaunchKernelGGL(k_ballot, 1, 64, 0, 0, iters, out); report("v_cmp -> ballot -> s_bcnt1 -> v_add chain");
  hipLaunchKernelGGL(k_branch, 1, 64, 0, 0, iters, out); report("uniform if / else + readfirstlane per iteration");
  hipLaunchKernelGGL(k_scratch, 1, 64, 0, 0, iters, out, 3); report("scratch store + dependent scratch load");
  return 0;
}
