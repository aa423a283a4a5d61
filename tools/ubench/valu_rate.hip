// VALU issue-rate microbenchmark (gfx950): cycles per wave64 VALU instruction per SIMD for a few
// instruction classes, at 1 / 2 / 4 / 8 waves per SIMD.  hipcc --offload-arch=gfx950 -O3 valu_rate.hip -o valu_rate
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>

template <int KIND>
__global__ void k(unsigned* out, int iters, unsigned seed) {
  unsigned a[8];
  float f[8];
  double d[4];
  unsigned long long q[4];
  unsigned sa[4] = {seed, seed * 3u, seed * 5u, seed * 7u};
#pragma unroll
  for (int i = 0; i < 8; ++i) { a[i] = seed + threadIdx.x * (i + 1); f[i] = (float)a[i]; }
#pragma unroll
  for (int i = 0; i < 4; ++i) { d[i] = (double)a[i]; q[i] = a[i] * 0x100000001ull; }
  long long t0 = clock64();
  for (int it = 0; it < iters; ++it) {
#pragma unroll
    for (int r = 0; r < 8; ++r) {
      if (KIND == 0) {
#pragma unroll
        for (int i = 0; i < 8; ++i) asm volatile("v_add_u32 %0, %0, %1" : "+v"(a[i]) : "v"(a[(i + 1) & 7]));
      } else if (KIND == 1) {
#pragma unroll
        for (int i = 0; i < 8; ++i) asm volatile("v_fma_f32 %0, %0, %1, %0" : "+v"(f[i]) : "v"(f[(i + 1) & 7]));
      } else if (KIND == 2) {
#pragma unroll
        for (int i = 0; i < 4; ++i) asm volatile("v_fma_f64 %0, %0, %1, %0" : "+v"(d[i]) : "v"(d[(i + 1) & 3]));
#pragma unroll
        for (int i = 0; i < 4; ++i) asm volatile("v_fma_f64 %0, %0, %1, %0" : "+v"(d[i]) : "v"(d[(i + 1) & 3]));
      } else if (KIND == 3) {
#pragma unroll
        for (int i = 0; i < 4; ++i) asm volatile("v_lshlrev_b64 %0, 1, %0" : "+v"(q[i]));
#pragma unroll
        for (int i = 0; i < 4; ++i) asm volatile("v_lshlrev_b64 %0, 1, %0" : "+v"(q[i]));
      } else if (KIND == 4) {
#pragma unroll
        for (int i = 0; i < 8; ++i) asm volatile("v_cndmask_b32 %0, %0, %1, vcc" : "+v"(a[i]) : "v"(a[(i + 1) & 7]));
      } else if (KIND == 5) {
#pragma unroll
        for (int i = 0; i < 8; ++i) asm volatile("v_cmp_lt_u32 vcc, %0, %1" : : "v"(a[i]), "v"(a[(i + 1) & 7]) : "vcc");
      } else if (KIND == 6) {
#pragma unroll
        for (int i = 0; i < 8; ++i) asm volatile("v_mul_lo_u32 %0, %0, %1" : "+v"(a[i]) : "v"(a[(i + 1) & 7]));
      } else if (KIND == 7) {
#pragma unroll
        for (int i = 0; i < 8; ++i) asm volatile("v_mul_f64 %0, %0, %1" : "+v"(d[i & 3]) : "v"(d[(i + 1) & 3]));
      } else if (KIND == 8) {
#pragma unroll
        for (int i = 0; i < 8; ++i) asm volatile("v_add_f64 %0, %0, %1" : "+v"(d[i & 3]) : "v"(d[(i + 1) & 3]));
      } else if (KIND == 10) {
#pragma unroll
        for (int i = 0; i < 8; ++i) {
          asm volatile("v_add_u32 %0, %0, %1" : "+v"(a[i]) : "v"(a[(i + 1) & 7]));
          asm volatile("s_add_u32 %0, %0, %1" : "+s"(sa[i & 3]) : "s"(sa[(i + 1) & 3]) : "scc");
        }
      } else if (KIND == 11) {
#pragma unroll
        for (int i = 0; i < 8; ++i) asm volatile("s_add_u32 %0, %0, %1" : "+s"(sa[i & 3]) : "s"(sa[(i + 1) & 3]) : "scc");
      } else if (KIND == 12) {
#pragma unroll
        for (int i = 0; i < 8; ++i) {
          asm volatile("v_add_u32 %0, %0, %1" : "+v"(a[i]) : "v"(a[(i + 1) & 7]));
          asm volatile("s_mov_b32 %0, %1" : "=s"(sa[i & 3]) : "s"(sa[(i + 1) & 3]));
        }
      } else if (KIND == 13) {
#pragma unroll
        for (int i = 0; i < 8; ++i) {
          asm volatile("v_add_u32 %0, %0, %1" : "+v"(a[i]) : "v"(a[(i + 1) & 7]));
          asm volatile("v_readlane_b32 %0, %1, 3" : "=s"(sa[i & 3]) : "v"(a[i]));
        }
      }
    }
  }
  long long t1 = clock64();
  unsigned acc = sa[0] + sa[1] + sa[2] + sa[3];
#pragma unroll
  for (int i = 0; i < 8; ++i) acc += a[i] + (unsigned)f[i];
#pragma unroll
  for (int i = 0; i < 4; ++i) acc += (unsigned)d[i] + (unsigned)q[i];
  if (threadIdx.x == 0) { out[2 * blockIdx.x] = (unsigned)(t1 - t0); out[2 * blockIdx.x + 1] = acc; }
}

template <int KIND>
void run(const char* name, int per_iter) {
  unsigned* d;
  hipMalloc(&d, 1 << 20);
  const int iters = 2000;
  printf("%-14s", name);
  for (int wps : {1, 2, 4, 8}) {
    // one workgroup of 256 * wps threads per CU -> wps waves on each SIMD
    int threads = 256 * wps;
    if (threads > 1024) { threads = 1024; }
    int blocks = 256 * (256 * wps / threads);
    hipLaunchKernelGGL(k<KIND>, dim3(blocks), dim3(threads), 0, 0, d, iters, 1u);
    hipDeviceSynchronize();
    hipEvent_t a, b; hipEventCreate(&a); hipEventCreate(&b);
    hipEventRecord(a);
    hipLaunchKernelGGL(k<KIND>, dim3(blocks), dim3(threads), 0, 0, d, iters, 1u);
    hipEventRecord(b); hipEventSynchronize(b);
    float ms; hipEventElapsedTime(&ms, a, b);
    std::vector<unsigned> h(2 * blocks);
    hipMemcpy(h.data(), d, h.size() * 4, hipMemcpyDeviceToHost);
    double cyc = 0; for (int i = 0; i < blocks; ++i) cyc += h[2 * i]; cyc /= blocks;
    double n = (double)iters * 8 * per_iter;   // instructions per wave
    // per-SIMD cycles per instruction = wave cycles / (n * wps)
    printf("  wps%d: %.2f cyc/inst/wave, %.2f cyc/inst/SIMD (%.0f us)", wps, cyc / n, cyc / (n * wps), ms * 1e3);
  }
  printf("\n");
  hipFree(d);
}

int main() {
  run<0>("v_add_u32", 8); run<1>("v_fma_f32", 8); run<2>("v_fma_f64", 8); run<3>("v_lshlrev_b64", 8);
  run<4>("v_cndmask_b32", 8); run<5>("v_cmp_lt_u32", 8); run<6>("v_mul_lo_u32", 8); run<7>("v_mul_f64", 8);
  run<8>("v_add_f64", 8); run<10>("valu+salu", 16); run<11>("salu", 8); run<12>("valu+s_mov", 16); run<13>("valu+readlane", 16);
  return 0;
}
