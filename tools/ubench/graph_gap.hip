// Does a captured HIP graph shorten the gaps between DEPENDENT kernels of one step() call on gfx950?  (VERDICT r02, next 2:
// "capture step + gate + follow + sort in a HIP graph: 18 us of stream gaps per call".)
// Three dependent kernels per "call" -- long (stands for the step kernel, ~750 us), short (the rasteriser, ~90 us), tiny (the cost
// sort, ~10 us) -- launched (a) on a stream, the host running ahead, (b) as one captured graph per call.  Prints the time per call and
// what is left after subtracting the kernels' own durations (measured alone).
// build + run: hipcc --offload-arch=gfx950 -O2 tools/ubench/graph_gap.hip -o /tmp/graph_gap && /tmp/graph_gap
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>
#include <functional>
__global__ void spin(long long cycles, int* sink) {
  const long long t0 = clock64();
  while (clock64() - t0 < cycles) {}
  if (sink && threadIdx.x == 0 && blockIdx.x == 0) sink[0] += 1;
}
#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e_)); return 1; } } while (0)
static float timed(hipStream_t s, int calls, const std::function<void()>& call) {
  hipEvent_t a, b; hipEventCreate(&a); hipEventCreate(&b);
  for (int i = 0; i < 20; ++i) call();
  hipStreamSynchronize(s);
  hipEventRecord(a, s);
  for (int i = 0; i < calls; ++i) call();
  hipEventRecord(b, s);
  hipEventSynchronize(b);
  float ms = 0; hipEventElapsedTime(&ms, a, b);
  return ms * 1000.f / calls;
}
#include <functional>
int main() {
  int* sink; CK(hipMalloc(&sink, 4)); CK(hipMemset(sink, 0, 4));
  hipStream_t s; CK(hipStreamCreate(&s));
  // ticks of clock64 per microsecond, calibrated: one wave spinning for 2e6 ticks
  long long f = 2000;
  {
    hipEvent_t a, b; hipEventCreate(&a); hipEventCreate(&b);
    hipLaunchKernelGGL(spin, dim3(1), dim3(64), 0, s, 2000000LL, sink);
    hipStreamSynchronize(s);
    hipEventRecord(a, s);
    hipLaunchKernelGGL(spin, dim3(1), dim3(64), 0, s, 2000000LL, sink);
    hipEventRecord(b, s);
    hipEventSynchronize(b);
    float ms = 0; hipEventElapsedTime(&ms, a, b);
    f = (long long)(2000000.0 / (ms * 1000.0));
    printf("clock64: %lld ticks per us\n", f);
  }
  const long long c_long = 750 * f, c_short = 90 * f, c_tiny = 10 * f;
  auto k = [&](long long c, int blocks) { hipLaunchKernelGGL(spin, dim3(blocks), dim3(64), 0, s, c, sink); };
  const int calls = 300;
  const float t_long = timed(s, calls, [&] { k(c_long, 4096); });
  const float t_short = timed(s, calls, [&] { k(c_short, 4096); });
  const float t_tiny = timed(s, calls, [&] { k(c_tiny, 1); });
  const float t_stream = timed(s, calls, [&] { k(c_long, 4096); k(c_short, 4096); k(c_tiny, 1); });
  hipGraph_t g; hipGraphExec_t ge;
  CK(hipStreamBeginCapture(s, hipStreamCaptureModeGlobal));
  k(c_long, 4096); k(c_short, 4096); k(c_tiny, 1);
  CK(hipStreamEndCapture(s, &g));
  CK(hipGraphInstantiate(&ge, g, nullptr, nullptr, 0));
  const float t_graph = timed(s, calls, [&] { hipGraphLaunch(ge, s); });
  printf("kernels alone, back to back (us per launch): long %.1f  short %.1f  tiny %.1f  sum %.1f\n", t_long, t_short, t_tiny, t_long + t_short + t_tiny);
  printf("three dependent kernels per call, stream launches: %.1f us per call (%.1f us over the sum)\n", t_stream, t_stream - (t_long + t_short + t_tiny));
  printf("the same as one captured graph per call:           %.1f us per call (%.1f us over the sum)\n", t_graph, t_graph - (t_long + t_short + t_tiny));
  return 0;
}
