// fn_bench.hip -- latency of single device functions of the step path on real env records (tools/fn_bench.py):
// one wavefront per env binds the record exactly like the step kernel, then calls ONE function `iters` times and
// reports cycles per call (s_memtime).  With one env per CU this is the function's own dependent chain; with a full
// batch, its cost beside two other waves of the SIMD.  Not part of the product library.
#include <hip/hip_runtime.h>
#define MOOG_WITH_MAZE 0
#include "../../moog.github.io_amd/csrc/moog_kernels.h"

enum { FN_FIND = -1, FN_SEARCH = 0, FN_PATH = 1, FN_PREFIX = 2, FN_INTEGRATE = 4, FN_SUBSTEP = 5, FN_BROAD = 6, FN_EMPTY = 9 };

template <int which>   // one kernel per function: each gets its own register allocation (a switch over all of them spills)
__global__ __attribute__((flatten)) __launch_bounds__(64, 3) void fn_bench_kernel(KArgs a, int iters, int* pairs, double* out) {
  const int env = blockIdx.x;
  if (env >= a.n_envs) return;
  Env e;
  bind_env(e, a, env);
  load_record(e, a.H, a.L, a.f64 + (size_t)env * a.L.f64_per_env, a.i32 + (size_t)env * a.L.i32_per_env);
  bbox_build_all(e);
  PProg P = e.P;
  const int K = uni(P->updates_per_env_step);
  const double dt = 1. / K;
  if (which == FN_FIND) {   // the first ordered pair of live sprites whose paths intersect
    int f0 = -1, f1 = -1;
    const int S = P->n_slots;
    for (int s0 = 0; s0 < S && f0 < 0; ++s0)
      for (int s1 = 0; s1 < S && f0 < 0; ++s1)
        if (s0 != s1 && ALIVE(s0) && ALIVE(s1) && P->slot_layer[s0] == 1 && P->slot_layer[s1] == 1 && overlaps(e, s0, s1)) { f0 = s0; f1 = s1; }
    if (e.lane == 0) { pairs[2 * env] = f0; pairs[2 * env + 1] = f1; }
    return;
  }
  const int s0 = uni(pairs[2 * env]), s1 = uni(pairs[2 * env + 1]);
  double acc = 0;
  if (s0 < 0 && which != FN_INTEGRATE && which != FN_SUBSTEP && which < 10 && which != FN_EMPTY && which != FN_BROAD) { if (e.lane == 0) out[env] = -1; return; }
  if (which == FN_PREFIX) {
    wsync();
    if (e.lane < 4) e.cand[e.lane] = (uint16_t)(((e.lane & 1) ? ((s1 << 8) | s0) : ((s0 << 8) | s1)));
    wsync();
  }
  const long long t0 = clock64();
  for (int it = 0; it < iters; ++it) {
    if constexpr (which == FN_SEARCH) { CVec c; get_collision_vectors(e, s0, s1, dt, c); acc += c.status + c.px; }
    else if constexpr (which == FN_PATH) acc += overlaps(e, s0, s1, true) ? 1 : 0;
    else if constexpr (which == FN_PREFIX) acc += narrow_reject_prefix(e, 0, 4) & 1023;
    else if constexpr (which == FN_INTEGRATE) integrate_all(e, dt);
    else if constexpr (which == FN_SUBSTEP) apply_physics<false>(e);
    else if constexpr (which == 10) { e.dbg = 1; apply_physics<false>(e); }    // forces + integrate only
    else if constexpr (which == 11) { e.dbg = 4; apply_physics<false>(e); }    // + broad phase and candidate lists, no narrow phase
    else if constexpr (which == 12) { e.dbg = 16; apply_physics<false>(e); }   // + narrow phase (batches, path tests), no search / response
    else if constexpr (which == 13) { e.dbg = 3; apply_physics<false>(e); }   // the force loop without collisions and integrate
    else if constexpr (which == 14) {   // the headers of the force loop alone
      const int n_forces = uni(P->n_forces);
      for (int fi = 0; fi < n_forces; ++fi) {
        PForce F = &P->forces[fi];
        const int n_a = uni(F->n_a), n_b = uni(F->n_b), kind = uni(F->kind);
        for (int a2 = 0; a2 < n_a; ++a2) {
          int la = uni(F->layers_a[a2]);
          int a0 = uni(P->layer_slot0[la]), a1 = a0 + uni(P->layer_nslots[la]);
          acc += a0 + a1 + kind;
          for (int b = 0; b < n_b; ++b) {
            int lb = uni(F->layers_b[b]);
            int b0 = uni(P->layer_slot0[lb]), b1 = b0 + uni(P->layer_nslots[lb]);
            acc += b0 + b1;
          }
        }
      }
    }
    else if constexpr (which == 15) {   // Drag on the agent's layer alone
      PForce F = &P->forces[0];
      int la = uni(F->layers_a[0]);
      int a0 = uni(P->layer_slot0[la]), a1 = a0 + uni(P->layer_nslots[la]);
      force_single_layer(e, F, a0, a1, K);
    }
    else wsync();
  }
  const long long t1 = clock64();
  if (e.lane == 0) { out[env] = (double)(t1 - t0) / iters; out[a.n_envs + env] = acc; }
}

extern "C" int moog_fn_bench(const moog_program_t* prog, double* f64_dev, int32_t* i32_dev, int n_envs, int which, int iters,
                             int* pairs_dev, double* out_dev) {
  static moog_program_t* d_prog = nullptr;
  static int16_t* d_vslot = nullptr;
  static FOp* d_fops = nullptr;
  static int n_fops = 0;
  moog_layout_t L;
  moog_layout(prog, &L);
  if (!d_prog) {
    hipMalloc(&d_prog, sizeof(moog_program_t));
    hipMemcpy(d_prog, prog, sizeof(moog_program_t), hipMemcpyHostToDevice);
    std::vector<int16_t> vs((size_t)(prog->n_total_verts > 0 ? prog->n_total_verts : 1), 0);
    for (int sl = 0; sl < prog->n_slots; ++sl)
      for (int k = 0; k < prog->slot_vcap[sl]; ++k) vs[prog->slot_voff[sl] + k] = (int16_t)sl;
    hipMalloc(&d_vslot, vs.size() * sizeof(int16_t));
    hipMemcpy(d_vslot, vs.data(), vs.size() * sizeof(int16_t), hipMemcpyHostToDevice);
    const std::vector<FOp> fops = moog_flatten_forces(prog);
    n_fops = (int)fops.size();
    hipMalloc(&d_fops, (fops.size() + 1) * sizeof(FOp));
    hipMemcpy(d_fops, fops.data(), fops.size() * sizeof(FOp), hipMemcpyHostToDevice);
  }
  KArgs a = {};
  a.P = d_prog; a.L = L; a.H = hot_layout(L); a.f64 = f64_dev; a.i32 = i32_dev;
  a.n_envs = n_envs; a.vslot = d_vslot; a.fops = d_fops; a.n_fops = n_fops;
  const moog_layout_t HL = a.H.L;
  size_t lds = (size_t)HL.f64_per_env * 8 + (size_t)HL.i32_per_env * 4 + (size_t)L.S * 4 * 8 +
               (size_t)((L.S + 3) & ~3) * 4 + CAND_CAP * 2 + 128 + 64 * 8 + 16;
#define FN_CASE(W) case W: hipFuncSetAttribute(reinterpret_cast<const void*>(fn_bench_kernel<W>), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds); \
    hipLaunchKernelGGL(fn_bench_kernel<W>, dim3(n_envs), dim3(64), lds, 0, a, iters, pairs_dev, out_dev); break;
  switch (which) {
    FN_CASE(-1) FN_CASE(0) FN_CASE(1) FN_CASE(2) FN_CASE(4) FN_CASE(5) FN_CASE(9) FN_CASE(10) FN_CASE(11) FN_CASE(12) FN_CASE(13) FN_CASE(14) FN_CASE(15)
    default: return -1;
  }
  return (int)hipDeviceSynchronize();
}
