// moog_raster_mask.h -- the mask rasteriser's kernel: one workgroup of RM_THREADS threads per frame, the phases of
// moog_raster_mask_core.h with barriers in between.  Included by moog_raster.hip.
#ifndef MOOG_RASTER_MASK_H_
#define MOOG_RASTER_MASK_H_
#ifndef RM_PERSIST
#define RM_PERSIST 0   // 1: a workgroup may draw several frames one after the other (MOOG_RASTER_PERSIST; measured in round 6: profiles/r06_raster.txt)
#endif
#include <hip/hip_runtime.h>

#include "moog_raster_mask_core.h"


extern __shared__ __attribute__((aligned(16))) unsigned char moog_lds[];

// BIG: the program has slots for polygons of more than RM_MAX_NV vertices (rm_p4_big); a kernel of its own so that everybody
// else's keeps its registers
// COMPACT: 4-byte edge records beside the integer points instead of 16-byte ones (RmEdgesCompact): programs whose edge records are what
// keeps frames off a CU
template <int WORDS, bool BIG, bool COMPACT>
__global__ __launch_bounds__(RM_THREADS, RM_WAVES_PER_SIMD) void moog_raster_mask_kernel(RmArgs a) {
  const RmCtx c = rm_ctx(a.plan, moog_lds);
  const int tid = (int)threadIdx.x, lane = tid & 63;
  // A workgroup draws the frames blockIdx.x, blockIdx.x + gridDim.x, ... (moog_raster_mask_launch: one frame per workgroup, or
  // -- MOOG_RASTER_PERSIST -- as many workgroups as are resident at once, each drawing its share one after the other)
#if RM_PERSIST
  for (int env = (int)blockIdx.x; env < a.n_envs; env += (int)gridDim.x) {
  if (env != (int)blockIdx.x) __syncthreads();   // (the tables are the previous frame's until every thread has stored its segments)
#else
  const int env = (int)blockIdx.x;
  if (env >= a.n_envs) return;
  {
#endif
  rm_load(a, c, env, tid, RM_THREADS);
  __syncthreads();
  if (a.debug_stop == 1 || a.debug_stop == 2) return;
  const int s_lo = __builtin_amdgcn_readfirstlane(rm_s_lo(a, c));
  const bool single = __builtin_amdgcn_readfirstlane(c.misc[6]) != 0;
  if (a.rows_seen && tid == 0 && c.rowoff[a.S] > a.cap_rows) {   // several passes: the engine may grow the records (mask_rows_grow)
    atomicMax(a.rows_seen, c.rowoff[a.S]);
    atomicAdd(a.rows_seen + 1, 1);
  }
  for (int base = 0;;) {
    const int end = __builtin_amdgcn_readfirstlane(rm_pass_end(a, c, base));
    const int total_rows = __builtin_amdgcn_readfirstlane(c.rowoff[end] - c.rowoff[base]);
    // (every wave assigns the pass's row records for itself: both write the same words, and a wave's LDS operations execute in
    //  order, so each reads back what it wrote -- no barrier between the assignment and the edges)
    if (!single) {   // (a one-pass frame's row records were assigned with the load)
      rm_p2_assign(a, c, base, end, s_lo, lane);
      __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "wavefront");
      __builtin_amdgcn_wave_barrier();
    }
    if (a.debug_stop == 3) return;
    rm_p3<WORDS, COMPACT>(a, c, base, end, s_lo, tid, RM_THREADS);
    __syncthreads();
    if (a.debug_stop == 4) return;
    if (a.cap_rows <= RM_SORT_ROUNDS * RM_THREADS) {
      RmSortKey sk;
      rm_p4a(c, total_rows, tid, RM_THREADS, sk);
      __syncthreads();
      rm_p4b(c, total_rows, tid, RM_THREADS, sk);
      __syncthreads();
    } else {
      rm_p4a_lds(c, total_rows, tid, RM_THREADS);
      __syncthreads();
      rm_p4b_lds(c, total_rows, tid, RM_THREADS);
      __syncthreads();
    }
    rm_p4<WORDS, COMPACT>(a, c, total_rows, tid, RM_THREADS, c.xx + (tid >> 6) * a.plan.xx_stride);
    if (BIG) rm_p4_big<WORDS, COMPACT>(a, c, tid, RM_THREADS, c.xx + (tid >> 6) * a.plan.xx_stride, reinterpret_cast<uint8_t*>(c.xx + (RM_THREADS / 64) * a.plan.xx_stride));
    __syncthreads();
    if (a.debug_stop == 5) return;
    rm_p5<WORDS>(a, c, env, base == 0, s_lo, tid, RM_THREADS);
    if (end >= a.S) break;
    base = end;
    __syncthreads();
    rm_next_pass(a, c, tid, RM_THREADS);
    __syncthreads();
  }
  }
}

// The draw records of frames the engine did not step itself (moog_engine_render after load_state or an edit of the state
// tensors, resets, programs whose step kernels do not emit): one wavefront per env runs the emitter on the record in HBM.
struct RmDeriveArgs { RmEmit em; const moog_program_t* P; moog_layout_t L; const double* f64; const int32_t* i32; const uint32_t* vinfo; int32_t n_envs; int32_t env0; };
__global__ __launch_bounds__(64) void moog_draw_derive_kernel(RmDeriveArgs d) {
  const int env = (int)blockIdx.x;
  if (env >= d.n_envs) return;
  RmSrcRecord src;
  src.P = d.P; src.L = &d.L; src.f = d.f64 + (size_t)env * d.L.f64_per_env; src.q = d.i32 + (size_t)env * d.L.i32_per_env; src.vi = d.vinfo;
  RmEmit em = d.em;
  if (em.rgb_override) em.rgb_override += (size_t)d.env0 * em.slots;   // (the override array is indexed by the engine's env; a chunk of envs starts at env0)
  RmEmitScratch sc;
  rm_emit_scratch(reinterpret_cast<int32_t*>(moog_lds), em.slots, em.ncopy, &sc);
  rm_emit(em, src, env, (int)threadIdx.x, sc, d.L.TOTV);
}

typedef void (*moog_raster_mask_fn)(RmArgs);
// [WORDS - 1][BIG][COMPACT]: a kernel per combination, so that frames that need none of it keep their registers
static inline moog_raster_mask_fn moog_raster_mask_pick(int words, bool big, bool compact) {
  static const moog_raster_mask_fn table[2][2][2] = {
      {{moog_raster_mask_kernel<1, false, false>, moog_raster_mask_kernel<1, false, true>},
       {moog_raster_mask_kernel<1, true, false>, moog_raster_mask_kernel<1, true, true>}},
      {{moog_raster_mask_kernel<2, false, false>, moog_raster_mask_kernel<2, false, true>},
       {moog_raster_mask_kernel<2, true, false>, moog_raster_mask_kernel<2, true, true>}}};
  return table[words - 1][big ? 1 : 0][compact ? 1 : 0];
}

static inline int moog_raster_mask_configure(size_t lds_bytes) {
  hipError_t err = hipSuccess;
  for (int k = 0; k < 8 && err == hipSuccess; ++k)
    err = hipFuncSetAttribute(reinterpret_cast<const void*>(moog_raster_mask_pick(1 + (k & 1), (k & 2) != 0, (k & 4) != 0)),
                              hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds_bytes);
  return (int)err;
}

static inline void moog_raster_mask_launch(const RmArgs& a, size_t lds_bytes, hipStream_t stream, int persist_slots = 0) {
  // persist_slots > 0: that many workgroups are resident on the device at once; the launch is cut into equal shares for at most
  // that many workgroups (4096 frames on 2560 slots: 2048 workgroups of two frames) so that it runs as ONE resident round
  unsigned g = (unsigned)a.n_envs;
  if (RM_PERSIST && persist_slots > 0 && a.n_envs > persist_slots) {
    const int per = (a.n_envs + persist_slots - 1) / persist_slots;
    g = (unsigned)((a.n_envs + per - 1) / per);
  }
  const dim3 grid(g);
  hipLaunchKernelGGL(moog_raster_mask_pick(a.W > 64 ? 2 : 1, a.big != 0, a.compact != 0), grid, dim3(RM_THREADS), lds_bytes, stream, a);
}

static inline void moog_draw_derive_launch(const RmDeriveArgs& d, hipStream_t stream) {
  hipLaunchKernelGGL(moog_draw_derive_kernel, dim3((unsigned)d.n_envs), dim3(64), 4u * (size_t)RM_EMIT_SCRATCH_WORDS(d.em.slots, d.em.S, d.em.ncopy), stream, d);
}

#endif  // MOOG_RASTER_MASK_H_
