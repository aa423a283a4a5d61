// moog_raster_mask.h -- the mask rasteriser's kernel: one workgroup of RM_THREADS threads per frame, the phases of
// moog_raster_mask_core.h with barriers in between.  Included by moog_raster.hip.
#ifndef MOOG_RASTER_MASK_H_
#define MOOG_RASTER_MASK_H_
#include <hip/hip_runtime.h>

#include "moog_raster_mask_core.h"


extern __shared__ __attribute__((aligned(16))) unsigned char moog_lds[];

// BIG: the program has slots for polygons of more than RM_MAX_NV vertices (rm_p4_big); a kernel of its own so that everybody
// else's keeps its registers
template <int WORDS, bool BIG, bool TORUS>
__global__ __launch_bounds__(RM_THREADS, RM_WAVES_PER_SIMD) void moog_raster_mask_kernel(RmArgs a) {
  const int env = (int)blockIdx.x;
  if (env >= a.n_envs) return;
  const RmCtx c = rm_ctx(a.plan, moog_lds);
  const int tid = (int)threadIdx.x, lane = tid & 63;
  RmThread th;
  rm_p0<WORDS>(a, c, env, tid, RM_THREADS, th);
  if (TORUS) {   // torus frames (a.ncopy = 9): nine copies per sprite, the visible ones become items
    if (tid < 64) rm_t0_slots(a, c, env, lane);
    __syncthreads();
    if (a.debug_stop == 1) return;
    rm_t1_bounds(a, c, env, tid, RM_THREADS);
    __syncthreads();
    if (tid < 64) rm_t2_items(a, c, lane);
    __syncthreads();
    rm_t3_points(a, c, env, tid, RM_THREADS);
    __syncthreads();
  } else {
    if (tid < 64) rm_p0_slots(a, c, env, lane, th);
    __syncthreads();
    if (a.debug_stop == 1) return;
    rm_p1<WORDS>(a, c, env, tid, RM_THREADS, th);
    __syncthreads();
  }
  if (a.debug_stop == 2) return;
  // (every wave scans the items for itself: both write the same words, and a wave's LDS operations execute in order,
  //  so each reads back what it wrote -- no barrier between the scan and the edges)
  const int s_lo = __builtin_amdgcn_readfirstlane(rm_s_lo(a, c));
  rm_p2_scan(a, c, s_lo, lane);
  __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "wavefront");
  __builtin_amdgcn_wave_barrier();
  if (a.rows_seen && tid == 0 && c.rowoff[a.S] > a.cap_rows) {   // several passes: the engine may grow the records (mask_rows_grow)
    atomicMax(a.rows_seen, c.rowoff[a.S]);
    atomicAdd(a.rows_seen + 1, 1);
  }
  for (int base = 0;;) {
    const int end = __builtin_amdgcn_readfirstlane(rm_pass_end(a, c, base));
    const int total_rows = __builtin_amdgcn_readfirstlane(c.rowoff[end] - c.rowoff[base]);
    rm_p2_assign(a, c, base, end, s_lo, lane);
    __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "wavefront");
    __builtin_amdgcn_wave_barrier();
    if (a.debug_stop == 3) return;
    rm_p3<WORDS>(a, c, base, end, s_lo, tid, RM_THREADS);
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
    rm_p4<WORDS>(a, c, total_rows, tid, RM_THREADS, c.xx + (tid >> 6) * a.plan.xx_stride);
    if (BIG) rm_p4_big<WORDS>(a, c, tid, RM_THREADS, c.xx + (tid >> 6) * a.plan.xx_stride, reinterpret_cast<uint8_t*>(c.xx + (RM_THREADS / 64) * a.plan.xx_stride));
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

typedef void (*moog_raster_mask_fn)(RmArgs);
// [WORDS - 1][BIG][TORUS]: a kernel per combination, so that frames that need neither keep their registers
static inline moog_raster_mask_fn moog_raster_mask_pick(int words, bool big, bool torus) {
  static const moog_raster_mask_fn table[2][2][2] = {
      {{moog_raster_mask_kernel<1, false, false>, moog_raster_mask_kernel<1, false, true>},
       {moog_raster_mask_kernel<1, true, false>, moog_raster_mask_kernel<1, true, true>}},
      {{moog_raster_mask_kernel<2, false, false>, moog_raster_mask_kernel<2, false, true>},
       {moog_raster_mask_kernel<2, true, false>, moog_raster_mask_kernel<2, true, true>}}};
  return table[words - 1][big ? 1 : 0][torus ? 1 : 0];
}

static inline int moog_raster_mask_configure(size_t lds_bytes) {
  hipError_t err = hipSuccess;
  for (int k = 0; k < 8 && err == hipSuccess; ++k)
    err = hipFuncSetAttribute(reinterpret_cast<const void*>(moog_raster_mask_pick(1 + (k & 1), (k & 2) != 0, (k & 4) != 0)),
                              hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds_bytes);
  return (int)err;
}

static inline void moog_raster_mask_launch(const RmArgs& a, size_t lds_bytes, hipStream_t stream) {
  const dim3 grid((unsigned)a.n_envs);
  hipLaunchKernelGGL(moog_raster_mask_pick(a.W > 64 ? 2 : 1, a.big != 0, a.ncopy > 1), grid, dim3(RM_THREADS), lds_bytes, stream, a);
}

#endif  // MOOG_RASTER_MASK_H_
