// moog_drawlist.h -- the per-env DRAW LIST: what PILRenderer.__call__ hands to ImageDraw.polygon
// (reference moog/observers/pil_renderer.py:100-110: for every sprite in layer order,
// `vertices = self._canvas_size * sprite.vertices`, then `tuple(v) for v in vertices` truncated by Pillow's
// (int) cast in ImagingDrawPolygon's caller), laid out for one wavefront per frame.
//
// The step kernel holds every world vertex in LDS when it stores the record, so it emits, for LIVE sprites
// only, the packed integer canvas points (int16 x | int16 y << 16, Pillow's x86 cast, clamped to +-32000
// exactly as the rasteriser's own vertex phase did) -- 8 bytes per live vertex instead of 16 bytes of f64 per
// vertex SLOT -- and the wave rasteriser (moog_raster_wave.h) starts from there.  `moog_drawlist_kernel`
// builds the same list from a state record in HBM for frames of states the step kernel did not produce
// (moog_engine_render on uploaded state, the frame of an explicit reset).
//
// Layout per env, 32-bit words, stride `dl_stride(..)`:
//   [0] rounds   [1] items (live sprites with vertices)   [2] live vertices   [3] spare
//   [4..7]   lanes used per round, one byte each (<= DL_MAX_ROUNDS rounds)
//   [8..11]  first item of every round, one byte each (0xff: the round is not used)
//   [12..27] item of slot s, one byte each (0xff: no item; items are the live sprites in slot = painter's order, <= 64)
//   [28..]   entries, 2 words each, 64 per round:
//              w0 = x | y << 16                      (int16 each)
//              w1 = item | k << 8 | nv << 16 | slot << 24   (k: index of the vertex within its sprite)
// A ROUND is 64 consecutive entries = what the 64 lanes of the rasteriser's wave process together.  A sprite's
// vertices never straddle two rounds (a sprite that does not fit into the current round starts the next one), so
// every neighbour a vertex needs (previous / next vertex, earlier vertices of the same polygon) is in the same
// round.  Entries of a round beyond its "lanes used" byte are not written and not read.
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>

#include "../../include/moog_engine.h"

#define DL_MAX_ROUNDS 16
#define DL_HDR 28          // words in front of the entries
#define DL_MAX_ITEMS 64
#define DL_SCRATCH_WORDS 96   // drawlist_emit: [0..63] per slot, [64..79] lanes per round, [80..95] first item per round
#define DL_MAX_NV 32       // vertices per sprite (one word of head bits per row in the rasteriser)

// Worst case of the greedy packing: a round is closed only when the next sprite (<= maxv vertices) does not
// fit, i.e. when it holds at least 65 - maxv entries.
__host__ __device__ inline int dl_max_rounds(int TOTV, int maxv) {
  const int per = 65 - (maxv < 1 ? 1 : maxv);
  const int r = (TOTV + per - 1) / per;
  return r < 1 ? 1 : r;
}
__host__ __device__ inline int dl_stride_words(int max_rounds) { return DL_HDR + 128 * max_rounds; }

// Pillow's (int) cast of a coordinate as x86-64 performs it (cvttsd2si): NaN and values outside the int range
// give INT_MIN (the reference can produce NaN sprite state); then the +-32000 clamp of the
// rasteriser's 16-bit points (such polygons are off the canvas either way).
__device__ __forceinline__ int dl_pil_int(double d) {
  return (d >= -2147483648.0 && d < 2147483648.0) ? (int)d : (int)0x80000000;
}
__device__ __forceinline__ unsigned dl_clamp16(int v) {
  return (unsigned)(unsigned short)(short)(v < -32000 ? -32000 : (v > 32000 ? 32000 : v));
}

// One wavefront builds one env's draw list.  flags / nverts / verts: the env's record fields (LDS in the step
// kernel, HBM in moog_drawlist_kernel); voff: vertex offset of every slot (S entries); vslot: vertex slot ->
// sprite slot (global table); tbl: DL_SCRATCH_WORDS words of LDS scratch.  S <= 64, every sprite <= DL_MAX_NV vertices.
template <bool THROUGH>
__device__ inline void drawlist_emit(uint32_t* __restrict__ out, const int S, const int TOTV, const int32_t* flags,
                                     const int32_t* nverts, const double* verts, const int32_t* voff,
                                     const int16_t* __restrict__ vslot, const int CW, const int CH, const int lane,
                                     uint32_t* tbl) {
  int nv = 0;
  if (lane < S) {
    const int f = flags[lane];
    nv = (f & MOOG_F_ALIVE) ? nverts[lane] : 0;
    if (nv < 0) nv = 0;
    if (nv > DL_MAX_NV) nv = DL_MAX_NV;
  }
  const unsigned long long live = __ballot(nv > 0);
  const int g = __popcll(live & ((1ull << lane) - 1ull));
  // greedy packing, in slot order (a scalar loop over the live sprites)
  int cur = 0, mypos = 0;
  for (unsigned long long m = live; m != 0ull; m &= m - 1ull) {
    const int t = __ffsll((long long)m) - 1;
    const int nvt = __builtin_amdgcn_readlane(nv, t);
    if ((cur & 63) + nvt > 64) cur = (cur + 63) & ~63;
    mypos = (lane == t) ? cur : mypos;
    cur += nvt;
  }
  const int n_rounds = (cur + 63) >> 6;
  if (lane < 32) tbl[64 + lane] = lane < 16 ? 0u : 255u;
  __builtin_amdgcn_fence(__ATOMIC_SEQ_CST, "workgroup", "local");
  __builtin_amdgcn_wave_barrier();
  tbl[lane] = (unsigned)mypos | ((unsigned)g << 10) | ((unsigned)nv << 16);
  if (nv > 0) {
    atomicMax(&tbl[64 + (mypos >> 6)], (unsigned)((mypos & 63) + nv));
    atomicMin(&tbl[80 + (mypos >> 6)], (unsigned)g);
  }
  __builtin_amdgcn_fence(__ATOMIC_SEQ_CST, "workgroup", "local");
  __builtin_amdgcn_wave_barrier();
  auto put = [&](uint32_t* p, unsigned v) {
    if (THROUGH) __hip_atomic_store(p, v, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    else *p = v;
  };
  if (lane == 0) { put(out + 0, (unsigned)n_rounds); put(out + 1, (unsigned)__popcll(live)); put(out + 2, (unsigned)cur); put(out + 3, 0u); }
  if (lane < 4) {
    const unsigned w = (tbl[64 + 4 * lane] & 255u) | ((tbl[64 + 4 * lane + 1] & 255u) << 8) |
                       ((tbl[64 + 4 * lane + 2] & 255u) << 16) | ((tbl[64 + 4 * lane + 3] & 255u) << 24);
    put(out + 4 + lane, w);
  }
  if (lane >= 4 && lane < 8) {
    const int b = 4 * (lane - 4);
    const unsigned w = (tbl[80 + b] & 255u) | ((tbl[80 + b + 1] & 255u) << 8) | ((tbl[80 + b + 2] & 255u) << 16) |
                       ((tbl[80 + b + 3] & 255u) << 24);
    put(out + 4 + lane, w);
  }
  {   // item of every slot: four slots per word
    const unsigned mine = nv > 0 ? (unsigned)g : 255u;
    const unsigned b1 = (unsigned)__shfl_down((int)mine, 1), b2 = (unsigned)__shfl_down((int)mine, 2), b3 = (unsigned)__shfl_down((int)mine, 3);
    if ((lane & 3) == 0) put(out + 12 + (lane >> 2), mine | (b1 << 8) | (b2 << 16) | (b3 << 24));
  }
  for (int idx = lane; idx < TOTV; idx += 64) {
    const int s = vslot[idx];
    const int k = idx - voff[s];
    const unsigned t = tbl[s];
    const int nvs = (int)((t >> 16) & 255u);
    if (k < nvs) {
      const double vx = verts[2 * idx], vy = verts[2 * idx + 1];
      const int ix = dl_pil_int((double)CW * vx), iy = dl_pil_int((double)CH * vy);
      const unsigned w0 = dl_clamp16(ix) | (dl_clamp16(iy) << 16);
      const unsigned w1 = ((t >> 10) & 63u) | ((unsigned)k << 8) | ((unsigned)nvs << 16) | ((unsigned)s << 24);
      uint32_t* p = out + DL_HDR + 2 * ((int)(t & 1023u) + k);
      if (THROUGH) {
        __hip_atomic_store(reinterpret_cast<unsigned long long*>(p), (unsigned long long)w0 | ((unsigned long long)w1 << 32),
                           __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
      } else {
        *reinterpret_cast<uint2*>(p) = make_uint2(w0, w1);
      }
    }
  }
}
