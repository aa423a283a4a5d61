// moog_draw_record.h -- the frame's DRAW RECORD: what the mask rasteriser (moog_raster_mask_core.h) reads instead of the state
// record.  Reference: moog/observers/pil_renderer.py:100-112 -- per sprite (and per copy a polygon modifier makes of it,
// polygon_modifiers.py:41-97) `vertices = canvas_size * sprite.vertices`, `color_to_rgb(sprite.color) + (opacity,)`,
// ImageDraw.polygon([tuple(v) for v in vertices]) which truncates every coordinate to int.
//
// A state record is 10.5 KB of float64 on the headline workload and the rasteriser wants ~1.6 KB of it: the live sprites'
// vertices as integer canvas points, their colours after the colour map, and per polygon which canvas rows it spans.  Until
// round 5 the raster kernel fetched the whole record and spent a fifth of its time turning it into that (VERDICT r05 item 1:
// phases p0 + p1, 12.2 of 56.6 us; 35 MB of reads per launch for 9 MB of input).  Now the kernels that already hold the
// record in LDS write the draw record when they store the record (moog_kernels.h: the step and reset kernels' epilogue,
// `rm_emit`), one wavefront per env; a caller that hands the engine a state of its own (moog_engine_render after
// load_state / an edit of the tensors) gets the same record from a small derive kernel (moog_raster.hip) that runs the same
// function on the records in HBM.
//
//   header  16 B   n_pts | total_rows | flags | 0          flags bit 0: the static prefix equals its reference record: its
//                                                           items are empty, the frame is composed on the cached picture
//   items   16 B x S (S = sprite slots x copies, painter's order; a dead / culled / prefix item is empty: nv = 0, no rows)
//           rowoff  exclusive prefix sum of the items' on-canvas row counts
//           pb_nv   first point | live vertices << 20
//           y01     smallest | largest << 16 integer y of its points (shorts; 32767 | -32768 when empty)
//           rgba    r | g << 8 | b << 16 | opacity << 24
//   points   4 B x n_pts   x | y << 16 (shorts: Pillow's (int) of the scaled coordinate, clamped to +-32000)
//   owner    1 B x n_pts   the item of every point
//
// Host + device code: tests/csrc/raster_mask_model.cpp runs the emitter on the CPU (lane = -1: one call does every lane's work).
#ifndef MOOG_DRAW_RECORD_H_
#define MOOG_DRAW_RECORD_H_
#include <math.h>
#include <stdint.h>
#include <string.h>

#include "../../include/moog_engine.h"

#if defined(__HIP_DEVICE_COMPILE__)
#define RM_DEV 1
#else
#define RM_DEV 0
#endif
#if defined(__HIPCC__)
#define RM_FN __host__ __device__ __forceinline__
#define RM_MEMBER __host__ __device__ __forceinline__
#define RM_SLOW __host__ __device__ __noinline__
#else
#define RM_FN static inline
#define RM_MEMBER inline
#define RM_SLOW static
#endif
#if RM_DEV
#define RM_ANY(x) __any((x))
#define RM_CONSTP(T) const __attribute__((address_space(4))) T*
#else
#define RM_ANY(x) (x)
#define RM_CONSTP(T) const T*
#endif

#define RM_MAX_NV 32          // vertices per polygon whose rows go by census words (edge index = bit of a word)
#define RM_BIG_NV 128         // vertices per polygon at most: longer ones (the 102-vertex annuli) take the cooperative row routine

struct RmDrawHdr { int32_t n_pts, total_rows, flags, pad; };
struct alignas(16) RmDrawItem { int32_t rowoff; uint32_t pb_nv; int32_t y01; uint32_t rgba; };
#define RM_DRAW_PREFIX_OK 1
#define RM_Y01_EMPTY ((int32_t)0x80007fff)   // ymin = 32767, ymax = -32768

struct RmDrawLayout { uint32_t o_items, o_pts, o_owner, stride; };   // byte offsets inside an env's record, bytes from one env's record to the next
static inline RmDrawLayout rm_draw_layout(int S, int pts_cap) {
  RmDrawLayout l;
  l.o_items = 16u;
  l.o_pts = l.o_items + 16u * (uint32_t)S;
  l.o_owner = l.o_pts + 4u * (uint32_t)pts_cap;
  l.stride = (l.o_owner + (uint32_t)pts_cap + 15u) & ~15u;
  return l;
}

// What the emitter needs to know about the renderer (by value in the kernels' arguments).  out == null: no draw records.
struct RmEmit {
  uint8_t* out;               // [n_envs][lay.stride]
  RmDrawLayout lay;
  int32_t S, slots, ncopy;    // items = slots x copies (1, or 9: polygon_modifiers.py TorusGeometry draws every sprite at the 3 x 3 offsets -1, 0, 1)
  int32_t W, H, scale_w;      // canvas in memory (width a multiple of 16), the width the vertices are scaled by (pil_renderer.py:65-66)
  int32_t cmap, first_person, fp_slot0, fp_nslots;
  // static prefix (moog_raster.h): the first n_static slots are in the cached picture when they equal the reference record
  int32_t n_static;
  const double* sref_v;       // reference world vertices, by vertex slot
  const double* sref_col;
  const int32_t* sref_flags;
  const int32_t* sref_nv;
  const int32_t* sref_opa;
  const uint32_t* rgb_override;   // [n_envs][slots] r | g << 8 | b << 16 instead of the colour map (moog_engine_set_color_override), or null
};

// ---- small helpers (shared with the rasteriser) ---------------------------------------------------------------------------
RM_FN uint32_t rm_f2u(float f) { uint32_t u; memcpy(&u, &f, 4); return u; }
RM_FN float rm_u2f(uint32_t u) { float f; memcpy(&f, &u, 4); return f; }
// Pillow's (int) cast of a coordinate as x86-64 performs it (cvttsd2si): NaN and out-of-range give INT_MIN
RM_FN int rm_pil_int(double d) { return (d >= -2147483648.0 && d < 2147483648.0) ? (int)d : (int)0x80000000; }
RM_FN int rm_clamp16(int v) { return v < -32000 ? -32000 : (v > 32000 ? 32000 : v); }

// color_maps.py:21-23 (colorsys.hsv_to_rgb, then uint8 truncation)
RM_FN uint32_t rm_hsv_rgb(double h, double s, double v) {
  double r, g, b;
  if (s == 0.0) { r = g = b = v; }
  else {
    int i = (int)(h * 6.0);
    double f = (h * 6.0) - i;
    double p = v * (1.0 - s), q = v * (1.0 - s * f), t = v * (1.0 - s * (1.0 - f));
    i = ((i % 6) + 6) % 6;
    switch (i) {
      case 0: r = v; g = t; b = p; break;
      case 1: r = q; g = v; b = p; break;
      case 2: r = p; g = v; b = t; break;
      case 3: r = p; g = q; b = v; break;
      case 4: r = t; g = p; b = v; break;
      default: r = v; g = p; b = q; break;
    }
  }
  return ((uint32_t)(int)(255 * r) & 255u) | (((uint32_t)(int)(255 * g) & 255u) << 8) | (((uint32_t)(int)(255 * b) & 255u) << 16);
}

// Inclusive scan over the wavefront in six DPP adds: within rows of 16 lanes, then across the rows
#if RM_DEV
RM_FN int rm_wave_scan(int v) {
  v += __builtin_amdgcn_update_dpp(0, v, 0x111, 0xf, 0xf, false);   // row_shr:1
  v += __builtin_amdgcn_update_dpp(0, v, 0x112, 0xf, 0xf, false);   // row_shr:2
  v += __builtin_amdgcn_update_dpp(0, v, 0x114, 0xf, 0xf, false);   // row_shr:4
  v += __builtin_amdgcn_update_dpp(0, v, 0x118, 0xf, 0xf, false);   // row_shr:8
  v += __builtin_amdgcn_update_dpp(0, v, 0x142, 0xa, 0xf, false);   // row_bcast:15 -> rows 1, 3
  v += __builtin_amdgcn_update_dpp(0, v, 0x143, 0xc, 0xf, false);   // row_bcast:31 -> rows 2, 3
  return v;
}
#endif

RM_FN void rm_min(int32_t* p, int32_t v) {
#if RM_DEV
  atomicMin(p, v);
#else
  if (v < *p) *p = v;
#endif
}
RM_FN void rm_max(int32_t* p, int32_t v) {
#if RM_DEV
  atomicMax(p, v);
#else
  if (v > *p) *p = v;
#endif
}
RM_FN void rm_or32(int32_t* p, int32_t v) {
#if RM_DEV
  atomicOr(p, v);
#else
  *p |= v;
#endif
}
// what one wavefront's LDS writes need before its other lanes read them (device); nothing on the host, where a phase is a loop
RM_FN void rm_wave_sync() {
#if RM_DEV
  __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "wavefront");
  __builtin_amdgcn_wave_barrier();
#endif
}
// a loop over i = 0 .. n-1 shared out to the lanes of a wavefront (host model: lane < 0, one call does every lane's work)
#if RM_DEV
#define RM_LANES(i, n, lane) _Pragma("unroll 1") for (int i = (lane); i < (n); i += 64)
#else
#define RM_LANES(i, n, lane) for (int i = 0; i < (n); ++i)
#endif

// ---- the emitter ------------------------------------------------------------------------------------------------------------
// One wavefront per env.  SRC says where the env's record lies (HBM, or the step kernel's LDS copy with the colours in HBM):
//   int flags(s), nv(s), opa(s), voff(s), vcap(s);  uint32_t vinfo(idx)  (vertex slot -> sprite slot | index within the sprite << 8);
//   double col(s, c);  const double* vbase()  (x, y of vertex slot 0);  const double* pos(s)
// Lanes are sprites where the work is per sprite (liveness, colour, item records) and VERTEX SLOTS where it is per vertex
// (the integer points, their bounds): every vertex slot of the record is looked at once, 64 at a time, its loads coalesced.
// `sc` is scratch in LDS: 2 words per sprite slot, 4 per item (RM_EMIT_SCRATCH_WORDS).
//   P  the static prefix against its reference record (slot words by slot, vertices by vertex slot)         -> s_lo
//   A  per slot: live vertices, colour; ONE copy per sprite: the items' first points (scan)
//   B  per vertex slot: its point(s) -> the items' bounds (LDS atomics); one copy per sprite: stored straight away
//   C  per item: (copies: does it touch the canvas? first point by scan) rows on the canvas (scan) -> the item record
//   D  (copies only) per vertex slot: the visible copies' points
#define RM_EMIT_SCRATCH_WORDS(slots, items, copies) (((copies) > 1 ? 8 * (slots) : 0) + 2 * (slots) + 4 * (items))
// key (copies only): per slot the order-preserving keys of its smallest / largest x and y among the ordinary vertices (four 64-bit words);
// slot: [2s] nvl | irregular << 16, [2s + 1] rgba;  item: [4g] first point, [4g + 1] nvl, [4g + 2] ymin, [4g + 3] ymax (copies, until C:
// the bounds the sprite's irregular vertices left, [4g] xmin [4g + 1] xmax)
struct RmEmitScratch { long long* key; int32_t* slot; int32_t* item; };
RM_FN void rm_emit_scratch(int32_t* base, int slots, int copies, RmEmitScratch* sc) {   // (base: 8-byte aligned)
  sc->key = reinterpret_cast<long long*>(base);
  sc->slot = base + (copies > 1 ? 8 * slots : 0);
  sc->item = sc->slot + 2 * slots;
}
RM_FN long long rm_key(double d) { long long b; memcpy(&b, &d, 8); return b ^ ((b >> 63) & 0x7fffffffffffffffll); }
RM_FN double rm_unkey(long long k) { const long long b = k ^ ((k >> 63) & 0x7fffffffffffffffll); double d; memcpy(&d, &b, 8); return d; }
RM_FN void rm_min64(long long* p, long long v) {
#if RM_DEV
  atomicMin(p, v);
#else
  if (v < *p) *p = v;
#endif
}
RM_FN void rm_max64(long long* p, long long v) {
#if RM_DEV
  atomicMax(p, v);
#else
  if (v > *p) *p = v;
#endif
}

// The integer canvas point of copy cp of a vertex (pil_renderer.py:104-108: the scaled doubles through Pillow's (int));
// copy c of a torus is drawn at the offset (c / 3 - 1, c % 3 - 1) (polygon_modifiers.py:88-97), the first-person modifier
// translates everything by (fpx, fpy) (polygon_modifiers.py:41-64); one of the two at most.
RM_FN uint32_t rm_emit_point(const RmEmit& a, double x, double y, int cp, double fpx, double fpy, int* ix_out, int* iy_out) {
  double px = x, py = y;
  if (a.first_person) { px = px + fpx; py = py + fpy; }
  if (a.ncopy > 1) { px = x + (double)(cp / 3 - 1); py = y + (double)(cp % 3 - 1); }
  const int ix = rm_clamp16(rm_pil_int((double)a.scale_w * px)), iy = rm_clamp16(rm_pil_int((double)a.H * py));
  *ix_out = ix; *iy_out = iy;
  return (uint32_t)(uint16_t)ix | ((uint32_t)(uint16_t)iy << 16);
}

RM_FN int rm_emit_nvl(int nvs, int vcap) {
  int nvl = nvs < 0 ? 0 : (nvs > RM_BIG_NV ? RM_BIG_NV : nvs);
  return nvl > vcap ? vcap : nvl;
}

// P: a prefix slot's words / a prefix vertex slot's coordinates against the reference record, bit for bit
template <class SRC>
RM_FN bool rm_emit_prefix_slot_differs(const RmEmit& a, const SRC& src, int s) {
  const int flags = src.flags(s), nvs = src.nv(s), opa = src.opa(s);
  const double c0 = src.col(s, 0), c1 = src.col(s, 1), c2 = src.col(s, 2);
  const double r0 = a.sref_col[3 * s], r1 = a.sref_col[3 * s + 1], r2 = a.sref_col[3 * s + 2];
  uint64_t b0, b1, b2, q0, q1, q2;
  memcpy(&b0, &c0, 8); memcpy(&b1, &c1, 8); memcpy(&b2, &c2, 8); memcpy(&q0, &r0, 8); memcpy(&q1, &r1, 8); memcpy(&q2, &r2, 8);
  return (((flags ^ a.sref_flags[s]) & MOOG_F_ALIVE) != 0) | (nvs != a.sref_nv[s]) | (opa != a.sref_opa[s]) | (b0 != q0) | (b1 != q1) | (b2 != q2);
}
template <class SRC>
RM_FN bool rm_emit_prefix_vertex_differs(const RmEmit& a, const SRC& src, int idx) {
  const uint32_t vi = src.vinfo(idx);
  const int s = (int)(vi & 0xffu), k = (int)(vi >> 8);
  const double x = src.vbase()[2 * idx], y = src.vbase()[2 * idx + 1];
  const double rx = a.sref_v[2 * idx], ry = a.sref_v[2 * idx + 1];
  const bool live = (src.flags(s) & MOOG_F_ALIVE) != 0 && k < rm_emit_nvl(src.nv(s), src.vcap(s));
  uint64_t b0, b1, q0, q1;
  memcpy(&b0, &x, 8); memcpy(&b1, &y, 8); memcpy(&q0, &rx, 8); memcpy(&q1, &ry, 8);
  return live & ((b0 != q0) | (b1 != q1));
}

RM_FN int rm_rows_on_canvas(int y0, int y1, int H) {
  if (y0 < 0) y0 = 0;
  if (y1 > H - 1) y1 = H - 1;   // rows >= H draw nothing (hline clips)
  return y1 >= y0 ? y1 - y0 + 1 : 0;
}

template <class SRC>
RM_FN void rm_emit(const RmEmit& a, const SRC& src, int env, int lane, const RmEmitScratch& sc, int totv, long long* clk = nullptr) {
#if RM_DEV
#define RM_CLK(i) do { if (clk) clk[i] = (long long)__builtin_readcyclecounter(); } while (0)
#else
#define RM_CLK(i) do { (void)clk; } while (0)
#endif
  RM_CLK(0);
  uint8_t* const rec = a.out + (size_t)env * a.lay.stride;
  RmDrawItem* const items = reinterpret_cast<RmDrawItem*>(rec + a.lay.o_items);
  uint32_t* const pts = reinterpret_cast<uint32_t*>(rec + a.lay.o_pts);
  uint8_t* const owner = rec + a.lay.o_owner;
  const bool copies = a.ncopy > 1;
  // (Code size matters here as much as instruction count: in the step kernel this runs once per env, at the very end, behind
  //  a quarter of a megabyte of hotter code -- every 64-byte line of it is an instruction-cache miss.  The loops over the
  //  vertex slots are kept rolled (one copy of their body), with the next round's slot number fetched a round ahead.)
  // first-person frames: everything is translated so that the agent layer's first sprite sits at (0.5, 0.5)
  double fpx = 0.0, fpy = 0.0;
  if (a.first_person) {
    for (int s = a.fp_slot0; s < a.fp_slot0 + a.fp_nslots; ++s)
      if (src.flags(s) & MOOG_F_ALIVE) { const double* p = src.pos(s); fpx = 0.5 - p[0]; fpy = 0.5 - p[1]; break; }
  }
  // P: the static prefix (never under copies: torus frames are drawn whole)
  const int NS = copies ? 0 : a.n_static;
  bool bad = false;
  if (NS > 0) {
    RM_LANES(s, NS, lane) bad = bad | rm_emit_prefix_slot_differs(a, src, s);
    const int nsv = src.voff(NS - 1) + src.vcap(NS - 1);   // (the prefix's vertex slots are the first of the record)
    RM_LANES(idx, nsv, lane) bad = bad | rm_emit_prefix_vertex_differs(a, src, idx);
  }
  RM_CLK(1);
  // A, first half: the slots' colours (in HBM for the step kernel: loaded before the vote on the prefix is waited for)
  int run_pts = 0;
#if RM_DEV
  for (int i0 = 0; i0 < a.slots; i0 += 64) {
    const int s = i0 + lane;
    const bool in = s < a.slots;
#else
  bad = RM_ANY(bad);
  for (int s = 0; s < a.slots; ++s) {
    const bool in = true;
#endif
    int nvl = 0;
    uint32_t rgba = 0u;
    if (in) {
      const int flags = src.flags(s), opa = src.opa(s);
      const double c0 = src.col(s, 0), c1 = src.col(s, 1), c2 = src.col(s, 2);
      const uint32_t ov = a.rgb_override ? a.rgb_override[(size_t)env * a.slots + s] : 0u;
      if (flags & MOOG_F_ALIVE) {
        uint32_t rgb;
        if (a.rgb_override) rgb = ov & 0xffffffu;
        else if (a.cmap == MOOG_CMAP_HSV) rgb = rm_hsv_rgb(c0, c1, c2);
        else rgb = ((uint32_t)(int)c0 & 255u) | (((uint32_t)(int)c1 & 255u) << 8) | (((uint32_t)(int)c2 & 255u) << 16);
        rgba = rgb | (((uint32_t)opa & 255u) << 24);
        nvl = rm_emit_nvl(src.nv(s), src.vcap(s));
      }
    }
#if RM_DEV
    if (i0 == 0) bad = RM_ANY(bad);   // (the vote, behind the first round's loads)
#endif
    const int s_lo = bad ? 0 : NS;
    if (s < s_lo) nvl = 0;
    int first = 0;
    if (!copies) {
#if RM_DEV
      const int inc = rm_wave_scan(nvl);
      first = run_pts + inc - nvl;
      run_pts += __builtin_amdgcn_readlane(inc, 63);
#else
      first = run_pts;
      run_pts += nvl;
#endif
    }
    if (in) {
      sc.slot[2 * s] = nvl; sc.slot[2 * s + 1] = (int32_t)rgba;
      if (copies) {
        long long* kk = sc.key + 4 * s;
        kk[0] = 0x7fffffffffffffffll; kk[1] = -0x7fffffffffffffffll - 1; kk[2] = kk[0]; kk[3] = kk[1];
      }
      for (int cp = 0; cp < a.ncopy; ++cp) {
        int32_t* it = sc.item + 4 * (s * a.ncopy + cp);
        it[0] = copies ? 0x7fffffff : first; it[1] = copies ? -0x7fffffff : nvl; it[2] = 0x7fffffff; it[3] = -0x7fffffff;
      }
    }
  }
  rm_wave_sync();
  RM_CLK(2);
  // B: per vertex slot.  One copy per sprite: the point, stored where the item's points go, and the item's y range.  Copies:
  // Pillow truncates the scaled coordinates towards zero, so a copy's integer points are not the sprite's shifted by a canvas --
  // but x -> (int)(W * (x + o)) is monotone, so a copy's integer bounds are those of the sprite's smallest and largest
  // coordinates: four 64-bit atomics per vertex on order-preserving keys.  A vertex with a coordinate that is not an ordinary number
  // (NaN, or beyond what (int) holds: Pillow's cast then gives INT_MIN, which is not monotone) leaves its nine points' y behind instead.
  auto vertex_b = [&](int idx, uint32_t vi, double x, double y) {
    const int s = (int)(vi & 0xffu), k = (int)(vi >> 8);
    const int nvl = sc.slot[2 * s] & 0xffff;
    if (k >= nvl) return;
    if (!copies) {
      int ix, iy;
      const uint32_t p = rm_emit_point(a, x, y, 0, fpx, fpy, &ix, &iy);
      int32_t* it = sc.item + 4 * s;
      pts[it[0] + k] = p; owner[it[0] + k] = (uint8_t)s;
      rm_min(it + 2, iy); rm_max(it + 3, iy);
    } else {
      // ordinary: every copy's scaled coordinate is far inside what (int) holds (NaN fails the comparisons)
      const bool ordinary = fabs(x) < 1.0e6 && fabs(y) < 1.0e6;
      if (ordinary) {
        long long* kk = sc.key + 4 * s;
        rm_min64(kk + 0, rm_key(x)); rm_max64(kk + 1, rm_key(x));
        rm_min64(kk + 2, rm_key(y)); rm_max64(kk + 3, rm_key(y));
      } else {
        rm_or32(sc.slot + 2 * s, 1 << 16);
        for (int cp = 0; cp < a.ncopy; ++cp) {   // (the bounds of such a sprite's copies: point by point; its x range is not tracked)
          int ix, iy;
          rm_emit_point(a, x, y, cp, fpx, fpy, &ix, &iy);
          int32_t* it = sc.item + 4 * (s * a.ncopy + cp);
          rm_min(it + 2, iy); rm_max(it + 3, iy);
        }
      }
    }
  };
#if RM_DEV
  {
    // (the table entries come from global memory, ~700 cycles each, a round's work is ~250: three rounds are in flight; the
    //  coordinates too when the record lies in global memory -- the derive kernel: SRC::kGlobalRecord -- one round ahead)
    uint32_t v0 = lane < totv ? src.vinfo(lane) : 0u, v1 = lane + 64 < totv ? src.vinfo(lane + 64) : 0u,
             v2 = lane + 128 < totv ? src.vinfo(lane + 128) : 0u;
    double xn = 0.0, yn = 0.0;
    if (SRC::kGlobalRecord && lane < totv) { xn = src.vbase()[2 * lane]; yn = src.vbase()[2 * lane + 1]; }
#pragma unroll 1
    for (int idx = lane; idx < totv; idx += 64) {
      const uint32_t v_now = v0;
      v0 = v1; v1 = v2;
      if (idx + 192 < totv) v2 = src.vinfo(idx + 192);
      double x, y;
      if (SRC::kGlobalRecord) {
        x = xn; y = yn;
        if (idx + 64 < totv) { xn = src.vbase()[2 * (idx + 64)]; yn = src.vbase()[2 * (idx + 64) + 1]; }
      } else { x = src.vbase()[2 * idx]; y = src.vbase()[2 * idx + 1]; }
      vertex_b(idx, v_now, x, y);
    }
  }
#else
  for (int idx = 0; idx < totv; ++idx) vertex_b(idx, src.vinfo(idx), src.vbase()[2 * idx], src.vbase()[2 * idx + 1]);
#endif
  rm_wave_sync();
  RM_CLK(3);
  // C: per item
  int run_rows = 0;
#if RM_DEV
  for (int i0 = 0; i0 < a.S; i0 += 64) {
    const int g = i0 + lane;
    const bool in = g < a.S;
#else
  for (int g = 0; g < a.S; ++g) {
    const bool in = true;
#endif
    const int s = in ? g / a.ncopy : 0, cp = in ? g - s * a.ncopy : 0;
    int32_t* it = sc.item + 4 * (in ? g : 0);
    int y0 = in ? it[2] : 0x7fffffff, y1 = in ? it[3] : -0x7fffffff;
    int nvl = in ? (sc.slot[2 * s] & 0xffff) : 0;
    int first = in ? it[0] : 0;
    if (copies) {
      // A copy whose points all lie two or more pixels beside the canvas paints nothing: its crossings are float32
      // interpolations between such points (off by far less than a pixel at these magnitudes), its heads lie between them.
      const bool irregular = in && (sc.slot[2 * s] >> 16) != 0;
      int x0 = 0x7fffffff, x1 = -0x7fffffff;
      const long long* kk = sc.key + 4 * s;
      if (in && kk[0] <= kk[1]) {   // the ordinary vertices: the copy's bounds from the sprite's extreme coordinates
        int ixa, iya, ixb, iyb;
        rm_emit_point(a, rm_unkey(kk[0]), rm_unkey(kk[2]), cp, fpx, fpy, &ixa, &iya);
        rm_emit_point(a, rm_unkey(kk[1]), rm_unkey(kk[3]), cp, fpx, fpy, &ixb, &iyb);
        x0 = ixa; x1 = ixb;
        y0 = iya < y0 ? iya : y0; y1 = iyb > y1 ? iyb : y1;
      }
      // (an irregular sprite keeps every copy its rows put on the canvas: its x range is not tracked)
      const bool vis = nvl > 0 && y1 >= 0 && y0 <= a.H - 1 && (irregular || (x1 >= -1 && x0 <= a.W));
      if (!vis) nvl = 0;
#if RM_DEV
      const int inc = rm_wave_scan(nvl);
      first = run_pts + inc - nvl;
      run_pts += __builtin_amdgcn_readlane(inc, 63);
#else
      first = run_pts;
      run_pts += nvl;
#endif
      if (in) { it[0] = first; it[1] = nvl; }
    }
    const int cnt = nvl > 0 ? rm_rows_on_canvas(y0, y1, a.H) : 0;
#if RM_DEV
    const int rinc = rm_wave_scan(cnt);
    const int rowoff = run_rows + rinc - cnt;
    run_rows += __builtin_amdgcn_readlane(rinc, 63);
#else
    const int rowoff = run_rows;
    run_rows += cnt;
#endif
    if (in) {
      RmDrawItem o;
      o.rowoff = rowoff;
      o.pb_nv = (uint32_t)first | ((uint32_t)nvl << 20);
      o.y01 = nvl > 0 ? (int32_t)((uint32_t)(uint16_t)y0 | ((uint32_t)(uint16_t)y1 << 16)) : RM_Y01_EMPTY;
      o.rgba = nvl > 0 ? (uint32_t)sc.slot[2 * s + 1] : 0u;
      items[g] = o;
    }
  }
  if (copies) {   // D: the visible copies' points
    rm_wave_sync();
    auto vertex_d = [&](int idx, uint32_t vi) {
      const int s = (int)(vi & 0xffu), k = (int)(vi >> 8);
      if (k >= (sc.slot[2 * s] & 0xffff)) return;
      const double x = src.vbase()[2 * idx], y = src.vbase()[2 * idx + 1];
      for (int cp = 0; cp < a.ncopy; ++cp) {
        const int g = s * a.ncopy + cp;
        const int32_t* it = sc.item + 4 * g;
        if (k >= it[1]) continue;
        int ix, iy;
        pts[it[0] + k] = rm_emit_point(a, x, y, cp, fpx, fpy, &ix, &iy);
        owner[it[0] + k] = (uint8_t)g;
      }
    };
#if RM_DEV
    uint32_t v_next = lane < totv ? src.vinfo(lane) : 0u;
#pragma unroll 1
    for (int idx = lane; idx < totv; idx += 64) {
      const uint32_t v_now = v_next;
      if (idx + 64 < totv) v_next = src.vinfo(idx + 64);
      vertex_d(idx, v_now);
    }
#else
    for (int idx = 0; idx < totv; ++idx) vertex_d(idx, src.vinfo(idx));
#endif
  }
  RM_CLK(4);
#if RM_DEV
  if (lane == 0)
#endif
  {
    RmDrawHdr h;
    h.n_pts = run_pts; h.total_rows = run_rows; h.flags = (NS > 0 && !bad) ? RM_DRAW_PREFIX_OK : 0; h.pad = 0;
    *reinterpret_cast<RmDrawHdr*>(rec) = h;
  }
}

// The record in HBM as the ABI lays it out (the derive kernel, the host model)
struct RmSrcRecord {
  static constexpr bool kGlobalRecord = true;   // (every read is a global load: the emitter fetches the coordinates a round ahead)
  const moog_program_t* P; const moog_layout_t* L; const double* f; const int32_t* q; const uint32_t* vi;   // vi: vertex slot -> sprite slot | index within the sprite << 8
  RM_MEMBER int flags(int s) const { return q[L->o_flags + s]; }
  RM_MEMBER int nv(int s) const { return q[L->o_nverts + s]; }
  RM_MEMBER int opa(int s) const { return q[L->o_opacity + s]; }
  RM_MEMBER int voff(int s) const { return P->slot_voff[s]; }
  RM_MEMBER int vcap(int s) const { return P->slot_vcap[s]; }
  RM_MEMBER double col(int s, int c) const { return f[L->o_color + 3 * s + c]; }
  RM_MEMBER uint32_t vinfo(int idx) const { return vi[idx]; }
  RM_MEMBER const double* vbase() const { return f + L->o_verts; }
  RM_MEMBER const double* pos(int s) const { return f + L->o_pos + 2 * s; }
};

#endif  // MOOG_DRAW_RECORD_H_
