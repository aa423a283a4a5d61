// moog_drawlist.h -- the per-env DRAW LIST: what PILRenderer.__call__ hands to ImageDraw.polygon
// (reference moog/observers/pil_renderer.py:100-110: for every sprite in layer order,
// `vertices = self._canvas_size * sprite.vertices`, then `tuple(v) for v in vertices` truncated by Pillow's (int) cast),
// already taken through the part of ImagingDrawPolygon / polygon_generic that only depends on the polygon itself:
//   * the integer canvas point of every LIVE vertex (Pillow's x86 cast, clamped to +-32000 as the rasterisers always did),
//   * the EDGE leaving it (Draw.c add_edge: end points, dx; horizontal runs merged into "heads" as ImagingDrawPolygon does),
//   * the edge's CORNER FIX-UP values (polygon_generic's "connect discontiguous corners": a search among the polygon's
//     earlier edges, quadratic in the polygon's size, independent of the rows it covers),
//   * every sprite's row range.
// The step kernel holds every world vertex in LDS when it stores the record, so it emits the list there (one wavefront,
// lanes = vertices, no barrier; 16 + 4 bytes per live vertex instead of 16 bytes of f64 per vertex SLOT), and the
// rasterisers start at the crossings.  `moog_drawlist_kernel` builds the same list from a state record in HBM for frames
// of states the step kernel did not produce (moog_engine_render on uploaded state, the frame of an explicit reset).
//
// Layout per env, 32-bit words, stride dl_stride_words(..):
//   [0] rounds   [1] items (live sprites with vertices)   [2] entries in use (incl. padding)   [3] spare
//   [4..7]    lanes used per round, one byte each (<= DL_MAX_ROUNDS rounds)
//   [8..11]   first item of every round, one byte each (0xff: the round is not used)
//   [12..27]  item of slot s, one byte each (0xff: none; items are the live sprites in slot = painter's order, <= 64)
//   [28..91]  row range of item g: ymin | ymax << 16 (int16 each)
//   [92..]    rounds of 64 entries: 64 edge records (REdge, 4 words each: x0 y0 x1 y1 | dx | vtop vbot), then 64 info
//             words: item | k << 8 | nv << 16 | head << 23 | slot << 24   (k: index of the vertex within its sprite;
//             head: the edge is a horizontal head, dx then holds xmin | xmax << 16)
// A ROUND is what the 64 lanes of a wavefront process together.  A sprite's vertices never straddle two rounds (a sprite
// that does not fit into the current round starts the next one), so every neighbour an edge needs -- next / previous
// vertex, the polygon's earlier edges -- is in the same round.  Entries of a round beyond its "lanes used" byte are not
// written and not read.
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>

#include "moog_raster.h"

#define DL_MAX_ROUNDS 16
#define DL_HDR 92          // words in front of the rounds
#define DL_ROUND_WORDS 320 // 64 edge records (4 words) + 64 info words
#define DL_MAX_ITEMS 64
#define DL_MAX_NV 32       // vertices per sprite (one word of head bits per row in the rasteriser)
#define DL_INFO_HEAD 0x800000u

// LDS scratch of drawlist_emit (bytes): A = edge attributes + item tables, B = per-slot table + the round's points
#define DL_SCRATCH_A (64 * 24 + 64 * 8 + 64 * 2 + 64)
#define DL_SCRATCH_B ((96 + 68) * 4)

// Worst case of the greedy packing: a round is closed only when the next sprite (<= maxv vertices) does not
// fit, i.e. when it holds at least 65 - maxv entries.
__host__ __device__ inline int dl_max_rounds(int TOTV, int maxv) {
  const int per = 65 - (maxv < 1 ? 1 : maxv);
  const int r = (TOTV + per - 1) / per;
  return r < 1 ? 1 : r;
}
__host__ __device__ inline int dl_stride_words(int max_rounds) { return DL_HDR + DL_ROUND_WORDS * max_rounds; }
__host__ __device__ inline const uint4* dl_edges(const uint32_t* dl, int r) { return reinterpret_cast<const uint4*>(dl + DL_HDR + DL_ROUND_WORDS * r); }
__host__ __device__ inline const uint32_t* dl_infos(const uint32_t* dl, int r) { return dl + DL_HDR + DL_ROUND_WORDS * r + 256; }

// Pillow's (int) cast of a coordinate as x86-64 performs it (cvttsd2si): NaN and values outside the int range
// give INT_MIN (the reference can produce NaN sprite state); then the +-32000 clamp of the
// rasteriser's 16-bit points (such polygons are off the canvas either way).
__device__ __forceinline__ int dl_pil_int(double d) {
  return (d >= -2147483648.0 && d < 2147483648.0) ? (int)d : (int)0x80000000;
}
__device__ __forceinline__ unsigned dl_clamp16(int v) {
  return (unsigned)(unsigned short)(short)(v < -32000 ? -32000 : (v > 32000 ? 32000 : v));
}
// Draw.c ROUND_UP (sign-symmetric)
__device__ __forceinline__ int dl_round_up(float f) { return (int)copysignf(floorf(fabsf(f) + 0.5f), f); }
__device__ __forceinline__ unsigned dl_pack(int x, int y) { return (unsigned)(unsigned short)x | ((unsigned)(unsigned short)y << 16); }

// Corner fix-up search.  polygon_generic lets an EARLIER table edge K decide the fix-up of edge E's first (last) row when
// K has the same upper (lower) end point, leans the same way (sign of dx, never 0) and crosses that row at the same x;
// the first such K in edge order decides (tip_decide in moog_raster_kernel.h, whose results this reproduces).  All of
// that is a property of the two edges alone, so every edge publishes an ATTRIBUTE record and an edge scans the records of
// its polygon's earlier edges once, in order (lanes of a polygon read the same record: a broadcast).  Edges that cannot
// decide anything (horizontal, vertical) publish points no vertex can have.
struct DLAttr { unsigned tw, xt, bw, xb, p0; float dx; };   // upper point, x on the first row, lower point, x on the last row, start point, dx
#define DL_NOPOINT 0x80008000u   // (canvas points are clamped to +-32000)

// tip_decide's replacement value once K is known to decide: E's and K's crossings of the row next to the tip row
__device__ __forceinline__ short dl_tip_value(float edx, int ex0, int ey0, const DLAttr& K, int y, float x, bool top) {
  const int off = top ? 1 : -1;
  const int kx0 = (short)(K.p0 & 0xffffu), ky0 = (short)(K.p0 >> 16);
  const float adj = (float)(y + off - ey0) * edx + (float)ex0;
  const float adjo = (float)(y + off - ky0) * K.dx + (float)kx0;
  short vv = R_NONE;
  if (adj > x && adjo > x) {
    const float v = (float)(dl_round_up(fminf(adj, adjo)) - 1);
    if (v > x) vv = (short)(int)v;
  } else if (adj < x && adjo < x) {
    const float v = (float)(dl_round_up(fmaxf(adj, adjo)) + 1);
    if (v < x) vv = (short)(int)v;
  }
  return vv;
}

// One wavefront builds one env's draw list.  flags / nverts / verts: the env's record fields (LDS in the step kernel, HBM
// in moog_drawlist_kernel); voff: vertex offset of every slot (S entries); sa / sb: DL_SCRATCH_A / DL_SCRATCH_B bytes of
// LDS scratch (16-byte aligned).  S <= 64, every sprite <= DL_MAX_NV vertices.  THROUGH: agent-scope stores (the list is
// about to be read by a kernel that is running now).
// deep = false: only the packed points and the info words are written (what the workgroup rasteriser reads); the edge stage
// -- five serial rounds of LDS round trips for a wavefront that, at the end of a heavy env's step, runs alone on its SIMD --
// costs the step kernel 4 % when it is on (785 -> 815 us on the headline workload), more than the wave rasteriser saves.
template <bool THROUGH>
__device__ inline void drawlist_emit(uint32_t* __restrict__ out, const int S, const int32_t* flags, const int32_t* nverts,
                                     const double* verts, const int32_t* voff, const int CW, const int CH, const int lane,
                                     unsigned char* sa, unsigned char* sb, const bool deep) {
  DLAttr* attr = reinterpret_cast<DLAttr*>(sa);
  int* iy = reinterpret_cast<int*>(sa + 64 * 24);                       // per item: ymin, ymax
  unsigned short* ipos = reinterpret_cast<unsigned short*>(sa + 64 * 24 + 64 * 8);   // per item: first entry
  unsigned char* islot = sa + 64 * 24 + 64 * 8 + 64 * 2;                // per item: slot
  uint32_t* tbl = reinterpret_cast<uint32_t*>(sb);                      // [0..63] per slot, [64..79] lanes per round, [80..95] first item per round
  unsigned* rpts = reinterpret_cast<unsigned*>(sb) + 96;                // the round's packed points (+ slack)
  auto lds_sync = [] {
    __builtin_amdgcn_fence(__ATOMIC_SEQ_CST, "workgroup", "local");
    __builtin_amdgcn_wave_barrier();
  };
  auto put = [&](uint32_t* p, unsigned v) {
    if (THROUGH) __hip_atomic_store(p, v, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    else *p = v;
  };
  int nv = 0;
  if (lane < S) {
    const int f = flags[lane];
    nv = (f & MOOG_F_ALIVE) ? nverts[lane] : 0;
    if (nv < 0) nv = 0;
    if (nv > DL_MAX_NV) nv = DL_MAX_NV;
  }
  const unsigned long long live = __ballot(nv > 0);
  const int g = __popcll(live & ((1ull << lane) - 1ull));
  const int n_items = __popcll(live);
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
  lds_sync();   // (the scratch may hold live data of the caller up to here)
  if (lane < 32) tbl[64 + lane] = lane < 16 ? 0u : 255u;
  iy[2 * lane] = 0x7fffffff; iy[2 * lane + 1] = -0x7fffffff;
  lds_sync();
  tbl[lane] = (unsigned)mypos | ((unsigned)g << 10) | ((unsigned)nv << 16);
  if (nv > 0) {
    atomicMax(&tbl[64 + (mypos >> 6)], (unsigned)((mypos & 63) + nv));
    atomicMin(&tbl[80 + (mypos >> 6)], (unsigned)g);
    ipos[g] = (unsigned short)mypos;
    islot[g] = (unsigned char)lane;
  }
  lds_sync();
  if (lane == 0) { put(out + 0, (unsigned)n_rounds); put(out + 1, (unsigned)n_items); put(out + 2, (unsigned)cur); put(out + 3, 0u); }
  if (lane < 8) {   // lanes used per round (words 4..7), first item per round (words 8..11)
    const int b = 16 * (lane >> 2) + 64 + 4 * (lane & 3);
    const unsigned w = (tbl[b] & 255u) | ((tbl[b + 1] & 255u) << 8) | ((tbl[b + 2] & 255u) << 16) | ((tbl[b + 3] & 255u) << 24);
    put(out + 4 + lane, w);
  }
  {   // item of every slot: four slots per word
    const unsigned mine = nv > 0 ? (unsigned)g : 255u;
    const unsigned b1 = (unsigned)__shfl_down((int)mine, 1), b2 = (unsigned)__shfl_down((int)mine, 2), b3 = (unsigned)__shfl_down((int)mine, 3);
    if ((lane & 3) == 0) put(out + 12 + (lane >> 2), mine | (b1 << 8) | (b2 << 16) | (b3 << 24));
  }
  // ---- rounds: lane = entry = the vertex and the edge leaving it --------------------------------------------------
  for (int r = 0; r < n_rounds; ++r) {
    const int cnt_r = (int)(tbl[64 + r] & 255u);
    const bool valid = lane < cnt_r;
    const int p = 64 * r + lane;
    int gi = (int)(tbl[80 + r] & 255u);   // the round's first item; the entry's item: the last one that starts at or before it
    if (valid) while (gi + 1 < n_items && (int)ipos[gi + 1] <= p) ++gi;
    const int s = valid ? (int)islot[gi] : 0;
    const int k = valid ? p - (int)ipos[gi] : 0;
    const int nvs = valid ? (int)((tbl[s] >> 16) & 255u) : 0;
    unsigned pt = 0u;
    if (valid) {
      const int idx = voff[s] + k;
      const double vx = verts[2 * idx], vy = verts[2 * idx + 1];
      pt = dl_clamp16(dl_pil_int((double)CW * vx)) | (dl_clamp16(dl_pil_int((double)CH * vy)) << 16);
    }
    const int x0 = (short)(pt & 0xffffu), y0 = (short)(pt >> 16);
    if (!deep) {   // the point and the info word only
      if (valid) {
        const unsigned info = (unsigned)gi | ((unsigned)k << 8) | ((unsigned)nvs << 16) | ((unsigned)s << 24);
        put(out + DL_HDR + DL_ROUND_WORDS * r + 4 * lane, pt);
        put(out + DL_HDR + DL_ROUND_WORDS * r + 256 + lane, info);
      }
      continue;
    }
    rpts[lane] = pt;
    if (valid) { atomicMin(&iy[2 * gi], y0); atomicMax(&iy[2 * gi + 1], y0); }
    lds_sync();
    // ---- the edge leaving the vertex (ImagingDrawPolygon: add_edge + merge of horizontal runs) --------------------
    const short2* pv = reinterpret_cast<const short2*>(rpts) + (lane - k);
    const int iy1 = valid ? iy[2 * gi + 1] : 0;
    const bool closing = (k + 1 == nvs);
    const short2 p1 = pv[closing ? 0 : k + 1];
    const short2 pp = pv[k >= 1 ? k - 1 : 0];
    const int x1 = p1.x, y1 = p1.y;
    const bool tbl_e = valid && y0 != y1;
    bool head = valid && y0 == y1 && !(closing && x0 == x1);   // last == first: no closing edge
    if (k >= 1 && !closing) {
      const bool ab = pp.y == y0 && ((x1 > x0 && x0 > pp.x) || (x1 < x0 && x0 < pp.x));
      // three equal vertices in a row (tiny circles): this zero-length head repeats the one before it
      const bool rep = pp.x == x0 && pp.y == y0 && x1 == x0;
      head = head && !ab && !rep;
    }
    float dx = ((float)(x1 - x0)) / (float)(tbl_e ? y1 - y0 : 1);
    if (!tbl_e) dx = 0.0f;
    float dxw = dx;   // what the record carries: a head's extent instead
    if (__any(head)) {
      if (head) {   // extend over the following absorbed edges (never the closing edge)
        short hx = (short)x1;
        int q = k + 1;
        short2 prev, curp;
        prev.x = (short)x0; prev.y = (short)y0; curp = p1;
        while (q <= nvs - 2) {
          const short2 nxt = pv[q + 1];
          const bool ab = (curp.y == nxt.y) && (prev.y == curp.y) &&
                          ((nxt.x > curp.x && curp.x > prev.x) || (nxt.x < curp.x && curp.x < prev.x));
          if (!ab) break;
          hx = nxt.x; prev = curp; curp = nxt; ++q;
        }
        const short xmin = x0 < hx ? (short)x0 : hx, xmax = x0 < hx ? hx : (short)x0;
        dxw = __int_as_float((int)((unsigned)(unsigned short)xmin | ((unsigned)(unsigned short)xmax << 16)));
      }
    }
    // ---- corner fix-up partners: publish the attributes, scan the polygon's earlier edges ---------------------------
    const int emin = y0 < y1 ? y0 : y1, emax = y0 < y1 ? y1 : y0;
    const bool lean = tbl_e && dx != 0.0f;
    DLAttr mine;
    {
      const bool up = y0 < y1;
      const unsigned w0 = dl_pack(x0, y0), w1 = dl_pack(x1, y1);
      mine.tw = lean ? (up ? w0 : w1) : DL_NOPOINT;
      mine.bw = lean ? (up ? w1 : w0) : DL_NOPOINT;
      mine.xt = __float_as_uint((float)(emin - y0) * dx + (float)x0);
      mine.xb = __float_as_uint((float)(emax - y0) * dx + (float)x0);
      mine.p0 = pt; mine.dx = dx;
    }
    attr[lane] = mine;
    lds_sync();
    const int pymax = iy1 > CH ? CH : iy1;    // polygon_generic clamps ymax to ysize
    bool open_top = lean && emin >= 0 && emin < CH, open_bot = lean && emax < CH && emax >= pymax;
    int kt = -1, kb = -1;
    {
      const DLAttr* pa = attr + (lane - k);
      const unsigned dxb = __float_as_uint(dx);
      for (int e = 0; __any(e < k && (open_top || open_bot)); ++e) {
        const DLAttr A = pa[e];
        const bool act = e < k && ((__float_as_uint(A.dx) ^ dxb) >> 31) == 0u;   // leans the same way (a partner's dx is never 0)
        const bool ht = act && open_top && A.tw == mine.tw && A.xt == mine.xt;
        const bool hb = act && open_bot && A.bw == mine.bw && A.xb == mine.xb;
        kt = ht ? e : kt; open_top = open_top && !ht;
        kb = hb ? e : kb; open_bot = open_bot && !hb;
      }
    }
    short vt = R_NONE, vb = R_NONE;
    if (__any(kt >= 0 || kb >= 0)) {
      const DLAttr* pa = attr + (lane - k);
      const DLAttr KT = pa[kt >= 0 ? kt : 0], KB = pa[kb >= 0 ? kb : 0];
      const short a = dl_tip_value(dx, x0, y0, KT, emin, __uint_as_float(mine.xt), true);
      const short b = dl_tip_value(dx, x0, y0, KB, emax, __uint_as_float(mine.xb), false);
      vt = kt >= 0 ? a : R_NONE;
      vb = kb >= 0 ? b : R_NONE;
    }
    if (valid) {
      uint32_t* pe = out + DL_HDR + DL_ROUND_WORDS * r + 4 * lane;
      const unsigned w0 = pt, w1 = dl_pack(x1, y1), w2 = __float_as_uint(dxw),
                     w3 = (unsigned)(unsigned short)vt | ((unsigned)(unsigned short)vb << 16);
      const unsigned info = (unsigned)gi | ((unsigned)k << 8) | ((unsigned)nvs << 16) | (head ? DL_INFO_HEAD : 0u) | ((unsigned)s << 24);
      uint32_t* pi = out + DL_HDR + DL_ROUND_WORDS * r + 256 + lane;
      if (THROUGH) {
        unsigned long long* p64 = reinterpret_cast<unsigned long long*>(pe);
        __hip_atomic_store(p64, (unsigned long long)w0 | ((unsigned long long)w1 << 32), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        __hip_atomic_store(p64 + 1, (unsigned long long)w2 | ((unsigned long long)w3 << 32), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        __hip_atomic_store(pi, info, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
      } else {
        *reinterpret_cast<uint4*>(pe) = make_uint4(w0, w1, w2, w3);
        *pi = info;
      }
    }
    lds_sync();   // (the next round overwrites the points and the attributes)
  }
  // every item's row range (items beyond n_items are never read)
  if (deep && lane < n_items) {
    const int a = iy[2 * lane], b = iy[2 * lane + 1];
    put(out + 28 + lane, dl_pack(a < -32000 ? -32000 : (a > 32000 ? 32000 : a), b < -32000 ? -32000 : (b > 32000 ? 32000 : b)));
  }
}
