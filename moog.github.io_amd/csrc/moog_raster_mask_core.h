// moog_raster_mask_core.h -- the "mask" rasteriser: Pillow-exact polygon fill without sorting crossings.
//
// Replaces, for one-tile frames (<= 128 x 128 canvas, polygons of <= 128 vertices -- census words up to 32, an indexed edge list
// beyond --, <= 256 items; plain frames, first-person frames and the nine copies per sprite of a torus), the push / sort / span
// pipeline of moog_raster_kernel.h.  Same contract: bit-exact with ImageDraw.polygon in RGBA blend mode as PILRenderer uses it
// (reference moog/observers/pil_renderer.py:88-120; Pillow Draw.c ImagingDrawPolygon / polygon_generic(hasAlpha = 1) /
// hline32rgba as restated in oracle/moog_oracle.c).
//
// What makes it cheap (the frames of the headline workload hold ~28 polygons that are 3-6 pixels tall):
//  * No crossing lists.  polygon_generic sorts a row's crossings xx[] and fills [ROUND_UP(xx[2q]), ROUND_DOWN(xx[2q+1])]
//    for every pair, left to right, never painting a pixel twice (x_pos).  The union of those spans is
//        XOR_x high(ROUND_DOWN(x) + 1)   |   OR_x pixels[ROUND_UP(x) .. ROUND_DOWN(x)]
//    (high(t) = pixels t, t + 1, ...): pixel p is covered iff the number of crossings with ROUND_DOWN(x) < p is odd, or
//    some crossing rounds onto p (the second term is one pixel, none for a positive half-integer).  Both terms are commutative, so a row's mask is accumulated in any order, in registers.
//    (A closed polygon crosses a row an even number of times with Pillow's counting -- an edge's last row counts twice,
//    a local extremum 2 or 4 times; a row with an odd count takes the generic routine.)
//  * Census by bit mask.  Polygons have <= 32 edges, so "which edges are active on row y", "which are horizontal heads
//    there" and "which have a corner there, leaning left / right" are four 32-bit words per (polygon, row), OR-ed
//    together by the edges (LDS atomics without return).  The row's thread then walks the set bits: only edges that
//    really cross the row cost anything.
//  * polygon_generic's corner fix-up ("connect discontiguous corners") is resolved per ROW from the two corner words:
//    fewer than two edges of one lean with a corner on the row = nothing to do (the common case, no edge is read).
//  * Horizontal heads (draw_horizontal_lines) only need the pen position x_pos at the moments Pillow looks at it; those
//    are the ends of the runs of the span mask, in order.
//  * Compose works on the frame's bytes as they lie in memory (48 bytes per 16 pixels): a 16-entry table turns four
//    coverage bits into three dwords of byte masks, one v_bfi_b32 per dword paints an opaque polygon.
//
// The file is host + device code: tests/test_raster_mask_model.py compiles it with g++ and runs whole frames through
// the same functions, thread by thread, against the oracle renderer (no GPU needed to find a logic error).
#ifndef MOOG_RASTER_MASK_CORE_H_
#define MOOG_RASTER_MASK_CORE_H_
#include "moog_draw_record.h"   // the frame's input: integer points, colours, row ranges (written by the step / reset / derive kernels)

#ifndef RM_LOAD_ASSIGN
#define RM_LOAD_ASSIGN 1      // 0: rm_p2_assign for every pass (A/B builds)
#endif
#ifndef RM_THREADS
#define RM_THREADS 128        // threads per frame (256 and 64 were measured: slower, profiles/r05_raster.txt)
#endif
#ifndef RM_WAVES_PER_SIMD
#define RM_WAVES_PER_SIMD 5   // register budget: 96 VGPRs
#endif
#define RM_SORT_ROUNDS_T (3 * RM_THREADS)   // = RM_SORT_ROUNDS * RM_THREADS (defined with the sort, below)
#define RM_XX (2 * RM_BIG_NV)  // crossing-list capacity of the generic row routine (2 per edge)

struct alignas(16) RmEdge { uint32_t w0, w1, w2, w3; };   // table edge: float x0 | float dx | y0, y1 (shorts) | 0;  head: xmin, xmax (shorts) | 0 | y, y | 1
struct alignas(16) RmRow { uint32_t act, heads, tipP, tipN; };   // census of a (polygon, row); after the row phase w0.. = the coverage mask
struct alignas(16) RmItem { int32_t rowbase; int32_t pb_nv; int32_t pymax; uint32_t rgba; };   // row record of row 0 | first vertex slot, live vertices << 20 | min(ymax, H) | colour

struct alignas(16) RmU4 { uint32_t x, y, z, w; };

struct RmPlan { uint32_t o_edge, o_ivert, o_rows, o_rowitem, o_info, o_item_y, o_rowoff, o_seg, o_lut, o_xx, o_misc, o_spare, o_owner, o_skey, xx_stride, total; };   // xx_stride: floats of scratch per wavefront

struct RmArgs {
  const uint8_t* draw;        // the frames' draw records (moog_draw_record.h), [n_envs][lay.stride]
  RmDrawLayout lay;
  uint8_t* image;
  int32_t n_envs;
  int32_t S;                  // items (polygons a frame may hold): slots * ncopy
  int32_t big;                // some slot may hold a polygon of more than RM_MAX_NV vertices (rm_p4_big)
  int32_t compact;            // the edge table holds 4-byte records (RmEdgesCompact; the plan was made for them)
  int32_t cap_rows;           // row records per pass (>= H)
  int32_t W, H;               // the canvas in memory (width a multiple of 16, <= 128)
  int32_t flip;               // rows are written bottom-up (np.flipud, pil_renderer.py:118)
  int32_t iwords;             // 32-bit words of a segment's item bit mask
  uint32_t bg;                // r | g << 8 | b << 16
  int32_t debug_stop;
  int32_t threads;            // threads per frame (64 * waves)
  // static prefix (moog_raster.h): a frame whose record says RM_DRAW_PREFIX_OK starts from the cached picture `sbg` (its prefix items are empty)
  int32_t n_static;
  const uint8_t* sbg;
  int32_t* rows_seen;         // two host-mapped words (or null): the most rows a frame wanted when that was more than cap_rows, and how many frames did (the engine may grow the records)
  RmPlan plan;
};

static inline uint32_t rm_align(uint32_t x) { return (x + 15u) & ~15u; }

// S: items; TOTV: points a frame may hold (the program's vertex slots x copies)
static inline void rm_plan(int S, int TOTV, int W, int H, int cap_rows, int iwords, int waves, int big, RmPlan* p, int compact = 0) {
  uint32_t o = 0;
  p->o_edge = o; o = rm_align(o + (uint32_t)TOTV * (compact ? 4u : (uint32_t)sizeof(RmEdge)));   // (compact: RmEdgesCompact)
  p->o_ivert = o; o = rm_align(o + (uint32_t)TOTV * 4u);
  p->o_rows = o; o = rm_align(o + (uint32_t)cap_rows * sizeof(RmRow));
  p->o_rowitem = o; o = rm_align(o + (uint32_t)cap_rows * 2u);
  p->o_info = o; o = rm_align(o + (uint32_t)S * sizeof(RmItem));
  p->o_item_y = o; o = rm_align(o + (uint32_t)S * 8u);
  p->o_rowoff = o; o = rm_align(o + (uint32_t)(S + 1) * 4u);
  p->o_seg = o; o = rm_align(o + (uint32_t)H * (uint32_t)(W / 16) * (uint32_t)iwords * 4u);
  p->o_lut = o; o = rm_align(o + 16u * 16u);
  p->xx_stride = big ? (uint32_t)RM_XX : 2u * RM_MAX_NV;   // per wave: the generic routine's crossing list (two entries per edge); for long polygons also rm_big_row's edge list and census words
  p->o_xx = o; o = rm_align(o + (uint32_t)waves * p->xx_stride * 4u + (big ? (uint32_t)waves * 64u * RM_MAX_NV : 0u));   // + a list of edge numbers per thread (rm_p4_big)
  p->o_misc = o; o = rm_align(o + 64u);
  p->o_owner = o; o = rm_align(o + (uint32_t)TOTV);   // the slot of every compact vertex number
  p->o_skey = o; o = rm_align(o + (cap_rows > RM_SORT_ROUNDS_T ? (uint32_t)cap_rows * 2u : 0u));   // the row sort of a long pass: bucket << 12 | place in the bucket
  {   // a word per thread (p3) / the sorted row list (p4)
    const uint32_t b1 = (uint32_t)waves * 64u * 4u, b2 = (uint32_t)cap_rows * 2u;
    p->o_spare = o; o = rm_align(o + (b1 > b2 ? b1 : b2));
  }   // a word per thread: where atomics with nothing to add go
  p->total = o;
}

struct RmCtx {
  RmEdge* edges; uint32_t* ivert; RmRow* rows; uint16_t* rowitem;   // rowitem: the row's item | 256 when a shallow edge has a corner on the row
  RmItem* info; int32_t* item_y; int32_t* rowoff;
  uint32_t* seg; uint32_t* lut; float* xx; uint32_t* spare; uint16_t* sorted; uint16_t* skey; uint8_t* owner; int32_t* misc;   // sorted (p4: the rows in order of their kind) shares the spare words' memory (p3)   // misc: [5] static prefix differs
};

RM_FN RmCtx rm_ctx(const RmPlan& pl, unsigned char* lds) {
  RmCtx c;
  c.edges = reinterpret_cast<RmEdge*>(lds + pl.o_edge);
  c.ivert = reinterpret_cast<uint32_t*>(lds + pl.o_ivert);
  c.rows = reinterpret_cast<RmRow*>(lds + pl.o_rows);
  c.rowitem = reinterpret_cast<uint16_t*>(lds + pl.o_rowitem);
  c.info = reinterpret_cast<RmItem*>(lds + pl.o_info);
  c.item_y = reinterpret_cast<int32_t*>(lds + pl.o_item_y);
  c.rowoff = reinterpret_cast<int32_t*>(lds + pl.o_rowoff);
  c.seg = reinterpret_cast<uint32_t*>(lds + pl.o_seg);
  c.lut = reinterpret_cast<uint32_t*>(lds + pl.o_lut);
  c.xx = reinterpret_cast<float*>(lds + pl.o_xx);
  c.misc = reinterpret_cast<int32_t*>(lds + pl.o_misc);
  c.spare = reinterpret_cast<uint32_t*>(lds + pl.o_spare);
  c.sorted = reinterpret_cast<uint16_t*>(lds + pl.o_spare);
  c.owner = reinterpret_cast<uint8_t*>(lds + pl.o_owner);
  c.skey = reinterpret_cast<uint16_t*>(lds + pl.o_skey);
  return c;
}

// ---- small helpers ---------------------------------------------------------------------------------------------------
RM_FN int rm_ffs(uint32_t m) { return __builtin_ctz(m); }
RM_FN void rm_or(uint32_t* p, uint32_t v) {
#if RM_DEV
  atomicOr(p, v);
#else
  *p |= v;
#endif
}
// float -> int the way v_cvt_i32_f32 does it (saturating): the host model and the kernel then agree on far-away crossings
RM_FN int rm_f2i(float f) {
  if (!(f > -2147483648.0f)) return (int)0x80000000;
  if (!(f < 2147483648.0f)) return 0x7fffffff;
  return (int)f;
}
// Draw.c ROUND_UP / ROUND_DOWN
RM_FN int rm_round_up(float f) { return rm_f2i(copysignf(floorf(fabsf(f) + 0.5f), f)); }
RM_FN int rm_round_down(float f) { return rm_f2i(copysignf(ceilf(fabsf(f) - 0.5f), f)); }
// ---- coverage masks -----------------------------------------------------------------------------------------------
template <int WORDS> struct RmMask { uint64_t w[WORDS]; };
RM_FN uint64_t rm_low(int t) { return t <= 0 ? 0ull : (t >= 64 ? ~0ull : ((1ull << t) - 1ull)); }   // pixels 0 .. t-1
RM_FN uint64_t rm_high(int t) { return t >= 64 ? 0ull : (~0ull << (t & 63)); }                       // pixels t .. 63 (t >= 0)
RM_FN uint64_t rm_high1(int t) { return t <= 64 ? ~0ull : rm_high(t - 64); }                           // the same of the second word
template <int WORDS> RM_FN void rm_clear(RmMask<WORDS>& m) { for (int i = 0; i < WORDS; ++i) m.w[i] = 0ull; }
template <int WORDS> RM_FN bool rm_bit(const RmMask<WORDS>& m, int p) { return (m.w[WORDS > 1 ? (p >> 6) : 0] >> (p & 63)) & 1ull; }
// pixels [a, b] clipped to [0, 64 * WORDS); nothing when a > b
template <int WORDS> RM_FN void rm_or_range(RmMask<WORDS>& m, int a, int b) {
  if (a < 0) a = 0;
  if (b > 64 * WORDS - 1) b = 64 * WORDS - 1;
  if (a > b) return;
  m.w[0] |= rm_low(b + 1) & ~rm_low(a);
  if (WORDS > 1) m.w[1] |= rm_low(b + 1 - 64) & ~rm_low(a - 64);
}

// ---- edges ------------------------------------------------------------------------------------------------------------
RM_FN int rm_y0(const RmEdge& e) { return (int)(int16_t)(e.w2 & 0xffffu); }
RM_FN int rm_y1(const RmEdge& e) { return (int)(int32_t)e.w2 >> 16; }
RM_FN float rm_xat(const RmEdge& e, int y) { return (float)(y - rm_y0(e)) * rm_u2f(e.w1) + rm_u2f(e.w0); }   // Draw.c: (ymin - y0) * dx + x0

// The polygons' edge records in LDS, two ways:
//   RmEdgesFull     16 bytes per vertex: the record as rm_build_edge / rm_p3 leave it, one 128-bit read per look.
//   RmEdgesCompact   4 bytes per vertex beside the integer points the frame holds anyway: dx of a table edge (x0, y0, y1 are the
//                    two points), xmin | xmax << 16 of a horizontal head, RM_NO_EDGE for a horizontal edge that is no head.  Three
//                    32-bit reads and a conversion per look -- a tenth more instructions in the rows phase -- for a quarter of the
//                    memory: the engine picks it for programs whose edge records are what keeps frames off a CU (falling_balls_64:
//                    1816 vertex slots, 29 of 47 KB; three resident frames per CU become six).
#define RM_NO_EDGE 0x7fff8000u   // (no head has xmin = -32768, xmax = 32767: points are clamped to +-32000)
struct RmEdgesFull {
  const RmEdge* pe;
  RM_MEMBER RmEdge operator[](int k) const { return pe[k]; }
};
struct RmEdgesCompact {
  const uint32_t* pv; const uint32_t* dx; int nv;
  RM_MEMBER RmEdge operator[](int k) const {
    const uint32_t q0 = pv[k], q1 = pv[(k + 1 >= nv) ? 0 : k + 1], w = dx[k];
    const bool table = (q0 >> 16) != (q1 >> 16), head = !table && w != RM_NO_EDGE;
    RmEdge E;
    E.w0 = table ? rm_f2u((float)(int)(int16_t)(q0 & 0xffffu)) : (head ? w : 0u);
    E.w1 = table ? w : 0u;
    E.w2 = table ? ((q0 >> 16) | (q1 & 0xffff0000u)) : (head ? ((q0 >> 16) | (q0 & 0xffff0000u)) : 0u);
    E.w3 = head ? 1u : 0u;
    return E;
  }
};
template <bool COMPACT> struct RmEdgeTab;
template <> struct RmEdgeTab<false> {
  typedef RmEdgesFull View;
  template <class C> static RM_MEMBER View view(const C& c, int first, int nv) { (void)nv; View v; v.pe = c.edges + first; return v; }
  template <class C> static RM_MEMBER void put(const C& c, int idx, const RmEdge& E, bool table, bool head) { (void)table; (void)head; c.edges[idx] = E; }
};
template <> struct RmEdgeTab<true> {
  typedef RmEdgesCompact View;
  template <class C> static RM_MEMBER View view(const C& c, int first, int nv) { View v; v.pv = c.ivert + first; v.dx = reinterpret_cast<const uint32_t*>(c.edges) + first; v.nv = nv; return v; }
  template <class C> static RM_MEMBER void put(const C& c, int idx, const RmEdge& E, bool table, bool head) {
    reinterpret_cast<uint32_t*>(c.edges)[idx] = table ? E.w1 : (head ? E.w0 : RM_NO_EDGE);
  }
};

// ImagingDrawPolygon's edge list, the edge that leaves vertex k of a ring of nv packed points (x | y << 16, shorts):
// 0 = no edge (the closing edge of a ring whose last point is its first, a horizontal edge merged into the run before
// it, or -- see below -- a repeated point), 1 = table edge, 2 = horizontal head [xmin, xmax].
RM_FN int rm_build_edge(const uint32_t* pv, int k, int nv, RmEdge* out) {
  const int k2 = (k + 1 == nv) ? 0 : k + 1;
  const uint32_t q0 = pv[k], q1 = pv[k2];
  const int x0 = (int16_t)(q0 & 0xffffu), y0 = (int16_t)(q0 >> 16), x1 = (int16_t)(q1 & 0xffffu), y1 = (int16_t)(q1 >> 16);
  const bool closing = (k == nv - 1);
  if (y0 != y1) {
    out->w0 = rm_f2u((float)x0);
    out->w1 = rm_f2u(((float)(x1 - x0)) / (float)(y1 - y0));
    out->w2 = (q0 >> 16) | (q1 & 0xffff0000u);
    out->w3 = 0u;
    return 1;
  }
  if (closing && x0 == x1) return 0;   // last == first: no closing edge
  if (k >= 1 && !closing) {
    const uint32_t qp = pv[k - 1];
    const int xp = (int16_t)(qp & 0xffffu), yp = (int16_t)(qp >> 16);
    // "horizontal line immediately following another horizontal line", same direction: the run before it grows
    if (yp == y0 && ((x1 > x0 && x0 > xp) || (x1 < x0 && x0 < xp))) return 0;
    // Three equal points in a row (tiny circles): this zero-length head repeats the one before it, which is visited
    // immediately before it with the same outcome either way -- dropped.
    if (xp == x0 && yp == y0 && x1 == x0) return 0;
  }
  int hx = x1;   // extend over the following merged edges (never the closing edge)
  {
    int q = k + 1;
    int px = x0, cx = x1;
    while (q <= nv - 2) {
      const uint32_t qn = pv[q + 1];
      const int nx = (int16_t)(qn & 0xffffu), ny = (int16_t)(qn >> 16);
      const bool ab = (ny == y0) && ((nx > cx && cx > px) || (nx < cx && cx < px));
      if (!ab) break;
      hx = nx; px = cx; cx = nx; ++q;
    }
  }
  const int xmin = x0 < hx ? x0 : hx, xmax = x0 < hx ? hx : x0;
  out->w0 = (uint32_t)(uint16_t)xmin | ((uint32_t)(uint16_t)xmax << 16);
  out->w1 = 0u;
  out->w2 = (q0 >> 16) | (q0 & 0xffff0000u);
  out->w3 = 1u;
  return 2;
}

// ---- one row of one polygon ---------------------------------------------------------------------------------------------
// polygon_generic verbatim for one row (any number of crossings, overwritten partner entries, odd counts): the rare rows.
// xx: RM_XX floats of scratch.
template <int WORDS, class PE>
RM_SLOW RmMask<WORDS> rm_row_generic(const PE pe, int nv, uint32_t heads, int y, int pymax, float* xx) {   // nv > RM_MAX_NV: `heads` is ignored, the heads of the row are found by looking   // (the mask comes back in registers: a reference would put the caller's copy in scratch memory)
  RmMask<WORDS> m;
  rm_clear(m);
  int j = 0;
  for (int i = 0; i < nv; ++i) {
    const RmEdge E = pe[i];
    const int y0 = rm_y0(E), y1 = rm_y1(E);
    if (y0 == y1) continue;   // (records of vertices without an edge are zero: y0 == y1)
    const int emin = y0 < y1 ? y0 : y1, emax = y0 < y1 ? y1 : y0;
    if (y < emin || y > emax) continue;
    const float dx = rm_u2f(E.w1);
    const float x = rm_xat(E, y);
    if (j < RM_XX) xx[j] = x;
    ++j;
    if (y == emax && y < pymax) {
      if (j < RM_XX) xx[j] = x;
      ++j;
    } else if (dx != 0.0f) {
      for (int k = 0; k < i; ++k) {
        const RmEdge K = pe[k];
        const int ky0 = rm_y0(K), ky1 = rm_y1(K);
        if (ky0 == ky1) continue;
        const int kmin = ky0 < ky1 ? ky0 : ky1, kmax = ky0 < ky1 ? ky1 : ky0;
        if (y < kmin || y > kmax) continue;
        const float kdx = rm_u2f(K.w1);
        if ((dx > 0 && kdx <= 0) || (dx < 0 && kdx >= 0)) continue;
        const bool top = (y == emin && y == kmin), bot = (y == emax && y == kmax);
        if (!(top || bot)) continue;
        if (x != rm_xat(K, y)) continue;
        const int off = top ? 1 : -1;
        const float adj = rm_xat(E, y + off), adjo = rm_xat(K, y + off);
        float vv = 0; bool have = false;
        if (adj > x && adjo > x) {
          const float v = (float)(rm_round_up(fminf(adj, adjo)) - 1);
          if (v > x) { vv = v; have = true; }
        } else if (adj < x && adjo < x) {
          const float v = (float)(rm_round_up(fmaxf(adj, adjo)) + 1);
          if (v < x) { vv = v; have = true; }
        }
        if (have) {   // the partner's (first) entry on this row is overwritten
          int kpos = 0;
          for (int q = 0; q < k; ++q) {
            const RmEdge Q = pe[q];
            const int qy0 = rm_y0(Q), qy1 = rm_y1(Q);
            if (qy0 == qy1) continue;
            const int qmin = qy0 < qy1 ? qy0 : qy1, qmax = qy0 < qy1 ? qy1 : qy0;
            if (y < qmin || y > qmax) continue;
            kpos += (y == qmax && y < pymax) ? 2 : 1;
          }
          if (kpos < RM_XX) xx[kpos] = vv;
        }
        break;
      }
    }
  }
  if (j > RM_XX) j = RM_XX;
  for (int q = 1; q < j; ++q) {   // qsort(x_cmp)
    const float key = xx[q];
    int r = q - 1;
    while (r >= 0 && xx[r] > key) { xx[r + 1] = xx[r]; --r; }
    xx[r + 1] = key;
  }
  int x_pos = (j == 0) ? -1 : 0;
  for (int i = 1; i <= j + 1; i += 2) {
    const bool last = (i >= j);   // the call behind the loop
    int x_end = 0;
    if (!last) {
      x_end = rm_round_down(xx[i]);
      if (x_end < x_pos) continue;
    }
    {   // draw_horizontal_lines: every head of the row, in edge order
      uint32_t hb = heads;
      const bool scan = nv > RM_MAX_NV;
      int ks = 0;
      while (scan ? ks < nv : hb != 0u) {
        int k;
        if (scan) { k = ks++; if (pe[k].w3 != 1u || rm_y0(pe[k]) != y) continue; }
        else { k = rm_ffs(hb); hb &= hb - 1u; }
        const RmEdge h = pe[k];
        int xmin = (int16_t)(h.w0 & 0xffffu);
        const int xmax = (int16_t)(h.w0 >> 16);
        if (x_pos != -1 && x_pos < xmin) continue;
        if (x_pos > xmin) {
          xmin = x_pos;
          if (xmax < xmin) continue;
        }
        rm_or_range(m, xmin, xmax);
        x_pos = xmax + 1;
      }
    }
    if (last) break;
    if (x_end < x_pos) continue;
    int x_start = rm_round_up(xx[i - 1]);
    if (x_pos > x_start) {
      x_start = x_pos;
      if (x_end < x_start) continue;
    }
    rm_or_range(m, x_start, x_end);
    x_pos = x_end + 1;
  }
  return m;
}

// The corner fix-up of a row, from the census word of one lean (bits = edges with a corner on this row): every edge but
// the first looks at the earlier ones in order; the first whose crossing is the same float decides, and what it decides
// is a new value for the PARTNER's entry (polygon_generic: xx[k] = ...).  At most two partners are kept; a third: generic.
struct RmFix { int k0, k1; float v0, v1; bool generic; };
RM_FN void rm_fix_put(RmFix& f, int k, float v) {
  if (f.k0 == k) f.v0 = v;
  else if (f.k1 == k) f.v1 = v;
  else if (f.k0 < 0) { f.k0 = k; f.v0 = v; }
  else if (f.k1 < 0) { f.k1 = k; f.v1 = v; }
  else f.generic = true;
}
template <class PE>
RM_FN void rm_fix_class(const PE pe, uint32_t m, int y, RmFix& f) {
  uint32_t rest = m & (m - 1u);
  while (rest) {
    const int i = rm_ffs(rest);
    rest &= rest - 1u;
    const RmEdge E = pe[i];
    const int y0 = rm_y0(E), y1 = rm_y1(E);
    const bool top = (y == (y0 < y1 ? y0 : y1));
    const float x = rm_xat(E, y);
    uint32_t cand = m & ((1u << i) - 1u);
    while (cand) {
      const int k = rm_ffs(cand);
      cand &= cand - 1u;
      const RmEdge K = pe[k];
      const int ky0 = rm_y0(K), ky1 = rm_y1(K);
      if (y != (top ? (ky0 < ky1 ? ky0 : ky1) : (ky0 < ky1 ? ky1 : ky0))) continue;
      if (x != rm_xat(K, y)) continue;
      const int off = top ? 1 : -1;
      const float adj = rm_xat(E, y + off), adjo = rm_xat(K, y + off);
      if (adj > x && adjo > x) {
        const float v = (float)(rm_round_up(fminf(adj, adjo)) - 1);
        if (v > x) rm_fix_put(f, k, v);
      } else if (adj < x && adjo < x) {
        const float v = (float)(rm_round_up(fmaxf(adj, adjo)) + 1);
        if (v < x) rm_fix_put(f, k, v);
      }
      break;
    }
  }
}

// draw_horizontal_lines with the pen at `pen`: heads that start behind the pen (or any head when the row has no
// crossing: pen == -1) are painted from the pen on and move it; the others stay pending.  x_pos only grows, so a head
// that was looked at with the pen at or behind its start is finished whether or not anything was painted.
template <int WORDS, class PE>
RM_FN void rm_heads(const PE pe, uint32_t& hb, int& pen, RmMask<WORDS>& m) {
  uint32_t bits = hb;
  while (bits) {
    const int k = rm_ffs(bits);
    bits &= bits - 1u;
    const RmEdge h = pe[k];
    const int xmin = (int16_t)(h.w0 & 0xffffu), xmax = (int16_t)(h.w0 >> 16);
    if (pen != -1 && pen < xmin) continue;
    hb &= ~(1u << k);
    const int hs = pen > xmin ? pen : xmin;
    if (xmax < hs) continue;
    rm_or_range(m, hs, xmax);
    pen = xmax + 1;
  }
}

// The crossings of a row, accumulated (see the head of the file).  HEADS: also where the spans end (seen / seen2).
template <int WORDS, bool HEADS, class PE>
RM_FN void rm_crossings(const PE pe, uint32_t m, int y, int pymax, int W, const RmFix& fix, RmMask<WORDS>& par, RmMask<WORDS>& pix,
                        RmMask<WORDS>& seen, RmMask<WORDS>& seen2, bool& odd) {
  const float wlim = (float)(W - 1);
  const bool anyheads = HEADS;
  // (wave-uniform loop without a branch inside: a lane whose row has no edge left goes through the motions on edge 0 with
  //  every effect masked off)
  const bool ylt = y < pymax;
  // (do-while: the plain while loop makes the compiler copy every loop-carried mask register once per iteration)
  if (RM_ANY(m != 0u)) do {
    const bool valid = m != 0u;
    const int k = valid ? rm_ffs(m) : 0;
    m &= m - 1u;
    const RmEdge E = pe[k];
    const int y0 = rm_y0(E), y1 = rm_y1(E);
    float x = rm_xat(E, y);
    x = (k == fix.k0) ? fix.v0 : x;
    x = (k == fix.k1) ? fix.v1 : x;
    const int emax = y0 < y1 ? y1 : y0;
    const bool dup = (y == emax) && ylt;   // "needed to draw consistent polygons": the edge's last row counts twice
    const float ax = fabsf(x);
    const float fu = floorf(ax + 0.5f), fd = ceilf(ax - 0.5f);
    const float fs = copysignf(fd, x);                 // ROUND_DOWN(x)
    const float fc = fminf(fmaxf(fs, -1.0f), wlim);    // clipped to [-1, W - 1]
    const int rc = (int)fc;
    const bool inr = valid && (fs == fc) && (fc >= 0.0f);   // ROUND_DOWN(x) is a pixel of the canvas
    // pixels p with ROUND_UP(x) <= p <= ROUND_DOWN(x): round(x) unless x is a half-integer -- a positive one has none, a
    // negative one two (Draw.c rounds halves away from zero going up, towards zero going down), of which only
    // ROUND_DOWN(-0.5) = 0 can be on the canvas
    const bool onpix = inr && ((fu == fd) || (x < 0.0f));
    const bool tog = valid && !dup;
    const int t = tog ? rc + 1 : 64 * WORDS;   // (nothing above 64 * WORDS - 1)
    par.w[0] ^= rm_high(t);
    if (WORDS > 1) par.w[1] ^= rm_high1(t);
    odd = odd != tog;
    const uint64_t b = (uint64_t)(inr ? 1u : 0u) << (rc & 63);
    const bool hiw = WORDS > 1 && rc >= 64;   // (which word of the mask the pixel is in)
    const uint64_t b0 = hiw ? 0ull : b, b1 = hiw ? b : 0ull;
    pix.w[0] |= onpix ? b0 : 0ull;
    if (WORDS > 1) pix.w[1] |= onpix ? b1 : 0ull;
    if (anyheads) {
      // Where Pillow's pen can come to rest: the last pixels of the spans = ROUND_DOWN of the crossings of odd rank.
      // seen: pixels some crossing rounds down onto; seen2: pixels two or more do (one of two neighbours in the sorted
      // list has odd rank; this includes the pairs of equal half-integers, whose span [n + 1, n] paints nothing and
      // still moves the pen).  A lone crossing has odd rank iff the parity mask covers its pixel (below).
      seen2.w[0] |= (seen.w[0] & b0) | (dup ? b0 : 0ull);
      seen.w[0] |= b0;
      if (WORDS > 1) { seen2.w[1] |= (seen.w[1] & b1) | (dup ? b1 : 0ull); seen.w[1] |= b1; }
    }
  } while (RM_ANY(m != 0u));
}

// The coverage mask of row y of a polygon from its census record.  Returns false when the row needs rm_row_generic.
template <int WORDS, class PE>
RM_FN bool rm_row_fast(const PE pe, const RmRow rec, int y, int pymax, int W, bool shallow, RmMask<WORDS>& out) {
  RmFix fix; fix.k0 = -1; fix.k1 = -1; fix.v0 = 0.0f; fix.v1 = 0.0f; fix.generic = false;
  {
    // (shallow: some edge with a corner on this row runs >= 1.49 pixels per row -- without one no fix-up moves anything)
    const bool tp = shallow && (rec.tipP & (rec.tipP - 1u)) != 0u, tn = shallow && (rec.tipN & (rec.tipN - 1u)) != 0u;
    if (RM_ANY(tp || tn)) {
      if (tp) rm_fix_class(pe, rec.tipP, y, fix);
      if (tn) rm_fix_class(pe, rec.tipN, y, fix);
    }
  }
  RmMask<WORDS> par, pix;
  rm_clear(par); rm_clear(pix);
  bool odd = false;
  RmMask<WORDS> seen, seen2;
  rm_clear(seen); rm_clear(seen2);
  const bool anyheads = RM_ANY(rec.heads != 0u);
  // (two copies of the loop: the bookkeeping for the heads is a fifth of its instructions, and after the row sort only
  //  the first wavefuls of a pass have rows with heads)
  if (anyheads) rm_crossings<WORDS, true>(pe, rec.act, y, pymax, W, fix, par, pix, seen, seen2, odd);
  else rm_crossings<WORDS, false>(pe, rec.act, y, pymax, W, fix, par, pix, seen, seen2, odd);
  const bool any = rec.act != 0u;
  for (int i = 0; i < WORDS; ++i) {
    const int wbits = W - 64 * i;
    out.w[i] = (par.w[i] | pix.w[i]) & rm_low(wbits);
  }
  if (anyheads) {
    if (rec.heads != 0u) {
      // Pillow's span loop, as far as the heads can tell: the spans' last pixels in order; a span that ends behind the
      // pen is passed over without a look at the heads
      RmMask<WORDS> res = out;
      uint32_t hb = rec.heads;
      int pen = any ? 0 : -1;
      for (int i = 0; i < WORDS; ++i) {
        uint64_t ends = seen2.w[i] | (seen.w[i] & par.w[i]);
        while (ends) {
          const int e = 64 * i + __builtin_ctzll(ends);
          ends &= ends - 1ull;
          if (e < pen) continue;
          rm_heads<WORDS, PE>(pe, hb, pen, res);
          if (e < pen) continue;
          pen = e + 1;
        }
      }
      rm_heads<WORDS, PE>(pe, hb, pen, res);
      for (int i = 0; i < WORDS; ++i) out.w[i] = res.w[i] & rm_low(W - 64 * i);
    }
  }
  return !(odd || fix.generic);
}

// ---- the frame, phase by phase ------------------------------------------------------------------------------------------
// One workgroup of `T` threads (T / 64 wavefronts) renders one frame.  Barriers stand between the phases:
//   load  clear the tables; the frame's draw record (moog_draw_record.h) -> items, integer points, row offsets
//   p2  (every wave for itself) item rows -> row records                 | p3  edges + census
//   p4  rows -> coverage masks, which 16-pixel segments an item touches  | p5  compose + store
// A frame with more rows than row records takes several passes over p2 .. p5, whole items at a time.
// (Rounds 1-5 read the state record here -- liveness, colour map, float64 vertices -> canvas points, per-item row ranges, the nine
//  copies of a torus: a fifth of the kernel and four fifths of its reads.  That work is the emitter's now, done where the
//  record is already on chip.)
RM_FN void rm_load(const RmArgs& a, const RmCtx& c, int env, int tid, int T) {
  const uint8_t* rec = a.draw + (size_t)env * a.lay.stride;
  const RmDrawHdr hdr = *reinterpret_cast<const RmDrawHdr*>(rec);
  const RmDrawItem* items = reinterpret_cast<const RmDrawItem*>(rec + a.lay.o_items);
  const uint32_t* pts = reinterpret_cast<const uint32_t*>(rec + a.lay.o_pts);
  const uint32_t* own = reinterpret_cast<const uint32_t*>(rec + a.lay.o_owner);
  // the loads go out first (HBM latency under the clearing)
  const int n_pts = hdr.n_pts;
  RmDrawItem it = {0, 0u, RM_Y01_EMPTY, 0u};
  if (tid < a.S) it = items[tid];
  uint32_t p0 = 0u, o0 = 0u;
  if (tid < n_pts) p0 = pts[tid];
  if (4 * tid < n_pts) o0 = own[tid];
  // A frame whose rows all fit in the row records (total_rows <= cap_rows: the usual case) is ONE pass over every item, and what
  // rm_p2_assign would do for that pass -- the item's first row record, its clamped ymax, the owner of its rows -- follows from
  // the item's own record: done below, by the thread that loaded it (the kernel then skips rm_p2_assign for the first pass).
  const bool single = RM_LOAD_ASSIGN && hdr.total_rows <= a.cap_rows;
  for (int i = tid; i < a.cap_rows; i += T) { RmRow z = {0u, 0u, 0u, 0u}; c.rows[i] = z; }
  // (the rows' flag bytes: a single-pass frame's are written with the owner bytes, 16 bits at a time, by the items' threads --
  //  clearing them here, from other threads, could land on top of that)
  if (!single) for (int i = tid; i < (a.cap_rows + 1) / 2; i += T) reinterpret_cast<uint32_t*>(c.rowitem)[i] = 0u;
  const int nseg = a.W >> 4;
  for (int i = tid; i < a.H * nseg * a.iwords; i += T) c.seg[i] = 0u;
  if (tid < 16) {   // four coverage bits -> byte masks of the 12 bytes of four RGB pixels
    const uint32_t b0 = tid & 1, b1 = (tid >> 1) & 1, b2 = (tid >> 2) & 1, b3 = (tid >> 3) & 1;
    c.lut[4 * tid + 0] = (b0 ? 0x00ffffffu : 0u) | (b1 ? 0xff000000u : 0u);
    c.lut[4 * tid + 1] = (b1 ? 0x0000ffffu : 0u) | (b2 ? 0xffff0000u : 0u);
    c.lut[4 * tid + 2] = (b2 ? 0x000000ffu : 0u) | (b3 ? 0xffffff00u : 0u);
    c.lut[4 * tid + 3] = 0u;
  }
  if (tid < 13) c.misc[tid] = tid == 0 ? n_pts : ((tid == 5 && !(hdr.flags & RM_DRAW_PREFIX_OK)) ? 1 : ((tid == 6 && single) ? 1 : 0));   // ([0] live vertices, [5] static prefix differs, [6] one pass: its row records are assigned, [8..11] rows per bucket of the sort, [12] rows of long polygons)
  if (tid == 0) c.rowoff[a.S] = hdr.total_rows;
  for (int g = tid; g < a.S; g += T) {
    if (g != tid) it = items[g];
    const int ymin = (int)(int16_t)((uint32_t)it.y01 & 0xffffu), ymax = (int)it.y01 >> 16;
    RmItem o;
    o.rowbase = 0; o.pb_nv = (int32_t)it.pb_nv; o.pymax = 0; o.rgba = it.rgba;
    if (single) {
      const int ys = ymin < 0 ? 0 : ymin;
      const int cnt = rm_rows_on_canvas(ymin, ymax, a.H);
      o.rowbase = it.rowoff - ys;
      o.pymax = ymax > a.H ? a.H : ymax;   // polygon_generic clamps ymax to ysize
      for (int j = 0; j < cnt; ++j) c.rowitem[it.rowoff + j] = (uint16_t)g;   // (owner | flags = 0)
    }
    c.info[g] = o;
    c.item_y[2 * g] = ymin; c.item_y[2 * g + 1] = ymax;
    c.rowoff[g] = it.rowoff;
  }
  for (int i = tid; i < n_pts; i += T) c.ivert[i] = (i == tid) ? p0 : pts[i];
  for (int i = tid; 4 * i < n_pts; i += T) reinterpret_cast<uint32_t*>(c.owner)[i] = (i == tid) ? o0 : own[i];
}

// Items below s_lo are in the cached picture of the static prefix (valid after the load's barrier; the emitter left them empty)
RM_FN int rm_s_lo(const RmArgs& a, const RmCtx& c) { return (a.n_static > 0 && c.misc[5] == 0) ? a.n_static : 0; }

// Rows an item occupies on the canvas: [ystart, ystart + cnt)
RM_FN int rm_item_rows(const RmArgs& a, const RmCtx& c, int g, int s_lo, int* ystart) {
  int y0 = c.item_y[2 * g], y1 = c.item_y[2 * g + 1];
  if (y0 < 0) y0 = 0;
  if (y1 > a.H - 1) y1 = a.H - 1;   // rows >= H draw nothing (hline clips)
  *ystart = y0;
  return (y1 >= y0 && g >= s_lo) ? (y1 - y0 + 1) : 0;
}

// The pass that starts with item `base`: as many whole items as fit in the row records.  Returns the item behind it.
RM_FN int rm_pass_end(const RmArgs& a, const RmCtx& c, int base) {
  const int r0 = c.rowoff[base];
  int lo = base + 1, hi = a.S;   // largest end with rowoff[end] - r0 <= cap_rows
  if (c.rowoff[a.S] - r0 <= a.cap_rows) return a.S;
  while (lo < hi) {
    const int mid = (lo + hi + 1) >> 1;
    if (c.rowoff[mid] - r0 <= a.cap_rows) lo = mid; else hi = mid - 1;
  }
  return lo;
}

// p2 (every wave for itself; device: `lane` = the thread's index in its wave; host model: lane = -1 does every lane's work): the
// pass's items get their row records (the row offsets come with the draw record).
RM_FN void rm_p2_assign(const RmArgs& a, const RmCtx& c, int base, int end, int s_lo, int lane) {
  const int r0 = c.rowoff[base];
#if RM_DEV
  for (int g = base + lane; g < end; g += 64) {
#else
  (void)lane;
  for (int g = base; g < end; ++g) {
#endif
    int ys;
    const int cnt = rm_item_rows(a, c, g, s_lo, &ys);
    const int first = c.rowoff[g] - r0;
    c.info[g].rowbase = first - ys;
    const int ymax = c.item_y[2 * g + 1];
    c.info[g].pymax = ymax > a.H ? a.H : ymax;   // polygon_generic clamps ymax to ysize
    // (the low byte only: the high byte is the "shallow corner" flag p3 sets, and the other wavefront may already be in p3 when
    //  this one gets here -- a 16-bit store would wipe its flags; the flags are cleared with the row records)
    for (int j = 0; j < cnt; ++j) reinterpret_cast<uint8_t*>(c.rowitem)[2 * (first + j)] = (uint8_t)g;
  }
}

// p3: the edge that leaves every vertex, and the census of the rows it touches.  Written for the wavefront: table edges
// (nine in ten) are built by every lane without a branch, the census is atomics whose operand is zero where a lane has
// nothing to say (on a spare word), so that the only divergent code is the horizontal edges' run merging.
template <int WORDS, bool COMPACT>
RM_FN void rm_p3(const RmArgs& a, const RmCtx& c, int base, int end, int s_lo, int tid, int T) {
  const int lo = s_lo > base ? s_lo : base;
  const int nlive = c.misc[0];
  uint32_t* const spare = c.spare + tid;   // (this thread's own: atomics on one address would queue up behind each other)
  for (int idx = tid; idx < nlive; idx += T) {   // compact vertex numbers: every one is a live vertex
    const int s = c.owner[idx];
    const RmItem it = c.info[s];
    const int nv = it.pb_nv >> 20, k = idx - (it.pb_nv & 0xfffff);
    const bool live = true;
    const bool mine = s >= lo && s < end;
    // A long polygon's row record is ONE 128-bit word: which of its edges cross the row or are heads on it (what kind, the row's
    // thread reads off the edge records: rm_p4_big).  A short polygon's is four 32-bit words, one per kind.
    const bool lng = mine && nv > RM_MAX_NV;
    const bool census = mine && !lng;
    const int wsel = lng ? (k >> 5) : 0;
    const uint32_t* pv = c.ivert + (idx - k);
    const uint32_t q0 = pv[k], q1 = pv[(k + 1 >= nv) ? 0 : k + 1];
    const int x0 = (int16_t)(q0 & 0xffffu), y0 = (int16_t)(q0 >> 16), x1 = (int16_t)(q1 & 0xffffu), y1 = (int16_t)(q1 >> 16);
    const bool table = mine && y0 != y1;
    RmEdge E;
    E.w0 = table ? rm_f2u((float)x0) : 0u;
    const float dx = ((float)(x1 - x0)) / (float)(table ? y1 - y0 : 1);
    E.w1 = table ? rm_f2u(dx) : 0u;
    E.w2 = table ? ((q0 >> 16) | (q1 & 0xffff0000u)) : 0u;
    E.w3 = 0u;
    const bool horiz = mine && y0 == y1;
    bool head = false;
    if (RM_ANY(horiz)) {
      if (horiz) head = rm_build_edge(pv, k, nv, &E) == 2;
    }
    if (live) RmEdgeTab<COMPACT>::put(c, idx, E, table, head);   // (a vertex without an edge keeps a zero record: y0 == y1, skipped by everyone)
    const uint32_t bit = 1u << (k & 31);
    RmRow* rr = c.rows + it.rowbase;
    const int emin = y0 < y1 ? y0 : y1, emax = y0 < y1 ? y1 : y0;
    const int ya = emin < 0 ? 0 : emin, yb = emax > a.H - 1 ? a.H - 1 : emax;
    {   // the rows the edge crosses
      const bool any = table && ya <= yb;
      const int last = any ? yb : ya - 1;
      for (int y = ya; RM_ANY(y <= last); ++y) {
        const bool on = y <= last;
        rm_or(on ? &rr[y].act + wsel : spare, on ? bit : 0u);   // (RmRow = four consecutive words)
      }
    }
    {   // a head's row
      const bool on = head && y0 >= 0 && y0 < a.H;
      rm_or(on ? (lng ? &rr[y0].act + wsel : &rr[y0].heads) : spare, on ? bit : 0u);
    }
    {   // rows on which the corner fix-up looks at this edge: its first row; its last if that is the polygon's
      const bool lean = census && table && dx != 0.0f, pos = dx > 0.0f;
      const bool t0 = lean && emin >= 0 && emin < a.H;
      const bool t1 = lean && emax == it.pymax && emax >= 0 && emax < a.H;
      rm_or(t0 ? (pos ? &rr[emin].tipP : &rr[emin].tipN) : spare, t0 ? bit : 0u);
      rm_or(t1 ? (pos ? &rr[emax].tipP : &rr[emax].tipN) : spare, t1 ? bit : 0u);
      // A fix-up moves a crossing only when both edges of the corner run at least 1.5 pixels sideways per row (the new
      // value is ROUND_UP of the nearer neighbour row's crossing -+ 1 and must lie beyond the corner): rows without such
      // an edge skip the partner search.  (1.49: the crossings are float32 sums, off by far less than that.)
      const bool shallow = fabsf(dx) >= 1.49f;
      uint8_t* const fl = reinterpret_cast<uint8_t*>(c.rowitem);
      uint8_t* const fspare = reinterpret_cast<uint8_t*>(spare);
      *((t0 && shallow) ? fl + 2 * (it.rowbase + emin) + 1 : fspare) = 1;
      *((t1 && shallow) ? fl + 2 * (it.rowbase + emax) + 1 : fspare) = 1;
    }
  }
}

#if !RM_DEV && defined(RM_STATS)
static long long rm_hist[16];   // rows by number of active edges
static long long rm_stats[16];   // host model only: [1] generic rows [3] active edges [4..7] rows by number of heads [8] rows with a corner pair [9] most active edges
#endif
// The rows of a pass differ a lot in what they cost -- two edges and nothing else (half of them), six or seven edges, a
// horizontal head or two, a corner fix-up -- and a wavefront pays for its most expensive lane in every loop.  So the rows
// are handed to p4's threads in order of their kind (a counting sort on four buckets): rows with heads or a possible
// fix-up first, then by the number of crossing edges.  The order of the rows has no effect on the picture.
//   p4a: every row's bucket, its place in the bucket (ballots; one LDS atomic per wave, round and bucket)
//   p4b (behind a barrier): the buckets' starts are known -> the sorted list
#define RM_SORT_ROUNDS 3   // rounds of T rows whose sort keys a thread keeps in registers (passes of more rows: rm_p4a_lds)
struct RmSortKey { int key[RM_SORT_ROUNDS]; int pos[RM_SORT_ROUNDS]; };

RM_FN int rm_row_bucket(const RmRow& rec, bool shallow) {
  const bool special = rec.heads != 0u || (shallow && (((rec.tipP & (rec.tipP - 1u)) | (rec.tipN & (rec.tipN - 1u))) != 0u));
  const int pc = __builtin_popcount(rec.act);
  return special ? 0 : (pc >= 5 ? 1 : (pc >= 3 ? 2 : 3));
}

RM_FN void rm_p4a(const RmCtx& c, int total_rows, int tid, int T, RmSortKey& sk) {
#if RM_DEV
  const int lane = tid & 63;
  for (int r = 0; r < RM_SORT_ROUNDS; ++r) {
    if (r * T >= total_rows) break;
    const int w = r * T + tid;
    const bool on = w < total_rows;
    int key = 5;
    if (on) {
      const RmRow rec = c.rows[w];
      const int ri = c.rowitem[w];
      key = (c.info[ri & 255].pb_nv >> 20) > RM_MAX_NV ? 4 : rm_row_bucket(rec, (ri >> 8) != 0);   // 4: a long polygon's row (rm_p4_big)
    }
    int pos = 0;
    for (int b = 0; b < 5; ++b) {
      const unsigned long long mb = __ballot(key == b);
      int base = 0;
      if (lane == 0 && mb) base = atomicAdd(&c.misc[8 + b], __builtin_popcountll(mb));
      base = __builtin_amdgcn_readfirstlane(base);
      if (key == b) pos = base + __builtin_amdgcn_mbcnt_hi((unsigned)(mb >> 32), __builtin_amdgcn_mbcnt_lo((unsigned)mb, 0u));
    }
    sk.key[r] = key; sk.pos[r] = pos;
  }
#else
  (void)sk; (void)T;
  if (tid != 0) return;   // host model: thread 0 sorts the whole pass
  int cnt[4] = {0, 0, 0, 0}, start[4];
  auto bucket = [&](int w) { return (c.info[c.rowitem[w] & 255].pb_nv >> 20) > RM_MAX_NV ? 4 : rm_row_bucket(c.rows[w], (c.rowitem[w] >> 8) != 0); };
  for (int w = 0; w < total_rows; ++w) if (bucket(w) < 4) cnt[bucket(w)]++;
  start[0] = 0; for (int b = 1; b < 4; ++b) start[b] = start[b - 1] + cnt[b - 1];
  for (int w = 0; w < total_rows; ++w) if (bucket(w) < 4) c.sorted[start[bucket(w)]++] = (uint16_t)w;
  for (int b = 0; b < 4; ++b) c.misc[8 + b] = cnt[b];
  int nb = 0;
  for (int w = 0; w < total_rows; ++w) if (bucket(w) == 4) c.sorted[cnt[0] + cnt[1] + cnt[2] + cnt[3] + nb++] = (uint16_t)w;
  c.misc[12] = nb;
#endif
}

RM_FN void rm_p4b(const RmCtx& c, int total_rows, int tid, int T, const RmSortKey& sk) {
#if RM_DEV
  const int n0 = c.misc[8], n1 = c.misc[9], n2 = c.misc[10];
  for (int r = 0; r < RM_SORT_ROUNDS; ++r) {
    if (r * T >= total_rows) break;
    const int key = sk.key[r];
    const int start = key == 0 ? 0 : (key == 1 ? n0 : (key == 2 ? n0 + n1 : (key == 3 ? n0 + n1 + n2 : n0 + n1 + n2 + c.misc[11])));
    if (key < 5) c.sorted[start + sk.pos[r]] = (uint16_t)(r * T + tid);
  }
#else
  (void)c; (void)total_rows; (void)tid; (void)T; (void)sk;
#endif
}

// The same with the keys in LDS instead of registers: passes of more than RM_SORT_ROUNDS * T rows (device only: the host model's
// rm_p4a sorts any number)
RM_FN void rm_p4a_lds(const RmCtx& c, int total_rows, int tid, int T) {
#if RM_DEV
  const int lane = tid & 63;
  for (int w0 = 0; w0 < total_rows; w0 += T) {
    const int w = w0 + tid;
    const bool on = w < total_rows;
    int key = 5;
    if (on) {
      const RmRow rec = c.rows[w];
      const int ri = c.rowitem[w];
      key = (c.info[ri & 255].pb_nv >> 20) > RM_MAX_NV ? 4 : rm_row_bucket(rec, (ri >> 8) != 0);   // 4: a long polygon's row (rm_p4_big)
    }
    int pos = 0;
    for (int b = 0; b < 5; ++b) {
      const unsigned long long mb = __ballot(key == b);
      int base = 0;
      if (lane == 0 && mb) base = atomicAdd(&c.misc[8 + b], __builtin_popcountll(mb));
      base = __builtin_amdgcn_readfirstlane(base);
      if (key == b) pos = base + __builtin_amdgcn_mbcnt_hi((unsigned)(mb >> 32), __builtin_amdgcn_mbcnt_lo((unsigned)mb, 0u));
    }
    if (on) c.skey[w] = (uint16_t)((key << 12) | pos);   // (a pass has at most 4096 rows: rm_plan's callers)
  }
#else
  (void)c; (void)total_rows; (void)tid; (void)T;
#endif
}

RM_FN void rm_p4b_lds(const RmCtx& c, int total_rows, int tid, int T) {
#if RM_DEV
  const int n0 = c.misc[8], n1 = c.misc[9], n2 = c.misc[10], n3 = c.misc[11];
  for (int w = tid; w < total_rows; w += T) {
    const int sk = c.skey[w], key = sk >> 12, pos = sk & 4095;
    const int start = key == 0 ? 0 : (key == 1 ? n0 : (key == 2 ? n0 + n1 : (key == 3 ? n0 + n1 + n2 : n0 + n1 + n2 + n3)));
    c.sorted[start + pos] = (uint16_t)w;
  }
#else
  (void)c; (void)total_rows; (void)tid; (void)T;
#endif
}

// p4: one thread per (item, row)
template <int WORDS, bool COMPACT>
RM_FN void rm_p4(const RmArgs& a, const RmCtx& c, int total_rows, int tid, int T, float* xx_wave) {
  typedef typename RmEdgeTab<COMPACT>::View EV;
  const int nseg = a.W >> 4;
  (void)total_rows;
  const int sorted_rows = c.misc[8] + c.misc[9] + c.misc[10] + c.misc[11];   // (the pass's rows but those of long polygons)
  for (int w0 = 0; w0 < sorted_rows; w0 += T) {
    const bool on = w0 + tid < sorted_rows;
    const int w = on ? c.sorted[w0 + tid] : 0;
    RmRow rec = {0u, 0u, 0u, 0u};
    RmItem it = {0, 0, 0, 0u};
    int g = 0;
    bool shallow = false;
    if (on) { rec = c.rows[w]; const int ri = c.rowitem[w]; g = ri & 255; shallow = (ri >> 8) != 0; it = c.info[g]; }
    const int y = w - it.rowbase;
    const EV pe = RmEdgeTab<COMPACT>::view(c, it.pb_nv & 0xfffff, it.pb_nv >> 20);
    RmMask<WORDS> m;
    const bool ok = rm_row_fast<WORDS, EV>(pe, rec, y, it.pymax, a.W, shallow, m);
#if !RM_DEV && defined(RM_STATS)
    if (on) {
      rm_stats[1] += ok ? 0 : 1;
      rm_stats[3] += __builtin_popcount(rec.act);
      const int nh = __builtin_popcount(rec.heads);
      rm_stats[4 + (nh > 3 ? 3 : nh)]++;
      rm_stats[8] += ((rec.tipP & (rec.tipP - 1u)) || (rec.tipN & (rec.tipN - 1u))) ? 1 : 0;
      { const int pc = __builtin_popcount(rec.act); rm_hist[pc > 15 ? 15 : pc]++; }
    }
#endif
#if RM_DEV
    unsigned long long gm = __ballot(on && !ok);
    while (gm) {   // the rare rows, one at a time (they share the wave's scratch list)
      const int l = __builtin_ctzll(gm);
      gm &= gm - 1ull;
      if ((tid & 63) == l) m = rm_row_generic<WORDS, EV>(pe, it.pb_nv >> 20, rec.heads, y, it.pymax, xx_wave);
    }
#else
    if (on && !ok) m = rm_row_generic<WORDS, EV>(pe, it.pb_nv >> 20, rec.heads, y, it.pymax, xx_wave);
#endif
    if (on) {
      uint64_t* mp = reinterpret_cast<uint64_t*>(c.rows + w);
      mp[0] = m.w[0];
      if (WORDS > 1) mp[1] = m.w[1];
      for (int sg = 0; sg < nseg; ++sg) {
        const uint64_t mw = m.w[WORDS > 1 ? (sg >> 2) : 0];
        if ((mw >> ((sg & 3) * 16)) & 0xffffull) rm_or(&c.seg[(y * nseg + sg) * a.iwords + (g >> 5)], 1u << (g & 31));
      }
    }
  }
}

// ---- long polygons (RM_MAX_NV < nv <= RM_BIG_NV: the 102-vertex annuli of the reference's fixation screens) ----------------
// A census word has 32 bits, so p3 leaves ONE 128-bit word per row of such a polygon: the edges that cross the row and the heads
// that lie on it.  Their rows come behind the others in the sorted list (bucket 4) and take a thread each like them: the thread
// walks the set bits, reads off every such edge's record what it is on this row (crossing, head, corner of which lean) and
// notes it in a list of edge numbers of its own (32 bytes of LDS), with the four census words over the list's positions.  polygon_generic only ever looks at a
// row's active edges and heads, in order, so the row routines of the short polygons run on the list unchanged (RmIndexed).
// A row with more than RM_MAX_NV such edges (a comb) goes to rm_row_generic over the whole polygon.
template <class EV> struct RmIndexed {
  EV pe; const uint8_t* idx;
  RM_MEMBER RmEdge operator[](int k) const { return pe[idx[k]]; }
};
// what an edge record says about row y without any arithmetic: a crossing, a head on the row, a corner the fix-up looks at
RM_FN void rm_big_classify(const RmEdge& E, int y, int pymax, bool& cross, bool& head, bool& tip_p, bool& tip_n, bool& shallow) {
  cross = false; head = false; tip_p = false; tip_n = false; shallow = false;
  const int y0 = rm_y0(E), y1 = rm_y1(E);
  if (E.w3 == 1u) { head = (y0 == y); return; }
  if (y0 == y1) return;
  const int emin = y0 < y1 ? y0 : y1, emax = y0 < y1 ? y1 : y0;
  if (y < emin || y > emax) return;
  cross = true;
  const float dx = rm_u2f(E.w1);
  const bool corner = (y == emin) || (y == emax && emax == pymax);
  if (dx != 0.0f && corner) { tip_p = dx > 0.0f; tip_n = !tip_p; shallow = fabsf(dx) >= 1.49f; }
}

// idx_all: 32 bytes per thread of the frame (behind the waves' crossing lists)
template <int WORDS, bool COMPACT>
RM_FN void rm_p4_big(const RmArgs& a, const RmCtx& c, int tid, int T, float* xx_wave, uint8_t* idx_all) {
  typedef typename RmEdgeTab<COMPACT>::View EV;
  typedef RmIndexed<EV> IX;
  const int nseg = a.W >> 4;
  const int first = c.misc[8] + c.misc[9] + c.misc[10] + c.misc[11], nbig = c.misc[12];
  uint8_t* const idx = idx_all + 32 * tid;
  for (int w0 = 0; w0 < nbig; w0 += T) {
    const bool on = w0 + tid < nbig;
    const int w = on ? c.sorted[first + w0 + tid] : 0;
    RmItem it = {0, 0, 0, 0u};
    int g = 0;
    if (on) { g = c.rowitem[w] & 255; it = c.info[g]; }
    const int y = w - it.rowbase, nv = on ? (it.pb_nv >> 20) : 0;
    const EV pe = RmEdgeTab<COMPACT>::view(c, it.pb_nv & 0xfffff, it.pb_nv >> 20);
    RmRow rec = {0u, 0u, 0u, 0u};
    bool shallow = false;
    int n_rel = 0;
    uint32_t bw[4] = {0u, 0u, 0u, 0u};
    if (on) { const RmRow r = c.rows[w]; bw[0] = r.act; bw[1] = r.heads; bw[2] = r.tipP; bw[3] = r.tipN; }   // the row's 128-bit word
    const int n_all = __builtin_popcount(bw[0]) + __builtin_popcount(bw[1]) + __builtin_popcount(bw[2]) + __builtin_popcount(bw[3]);
    if (n_all <= RM_MAX_NV) {
      for (int q = 0; q < 4; ++q) {
        uint32_t mq = q * 32 < nv ? bw[q] : 0u;
        if (RM_ANY(mq != 0u)) do {   // (wave-uniform loop: a lane without a bit left goes through the motions on edge 0)
          const bool valid = mq != 0u;
          const int k = valid ? q * 32 + rm_ffs(mq) : 0;
          mq &= mq - 1u;
          bool cr, hd, tp, tn, sh;
          rm_big_classify(pe[k], y, it.pymax, cr, hd, tp, tn, sh);
          if (valid) {
            idx[n_rel] = (uint8_t)k;
            const uint32_t bit = 1u << n_rel;
            rec.act |= cr ? bit : 0u; rec.heads |= hd ? bit : 0u; rec.tipP |= tp ? bit : 0u; rec.tipN |= tn ? bit : 0u;
            shallow = shallow || sh;
            ++n_rel;
          }
        } while (RM_ANY(mq != 0u));
      }
    } else n_rel = n_all;
    IX list; list.pe = pe; list.idx = idx;
    RmMask<WORDS> m;
    rm_clear(m);
    const bool all = on && n_rel > RM_MAX_NV;
    if (all) { rec.act = 0u; rec.heads = 0u; rec.tipP = 0u; rec.tipN = 0u; }
    const bool ok = rm_row_fast<WORDS, IX>(list, rec, y, it.pymax, a.W, shallow, m);
#if RM_DEV
    unsigned long long gm = __ballot(on && (all || !ok));
    while (gm) {   // the rare rows, one at a time (they share the wave's scratch list)
      const int l = __builtin_ctzll(gm);
      gm &= gm - 1ull;
      if ((tid & 63) == l) {
        if (all) m = rm_row_generic<WORDS, EV>(pe, nv, 0u, y, it.pymax, xx_wave);
        else m = rm_row_generic<WORDS, IX>(list, n_rel, rec.heads, y, it.pymax, xx_wave);
      }
    }
#else
    if (on && all) m = rm_row_generic<WORDS, EV>(pe, nv, 0u, y, it.pymax, xx_wave);
    else if (on && !ok) m = rm_row_generic<WORDS, IX>(list, n_rel, rec.heads, y, it.pymax, xx_wave);
#if defined(RM_STATS)
    if (on) { rm_stats[1] += (all || !ok) ? 1 : 0; rm_stats[10] += 1; rm_stats[11] += all ? 1 : 0; }
#endif
#endif
    if (on) {
      uint64_t* mp = reinterpret_cast<uint64_t*>(c.rows + w);
      mp[0] = m.w[0];
      if (WORDS > 1) mp[1] = m.w[1];
      for (int sg = 0; sg < nseg; ++sg) {
        const uint64_t mw = m.w[WORDS > 1 ? (sg >> 2) : 0];
        if ((mw >> ((sg & 3) * 16)) & 0xffffull) rm_or(&c.seg[(y * nseg + sg) * a.iwords + (g >> 5)], 1u << (g & 31));
      }
    }
  }
}

// Draw.c BLEND8 / DIV255 on one channel
RM_FN uint32_t rm_blend8(uint32_t bg, uint32_t fg, uint32_t al) {
  const uint32_t t = bg * (255u - al) + fg * al + 128u;
  return ((t >> 8) + t) >> 8;
}

// BLEND8 on the four bytes of a dword at once (bytes 0 and 2, then 1 and 3, in 16-bit lanes: bg * (255 - a) + fg * a + 128
// <= 65153 and the DIV255 sum stay below 65536, so nothing carries from one lane into the next)
RM_FN uint32_t rm_blend_dword(uint32_t o, uint32_t f, uint32_t al) {
  const uint32_t na = 255u - al, k = 0x00ff00ffu;
  uint32_t e = (o & k) * na + (f & k) * al + 0x00800080u;
  e = ((((e >> 8) & k) + e) >> 8) & k;
  uint32_t h = ((o >> 8) & k) * na + ((f >> 8) & k) * al + 0x00800080u;
  h = ((((h >> 8) & k) + h) >> 8) & k;
  return e | (h << 8);
}

// p5: compose (painter's order = item order) and store: one 16-pixel segment = 48 bytes = 12 dwords per thread.
// first_pass: the picture starts from the background colour / the cached static prefix, else from what `image` holds.
// ITEMS = false: a segment no item touches (a copy of the background).
template <int WORDS, bool ITEMS>
RM_FN void rm_p5_segment(const RmArgs& a, const RmCtx& c, uint8_t* out, bool first_pass, bool from_cache, int seg, bool on) {
  const int nseg = a.W >> 4;
  const uint32_t bg0 = (a.bg & 0xffffffu) | (a.bg << 24), bg1 = ((a.bg >> 8) & 0xffffu) | (a.bg << 16), bg2 = ((a.bg >> 16) & 0xffu) | (a.bg << 8);
  const int y = seg / nseg, sg = seg - y * nseg, x0 = sg * 16;
  const size_t off = ((size_t)(a.flip ? a.H - 1 - y : y) * a.W + x0) * 3;   // (a multiple of 48)
  RmU4* dst = reinterpret_cast<RmU4*>(out + off);
  uint32_t d[12];
  if (first_pass && !from_cache) {
    for (int q = 0; q < 4; ++q) { d[3 * q] = bg0; d[3 * q + 1] = bg1; d[3 * q + 2] = bg2; }
  } else if (on) {
    const RmU4* src = from_cache ? reinterpret_cast<const RmU4*>(a.sbg + off) : dst;
    const RmU4 u0 = src[0], u1 = src[1], u2 = src[2];
    d[0] = u0.x; d[1] = u0.y; d[2] = u0.z; d[3] = u0.w; d[4] = u1.x; d[5] = u1.y; d[6] = u1.z; d[7] = u1.w;
    d[8] = u2.x; d[9] = u2.y; d[10] = u2.z; d[11] = u2.w;
  } else {
    for (int q = 0; q < 12; ++q) d[q] = 0u;
  }
  if (ITEMS) {
    for (int iw = 0; iw < a.iwords; ++iw) {
      uint32_t bitsw = on ? c.seg[seg * a.iwords + iw] : 0u;
      // (wave-uniform loop, no branch inside for opaque items: a lane without an item left paints with an empty mask)
      // (do-while: with a plain while loop the compiler copies all twelve dwords of the segment once per iteration)
      if (RM_ANY(bitsw != 0u)) do {
        const bool valid = bitsw != 0u;
        const int g = valid ? iw * 32 + rm_ffs(bitsw) : 0;
        bitsw &= bitsw - 1u;
        const RmItem it = c.info[g];
        const uint32_t* mrow = reinterpret_cast<const uint32_t*>(c.rows + (valid ? it.rowbase + y : 0));
        const uint32_t bits = valid ? (mrow[x0 >> 5] >> (x0 & 31)) & 0xffffu : 0u;
        const uint32_t rgb = it.rgba, al = it.rgba >> 24;
        const uint32_t c0 = (rgb & 0xffffffu) | (rgb << 24), c1 = ((rgb >> 8) & 0xffffu) | (rgb << 16), c2 = ((rgb >> 16) & 0xffu) | (rgb << 8);
        const bool blend = valid && al != 255u;
        if (RM_ANY(blend)) {   // a translucent sprite in this wavefront: hline32rgba, one BLEND8 per channel of every covered pixel
          for (int q = 0; q < 4; ++q) {
            const RmU4 lm = *reinterpret_cast<const RmU4*>(c.lut + 4 * ((bits >> (4 * q)) & 15u));
            const uint32_t s0 = blend ? rm_blend_dword(d[3 * q], c0, al) : c0, s1 = blend ? rm_blend_dword(d[3 * q + 1], c1, al) : c1,
                           s2 = blend ? rm_blend_dword(d[3 * q + 2], c2, al) : c2;
            d[3 * q] = (s0 & lm.x) | (d[3 * q] & ~lm.x);
            d[3 * q + 1] = (s1 & lm.y) | (d[3 * q + 1] & ~lm.y);
            d[3 * q + 2] = (s2 & lm.z) | (d[3 * q + 2] & ~lm.z);
          }
        } else {
          for (int q = 0; q < 4; ++q) {
            const RmU4 lm = *reinterpret_cast<const RmU4*>(c.lut + 4 * ((bits >> (4 * q)) & 15u));
            d[3 * q] = (c0 & lm.x) | (d[3 * q] & ~lm.x);
            d[3 * q + 1] = (c1 & lm.y) | (d[3 * q + 1] & ~lm.y);
            d[3 * q + 2] = (c2 & lm.z) | (d[3 * q + 2] & ~lm.z);
          }
        }
      } while (RM_ANY(bitsw != 0u));
    }
  }
  if (on) {
    RmU4 v0, v1, v2;
    v0.x = d[0]; v0.y = d[1]; v0.z = d[2]; v0.w = d[3]; v1.x = d[4]; v1.y = d[5]; v1.z = d[6]; v1.w = d[7];
    v2.x = d[8]; v2.y = d[9]; v2.z = d[10]; v2.w = d[11];
    dst[0] = v0; dst[1] = v1; dst[2] = v2;
  }
}

template <int WORDS>
RM_FN void rm_p5(const RmArgs& a, const RmCtx& c, int env, bool first_pass, int s_lo, int tid, int T) {
  // (Listing the segments that hold a sprite -- four in ten -- and composing those with every lane busy was measured:
  //  the copies of the empty ones and the list cost what the denser loop saves, profiles/r05_raster.txt.)
  const int nseg = a.W >> 4, segs = a.H * nseg;
  uint8_t* out = a.image + (size_t)env * a.H * a.W * 3;
  const bool from_cache = first_pass && s_lo > 0;
  for (int seg = tid; seg < segs; seg += T) rm_p5_segment<WORDS, true>(a, c, out, first_pass, from_cache, seg, true);
}

// Later passes start from clean row records / segment words
RM_FN void rm_next_pass(const RmArgs& a, const RmCtx& c, int tid, int T) {
  if (tid < 5) c.misc[8 + tid] = 0;
  for (int i = tid; i < a.cap_rows; i += T) { RmRow z = {0u, 0u, 0u, 0u}; c.rows[i] = z; }
  for (int i = tid; i < (a.cap_rows + 1) / 2; i += T) reinterpret_cast<uint32_t*>(c.rowitem)[i] = 0u;   // (the rows' flag bytes)
  const int nseg = a.W >> 4;
  for (int i = tid; i < a.H * nseg * a.iwords; i += T) c.seg[i] = 0u;
}

#endif  // MOOG_RASTER_MASK_CORE_H_
