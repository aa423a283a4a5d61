// moog_raster_kernel.h -- the rasteriser's device code (see moog_raster.h for the design): one workgroup renders one
// tile of one frame.  Included by moog_raster.hip (the stand-alone kernel) and by the fused step + raster launch.
#ifndef MOOG_RASTER_KERNEL_H_
#define MOOG_RASTER_KERNEL_H_
#include "moog_device.h"
#include "moog_raster.h"

extern __shared__ __attribute__((aligned(16))) unsigned char moog_lds[];

// Draw.c ROUND_UP / ROUND_DOWN: sign-symmetric, so branch-free with copysign
__device__ __forceinline__ int pil_round_up(float f) {
  return (int)copysignf(floorf(fabsf(f) + 0.5f), f);
}
__device__ __forceinline__ int pil_round_down(float f) {
  return (int)copysignf(ceilf(fabsf(f) - 0.5f), f);
}
// key <-> the two roundings: s = up + down; the roundings differ (by one, away from /
// towards zero) only for an exact half-integer x, which is when s is odd.
__device__ __forceinline__ int key_up(unsigned k) {
  int s = (int)k - R_KEY_BIAS;
  return s >= 0 ? (s + 1) >> 1 : s >> 1;
}
__device__ __forceinline__ int key_down(unsigned k) {
  int s = (int)k - R_KEY_BIAS;
  return s >= 0 ? s >> 1 : (s + 1) >> 1;
}

// color_maps.py:21-23 (colorsys.hsv_to_rgb, then uint8 truncation)
__device__ inline void hsv_to_rgb_u8(double h, double s, double v, unsigned& r8, unsigned& g8,
                                     unsigned& b8) {
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
  r8 = (unsigned)(int)(255 * r) & 255u; g8 = (unsigned)(int)(255 * g) & 255u;
  b8 = (unsigned)(int)(255 * b) & 255u;
}

// Pillow's (int) cast of a coordinate as x86-64 performs it (cvttsd2si): NaN and values
// outside the int range give INT_MIN (the reference can produce NaN sprite state)
__device__ __forceinline__ int pil_int(double d) {
  return (d >= -2147483648.0 && d < 2147483648.0) ? (int)d : (int)0x80000000;
}

__device__ inline short clamp16(int v) { return (short)(v < -32000 ? -32000 : (v > 32000 ? 32000 : v)); }

struct RMask { unsigned long long w0, w1; };

// (W = width of the tile the workgroup renders, xoff = its first canvas column: bit i of the mask is column xoff + i)
__device__ inline void mask_fill(RMask& m, int W, int xoff, int x0, int x1) {
  x0 -= xoff; x1 -= xoff;
  if (x0 < 0) x0 = 0; else if (x0 >= W) return;
  if (x1 < 0) return; else if (x1 >= W) x1 = W - 1;
  if (x0 > x1) return;
  // bits [x0, x1] of a 128-bit mask
  if (x0 < 64) {
    int hi = x1 < 63 ? x1 : 63;
    m.w0 |= (~0ull << x0) & (~0ull >> (63 - hi));
  }
  if (x1 >= 64) {
    int lo = x0 > 64 ? x0 - 64 : 0, hi = x1 - 64;
    m.w1 |= (~0ull << lo) & (~0ull >> (63 - hi));
  }
}

// view of one polygon's edge records in LDS
struct RPoly {
  const REdge* e;            // n records, one per vertex
  int n;
  const unsigned* head;      // bit k: record k is a horizontal head (null: one word, all ones)
  int hwords;
  unsigned rowbits;          // heads on the row being drawn, bit (k mod 32) (~0u: not known)
};

__device__ __forceinline__ bool r_is_table(const REdge& E) { return E.y0 != E.y1; }

// Draw.c draw_horizontal_lines (heads visited in edge order)
__device__ inline void draw_horizontal(const RPoly& p, int y, int* x_pos, RMask& m, int W, int xoff) {
  for (int hw = 0; hw < p.hwords; ++hw) {
    unsigned bits = (p.head ? p.head[hw] : ~0u) & p.rowbits;
    while (bits) {
      int k = hw * 32 + __ffs((int)bits) - 1;
      bits &= bits - 1u;
      REdge h = p.e[k];
      if (h.y0 != y) continue;
      unsigned xb = (unsigned)__float_as_int(h.dx);
      int xmin = (short)(xb & 0xffffu), xmax = (short)(xb >> 16);
      if (*x_pos != -1 && *x_pos < xmin) continue;
      if (*x_pos > xmin) {
        xmin = *x_pos;
        if (xmax < xmin) continue;
      }
      mask_fill(m, W, xoff, xmin, xmax);
      *x_pos = xmax + 1;
    }
  }
}

// polygon_generic's corner fix-up, one candidate: E is the edge whose end row (its first
// row when `top`, else its last) is being processed, K an EARLIER table edge.  K decides
// the fix-up when it is active on the row with the same tip point, leans the same way and
// crosses the row at the same x; the reference stops at the first such edge whether or not
// a replacement results.  Returns true when K decides; *vv = replacement or R_NONE.
__device__ __forceinline__ bool tip_decide(const REdge& E, const REdge& K, bool top, short* vv) {
  const int y0 = E.y0, y1 = E.y1, ky0 = K.y0, ky1 = K.y1;
  const int y = top ? (y0 < y1 ? y0 : y1) : (y0 < y1 ? y1 : y0);
  const int ktip = top ? (ky0 < ky1 ? ky0 : ky1) : (ky0 < ky1 ? ky1 : ky0);
  if (ktip != y) return false;
  const int tipx = (y == y0) ? E.x0 : E.x1;
  const int ktx = (ktip == ky0) ? K.x0 : K.x1;
  if (ktx != tipx) return false;
  const float dx = E.dx, kdx = K.dx;
  if ((dx > 0 && kdx <= 0) || (dx < 0 && kdx >= 0)) return false;
  const float x = (float)(y - y0) * dx + (float)E.x0;
  if (x != (float)(y - ky0) * kdx + (float)K.x0) return false;
  const int off = top ? 1 : -1;
  const float adj = (float)(y + off - y0) * dx + (float)E.x0;
  const float adjo = (float)(y + off - ky0) * kdx + (float)K.x0;
  *vv = R_NONE;
  if (adj > x && adjo > x) {
    float v = (float)(pil_round_up(fminf(adj, adjo)) - 1);
    if (v > x) *vv = (short)(int)v;
  } else if (adj < x && adjo < x) {
    float v = (float)(pil_round_up(fmaxf(adj, adjo)) + 1);
    if (v < x) *vv = (short)(int)v;
  }
  return true;
}

// Fix-up replacements of table edge k (dx != 0) for its top and bottom rows.  An earlier
// edge can only qualify if it touches the tip point, i.e. if one of the vertices 0..k is
// that point: the loop compares packed vertices and looks at the two edges incident to a
// matching vertex (in edge order, so the first qualifying edge decides).
__device__ inline void tip_replacements(const REdge* reg, const unsigned* pts, int k, const REdge& E,
                                        bool need_top, bool need_bot, short* vtop, short* vbot) {
  *vtop = R_NONE; *vbot = R_NONE;
  const unsigned w0 = (unsigned short)E.x0 | ((unsigned)(unsigned short)E.y0 << 16);
  const unsigned w1 = (unsigned short)E.x1 | ((unsigned)(unsigned short)E.y1 << 16);
  const bool up = E.y0 < E.y1;
  const unsigned tw = up ? w0 : w1, bw = up ? w1 : w0;
  bool done_top = !need_top || E.dx == 0.0f, done_bot = !need_bot || E.dx == 0.0f;
  // Vertices 0 .. k-1 coincide with a tip only in degenerate (truncated) polygons.  The
  // scan for the next coinciding vertex is wave uniform (two packed points per step, no
  // work inside); the rare hits are examined between scans.
  int j0 = 0;
  for (;;) {
    const int lim = (done_top && done_bot) ? 0 : k;
    int found = lim;
    for (int j = j0; __any(j < found); j += 2) {
      unsigned v0 = pts[j], v1 = pts[j + 1];
      bool m1 = (j + 1 < found) && ((!done_top && v1 == tw) || (!done_bot && v1 == bw));
      bool m0 = (j < found) && ((!done_top && v0 == tw) || (!done_bot && v0 == bw));
      found = m1 ? j + 1 : found;
      found = m0 ? j : found;
    }
    if (!__any(found < lim)) break;
    // A coinciding vertex matters only if one of its two edges is a table edge with that
    // tip and E's lean: decided from the three packed points around it (cheap; most hits
    // are runs of equal points on tiny circles and end here).
    bool useful = false;
    if (found < lim) {
      const unsigned pj = pts[found];
      const bool top = !done_top && pj == tw;   // else the bottom point matched
      const int jx = (short)(pj & 0xffffu), jy = (short)(pj >> 16);
      const int lean = E.dx > 0.0f ? 1 : -1;
      {   // edge found -> found + 1 (an earlier edge since found < k)
        const unsigned pn = pts[found + 1];
        const int dxi = (short)(pn & 0xffffu) - jx, dyi = (short)(pn >> 16) - jy;
        const bool tip_here = top ? dyi > 0 : dyi < 0;
        useful = tip_here && ((dxi > 0) == (dyi > 0) ? 1 : -1) == lean && dxi != 0;
      }
      if (found > 0) {   // edge found - 1 -> found
        const unsigned pp = pts[found - 1];
        const int dxi = jx - (short)(pp & 0xffffu), dyi = jy - (short)(pp >> 16);
        const bool tip_here = top ? dyi < 0 : dyi > 0;
        useful = useful || (tip_here && ((dxi > 0) == (dyi > 0) ? 1 : -1) == lean && dxi != 0);
      }
    }
    if (__any(useful)) {
      if (useful) {
        for (int e = (found > 0 ? found - 1 : 0); e <= found; ++e) {
          REdge K = reg[e];
          if (!r_is_table(K)) continue;
          if (!done_top && tip_decide(E, K, true, vtop)) done_top = true;
          if (!done_bot && tip_decide(E, K, false, vbot)) done_bot = true;
        }
      }
    }
    j0 = found < lim ? found + 1 : k;
  }
  // vertex k is this edge's own start point: the edge arriving there (all lanes together)
  // (vertex k is this edge's top if the edge runs downwards, else its bottom: one test)
  if (k >= 1 && !(up ? done_top : done_bot)) {
    REdge K = reg[k - 1];
    short vv = R_NONE;
    if (r_is_table(K)) tip_decide(E, K, up, &vv);
    if (up) *vtop = vv; else *vbot = vv;
  }
}

// Marks a row for the generic routine; the first marker queues it.
__device__ __forceinline__ void make_generic(RRow* r, int w, unsigned short* queue, int* misc) {
  unsigned old = atomicOr(&r->cnt, R_GENERIC);
  if (!(old & R_GENERIC)) queue[atomicAdd(&misc[2], 1)] = (unsigned short)w;
}

// One crossing of table edge E with row y, in two halves so that a thread with several rows has
// all its slot requests (LDS atomics with return) in flight before it needs the first answer.
struct RPush { unsigned key, n; RRow* r; bool on, fix, far; unsigned pos; };

__device__ __forceinline__ RPush push_prepare(RRow* rows, int rb, const REdge& E, int emin, int emax, int pymax,
                                              short vtop, short vbot, int y, bool on) {
  RPush p;
  float x = (float)(y - (int)E.y0) * E.dx + (float)E.x0;
  const bool bot = (y == emax);
  const bool dup = bot && (y < pymax);      // polygon_generic: an edge's last row counts twice
  const short vv = (y == emin) ? vtop : ((bot && !dup) ? vbot : R_NONE);
  p.n = dup ? 2u : 1u;
  p.fix = on && (vv != R_NONE);
  if (vv != R_NONE) x = (float)vv;
  p.far = !(fabsf(x) <= R_XLIM);
  p.key = (unsigned)(pil_round_up(x) + pil_round_down(x) + R_KEY_BIAS);
  p.r = rows + (rb + y);
  p.on = on;
  p.pos = 0u;
  return p;
}

__device__ __forceinline__ void push_commit(const RPush& p, const REdge& E, int rb, int y, int g,
                                            unsigned short* queue, int* misc) {
  if (!p.on) return;
  const unsigned pos = p.pos & R_CNT_MASK;
  if (pos < R_CAP) p.r->key[pos] = (unsigned short)p.key;
  if (p.n == 2u && pos + 1 < R_CAP) p.r->key[pos + 1] = (unsigned short)p.key;
  if (pos == 0u) atomicOr(&p.r->cnt, (unsigned)(g + 1) << R_ITEM_SHIFT);   // the first arrival names the item
  bool gen = p.far || (pos <= R_CAP && pos + p.n > R_CAP);
  if (p.fix) {
    // Two fix-ups on one row are independent unless they belong to the same tip point (then
    // the reference overwrites one partner entry twice): remember the tip columns mod 8.
    const int tipx = (y == E.y0) ? E.x0 : E.x1;
    const unsigned bit = R_FIX_ONE << (tipx & 7);
    gen = gen || (atomicOr(&p.r->cnt, bit) & bit);
  }
  if (gen) make_generic(p.r, rb + y, queue, misc);
}

__device__ __forceinline__ void push_crossing(RRow* rows, int rb, const REdge& E, int emin, int emax,
                                              int pymax, short vtop, short vbot, int g, int y,
                                              unsigned short* queue, int* misc) {
  RPush p = push_prepare(rows, rb, E, emin, emax, pymax, vtop, vbot, y, true);
  p.pos = atomicAdd(&p.r->cnt, p.n);
  push_commit(p, E, rb, y, g, queue, misc);
}

// Generic scanline (any number of crossings, several fix-ups): crossing list in LDS.
// xx: this thread's crossing list, element j at xx[j * R_SLOW].
__device__ inline RMask scanline_mask_generic(const RPoly& p, int y, int poly_ymax, float* xx, int W, int xoff,
                                              const int R_XX) {
  RMask m = {0ull, 0ull};
  int j = 0;
  for (int i = 0; i < p.n; ++i) {
    REdge E = p.e[i];
    if (!r_is_table(E)) continue;
    int y0 = E.y0, y1 = E.y1;
    int emin = y0 < y1 ? y0 : y1, emax = y0 < y1 ? y1 : y0;
    if (y < emin || y > emax) continue;
    float dx = E.dx;
    float x = (float)(y - y0) * dx + (float)E.x0;
    if (j < R_XX) xx[j * R_SLOW] = x;
    ++j;
    if (y == emax && y < poly_ymax) {
      if (j < R_XX) xx[j * R_SLOW] = x;
      ++j;
    } else if (dx != 0.0f && (y == emin || y == emax)) {
      // connect discontiguous corners: the partner's entry on this row is overwritten
      short vv = R_NONE;
      int kt = -1;
      for (int k = 0; k < i; ++k) {
        REdge K = p.e[k];
        if (!r_is_table(K)) continue;
        if (tip_decide(E, K, y == emin, &vv)) { kt = k; break; }
      }
      if (kt >= 0 && vv != R_NONE) {
        int kpos = 0;
        for (int k = 0; k < kt; ++k) {
          REdge K = p.e[k];
          if (!r_is_table(K)) continue;
          int kmin = K.y0 < K.y1 ? K.y0 : K.y1, kmax = K.y0 < K.y1 ? K.y1 : K.y0;
          if (y < kmin || y > kmax) continue;
          kpos += (y == kmax && y < poly_ymax) ? 2 : 1;
        }
        if (kpos < R_XX) xx[kpos * R_SLOW] = (float)vv;
      }
    }
  }
  if (j > R_XX) j = R_XX;
  for (int q = 1; q < j; ++q) {  // insertion sort (qsort with x_cmp)
    float key = xx[q * R_SLOW];
    int r = q - 1;
    while (r >= 0 && xx[r * R_SLOW] > key) { xx[(r + 1) * R_SLOW] = xx[r * R_SLOW]; --r; }
    xx[(r + 1) * R_SLOW] = key;
  }
  int x_pos = (j == 0) ? -1 : 0;
  for (int i = 1; i < j; i += 2) {
    int x_end = pil_round_down(xx[i * R_SLOW]);
    if (x_end < x_pos) continue;
    draw_horizontal(p, y, &x_pos, m, W, xoff);
    if (x_end < x_pos) continue;
    int x_start = pil_round_up(xx[(i - 1) * R_SLOW]);
    if (x_pos > x_start) {
      x_start = x_pos;
      if (x_end < x_start) continue;
    }
    mask_fill(m, W, xoff, x_start, x_end);
    x_pos = x_end + 1;
  }
  draw_horizontal(p, y, &x_pos, m, W, xoff);
  return m;
}

// Batcher's odd-even merge sort on N registers (N a power of two); entries that are the
// compile-time constant 0xffffffff fold away, so padding to N costs nothing.
template <int N>
__device__ __forceinline__ void sort_network(unsigned (&k)[N]) {
#pragma unroll
  for (int p = 1; p < N; p *= 2) {
#pragma unroll
    for (int q = p; q >= 1; q /= 2) {
#pragma unroll
      for (int j = q % p; j <= N - 1 - q; j += 2 * q) {
#pragma unroll
        for (int i = 0; i < q; ++i) {
          if (i + j + q <= N - 1 && (i + j) / (2 * p) == (i + j + q) / (2 * p)) {
            unsigned lo = min(k[i + j], k[i + j + q]), hi = max(k[i + j], k[i + j + q]);
            k[i + j] = lo; k[i + j + q] = hi;
          }
        }
      }
    }
  }
}

// bits [xs, xe] of the row mask, clipped to the canvas; nothing when xs > xe or !pred.  Branch free.
template <int WORDS>
__device__ __forceinline__ void mask_or_range(RMask& m, int W, int xoff, int xs, int xe, bool pred) {
  xs -= xoff; xe -= xoff;
  const int a = xs < 0 ? 0 : xs, b = xe > W - 1 ? W - 1 : xe;
  pred = pred && (a <= b);
  {
    const int hi = b < 63 ? b : 63;
    unsigned long long bits = (~0ull << (a & 63)) & (~0ull >> ((63 - hi) & 63));
    m.w0 |= (pred && a < 64) ? bits : 0ull;
  }
  if (WORDS > 1) {
    const int lo = a > 64 ? a - 64 : 0, hi = b - 64;
    unsigned long long bits = (~0ull << (lo & 63)) & (~0ull >> ((63 - hi) & 63));
    m.w1 |= (pred && b >= 64) ? bits : 0ull;
  }
}

// draw_horizontal_lines for a row with exactly one head [hxmin, hxmax]; `act`: the call happens
template <int WORDS>
__device__ __forceinline__ void draw_one_head(bool act, int hxmin, int hxmax, int& x_pos, RMask& m, int W, int xoff) {
  bool hv = act && (x_pos == -1 || x_pos >= hxmin);
  const int hs = x_pos > hxmin ? x_pos : hxmin;
  hv = hv && !(x_pos > hxmin && hxmax < hs);
  if (__any(hv)) mask_or_range<WORDS>(m, W, xoff, hs, hxmax, hv);
  x_pos = hv ? hxmax + 1 : x_pos;
}

// Pillow's span loop on NP sorted pairs (polygon_generic after qsort), predicated instead of
// branching: lanes of a wave work on rows of different polygons.
template <int NP, int N, int WORDS>
__device__ __forceinline__ RMask span_loop(const unsigned (&k)[N], int cnt, bool head, int hxmin, int hxmax, int W, int xoff) {
  RMask m = {0ull, 0ull};
  int x_pos = (cnt == 0) ? -1 : 0;
  const bool anyhead = __any(head);
#pragma unroll
  for (int q = 0; q < NP; ++q) {
    if (q >= 1 && !__any(2 * q + 1 < cnt)) break;   // no row of the wave has another pair
    const int x_end = key_down(k[2 * q + 1]);
    bool act = (2 * q + 1 < cnt) && (x_end >= x_pos);
    if (anyhead) {
      draw_one_head<WORDS>(act && head, hxmin, hxmax, x_pos, m, W, xoff);
      act = act && (x_end >= x_pos);
    }
    const int x_start = key_up(k[2 * q]);
    const bool gt = x_pos > x_start;
    const int xs = gt ? x_pos : x_start;
    act = act && !(gt && x_end < xs);
    mask_or_range<WORDS>(m, W, xoff, xs, x_end, act);   // empty when x_start > x_end, x_pos still moves
    x_pos = act ? x_end + 1 : x_pos;
  }
  if (anyhead) draw_one_head<WORDS>(head, hxmin, hxmax, x_pos, m, W, xoff);
  return m;
}

// draw_horizontal_lines for a row whose heads are bits of `hb` (polygons of <= 32 edges).  x_pos
// only grows, so a head that is not "after the current position" is finished after this call
// whether it was drawn or not: its bit is cleared and later calls do not visit it again.
template <int WORDS>
__device__ __forceinline__ void draw_pending_heads(const REdge* pe, unsigned& hb, int& x_pos, RMask& m, int W, int xoff) {
  unsigned bits = hb;
  while (bits) {
    const int k = __ffs((int)bits) - 1;
    bits &= bits - 1u;
    const unsigned xb = (unsigned)__float_as_int(pe[k].dx);
    const int xmin = (short)(xb & 0xffffu), xmax = (short)(xb >> 16);
    if (x_pos != -1 && x_pos < xmin) continue;   // after the current position: stays pending
    hb &= ~(1u << k);
    const int hs = x_pos > xmin ? x_pos : xmin;
    const bool draw = !(x_pos > xmin && xmax < hs);
    mask_or_range<WORDS>(m, W, xoff, hs, xmax, draw);
    x_pos = draw ? xmax + 1 : x_pos;
  }
}

template <int NP, int N, int WORDS>
__device__ __forceinline__ RMask span_loop_pending(const unsigned (&k)[N], int cnt, const REdge* pe, unsigned hb, int W, int xoff) {
  RMask m = {0ull, 0ull};
  int x_pos = (cnt == 0) ? -1 : 0;
#pragma unroll
  for (int q = 0; q < NP; ++q) {
    if (q >= 1 && !__any(2 * q + 1 < cnt)) break;
    const int x_end = key_down(k[2 * q + 1]);
    bool act = (2 * q + 1 < cnt) && (x_end >= x_pos);
    if (act) draw_pending_heads<WORDS>(pe, hb, x_pos, m, W, xoff);
    act = act && (x_end >= x_pos);
    const int x_start = key_up(k[2 * q]);
    const bool gt = x_pos > x_start;
    const int xs = gt ? x_pos : x_start;
    act = act && !(gt && x_end < xs);
    mask_or_range<WORDS>(m, W, xoff, xs, x_end, act);
    x_pos = act ? x_end + 1 : x_pos;
  }
  draw_pending_heads<WORDS>(pe, hb, x_pos, m, W, xoff);
  return m;
}

// The same with any number of heads in the row, visited through the polygon's head list
template <int NP, int N>
__device__ __forceinline__ RMask span_loop_poly(const unsigned (&k)[N], int cnt, const RPoly& p, int y, int W, int xoff) {
  RMask m = {0ull, 0ull};
  int x_pos = (cnt == 0) ? -1 : 0;
#pragma unroll
  for (int q = 0; q < NP; ++q) {
    if (q >= 1 && !__any(2 * q + 1 < cnt)) break;   // no row of the wave has another pair
    if (2 * q + 1 < cnt) {
      int x_end = key_down(k[2 * q + 1]);
      if (x_end >= x_pos) {
        draw_horizontal(p, y, &x_pos, m, W, xoff);
        if (x_end >= x_pos) {
          int x_start = key_up(k[2 * q]);
          bool skip = false;
          if (x_pos > x_start) { x_start = x_pos; skip = (x_end < x_start); }
          if (!skip) {
            mask_fill(m, W, xoff, x_start, x_end);
            x_pos = x_end + 1;
          }
        }
      }
    }
  }
  draw_horizontal(p, y, &x_pos, m, W, xoff);
  return m;
}

// Draw.c BLEND8 / DIV255 on one channel
__device__ __forceinline__ unsigned blend8(unsigned bg, unsigned fg, unsigned al) {
  unsigned t = bg * (255u - al) + fg * al + 128u;
  return ((t >> 8) + t) >> 8;
}

// Later passes of a frame with more rows than records start from clean row records (rare; kept
// out of line so that its address arithmetic is not hoisted into the common path's registers).
__device__ __noinline__ void r_next_pass(RRow* rows, int cap_rows, int* misc, int tid) {
  for (int i = tid; i < cap_rows; i += R_THREADS) {
    uint4* r = reinterpret_cast<uint4*>(rows + i);
    r[0] = make_uint4(~0u, ~0u, ~0u, ~0u);
    r[1] = make_uint4(~0u, ~0u, 0u, 0u);
  }
  if (tid == 0) { misc[1] = 0; misc[2] = 0; misc[3] = 0; misc[4] = 0; }
}

// block = index of the workgroup among the launch's raster workgroups; env_of_block >= 0 names the env of a
// one-tile frame directly (the fused launch renders envs in its own order)
template <int WORDS>
__device__ __forceinline__ void raster_block(const RArgs& a, const int block, const int env_of_block) {
  // one workgroup = one tile (<= 128 columns x band_h rows) of one env's frame; frames up to 128 x 128 are one tile
  const int tiles = a.tiles_x * a.bands;
  const int benv = a.tiles_x * a.bands == 1 ? block : block / tiles;
  if (benv >= a.n_envs) return;
  const int tile_id = block - benv * tiles;
  const int env = env_of_block >= 0 ? env_of_block : benv;
  if (a.build && a.env_build && a.env_build[env] == 0) return;   // (per-env prefix: this env's picture is up to date)
  const int band = tile_id / a.tiles_x;
  PProg P = as_const_prog(a.P);
  const int WF = a.canvas_w, H = a.canvas_h;   // the whole canvas (anti_aliasing x the observation)
  const int W = a.tile_w;                                  // this tile: columns [xoff, xoff + W), rows [yb0, yb1)
  const int xoff = (tile_id - band * a.tiles_x) * a.tile_w;
  const int yb0 = band * a.band_h, yb1 = (yb0 + a.band_h < H) ? yb0 + a.band_h : H;
  const int S = P->n_slots, TOTV = a.L.TOTV;
  const bool torus = (P->render.polymod == MOOG_POLYMOD_TORUS);
  const int ncopy = torus ? 9 : 1;
  const int words = WORDS, iwords = a.iwords, hwords = a.hwords, cap_rows = a.chunk;
  const int nseg = W >> 4;
  const int items = S * ncopy;
  const double* gf = a.f64 + (size_t)env * a.L.f64_per_env;
  const int32_t* gq = a.i32 + (size_t)env * a.L.i32_per_env;
  const int tid = threadIdx.x, lane = tid & 63;
#ifdef MOOG_RASTER_PROFILE   // tools/raster_profile.sh: phase clocks / work counters instead of a frame
  const bool clk = (a.debug_stop == 11);
  const long long T0 = clk ? clock64() : 0;
  long long T1 = 0, T2 = 0, T3 = 0, T4 = 0, T5 = 0, T6 = 0, TA = 0, TB = 0;
#define R_CLK(t) if (clk) t = clock64()
#else
#define R_CLK(t)
#endif

  const RPlan& pl = a.plan;
  REdge* edges = reinterpret_cast<REdge*>(moog_lds + pl.o_edge);
  short2* ivert = reinterpret_cast<short2*>(moog_lds + pl.o_ivert);
  unsigned* longlist = reinterpret_cast<unsigned*>(moog_lds + pl.o_long);
  unsigned* list = reinterpret_cast<unsigned*>(moog_lds + pl.o_list);
  float* xxs = reinterpret_cast<float*>(moog_lds + pl.o_xx);
  int* pbase = reinterpret_cast<int*>(moog_lds + pl.o_pbase);
  int* item_y = reinterpret_cast<int*>(moog_lds + pl.o_item_y);
  unsigned* item_rgba = reinterpret_cast<unsigned*>(moog_lds + pl.o_item_rgba);
  int* rowbase = reinterpret_cast<int*>(moog_lds + pl.o_rowbase);
  int* rowoff = reinterpret_cast<int*>(moog_lds + pl.o_rowoff);
  unsigned* headmask = reinterpret_cast<unsigned*>(moog_lds + pl.o_head);
  RRow* rows = reinterpret_cast<RRow*>(moog_lds + pl.o_rows);
  unsigned* segitems = reinterpret_cast<unsigned*>(moog_lds + pl.o_seg);
  unsigned short* queue = reinterpret_cast<unsigned short*>(moog_lds + pl.o_queue);
  int* misc = reinterpret_cast<int*>(moog_lds + pl.o_misc);   // [0] list length, [1] long list, [2] generic queue, [3] multi-head queue, [4] very long edges

  // phase 1's first loads go out before the tables are cleared (HBM latency under the clearing)
  unsigned vi_next = 0u;
  double2 v_next = make_double2(0.0, 0.0);
  // static prefix: this thread's share of the comparison with the reference record
  const bool per_env = a.sbg_env_stride != 0;   // the env's own picture, validated by the check launch: nothing to compare here
  const int NS = (a.build || per_env) ? 0 : a.n_static;
  const int NSE = (per_env && !a.build) ? a.n_static : 0;   // slots that are in the env's picture: dead as far as this frame goes
  // ... and whole rounds of them are left out of the per-slot and per-vertex loops (their vertex slots are the record's first)
  const int S0 = NSE & ~63, V0 = NSE == 0 ? 0 : ((int)a.nsv / R_THREADS) * R_THREADS;
  const uint8_t* sbg = a.sbg + (size_t)env * a.sbg_env_stride;
  bool st_bad = false;
  if (V0 + tid < TOTV) {
    vi_next = a.vinfo[V0 + tid];
    v_next = *reinterpret_cast<const double2*>(gf + a.L.o_verts + 2 * (V0 + tid));
  }
  // ---- 0: clear the per-item tables and the row records ---------------------------------
  for (int i = tid; i < items; i += R_THREADS) { item_y[2 * i] = 0x7fffffff; item_y[2 * i + 1] = -0x7fffffff; }
  for (int i = tid; i < items * hwords; i += R_THREADS) headmask[i] = 0u;
  for (int i = tid; i < cap_rows; i += R_THREADS) {
    uint4* r = reinterpret_cast<uint4*>(rows + i);
    r[0] = make_uint4(~0u, ~0u, ~0u, ~0u);
    r[1] = make_uint4(~0u, ~0u, 0u, 0u);
  }
  if (tid < 8) misc[tid] = 0;   // ([5]: some thread found the static prefix different from the reference)
  // per-sprite colour (the last wave: it has the fewest vertices to convert)
  for (int s = tid; s < S0; s += R_THREADS) {   // (slots of the env's picture: no vertices, no colour)
    pbase[s] = P->slot_voff[s];
    for (int c = 0; c < ncopy; ++c) item_rgba[s * ncopy + c] = 0u;
  }
  for (int s = S0 + tid - (R_THREADS - 64); s >= S0 && s < S; s += 64) {
    // (every load of the slot goes out at once: one trip to HBM, not one per dependent step)
    const int flags = gq[a.L.o_flags + s];
    const int nvs = gq[a.L.o_nverts + s], opa = gq[a.L.o_opacity + s];
    const double* col = gf + a.L.o_color + 3 * s;
    const double c0 = col[0], c1 = col[1], c2 = col[2];
    const bool alive = (flags & MOOG_F_ALIVE) != 0 && !(a.build && s >= a.n_static) && s >= NSE;
    if (s < NS) {
      const double* rc = a.sref_col + 3 * s;
      st_bad = st_bad || ((flags ^ a.sref_flags[s]) & MOOG_F_ALIVE) != 0 || nvs != a.sref_nv[s] || opa != a.sref_opa[s] ||
               __double_as_longlong(c0) != __double_as_longlong(rc[0]) ||
               __double_as_longlong(c1) != __double_as_longlong(rc[1]) ||
               __double_as_longlong(c2) != __double_as_longlong(rc[2]);
    }
    // first edge record of the slot | live vertex count << 20 (0 for a dead sprite): the later
    // phases take both from LDS instead of chasing the record's flag words through HBM
    // A canvas cut into tiles: a sprite whose bounding circle (position, _max_radius: the bound overlaps_sprite
    // itself relies on) misses this tile by more than two pixels has no pixel in it and is left out of the
    // tile's vertex, edge and row work.  (NaN coordinates fail every comparison and stay in.)
    bool in_tile = true;
    if (tiles > 1 && alive && !torus && P->render.polymod != MOOG_POLYMOD_FIRST_PERSON) {
      const double px = gf[a.L.o_pos + 2 * s], py = gf[a.L.o_pos + 2 * s + 1], rad = gf[a.L.o_maxr + s];
      const double x0 = (px - rad) * (double)a.scale_w - 2.0, x1 = (px + rad) * (double)a.scale_w + 2.0;
      const double y0 = (py - rad) * (double)H - 2.0, y1 = (py + rad) * (double)H + 2.0;
      if (x1 < (double)xoff || x0 > (double)(xoff + W) || y1 < (double)yb0 || y0 > (double)yb1) in_tile = false;
    }
    pbase[s] = P->slot_voff[s] | (((alive && in_tile) ? nvs : 0) << 20);
    unsigned rgba = 0u;
    if (alive) {
      unsigned r8, g8, b8;
      if (a.rgb_override) {   // the host evaluated PILRenderer(color_to_rgb=<a callable>) for this sprite's colour
        const unsigned o = a.rgb_override[(size_t)env * S + s];
        r8 = o & 255u; g8 = (o >> 8) & 255u; b8 = (o >> 16) & 255u;
      } else if (P->render.cmap == MOOG_CMAP_HSV) hsv_to_rgb_u8(c0, c1, c2, r8, g8, b8);
      else { r8 = (unsigned)(int)c0 & 255u; g8 = (unsigned)(int)c1 & 255u; b8 = (unsigned)(int)c2 & 255u; }
      rgba = r8 | (g8 << 8) | (b8 << 16) | (((unsigned)opa & 255u) << 24);
    }
    for (int c = 0; c < ncopy; ++c) item_rgba[s * ncopy + c] = rgba;
  }
  // FirstPersonAgent (polygon_modifiers.py:41-64): every polygon is translated so that the
  // agent layer's first sprite sits at (0.5, 0.5)
  const bool first_person = (P->render.polymod == MOOG_POLYMOD_FIRST_PERSON);
  double fpx = 0, fpy = 0;
  if (first_person) {
    int l = P->render.polymod_layer;
    for (int s = P->layer_slot0[l]; s < P->layer_slot0[l] + P->layer_nslots[l]; ++s)
      if (gq[a.L.o_flags + s] & MOOG_F_ALIVE) {
        fpx = 0.5 - gf[a.L.o_pos + 2 * s]; fpy = 0.5 - gf[a.L.o_pos + 2 * s + 1];
        break;
      }
  }
  __syncthreads();
  if (a.debug_stop == 1) return;

  // ---- 1: vertices -> integer canvas coordinates; item row ranges ----------------------
  unsigned vi_keep0 = 0u, vi_keep1 = 0u;   // the first two rounds' table entries, reused by phase 2
  for (int idx = V0 + tid; idx < TOTV; idx += R_THREADS) {
    const unsigned vi = vi_next;
    const double2 v = v_next;
    if (idx == V0 + tid) vi_keep0 = vi; else if (idx == V0 + tid + R_THREADS) vi_keep1 = vi;
    if (idx + R_THREADS < TOTV) {   // the next round's loads
      vi_next = a.vinfo[idx + R_THREADS];
      v_next = *reinterpret_cast<const double2*>(gf + a.L.o_verts + 2 * (idx + R_THREADS));
    }
    int s = vi & 0xffu, k = (vi >> 8) & 0xffu;
    if (k >= (pbase[s] >> 20)) continue;
    if (idx < a.nsv && NS > 0) {   // (the reference is the same for every frame: cache resident)
      const double2 r = *reinterpret_cast<const double2*>(a.sref_v + 2 * idx);
      st_bad = st_bad || __double_as_longlong(v.x) != __double_as_longlong(r.x) ||
               __double_as_longlong(v.y) != __double_as_longlong(r.y);
    }
    for (int c = 0; c < ncopy; ++c) {
      double vx = v.x, vy = v.y;
      if (torus) { vx = vx + (double)(c / 3 - 1); vy = vy + (double)(c % 3 - 1); }
      if (first_person) { vx = vx + fpx; vy = vy + fpy; }
      int ix = pil_int((double)a.scale_w * vx), iy = pil_int((double)H * vy);
      short2 o; o.x = clamp16(ix); o.y = clamp16(iy);
      ivert[c * TOTV + idx] = o;
      int it = s * ncopy + c;
      atomicMin(&item_y[2 * it], (int)o.y);
      atomicMax(&item_y[2 * it + 1], (int)o.y);
    }
  }
  if (st_bad) misc[5] = 1;   // (cleared before the previous barrier)
  __syncthreads();
  if (a.debug_stop == 2) return;
  // slots below s_lo are already in the cached picture
  const int s_lo = per_env ? NSE : ((NS > 0 && misc[5] == 0) ? NS : 0);

  // ---- 2b: the edge leaving every vertex (ImagingDrawPolygon: add_edge + merge of
  //          horizontal runs); table edges and horizontal heads join the compact list
  for (int c = 0; c < ncopy; ++c) {
    for (int base0 = V0; base0 < TOTV; base0 += R_THREADS) {
      int idx = base0 + tid;
      int kind = 0;   // 1 table edge, 2 horizontal head
      unsigned vi = 0u;
      if (idx < TOTV) vi = base0 == V0 ? vi_keep0 : (base0 == V0 + R_THREADS ? vi_keep1 : a.vinfo[idx]);
      if (idx < TOTV) {
        int s = vi & 0xffu, k = (vi >> 8) & 0xffu;
        int nv = pbase[s] >> 20;
        if (k < nv && s >= s_lo) {
          const short2* pv = ivert + c * TOTV + (idx - k);
          int k2 = (k + 1 == nv) ? 0 : k + 1;
          short2 p0 = pv[k], p1 = pv[k2];
          bool closing = (k == nv - 1);
          REdge E;
          E.x0 = p0.x; E.y0 = p0.y; E.x1 = p1.x; E.y1 = p1.y; E.dx = 0.0f; E.vtop = R_NONE; E.vbot = R_NONE;
          if (p0.y != p1.y) {
            kind = 1;
            E.dx = ((float)(p1.x - p0.x)) / (float)(p1.y - p0.y);
          } else if (!(closing && p0.x == p1.x)) {   // last == first: no closing edge
            bool absorbed = false;
            if (k >= 1 && !closing) {
              short2 pp = pv[k - 1];
              if (pp.y == p0.y) absorbed = (p1.x > p0.x && p0.x > pp.x) || (p1.x < p0.x && p0.x < pp.x);
              // Three equal vertices in a row (tiny circles): this zero-length head repeats the one
              // before it, which is visited immediately before it with the same result either way
              // (drawn: x_pos has passed it; not drawn: same x_pos, same decision) -- drop it.
              if (pp.x == p0.x && pp.y == p0.y && p1.x == p0.x) absorbed = true;
            }
            if (!absorbed) {
              // extend over the following absorbed edges (never the closing edge)
              kind = 2;
              short hx = p1.x;
              int q = k + 1;
              short2 prev = p0, cur = p1;
              while (q <= nv - 2) {
                short2 nxt = pv[q + 1];
                bool ab = (cur.y == nxt.y) && (prev.y == cur.y) &&
                          ((nxt.x > cur.x && cur.x > prev.x) || (nxt.x < cur.x && cur.x < prev.x));
                if (!ab) break;
                hx = nxt.x; prev = cur; cur = nxt; ++q;
              }
              short xmin = p0.x < hx ? p0.x : hx, xmax = p0.x < hx ? hx : p0.x;
              E.dx = __int_as_float((int)((unsigned)(unsigned short)xmin | ((unsigned)(unsigned short)xmax << 16)));
              atomicOr(&headmask[(s * ncopy + c) * hwords + (k >> 5)], 1u << (k & 31));
            }
          }
          edges[c * TOTV + idx] = E;
        }
      }
      unsigned long long m = __ballot(kind != 0);
      if (m) {
        int pos0 = 0;
        if (lane == 0) pos0 = atomicAdd(&misc[0], __popcll(m));
        pos0 = __shfl(pos0, 0);
        if (kind) list[pos0 + __popcll(m & ((1ull << lane) - 1ull))] =
            (vi & 0xffffu) | ((unsigned)c << 16) | (kind == 2 ? 0x80000000u : 0u);
      }
    }
  }
  // ---- 2a: exclusive scan of the clamped row counts of all items (the last wave, whose second
  //          round of edges is nearly empty; needed after the barrier only) ----------------------
  if (tid >= R_THREADS - 64) {
    const int tid = lane;
    int run = 0;
    for (int i0 = 0; i0 < items; i0 += 64) {
      int it = i0 + tid;
      int cnt = 0, ystart = 0;
      if (it < items) {
        int y0 = item_y[2 * it], y1 = item_y[2 * it + 1];
        if (y0 < yb0) y0 = yb0;
        if (y1 > yb1 - 1) y1 = yb1 - 1;   // rows >= H draw nothing (hline clips); other bands have their own workgroups
        cnt = (y1 >= y0 && it >= s_lo * ncopy) ? (y1 - y0 + 1) : 0;
        ystart = y0;
      }
      int inc = cnt;
      for (int o = 1; o < 64; o <<= 1) {
        int t = __shfl_up(inc, o);
        if (tid >= o) inc += t;
      }
      if (it < items) { rowoff[it] = run + inc - cnt; rowbase[it] = run + inc - cnt - ystart; }
      run += __shfl(inc, 63);
    }
    if (tid == 0) rowoff[items] = run;
  }
  __syncthreads();
  if (a.debug_stop == 3) return;
  R_CLK(T1);

  const int nlist = misc[0];
  const int nvtot = TOTV * ncopy;
  const int segs = (yb1 - yb0) * nseg;   // 16-pixel row segments of this tile
  const unsigned bgx = ((unsigned)P->render.bg[0] & 255u) | (((unsigned)P->render.bg[1] & 255u) << 8) |
                       (((unsigned)P->render.bg[2] & 255u) << 16);

  // passes: as many whole items as fit in the row records (cap_rows >= H); with more
  // than one pass the partially composed frame round-trips through `out` (L2)
  for (int base = 0;;) {
    const int r0 = rowoff[base];
    int lo = base + 1, hi = items;   // largest end with rowoff[end] - r0 <= cap_rows
    if (rowoff[items] - r0 <= cap_rows) lo = hi;
    while (lo < hi) {
      int mid = (lo + hi + 1) >> 1;
      if (rowoff[mid] - r0 <= cap_rows) lo = mid; else hi = mid - 1;
    }
    const int end = lo;
    const int total_rows = rowoff[end] - r0;

    // ---- 3a: fix-up partners, then every listed edge pushes its first rows ---------------
    for (int ei = tid; ei < nlist; ei += R_THREADS) {
      unsigned entry = list[ei];   // slot | index within the sprite << 8 | copy << 16 | head << 31
      const int s = entry & 0xffu, k = (entry >> 8) & 0xffu, c = (entry >> 16) & 0xfu;
      const int idx = (pbase[s] & 0xfffff) + k;
      int g = s * ncopy + c;
      REdge E = edges[c * TOTV + idx];
      const int rb = rowbase[g] - r0;
      if (entry >> 31) {
        if (g < base || g >= end) continue;
        int y = E.y0;
        if (y >= yb0 && y < yb1) {
          RRow* r = rows + (rb + y);
          unsigned old = atomicOr(&r->hbits, 1u << (k & 31));
          atomicOr(&r->cnt, (unsigned)(g + 1) << R_ITEM_SHIFT);
          // the row's second head queues it (once); with more than 32 edges per polygon a bit
          // stands for several edges, so the first head does
          if (hwords > 1 ? old == 0u : (old != 0u && (old & (old - 1u)) == 0u))
            queue[2 * cap_rows - 1 - atomicAdd(&misc[3], 1)] = (unsigned short)(rb + y);
        }
        continue;
      }
      const int iymax = item_y[2 * g + 1];
      const int pymax = iymax > H ? H : iymax;    // polygon_generic clamps ymax to ysize
      const int emin = E.y0 < E.y1 ? E.y0 : E.y1, emax = E.y0 < E.y1 ? E.y1 : E.y0;
      short vtop = E.vtop, vbot = E.vbot;
      if (base == 0) {   // every edge of the frame comes by in the first pass; later passes reuse the record
        tip_replacements(edges + c * TOTV + (idx - k), reinterpret_cast<const unsigned*>(ivert + c * TOTV + (idx - k)),
                         k, E, emin >= 0 && emin < H, emax < H && emax >= pymax, &vtop, &vbot);
        edges[c * TOTV + idx].vtop = vtop; edges[c * TOTV + idx].vbot = vbot;
      }
      if (g < base || g >= end) continue;
      R_CLK(TA);
      const int ya = emin < yb0 ? yb0 : emin, yb = emax > yb1 - 1 ? yb1 - 1 : emax;
      RPush pp[4];
#pragma unroll
      for (int j = 0; j < 4; ++j) pp[j] = push_prepare(rows, rb, E, emin, emax, pymax, vtop, vbot, ya + j, ya + j <= yb);
#pragma unroll
      for (int j = 0; j < 4; ++j)   // (rows the edge does not have add 0 to a spare word: no branch, one wait for all four)
        pp[j].pos = atomicAdd(pp[j].on ? &pp[j].r->cnt : reinterpret_cast<unsigned*>(misc + 7), pp[j].on ? pp[j].n : 0u);
#pragma unroll
      for (int j = 0; j < 4; ++j) push_commit(pp[j], E, rb, ya + j, g, queue, misc);
      R_CLK(TB);
      if (yb - ya >= 4) {
        if (yb - ya >= 12) longlist[nvtot - 1 - atomicAdd(&misc[4], 1)] = entry;   // very long: from the back
        else longlist[atomicAdd(&misc[1], 1)] = entry;
      }
    }
    R_CLK(T2);
    __syncthreads();
    // ---- 3b: the remaining rows of long edges: eight lanes per edge with up to 12 rows, a
    //          whole wave per longer edge (walls)
    for (int i = tid; i < segs * iwords; i += R_THREADS) segitems[i] = 0u;   // (aliases the integer vertices)
    {
      const int nlong = misc[1], nvlong = misc[4], sub = tid & 7;
      const int ngroups = (nlong + 7) >> 3;   // waves' worth of 8-lane groups come first, then whole-wave edges
      for (int u = tid >> 6; u < ngroups + nvlong; u += R_THREADS / 64) {
        const bool vl = (u >= ngroups);
        const int li = vl ? u - ngroups : u * 8 + (lane >> 3);
        if (!vl && li >= nlong) continue;
        unsigned entry = vl ? longlist[nvtot - 1 - li] : longlist[li];
        const int s = entry & 0xffu, c = (entry >> 16) & 0xfu;
        const int idx = (pbase[s] & 0xfffff) + ((entry >> 8) & 0xffu);
        int g = s * ncopy + c;
        REdge E = edges[c * TOTV + idx];
        const int rb = rowbase[g] - r0;
        const int iymax = item_y[2 * g + 1];
        const int pymax = iymax > H ? H : iymax;
        const int emin = E.y0 < E.y1 ? E.y0 : E.y1, emax = E.y0 < E.y1 ? E.y1 : E.y0;
        const int ya = emin < yb0 ? yb0 : emin, yb = emax > yb1 - 1 ? yb1 - 1 : emax;
        if (vl) {
          for (int y = ya + 4 + lane; y <= yb; y += 64)
            push_crossing(rows, rb, E, emin, emax, pymax, E.vtop, E.vbot, g, y, queue, misc);
        } else if (ya + 4 + sub <= yb) {
          push_crossing(rows, rb, E, emin, emax, pymax, E.vtop, E.vbot, g, ya + 4 + sub, queue, misc);
        }
      }
    }
    __syncthreads();
    if (a.debug_stop == 4) return;
    R_CLK(T3);

    // ---- 4: coverage masks, one thread per (item, row) ------------------------------------
    // A handful of multi-head rows (below) is one long dependent chain on a single wave: then the
    // last wave does only those, beside the other three waves' main loop.
    const bool spare_wave = misc[3] > 0 && misc[3] <= 64;
    const int main_threads = spare_wave ? R_THREADS - 64 : R_THREADS;
    for (int w0 = (tid < main_threads) ? 0 : total_rows; w0 < total_rows; w0 += main_threads) {
      const int w = w0 + tid;
      unsigned k[16];
      int cnt = 0, g = -1, hxmin = 0, hxmax = 0;
      unsigned hbits = 0u;
      bool head = false, slow = false, multi = false;
      if (w < total_rows) {
        const uint4* rr = reinterpret_cast<const uint4*>(rows + w);
        uint4 q0 = rr[0], q1 = rr[1];
        k[0] = q0.x & 0xffffu; k[1] = q0.x >> 16; k[2] = q0.y & 0xffffu; k[3] = q0.y >> 16;
        k[4] = q0.z & 0xffffu; k[5] = q0.z >> 16; k[6] = q0.w & 0xffffu; k[7] = q0.w >> 16;
        k[8] = q1.x & 0xffffu; k[9] = q1.x >> 16; k[10] = q1.y & 0xffffu; k[11] = q1.y >> 16;
        hbits = q1.z;
        unsigned cw = q1.w;
        g = (int)(cw >> R_ITEM_SHIFT) - 1;
        cnt = cw & R_CNT_MASK;
        slow = (cw & R_GENERIC) != 0u;
        head = !slow && hwords == 1 && hbits != 0u && (hbits & (hbits - 1u)) == 0u;
        multi = !slow && hbits != 0u && !head;
      } else {
#pragma unroll
        for (int q = 0; q < R_CAP; ++q) k[q] = 0xffffu;
      }
#pragma unroll
      for (int q = R_CAP; q < 16; ++q) k[q] = 0xffffffffu;
      const bool live = (g >= 0);
      if (slow || multi) cnt = 0;
      const int y = live ? w + r0 - rowbase[g] : 0;
      if (head) {   // the row's one head: its record
        const int s = torus ? g / 9 : g, c = g - s * ncopy;
        unsigned xb = (unsigned)__float_as_int(edges[c * TOTV + (pbase[s] & 0xfffff) + __ffs((int)hbits) - 1].dx);
        hxmin = (short)(xb & 0xffffu); hxmax = (short)(xb >> 16);
      }
      RMask m;
      if (__any(cnt > 8)) {
        sort_network<16>(k);
        m = span_loop<R_CAP / 2, 16, WORDS>(k, cnt, head, hxmin, hxmax, W, xoff);
      } else {
        unsigned k8[8] = {k[0], k[1], k[2], k[3], k[4], k[5], k[6], k[7]};
        sort_network<8>(k8);
        m = span_loop<4, 8, WORDS>(k8, cnt, head, hxmin, hxmax, W, xoff);
      }
      if (w < total_rows) {
        if (!slow && !multi) {   // (those rows were queued by the push phase)
          unsigned long long* mp = reinterpret_cast<unsigned long long*>(rows + w);
          mp[0] = m.w0;
          if (words > 1) mp[1] = m.w1;
          if (live) {
            for (int sg = 0; sg < nseg; ++sg) {
              unsigned long long mw = (sg < 4) ? m.w0 : m.w1;
              if ((mw >> ((sg & 3) * 16)) & 0xffffull)
                atomicOr(&segitems[((y - yb0) * nseg + sg) * iwords + (g >> 5)], 1u << (g & 31));
            }
          }
        }
      }
    }
    // The rare rows, beside the main loop's tail (no barrier in between: their queues were
    // complete before phase 4): rows with several horizontal heads, handed out from the last
    // wave backwards (same keys, heads through the row's head bits); rows for the generic
    // scanline on the third wave.
    if (a.debug_stop == 6) { __syncthreads(); return; }
    R_CLK(T4);
    {
      const int nmulti = misc[3];
      for (int qi = R_THREADS - 1 - tid; qi < nmulti; qi += R_THREADS) {   // the last wave first: it is free soonest
        const int w = queue[2 * cap_rows - 1 - qi];
        const uint4* rr = reinterpret_cast<const uint4*>(rows + w);
        uint4 q0 = rr[0], q1 = rr[1];
        if (q1.w & R_GENERIC) continue;   // the generic routine draws this row
        unsigned k[16];
        k[0] = q0.x & 0xffffu; k[1] = q0.x >> 16; k[2] = q0.y & 0xffffu; k[3] = q0.y >> 16;
        k[4] = q0.z & 0xffffu; k[5] = q0.z >> 16; k[6] = q0.w & 0xffffu; k[7] = q0.w >> 16;
        k[8] = q1.x & 0xffffu; k[9] = q1.x >> 16; k[10] = q1.y & 0xffffu; k[11] = q1.y >> 16;
#pragma unroll
        for (int q = R_CAP; q < 16; ++q) k[q] = 0xffffffffu;
        const int cnt = q1.w & R_CNT_MASK, g = (int)(q1.w >> R_ITEM_SHIFT) - 1;
        const int s = torus ? g / 9 : g, c = g - s * ncopy;
        const int y = w + r0 - rowbase[g];
        const unsigned hbits = q1.z;   // the heads on this row (the polygon has <= 32 edges)
        const REdge* pe = edges + c * TOTV + (pbase[s] & 0xfffff);
        RMask m;
        if (hwords > 1) {
          sort_network<16>(k);
          RPoly poly = {pe, 32, headmask + g * hwords, hwords, hbits};
          m = span_loop_poly<R_CAP / 2, 16>(k, cnt, poly, y, W, xoff);
        } else if (__any(cnt > 8)) {
          sort_network<16>(k);
          m = span_loop_pending<R_CAP / 2, 16, WORDS>(k, cnt, pe, hbits, W, xoff);
        } else {   // this is one wave's dependent chain: the short network when it suffices
          unsigned k8[8] = {k[0], k[1], k[2], k[3], k[4], k[5], k[6], k[7]};
          sort_network<8>(k8);
          m = span_loop_pending<4, 8, WORDS>(k8, cnt, pe, hbits, W, xoff);
        }
        unsigned long long* mp = reinterpret_cast<unsigned long long*>(rows + w);
        mp[0] = m.w0;
        if (words > 1) mp[1] = m.w1;
        for (int sg = 0; sg < nseg; ++sg) {
          unsigned long long mw = (sg < 4) ? m.w0 : m.w1;
          if ((mw >> ((sg & 3) * 16)) & 0xffffull)
            atomicOr(&segitems[((y - yb0) * nseg + sg) * iwords + (g >> 5)], 1u << (g & 31));
        }
      }
    }
    if (tid >= R_THREADS - 128 && tid < R_THREADS - 64 && a.debug_stop != 7) {
      {
        const int qn = misc[2], t = lane < R_SLOW ? lane : qn;
        for (int qi = t; qi < qn; qi += R_SLOW) {
          int w = queue[qi];
          int g = (int)(rows[w].cnt >> R_ITEM_SHIFT) - 1;
          int s = torus ? g / 9 : g, c = g - s * ncopy;
          int y = w + r0 - rowbase[g];
          int iymax = item_y[2 * g + 1];
          int pymax = iymax > H ? H : iymax;
          RPoly poly = {edges + c * TOTV + (pbase[s] & 0xfffff), pbase[s] >> 20, headmask + g * hwords, hwords, ~0u};
          RMask m = scanline_mask_generic(poly, y, pymax, xxs + t, W, xoff, a.xxcap);
          unsigned long long* mp = reinterpret_cast<unsigned long long*>(rows + w);
          mp[0] = m.w0;
          if (words > 1) mp[1] = m.w1;
          for (int sg = 0; sg < nseg; ++sg) {
            unsigned long long mw = (sg < 4) ? m.w0 : m.w1;
            if ((mw >> ((sg & 3) * 16)) & 0xffffull)
              atomicOr(&segitems[((y - yb0) * nseg + sg) * iwords + (g >> 5)], 1u << (g & 31));
          }
        }
      }
    }
    R_CLK(T5);
    __syncthreads();
    R_CLK(T6);
    if (a.debug_stop == 5) return;
#ifdef MOOG_RASTER_PROFILE
    if (clk) {   // phase clocks of waves 0 and 3 instead of a frame
      if (lane == 0 && (tid == 0 || tid == 192)) {
        unsigned* o32 = reinterpret_cast<unsigned*>(a.image + (size_t)env * H * WF * 3) + (tid ? 8 : 0);
        o32[0] = (unsigned)(T1 - T0); o32[1] = (unsigned)(T2 - T1); o32[2] = (unsigned)(T3 - T2); o32[3] = (unsigned)(T4 - T3);
        o32[4] = (unsigned)(T5 - T4); o32[5] = (unsigned)(T6 - T5); o32[6] = (unsigned)(clock64() - T0); o32[7] = (unsigned)(TB - TA);
      }
      return;
    }
    if (a.debug_stop == 10) { uint8_t* out = a.image + (size_t)env * H * WF * 3; if (tid < 8) out[tid] = (uint8_t)(tid == 5 ? (rowoff[items] >> 2) : misc[tid]); return; }   // counters instead of a frame
#endif

    // ---- 5: compose (painter's order = item order), pack RGB, store flipped ------------------
    uint8_t* out = a.image + (size_t)env * H * WF * 3;
    const bool from_cache = (base == 0 && s_lo > 0);
    for (int seg = tid; seg < segs; seg += R_THREADS) {
      const int yr = seg / nseg, sg = seg - yr * nseg, x0 = sg * 16, y = yb0 + yr;
      uint4* dst = reinterpret_cast<uint4*>(out + ((size_t)(a.flip ? H - 1 - y : y) * WF + xoff + x0) * 3);
      if (from_cache) {   // a segment no sprite touches is a copy of the cached picture
        bool any = false;
        for (int iw = 0; iw < iwords; ++iw) any = any || segitems[seg * iwords + iw] != 0u;
        if (!any) {
          const uint4* src = reinterpret_cast<const uint4*>(sbg + ((size_t)(a.flip ? H - 1 - y : y) * WF + xoff + x0) * 3);
          const uint4 c0 = src[0], c1 = src[1], c2 = src[2];
          dst[0] = c0; dst[1] = c1; dst[2] = c2;
          continue;
        }
      }
      unsigned px[16];
      if (base == 0 && !from_cache) {
#pragma unroll
        for (int i = 0; i < 16; ++i) px[i] = bgx;
      } else {  // continue from the previous pass, or from the cached picture of the static prefix
        const uint4* src = from_cache ? reinterpret_cast<const uint4*>(sbg + ((size_t)(a.flip ? H - 1 - y : y) * WF + xoff + x0) * 3) : dst;
        uint4 q0 = src[0], q1 = src[1], q2 = src[2];
        unsigned d[12] = {q0.x, q0.y, q0.z, q0.w, q1.x, q1.y, q1.z, q1.w, q2.x, q2.y, q2.z, q2.w};
#pragma unroll
        for (int q = 0; q < 4; ++q) {
          px[4 * q] = d[3 * q] & 0xFFFFFFu;
          px[4 * q + 1] = (d[3 * q] >> 24) | ((d[3 * q + 1] & 0xFFFFu) << 8);
          px[4 * q + 2] = (d[3 * q + 1] >> 16) | ((d[3 * q + 2] & 0xFFu) << 16);
          px[4 * q + 3] = d[3 * q + 2] >> 8;
        }
      }
      for (int iw = 0; iw < iwords; ++iw) {
        unsigned bitsw = segitems[seg * iwords + iw];
        while (bitsw) {
          int g = iw * 32 + __ffs((int)bitsw) - 1;
          bitsw &= bitsw - 1;
          const unsigned* mrow = reinterpret_cast<const unsigned*>(rows + (rowbase[g] - r0 + y));
          unsigned bits = (mrow[x0 >> 5] >> (x0 & 31)) & 0xFFFFu;
          unsigned rgba = item_rgba[g];
          unsigned al = rgba >> 24;
          if (al == 255u) {
            unsigned fg = rgba & 0xFFFFFFu;
#pragma unroll
            for (int i = 0; i < 16; ++i) px[i] = (bits & (1u << i)) ? fg : px[i];
          } else {
            unsigned f0 = rgba & 255u, f1 = (rgba >> 8) & 255u, f2 = (rgba >> 16) & 255u;
#pragma unroll
            for (int i = 0; i < 16; ++i) {
              if (bits & (1u << i)) {
                unsigned o = px[i];
                px[i] = blend8(o & 255u, f0, al) | (blend8((o >> 8) & 255u, f1, al) << 8) |
                        (blend8((o >> 16) & 255u, f2, al) << 16);
              }
            }
          }
        }
      }
      unsigned d[12];
#pragma unroll
      for (int q = 0; q < 4; ++q) {   // 4 pixels (RGBX) -> 3 dwords (RGB)
        unsigned p0 = px[4 * q], p1 = px[4 * q + 1], p2 = px[4 * q + 2], p3 = px[4 * q + 3];
        d[3 * q] = (p0 & 0xFFFFFFu) | (p1 << 24);
        d[3 * q + 1] = ((p1 >> 8) & 0xFFFFu) | (p2 << 16);
        d[3 * q + 2] = ((p2 >> 16) & 0xFFu) | (p3 << 8);
      }
      dst[0] = make_uint4(d[0], d[1], d[2], d[3]);
      dst[1] = make_uint4(d[4], d[5], d[6], d[7]);
      dst[2] = make_uint4(d[8], d[9], d[10], d[11]);
    }
    if (end >= items) break;
    base = end;
    __syncthreads();
    r_next_pass(rows, cap_rows, misc, tid);
    __syncthreads();
  }
}

#endif  // MOOG_RASTER_KERNEL_H_
