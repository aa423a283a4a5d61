// moog_raster.h -- HIP scanline polygon rasteriser (gfx950), bit-exact with
// Pillow's ImageDraw.polygon in RGBA blend mode as used by PILRenderer
// (reference moog/observers/pil_renderer.py:88-120; Pillow Draw.c
// ImagingDrawPolygon / polygon_generic(hasAlpha=1) / hline32rgba as restated and
// fuzz-validated against Pillow 12.2.0 in oracle/moog_oracle.c).
//
// One 256-thread workgroup renders one env's frame.  The kernel is bound by LDS /
// issue latency of short dependent chains, so the design goal is occupancy: the LDS
// working set is ~15 KB per env (integer vertices, 16-byte edge records, row masks;
// no per-thread scratch), which lets 8 workgroups = 32 waves share a CU.
// Everything between reading the sprite vertices (coalesced 16 B/lane) and
// writing the uint8 frame (each byte written exactly once, 3 x dwordx4 per lane
// over contiguous 3 KB spans per wave) stays in LDS / registers:
//   1  vertices -> integer canvas coordinates ((int)(W*x), one thread per vertex
//      per polygon copy), per-item row ranges by LDS atomics, per-item RGBA
//   2  one thread per edge: slope, horizontal-run merging (ImagingDrawPolygon)
//   3  compact work list of (item, row) pairs that actually intersect the canvas
//   4  one thread per (item, row): Pillow's scanline -> 64/128-bit coverage mask
//   5  one thread per 16-pixel row segment: compose the covering items in
//      painter's order (one blend per covered pixel), RGBX in registers, pack
//      to RGB and store to the flipped row (np.flipud)
// HBM-bound by construction: algorithmic bytes = H*W*3 + live vertices * 16.
#pragma once
#include "moog_device.h"

#define R_THREADS 256
#define R_SLOW 16     // lanes that run the generic scanline concurrently (bounds its LDS scratch)

struct RArgs {
  const moog_program_t* P;
  moog_layout_t L;
  const double* f64;
  const int32_t* i32;
  uint8_t* image;
  const int16_t* vslot;
  int32_t n_envs;
  int32_t chunk;       // row capacity of the coverage-mask buffer (rows per pass)
  int32_t words;       // 64-bit words per row mask
  int32_t iwords;      // 32-bit words of the per-row item bitmask
  int32_t max_items;   // S * copies
  int32_t debug_stop;  // >0: return after that phase (profiling aid)
  int32_t xxcap;       // crossing-list capacity of the generic scanline = 2 * max vertices per sprite
};

// LDS plan shared by host (sizes) and device (carve-up)
struct RPlan {
  size_t o_ivert, o_edge, o_eflag, o_slotinfo, o_item_slot, o_item_y, o_item_rgba, o_item_cnt,
      o_rowoff, o_rowitems, o_masks, o_xx, o_frame, o_misc, o_carry, o_queue, total;
};

// Edge record, 16 bytes.  Table (non-horizontal) edges: x0, y0, y1, dx.
// Horizontal heads: x0 = xmin, y0 = y, y1 = xmax (dx unused).  Per polygon the
// table edges are packed from the front of its region (in edge order), the
// horizontal heads from the back (in edge order going backwards).
struct REdge { short x0, y0, y1, x1; float dx; float pad; };

__host__ __device__ inline size_t r_align(size_t x) { return (x + 15) & ~(size_t)15; }

__host__ __device__ inline void raster_plan(int S, int TOTV, int ncopy, int W, int H, int cap_rows,
                                            int words, int iwords, int xxcap, RPlan* p) {
  size_t o = 0;
  size_t nv = (size_t)TOTV * ncopy, items = (size_t)S * ncopy;
  p->o_edge = o; o = r_align(o + nv * sizeof(REdge)); // packed edge records
  p->o_slotinfo = o; o = r_align(o + (size_t)S * 16); // per slot: rank, nverts, vertex offset, -
  p->o_item_slot = o; o = r_align(o + items * 4);     // slot | copy << 16
  p->o_item_y = o; o = r_align(o + items * 8);        // ymin, ymax (ints, atomics)
  p->o_item_rgba = o; o = r_align(o + items * 4);
  p->o_item_cnt = o; o = r_align(o + items * 4);      // n_table | n_heads << 16
  p->o_rowoff = o; o = r_align(o + (items + 1) * 4);
  p->o_rowitems = o; o = r_align(o + (size_t)H * iwords * 4);
  p->o_misc = o; o = r_align(o + 64);
  p->o_carry = o; o = r_align(o + (size_t)S * 16);    // counts of a polygon's earlier 64-vertex chunks
  p->o_queue = o; o = r_align(o + (size_t)cap_rows * 2 + 16);   // rows that need the generic scanline
  p->o_xx = o; o = r_align(o + (size_t)xxcap * R_SLOW * 4);
  // union: the integer vertices are dead once the edge records are packed; the
  // coverage masks live only afterwards
  size_t u = o;
  p->o_ivert = u; size_t e1 = r_align(u + nv * 4);
  p->o_eflag = e1;
  p->o_masks = u; size_t e2 = r_align(u + (size_t)cap_rows * words * 8);
  p->o_frame = 0;
  p->total = e1 > e2 ? e1 : e2;
}

// Draw.c ROUND_UP / ROUND_DOWN: sign-symmetric, so branch-free with copysign
__device__ __forceinline__ int pil_round_up(float f) {
  return (int)copysignf(floorf(fabsf(f) + 0.5f), f);
}
__device__ __forceinline__ int pil_round_down(float f) {
  return (int)copysignf(ceilf(fabsf(f) - 0.5f), f);
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

__device__ inline short clamp16(int v) { return (short)(v < -32000 ? -32000 : (v > 32000 ? 32000 : v)); }

struct RMask { unsigned long long w0, w1; };

__device__ inline void mask_fill(RMask& m, int W, int x0, int x1) {
  if (x0 < 0) x0 = 0; else if (x0 >= W) return;
  if (x1 < 0) return; else if (x1 >= W) x1 = W - 1;
  if (x0 > x1) return;
  // bits [x0, x1] of a 128-bit mask
  if (x0 < 64) {
    int hi = x1 < 63 ? x1 : 63;
    unsigned long long bits = (hi - x0 == 63) ? ~0ull : (((1ull << (hi - x0 + 1)) - 1ull) << x0);
    m.w0 |= bits;
  }
  if (x1 >= 64) {
    int lo = x0 > 64 ? x0 - 64 : 0, hi = x1 - 64;
    unsigned long long bits = (hi - lo == 63) ? ~0ull : (((1ull << (hi - lo + 1)) - 1ull) << lo);
    m.w1 |= bits;
  }
}

// view of one polygon's packed edge records in LDS
struct RPoly {
  const REdge* e;   // region of n records
  int n;            // region size (= vertex count)
  int nt;           // table edges  e[0 .. nt)
  int nh;           // horizontal heads e[n-1], e[n-2], ... (nh of them)
};

// Draw.c draw_horizontal_lines (heads visited in edge order)
__device__ inline void draw_horizontal(const RPoly& p, int y, int* x_pos, RMask& m, int W) {
  for (int i = 0; i < p.nh; ++i) {
    REdge h = p.e[p.n - 1 - i];
    if (h.y0 != y) continue;
    int xmin = h.x0, xmax = h.y1;
    if (*x_pos != -1 && *x_pos < xmin) continue;
    if (*x_pos > xmin) {
      xmin = *x_pos;
      if (xmax < xmin) continue;
    }
    mask_fill(m, W, xmin, xmax);
    *x_pos = xmax + 1;
  }
}

// polygon_generic's corner fix-up for table edge i on row y (its first row when
// `top`, else its last row): the first earlier table edge k that is active on the
// row, leans the same way, shares the tip and crosses the row at the same x gets
// its entry replaced so that the tip row's span meets the adjacent row's span.
// Returns k (or -1) and the replacement value.
__device__ inline int tip_partner(const REdge* e, int i, bool top, float* vv_out) {
  REdge E = e[i];
  float dx = E.dx;
  if (dx == 0.0f) return -1;
  int y0 = E.y0, y1 = E.y1;
  int y = top ? (y0 < y1 ? y0 : y1) : (y0 < y1 ? y1 : y0);
  float x = (float)(y - y0) * dx + (float)E.x0;
  // An edge's crossing of its own end row is within 1e-3 of that end point, so
  // only edges ending at the same integer point can compare equal below.
  const int tipx = (y == y0) ? E.x0 : E.x1;
  const short* raw = reinterpret_cast<const short*>(e);
  for (int k = 0; k < i; ++k) {
    int ky0 = raw[8 * k + 1], ky1 = raw[8 * k + 2];
    int ktip = top ? (ky0 < ky1 ? ky0 : ky1) : (ky0 < ky1 ? ky1 : ky0);
    if (ktip != y) continue;
    int ktx = (ktip == ky0) ? raw[8 * k] : raw[8 * k + 3];
    if (ktx != tipx) continue;
    REdge K = e[k];
    float kdx = K.dx;
    if ((dx > 0 && kdx <= 0) || (dx < 0 && kdx >= 0)) continue;
    if (x != (float)(y - ky0) * kdx + (float)K.x0) continue;
    int off = top ? 1 : -1;
    float adj = (float)(y + off - y0) * dx + (float)E.x0;
    float adjo = (float)(y + off - ky0) * kdx + (float)K.x0;
    if (adj > x && adjo > x) {
      float vv = (float)(pil_round_up(fminf(adj, adjo)) - 1);
      if (vv > x) { *vv_out = vv; return k; }
    } else if (adj < x && adjo < x) {
      float vv = (float)(pil_round_up(fmaxf(adj, adjo)) + 1);
      if (vv < x) { *vv_out = vv; return k; }
    }
    return -1;   // the reference stops at the first matching edge
  }
  return -1;
}

// Generic scanline (any number of crossings, corner fix-ups): crossing list in LDS.
// xx: this thread's crossing list, element j at xx[j * R_SLOW].
__device__ inline RMask scanline_mask_generic(const RPoly& p, int y, int poly_ymax, float* xx, int W,
                                              const int R_XX) {
  RMask m = {0ull, 0ull};
  int j = 0;
  for (int i = 0; i < p.nt; ++i) {
    REdge E = p.e[i];
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
      // connect discontiguous corners: only a row at an end point of this edge can
      // share a tip; the partner's entry on this row is overwritten
      float vv = 0.0f;
      int kt = tip_partner(p.e, i, y == emin, &vv);
      if (kt >= 0) {
        int kpos = 0;
        for (int k = 0; k < kt; ++k) {
          REdge K = p.e[k];
          int kmin = K.y0 < K.y1 ? K.y0 : K.y1, kmax = K.y0 < K.y1 ? K.y1 : K.y0;
          if (y < kmin || y > kmax) continue;
          kpos += (y == kmax && y < poly_ymax) ? 2 : 1;
        }
        if (kpos < R_XX) xx[kpos * R_SLOW] = vv;
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
    if (p.nh) draw_horizontal(p, y, &x_pos, m, W);
    if (x_end < x_pos) continue;
    int x_start = pil_round_up(xx[(i - 1) * R_SLOW]);
    if (x_pos > x_start) {
      x_start = x_pos;
      if (x_end < x_start) continue;
    }
    mask_fill(m, W, x_start, x_end);
    x_pos = x_end + 1;
  }
  if (p.nh) draw_horizontal(p, y, &x_pos, m, W);
  return m;
}

// Coverage of scanline y of one polygon: polygon_generic(hasAlpha=1), one row.
// Fast path: up to N crossings kept sorted in registers (insertion by a min/max
// chain; the sorted multiset is all the span loop needs).  The edge loop is branch
// free: an inactive edge inserts +inf, which leaves the registers unchanged, so lanes
// working on different polygons do not diverge.  A corner fix-up replaces the
// partner's entry, whose value equals this edge's crossing x, i.e. "remove one x,
// insert vv".  More than N crossings or a second fix-up on the same row (the
// partner's entry might already be modified) report `overflow`.
// BITS: `bits` has bit i set for every table edge i whose row range contains y (built once per
// frame by the edge threads, phase 3b), so the loop visits only the edges that cross the row
// instead of the whole table (a 30-gon has 28 table edges, a row crosses ~6 of them).
template <int N, bool BITS>
__device__ inline RMask scanline_regs(const RPoly& p, int y, int poly_ymax, int W, bool* overflow,
                                      unsigned long long bits) {
  const float INF = __builtin_inff();
  float r[N];
#pragma unroll
  for (int q = 0; q < N; ++q) r[q] = INF;
  int j = 0, nfix = 0;
  for (int i = 0; BITS ? (bits != 0ull) : (i < p.nt); ++i) {
    if (BITS) { i = __ffsll((long long)bits) - 1; bits &= bits - 1ull; }
    REdge E = p.e[i];
    int y0 = E.y0, y1 = E.y1;
    int emin = y0 < y1 ? y0 : y1, emax = y0 < y1 ? y1 : y0;
    bool active = BITS || ((y >= emin) && (y <= emax));
    float x = (float)(y - y0) * E.dx + (float)E.x0;
    bool dup = active && (y == emax) && (y < poly_ymax);
    j += (active ? 1 : 0) + (dup ? 1 : 0);
    float ta = active ? x : INF, td = dup ? x : INF;
#pragma unroll
    for (int q = 0; q < N; ++q) {
      float lo = fminf(r[q], ta); ta = fmaxf(r[q], ta);
      float lo2 = fminf(lo, td); td = fmaxf(lo, td);
      r[q] = lo2;
    }
    int flag = __float_as_int(E.pad);
    if (active && !dup && flag && ((y == emin && (flag & 1)) || (y == emax && (flag & 2)))) {
      float vv = 0.0f;
      int kt = tip_partner(p.e, i, y == emin, &vv);
      if (kt >= 0) {
        ++nfix;
        bool f = false;   // remove one instance of x (the partner's entry) ...
#pragma unroll
        for (int q = 0; q < N - 1; ++q) { f = f || (r[q] == x); r[q] = f ? r[q + 1] : r[q]; }
        r[N - 1] = INF;
        float t = vv;     // ... then insert vv
#pragma unroll
        for (int q = 0; q < N; ++q) { float lo = fminf(r[q], t); t = fmaxf(r[q], t); r[q] = lo; }
      }
    }
  }
  RMask m = {0ull, 0ull};
  *overflow = (nfix > 1 || j > N);
  if (*overflow) return m;
  bool head_here = false;
  for (int i = 0; i < p.nh; ++i) head_here = head_here || (p.e[p.n - 1 - i].y0 == y);
  int x_pos = (j == 0) ? -1 : 0;
#pragma unroll
  for (int q = 0; q < N / 2; ++q) {
    if (2 * q + 1 < j) {
      int x_end = pil_round_down(r[2 * q + 1]);
      if (x_end >= x_pos) {
        if (head_here) draw_horizontal(p, y, &x_pos, m, W);
        if (x_end >= x_pos) {
          int x_start = pil_round_up(r[2 * q]);
          bool skip = false;
          if (x_pos > x_start) { x_start = x_pos; skip = (x_end < x_start); }
          if (!skip) {
            mask_fill(m, W, x_start, x_end);   // empty when x_start > x_end, x_pos still moves
            x_pos = x_end + 1;
          }
        }
      }
    }
  }
  if (head_here) draw_horizontal(p, y, &x_pos, m, W);
  return m;
}

// 8 sorted registers cover ~99 % of the rows; rows with 9..16 crossings (spoked
// shapes) re-run with 16; anything beyond is queued for the generic LDS routine.
template <bool BITS>
__device__ inline RMask scanline_mask(const RPoly& p, int y, int poly_ymax, int W, bool* need_generic,
                                      unsigned long long bits) {
  bool over = false;
  RMask m;
  // A wave whose rows include one with many crossings (the top / bottom rows of a small
  // circle hold a dozen sub-pixel edges) would run the 8-register pass for nothing.
  if (BITS && __any(__popcll(bits) > 6)) {
    m = scanline_regs<16, BITS>(p, y, poly_ymax, W, &over, bits);
  } else {
    m = scanline_regs<8, BITS>(p, y, poly_ymax, W, &over, bits);
    if (over) m = scanline_regs<16, BITS>(p, y, poly_ymax, W, &over, bits);
  }
  *need_generic = over;
  return m;
}

// Draw.c BLEND8 / DIV255 on one channel
__device__ __forceinline__ unsigned blend8(unsigned bg, unsigned fg, unsigned al) {
  unsigned t = bg * (255u - al) + fg * al + 128u;
  return ((t >> 8) + t) >> 8;
}

__global__ __launch_bounds__(R_THREADS) void moog_raster_kernel(RArgs a) {
  const int env = blockIdx.x;
  if (env >= a.n_envs) return;
  PProg P = as_const_prog(a.P);
  const int W = P->render.width, H = P->render.height;
  const int S = P->n_slots, TOTV = a.L.TOTV;
  const bool torus = (P->render.polymod == MOOG_POLYMOD_TORUS);
  const int ncopy = torus ? 9 : 1;
  const int words = a.words, iwords = a.iwords, cap_rows = a.chunk;
  const double* gf = a.f64 + (size_t)env * a.L.f64_per_env;
  const int32_t* gq = a.i32 + (size_t)env * a.L.i32_per_env;
  const int tid = threadIdx.x, lane = tid & 63;

  RPlan pl;
  raster_plan(S, TOTV, ncopy, W, H, cap_rows, words, iwords, a.xxcap, &pl);
  short2* ivert = reinterpret_cast<short2*>(moog_lds + pl.o_ivert);
  REdge* edges = reinterpret_cast<REdge*>(moog_lds + pl.o_edge);
  int* slotinfo = reinterpret_cast<int*>(moog_lds + pl.o_slotinfo);
  int* item_slot = reinterpret_cast<int*>(moog_lds + pl.o_item_slot);
  int* item_y = reinterpret_cast<int*>(moog_lds + pl.o_item_y);
  unsigned* item_rgba = reinterpret_cast<unsigned*>(moog_lds + pl.o_item_rgba);
  int* item_cnt = reinterpret_cast<int*>(moog_lds + pl.o_item_cnt);
  int* rowoff = reinterpret_cast<int*>(moog_lds + pl.o_rowoff);
  unsigned* rowitems = reinterpret_cast<unsigned*>(moog_lds + pl.o_rowitems);
  unsigned long long* masks = reinterpret_cast<unsigned long long*>(moog_lds + pl.o_masks);
  float* xxs = reinterpret_cast<float*>(moog_lds + pl.o_xx);
  int* misc = reinterpret_cast<int*>(moog_lds + pl.o_misc);   // [0] n_live, [1] queue length
  int* carry = reinterpret_cast<int*>(moog_lds + pl.o_carry);
  unsigned short* queue = reinterpret_cast<unsigned short*>(moog_lds + pl.o_queue);

  // ---- 0: live sprites in slot (= layer, list) order; per-sprite colour (wave 0) --------
  if (tid < 64) {
    int n_live = 0;
    for (int s0 = 0; s0 < S; s0 += 64) {
      int s = s0 + tid;
      bool live = false;
      int nv = 0;
      if (s < S) { live = (gq[a.L.o_flags + s] & MOOG_F_ALIVE) != 0; nv = gq[a.L.o_nverts + s]; }
      unsigned long long bal = __ballot(live);
      int rank = n_live + __popcll(bal & ((1ull << tid) - 1ull));
      if (s < S) { slotinfo[4 * s] = live ? rank : -1; slotinfo[4 * s + 1] = nv; slotinfo[4 * s + 2] = P->slot_voff[s]; }
      if (live) {
        unsigned r8, g8, b8;
        const double* col = gf + a.L.o_color + 3 * s;
        if (P->render.cmap == MOOG_CMAP_HSV) hsv_to_rgb_u8(col[0], col[1], col[2], r8, g8, b8);
        else { r8 = (unsigned)(int)col[0] & 255u; g8 = (unsigned)(int)col[1] & 255u; b8 = (unsigned)(int)col[2] & 255u; }
        unsigned a8 = (unsigned)gq[a.L.o_opacity + s] & 255u;
        unsigned rgba = r8 | (g8 << 8) | (b8 << 16) | (a8 << 24);
        for (int c = 0; c < ncopy; ++c) {
          int it = rank * ncopy + c;
          item_slot[it] = s | (c << 16);
          item_rgba[it] = rgba;
          item_y[2 * it] = 0x7fffffff;
          item_y[2 * it + 1] = -0x7fffffff;
        }
      }
      n_live += __popcll(bal);
    }
    if (tid == 0) misc[0] = n_live;
  }
  __syncthreads();
  const int total_items = misc[0] * ncopy;
  if (a.debug_stop == 1) return;

  // FirstPersonAgent (polygon_modifiers.py:41-64): every polygon is translated so that the
  // agent layer's first sprite sits at (0.5, 0.5)
  const bool first_person = (P->render.polymod == MOOG_POLYMOD_FIRST_PERSON);
  double fpx = 0, fpy = 0;
  if (first_person) {
    int l = P->render.polymod_layer;
    for (int s = P->layer_slot0[l]; s < P->layer_slot0[l] + P->layer_nslots[l]; ++s)
      if (slotinfo[4 * s] >= 0) {
        fpx = 0.5 - gf[a.L.o_pos + 2 * s]; fpy = 0.5 - gf[a.L.o_pos + 2 * s + 1];
        break;
      }
  }
  // ---- 1: vertices -> integer canvas coordinates; item row ranges ----------------------
  for (int idx = tid; idx < TOTV; idx += R_THREADS) {
    int s = a.vslot[idx];
    int rank = slotinfo[4 * s], nv = slotinfo[4 * s + 1];
    int k = idx - slotinfo[4 * s + 2];
    if (rank < 0 || k >= nv) continue;
    double2 v = *reinterpret_cast<const double2*>(gf + a.L.o_verts + 2 * idx);
    for (int c = 0; c < ncopy; ++c) {
      double vx = v.x, vy = v.y;
      if (torus) { vx = vx + (double)(c / 3 - 1); vy = vy + (double)(c % 3 - 1); }
      if (first_person) { vx = vx + fpx; vy = vy + fpy; }
      int ix = (int)((double)W * vx), iy = (int)((double)H * vy);
      short2 o; o.x = clamp16(ix); o.y = clamp16(iy);
      ivert[c * TOTV + idx] = o;
      int it = rank * ncopy + c;
      atomicMin(&item_y[2 * it], (int)o.y);
      atomicMax(&item_y[2 * it + 1], (int)o.y);
    }
  }
  __syncthreads();
  if (a.debug_stop == 2) return;
  // ---- 3 (ahead of 2, whose barriers publish it): exclusive scan of the clamped row counts
  //         of all items (wave 0)
  if (tid < 64) {
    int run = 0;
    for (int i0 = 0; i0 < total_items; i0 += 64) {
      int it = i0 + tid;
      int cnt = 0;
      if (it < total_items) {
        int y0 = item_y[2 * it], y1 = item_y[2 * it + 1];
        if (y0 < 0) y0 = 0;
        if (y1 > H - 1) y1 = H - 1;   // rows >= H draw nothing (hline clips)
        cnt = (y1 >= y0) ? (y1 - y0 + 1) : 0;
      }
      int inc = cnt;
      for (int o = 1; o < 64; o <<= 1) {
        int t = __shfl_up(inc, o);
        if (tid >= o) inc += t;
      }
      if (it < total_items) rowoff[it] = run + inc - cnt;
      run += __shfl(inc, 63);
    }
    if (tid == 0) rowoff[total_items] = run;
  }
  // ---- 2: edges (ImagingDrawPolygon: add_edge + merge of horizontal runs), packed per
  //         polygon: table edges from the front, horizontal heads from the back.  The
  //         in-polygon ranks come from wave ballots (a polygon's vertices are contiguous
  //         lanes); a polygon straddling 64-vertex chunks gets the counts of its earlier
  //         parts through `carry` (one entry per earlier chunk; it spans at most three
  //         chunks: <= 128 vertices).
  for (int c = 0; c < ncopy; ++c) {
    for (int base0 = 0; base0 < TOTV; base0 += R_THREADS) {
      const int base = base0 + (tid & ~63);   // this wave's 64-vertex chunk
      int idx = base + lane;
      int s = (idx < TOTV) ? (int)a.vslot[idx] : 0;
      int rank = slotinfo[4 * s], nv = slotinfo[4 * s + 1];
      int v0 = slotinfo[4 * s + 2];
      int k = idx - v0;
      bool valid = (idx < TOTV) && rank >= 0 && k < nv;
      const short2* pv = ivert + c * TOTV + v0;
      unsigned char fl = 0;
      short2 p0 = make_short2(0, 0), p1 = p0;
      bool closing = false;
      if (valid) {
        int k2 = (k + 1 == nv) ? 0 : k + 1;
        p0 = pv[k]; p1 = pv[k2];
        closing = (k == nv - 1);
        bool horiz = (p0.y == p1.y);
        if (closing && p0.x == p1.x && p0.y == p1.y) fl = 0;   // last == first: no closing edge
        else if (!horiz) fl = 1;
        else {
          bool absorbed = false;
          if (k >= 1 && !closing) {
            short2 pp = pv[k - 1];
            if (pp.y == p0.y) absorbed = (p1.x > p0.x && p0.x > pp.x) || (p1.x < p0.x && p0.x < pp.x);
          }
          fl = absorbed ? 0 : 2;
        }
      }
      unsigned long long m1 = __ballot(valid && fl == 1), m2 = __ballot(valid && fl == 2);
      int lo_lane = v0 - base;
      if (lo_lane < 0) lo_lane = 0;
      unsigned long long below = ((1ull << lane) - 1ull) & ~((1ull << lo_lane) - 1ull);
      int nt = __popcll(m1 & below), nh = __popcll(m2 & below);
      // the polygon of the chunk's last lane may continue in the next chunk: publish this
      // chunk's own counts under the chunk's ordinal within the polygon
      const int ord = (base >> 6) - (v0 >> 6);
      if (lane == 63 && valid && (v0 + nv > base + 64) && ord < 2) {
        carry[4 * s + 2 * ord] = nt + (fl == 1);
        carry[4 * s + 2 * ord + 1] = nh + (fl == 2);
      }
      __syncthreads();
      if (valid) {
        for (int j = 0; j < ord && j < 2; ++j) { nt += carry[4 * s + 2 * j]; nh += carry[4 * s + 2 * j + 1]; }
        REdge* reg = edges + c * TOTV + v0;
        if (fl == 1) {
          REdge E; E.x0 = p0.x; E.y0 = p0.y; E.y1 = p1.y; E.x1 = p1.x; E.pad = 0.0f;
          E.dx = ((float)(p1.x - p0.x)) / (float)(p1.y - p0.y);
          reg[nt] = E;
        } else if (fl == 2) {
          // extend over the following absorbed edges (never the closing edge)
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
          REdge E;
          E.x0 = p0.x < hx ? p0.x : hx; E.y0 = p0.y; E.y1 = p0.x < hx ? hx : p0.x; E.dx = 0.0f;
          E.x1 = 0; E.pad = 0.0f;
          reg[nv - 1 - nh] = E;
        }
        if (closing) item_cnt[rank * ncopy + c] = (nt + (fl == 1)) | ((nh + (fl == 2)) << 16);
      }
      __syncthreads();   // carries are consumed before the next chunk round overwrites them
    }
  }
  // ---- 2c: flag table edges that can take part in polygon_generic's corner fix-up: an
  //         earlier table edge with the same lean whose top (bit 0) / bottom (bit 1) end
  //         point is the same integer point.  Rows without a flagged event take the
  //         register fast path in scanline_mask().
  for (int c = 0; c < ncopy; ++c) {
    for (int idx = tid; idx < TOTV; idx += R_THREADS) {
      int s = a.vslot[idx];
      int rank = slotinfo[4 * s];
      if (rank < 0) continue;
      int v0 = slotinfo[4 * s + 2];
      int k = idx - v0;
      int nt = item_cnt[rank * ncopy + c] & 0xffff;
      if (k >= nt) continue;
      REdge* reg = edges + c * TOTV + v0;
      REdge E = reg[k];
      if (E.dx == 0.0f) continue;
      bool up = E.y0 < E.y1;
      int tx = up ? E.x0 : E.x1, ty = up ? E.y0 : E.y1;   // top end point
      int bx = up ? E.x1 : E.x0, by = up ? E.y1 : E.y0;   // bottom end point
      int flag = 0;
      for (int q = 0; q < k; ++q) {
        REdge K = reg[q];
        if ((E.dx > 0 && K.dx <= 0) || (E.dx < 0 && K.dx >= 0)) continue;
        bool kup = K.y0 < K.y1;
        int ktx = kup ? K.x0 : K.x1, kty = kup ? K.y0 : K.y1;
        int kbx = kup ? K.x1 : K.x0, kby = kup ? K.y1 : K.y0;
        if (ktx == tx && kty == ty) flag |= 1;
        if (kbx == bx && kby == by) flag |= 2;
      }
      if (flag) reg[k].pad = __int_as_float(flag);
    }
  }
  if (a.debug_stop == 3) return;

  __syncthreads();   // also: the integer vertices are dead from here on (masks alias them)
  if (a.debug_stop == 4) return;

  // ---- 3b: per (item, row) the set of table edges whose row range contains the row, as a
  //          64-bit mask kept where the row's coverage mask goes later (each row's thread
  //          reads it before it writes the mask).  One thread per table edge ORs its bit
  //          into the rows it spans.  Used when all rows fit in one pass and no polygon has
  //          more than 64 edges; the scanline then visits only those edges.
  const bool use_bits = (rowoff[total_items] <= cap_rows) && (a.xxcap <= 128);
  if (use_bits) {
    const int nrows = rowoff[total_items];
    for (int i = tid; i < nrows; i += R_THREADS) masks[(size_t)i * words] = 0ull;
    __syncthreads();
    for (int c = 0; c < ncopy; ++c) {
      for (int idx = tid; idx < TOTV; idx += R_THREADS) {
        int s = a.vslot[idx];
        int rank = slotinfo[4 * s];
        if (rank < 0) continue;
        int g = rank * ncopy + c;
        int k = idx - slotinfo[4 * s + 2];
        if (k >= (item_cnt[g] & 0xffff)) continue;
        REdge E = edges[c * TOTV + idx];
        int emin = E.y0 < E.y1 ? E.y0 : E.y1, emax = E.y0 < E.y1 ? E.y1 : E.y0;
        int ymin = item_y[2 * g];
        int wbase = rowoff[g] - (ymin < 0 ? 0 : ymin);
        int ya = emin < 0 ? 0 : emin, yb = emax > H - 1 ? H - 1 : emax;
        const unsigned long long bit = 1ull << k;
        for (int y = ya; y <= yb; ++y) atomicOr(&masks[(size_t)(wbase + y) * words], bit);
      }
    }
  }

  const int segs = (H * W) / 16;   // 16-pixel row segments
  const unsigned bgx = ((unsigned)P->render.bg[0] & 255u) | (((unsigned)P->render.bg[1] & 255u) << 8) |
                       (((unsigned)P->render.bg[2] & 255u) << 16);
  uint8_t* out = a.image + (size_t)env * H * W * 3;

  // passes: as many whole items as fit in the mask buffer (cap_rows >= H); with more
  // than one pass the partially composed frame round-trips through `out` (L2)
  for (int base = 0;;) {
    const int r0 = rowoff[base];
    int lo = base + 1, hi = total_items;   // largest end with rowoff[end] - r0 <= cap_rows
    if (total_items == 0) { lo = hi = 0; }
    while (lo < hi) {
      int mid = (lo + hi + 1) >> 1;
      if (rowoff[mid] - r0 <= cap_rows) lo = mid; else hi = mid - 1;
    }
    const int end = lo;
    const int total_rows = rowoff[end] - r0;
    for (int i = tid; i < H * iwords; i += R_THREADS) rowitems[i] = 0u;
    if (tid == 0) misc[1] = 0;
    __syncthreads();
    // ---- 4: coverage masks, one thread per (item, row) ------------------------------------
    for (int w = tid; w < total_rows; w += R_THREADS) {
      int l2 = base, h2 = end - 1;   // last item with rowoff <= r0 + w
      while (l2 < h2) {
        int mid = (l2 + h2 + 1) >> 1;
        if (rowoff[mid] - r0 <= w) l2 = mid; else h2 = mid - 1;
      }
      int g = l2, it = g - base;
      int ymin = item_y[2 * g], ymax = item_y[2 * g + 1];
      int ystart = ymin < 0 ? 0 : ymin;
      int y = ystart + (w - (rowoff[g] - r0));
      int pymax = ymax > H ? H : ymax;              // polygon_generic clamps ymax to ysize
      int sc = item_slot[g];
      int s = sc & 0xffff, c = sc >> 16;
      int cnt = item_cnt[g];
      RPoly poly = {edges + c * TOTV + slotinfo[4 * s + 2], slotinfo[4 * s + 1], cnt & 0xffff, cnt >> 16};
      bool generic = false;
      RMask m = use_bits ? scanline_mask<true>(poly, y, pymax, W, &generic, masks[(size_t)w * words])
                         : scanline_mask<false>(poly, y, pymax, W, &generic, 0ull);
      if (generic) {
        queue[atomicAdd(&misc[1], 1)] = (unsigned short)w;
      } else {
        masks[(size_t)w * words] = m.w0;
        if (words > 1) masks[(size_t)w * words + 1] = m.w1;
        if (m.w0 | m.w1) atomicOr(&rowitems[y * iwords + (it >> 5)], 1u << (it & 31));
      }
    }
    __syncthreads();
    // rows that need the generic scanline (corner fix-ups, > 8 crossings): R_SLOW lanes
    if (tid < R_SLOW) {
      const int qn = misc[1];
      for (int qi = tid; qi < qn; qi += R_SLOW) {
        int w = queue[qi];
        int l2 = base, h2 = end - 1;
        while (l2 < h2) {
          int mid = (l2 + h2 + 1) >> 1;
          if (rowoff[mid] - r0 <= w) l2 = mid; else h2 = mid - 1;
        }
        int g = l2, it = g - base;
        int ymin = item_y[2 * g], ymax = item_y[2 * g + 1];
        int ystart = ymin < 0 ? 0 : ymin;
        int y = ystart + (w - (rowoff[g] - r0));
        int pymax = ymax > H ? H : ymax;
        int sc = item_slot[g];
        int s = sc & 0xffff, c = sc >> 16;
        int cnt = item_cnt[g];
        RPoly poly = {edges + c * TOTV + slotinfo[4 * s + 2], slotinfo[4 * s + 1], cnt & 0xffff, cnt >> 16};
        RMask m = scanline_mask_generic(poly, y, pymax, xxs + tid, W, a.xxcap);
        masks[(size_t)w * words] = m.w0;
        if (words > 1) masks[(size_t)w * words + 1] = m.w1;
        if (m.w0 | m.w1) atomicOr(&rowitems[y * iwords + (it >> 5)], 1u << (it & 31));
      }
    }
    __syncthreads();
    if (a.debug_stop == 5) return;
    // ---- 5: compose (painter's order = item order), pack RGB, store flipped ------------------
    for (int seg = tid; seg < segs; seg += R_THREADS) {
      int y = (seg * 16) / W, x0 = (seg * 16) % W;
      uint4* dst = reinterpret_cast<uint4*>(out + ((size_t)(H - 1 - y) * W + x0) * 3);
      unsigned px[16];
      if (base == 0) {
#pragma unroll
        for (int i = 0; i < 16; ++i) px[i] = bgx;
      } else {  // continue from the previous pass
        uint4 q0 = dst[0], q1 = dst[1], q2 = dst[2];
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
        unsigned bitsw = rowitems[y * iwords + iw];
        while (bitsw) {
          int b = __ffs((int)bitsw) - 1;
          bitsw &= bitsw - 1;
          int g = base + iw * 32 + b;
          int ymin = item_y[2 * g];
          int ystart = ymin < 0 ? 0 : ymin;
          size_t w = (size_t)(rowoff[g] - r0) + (y - ystart);
          unsigned long long mw = masks[w * words + (x0 >> 6)];
          unsigned bits = (unsigned)(mw >> (x0 & 63)) & 0xFFFFu;
          if (!bits) continue;
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
    if (end >= total_items) break;
    base = end;
    __syncthreads();
  }
}
