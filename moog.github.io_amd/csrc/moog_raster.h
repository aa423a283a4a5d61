// moog_raster.h -- HIP scanline polygon rasteriser (gfx950), bit-exact with
// Pillow's ImageDraw.polygon in RGBA blend mode as used by PILRenderer
// (reference moog/observers/pil_renderer.py:88-120; Pillow Draw.c
// ImagingDrawPolygon / polygon_generic(hasAlpha=1) / hline32rgba as restated and
// fuzz-validated against Pillow 12.2.0 in oracle/moog_oracle.c).
//
// One 256-thread workgroup renders one env's frame; everything between reading
// the sprite vertices (coalesced 16 B/lane) and writing the uint8 frame
// (coalesced 16 B/lane, written exactly once) stays in LDS / registers:
//   1  vertices -> integer canvas coordinates ((int)(W*x), one thread per vertex
//      per polygon copy), per-item row ranges by LDS atomics, per-item RGBA
//   2  one thread per edge: slope, horizontal-run merging (ImagingDrawPolygon)
//   3  compact work list of (item, row) pairs that actually intersect the canvas
//   4  one thread per (item, row): Pillow's scanline -> 64/128-bit coverage mask
//   5  one thread per 16-pixel row segment: compose the covering items in
//      painter's order (one blend per covered pixel), RGBX in registers
//   6  pack to RGB, flip rows (np.flipud) in LDS, stream out with dwordx4 stores
// HBM-bound by construction: algorithmic bytes = H*W*3 + live vertices * 16.
#pragma once
#include "moog_device.h"

#define R_THREADS 256
#define R_XX 24      // max crossings kept per scanline
#define R_MAXSEG 4   // 16-pixel segments per thread (<= 128x128 frames)

struct RArgs {
  const moog_program_t* P;
  moog_layout_t L;
  const double* f64;
  const int32_t* i32;
  uint8_t* image;
  const int16_t* vslot;
  int32_t n_envs;
  int32_t chunk;       // items per pass (masks are sized for chunk * H rows)
  int32_t words;       // 64-bit words per row mask
  int32_t iwords;      // 32-bit words of the per-row item bitmask
  int32_t max_items;   // S * copies
};

// LDS plan shared by host (sizes) and device (carve-up)
struct RPlan {
  size_t o_ivert, o_dx, o_eflag, o_hx1, o_slotinfo, o_item_slot, o_item_y, o_item_rgba, o_rowoff,
      o_rowitems, o_masks, o_xx, o_frame, o_misc, total;
};

__host__ __device__ inline size_t r_align(size_t x) { return (x + 15) & ~(size_t)15; }

__host__ __device__ inline void raster_plan(int S, int TOTV, int ncopy, int W, int H, int chunk,
                                            int words, int iwords, RPlan* p) {
  size_t o = 0;
  size_t nv = (size_t)TOTV * ncopy, items = (size_t)S * ncopy;
  p->o_ivert = o; o = r_align(o + nv * 4);            // short2 per copy-vertex
  p->o_dx = o; o = r_align(o + nv * 4);               // float slope of edge k -> k+1
  p->o_hx1 = o; o = r_align(o + nv * 2);              // short: merged end x of horizontal heads
  p->o_eflag = o; o = r_align(o + nv);                // 0 none, 1 table edge, 2 horizontal head
  p->o_slotinfo = o; o = r_align(o + (size_t)S * 8);  // per slot: rank (int), nverts (int)
  p->o_item_slot = o; o = r_align(o + items * 4);     // slot | copy << 16
  p->o_item_y = o; o = r_align(o + items * 8);        // ymin, ymax (ints, atomics)
  p->o_item_rgba = o; o = r_align(o + items * 4);
  p->o_rowoff = o; o = r_align(o + (items + 1) * 4);
  p->o_rowitems = o; o = r_align(o + (size_t)H * iwords * 4);
  p->o_masks = o; o = r_align(o + (size_t)chunk * H * words * 8);
  p->o_xx = o; o = r_align(o + (size_t)R_XX * R_THREADS * 4);
  p->o_frame = o; o = r_align(o + (size_t)H * W * 3);
  p->o_misc = o; o = r_align(o + 64);
  p->total = o;
}

__device__ inline int pil_round_up(float f) {
  return (int)((f >= 0.0f) ? floorf(f + 0.5f) : -floorf(fabsf(f) + 0.5f));
}
__device__ inline int pil_round_down(float f) {
  return (int)((f >= 0.0f) ? ceilf(f - 0.5f) : -ceilf(fabsf(f) - 0.5f));
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

// view of one polygon's edge data in LDS (edge k runs vertex k -> (k+1) % n)
struct RPoly {
  const short2* v;
  const float* dx;
  const short* hx1;
  const unsigned char* fl;
  int n;
};

// Draw.c draw_horizontal_lines
__device__ inline void draw_horizontal(const RPoly& p, int y, int* x_pos, RMask& m, int W) {
  for (int i = 0; i < p.n; ++i) {
    if (p.fl[i] != 2) continue;
    short2 a = p.v[i];
    if (a.y != y) continue;
    int xa = a.x, xb = p.hx1[i];
    int xmin = xa < xb ? xa : xb, xmax = xa < xb ? xb : xa;
    if (*x_pos != -1 && *x_pos < xmin) continue;
    if (*x_pos > xmin) {
      xmin = *x_pos;
      if (xmax < xmin) continue;
    }
    mask_fill(m, W, xmin, xmax);
    *x_pos = xmax + 1;
  }
}

// Coverage of scanline y of one polygon: polygon_generic(hasAlpha=1), one row.
// xx: this thread's crossing list, element j at xx[j * R_THREADS].
__device__ inline RMask scanline_mask(const RPoly& p, int y, int poly_ymax, float* xx, int W) {
  RMask m = {0ull, 0ull};
  int j = 0;
  for (int i = 0; i < p.n; ++i) {
    if (p.fl[i] != 1) continue;
    int i2 = (i + 1 == p.n) ? 0 : i + 1;
    short2 a = p.v[i], b = p.v[i2];
    int y0 = a.y, y1 = b.y;
    int emin = y0 < y1 ? y0 : y1, emax = y0 < y1 ? y1 : y0;
    if (y < emin || y > emax) continue;
    float dx = p.dx[i];
    float x = (float)(y - y0) * dx + (float)a.x;
    if (j < R_XX) xx[j * R_THREADS] = x;
    ++j;
    if (y == emax && y < poly_ymax) {
      if (j < R_XX) xx[j * R_THREADS] = x;
      ++j;
    } else if (dx != 0.0f) {
      // connect discontiguous corners (a tip whose two edges lean the same way)
      int jj = 0;
      for (int k = 0; k < i; ++k) {
        if (p.fl[k] != 1) continue;
        int k2 = (k + 1 == p.n) ? 0 : k + 1;
        short2 ka = p.v[k], kb = p.v[k2];
        int ky0 = ka.y, ky1 = kb.y;
        int kmin = ky0 < ky1 ? ky0 : ky1, kmax = ky0 < ky1 ? ky1 : ky0;
        if (y < kmin || y > kmax) continue;
        int kpos = jj;
        jj += (y == kmax && y < poly_ymax) ? 2 : 1;
        float kdx = p.dx[k];
        if ((dx > 0 && kdx <= 0) || (dx < 0 && kdx >= 0)) continue;
        bool top = (y == emin && y == kmin), bot = (y == emax && y == kmax);
        if (!(top || bot)) continue;
        if (x != (float)(y - ky0) * kdx + (float)ka.x) continue;
        int off = top ? 1 : -1;
        float adj = (float)(y + off - y0) * dx + (float)a.x;
        float adjo = (float)(y + off - ky0) * kdx + (float)ka.x;
        if (adj > x && adjo > x) {
          float vv = (float)(pil_round_up(fminf(adj, adjo)) - 1);
          if (vv > x && kpos < R_XX) xx[kpos * R_THREADS] = vv;
        } else if (adj < x && adjo < x) {
          float vv = (float)(pil_round_up(fmaxf(adj, adjo)) + 1);
          if (vv < x && kpos < R_XX) xx[kpos * R_THREADS] = vv;
        }
        break;
      }
    }
  }
  if (j > R_XX) j = R_XX;
  for (int q = 1; q < j; ++q) {  // insertion sort (qsort with x_cmp)
    float key = xx[q * R_THREADS];
    int r = q - 1;
    while (r >= 0 && xx[r * R_THREADS] > key) { xx[(r + 1) * R_THREADS] = xx[r * R_THREADS]; --r; }
    xx[(r + 1) * R_THREADS] = key;
  }
  int x_pos = (j == 0) ? -1 : 0;
  for (int i = 1; i < j; i += 2) {
    int x_end = pil_round_down(xx[i * R_THREADS]);
    if (x_end < x_pos) continue;
    draw_horizontal(p, y, &x_pos, m, W);
    if (x_end < x_pos) continue;
    int x_start = pil_round_up(xx[(i - 1) * R_THREADS]);
    if (x_pos > x_start) {
      x_start = x_pos;
      if (x_end < x_start) continue;
    }
    mask_fill(m, W, x_start, x_end);
    x_pos = x_end + 1;
  }
  draw_horizontal(p, y, &x_pos, m, W);
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
  const moog_program_t* P = a.P;
  const int W = P->render.width, H = P->render.height;
  const int S = P->n_slots, TOTV = a.L.TOTV;
  const bool torus = (P->render.polymod == MOOG_POLYMOD_TORUS);
  const int ncopy = torus ? 9 : 1;
  const int words = a.words, iwords = a.iwords, chunk = a.chunk;
  const double* gf = a.f64 + (size_t)env * a.L.f64_per_env;
  const int32_t* gq = a.i32 + (size_t)env * a.L.i32_per_env;
  const int tid = threadIdx.x;

  RPlan pl;
  raster_plan(S, TOTV, ncopy, W, H, chunk, words, iwords, &pl);
  short2* ivert = reinterpret_cast<short2*>(moog_lds + pl.o_ivert);
  float* edx = reinterpret_cast<float*>(moog_lds + pl.o_dx);
  short* hx1 = reinterpret_cast<short*>(moog_lds + pl.o_hx1);
  unsigned char* eflag = moog_lds + pl.o_eflag;
  int* slotinfo = reinterpret_cast<int*>(moog_lds + pl.o_slotinfo);
  int* item_slot = reinterpret_cast<int*>(moog_lds + pl.o_item_slot);
  int* item_y = reinterpret_cast<int*>(moog_lds + pl.o_item_y);
  unsigned* item_rgba = reinterpret_cast<unsigned*>(moog_lds + pl.o_item_rgba);
  int* rowoff = reinterpret_cast<int*>(moog_lds + pl.o_rowoff);
  unsigned* rowitems = reinterpret_cast<unsigned*>(moog_lds + pl.o_rowitems);
  unsigned long long* masks = reinterpret_cast<unsigned long long*>(moog_lds + pl.o_masks);
  float* xxs = reinterpret_cast<float*>(moog_lds + pl.o_xx);
  uint8_t* frame = moog_lds + pl.o_frame;
  int* misc = reinterpret_cast<int*>(moog_lds + pl.o_misc);  // [0] n_live

  // ---- 0: live sprites in slot (= layer, list) order; per-sprite colour ----------------
  if (tid < 64) {
    int base = 0;
    for (int s0 = 0; s0 < S; s0 += 64) {
      int s = s0 + tid;
      bool live = false;
      int nv = 0;
      if (s < S) { live = (gq[a.L.o_flags + s] & MOOG_F_ALIVE) != 0; nv = gq[a.L.o_nverts + s]; }
      unsigned long long bal = __ballot(live);
      int rank = base + __popcll(bal & ((1ull << tid) - 1ull));
      if (s < S) { slotinfo[2 * s] = live ? rank : -1; slotinfo[2 * s + 1] = nv; }
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
      base += __popcll(bal);
    }
    if (tid == 0) misc[0] = base;
  }
  __syncthreads();
  const int n_live = misc[0];
  const int total_items = n_live * ncopy;

  // ---- 1: vertices -> integer canvas coordinates; item row ranges ----------------------
  for (int idx = tid; idx < TOTV; idx += R_THREADS) {
    int s = a.vslot[idx];
    int rank = slotinfo[2 * s], nv = slotinfo[2 * s + 1];
    int k = idx - P->slot_voff[s];
    if (rank < 0 || k >= nv) continue;
    double2 v = *reinterpret_cast<const double2*>(gf + a.L.o_verts + 2 * idx);
    for (int c = 0; c < ncopy; ++c) {
      double vx = v.x, vy = v.y;
      if (torus) { vx = vx + (double)(c / 3 - 1); vy = vy + (double)(c % 3 - 1); }
      int ix = (int)((double)W * vx), iy = (int)((double)H * vy);
      short2 o; o.x = clamp16(ix); o.y = clamp16(iy);
      ivert[c * TOTV + idx] = o;
      int it = rank * ncopy + c;
      atomicMin(&item_y[2 * it], (int)o.y);
      atomicMax(&item_y[2 * it + 1], (int)o.y);
    }
  }
  __syncthreads();
  // ---- 2: edges (ImagingDrawPolygon: add_edge + merge of consecutive horizontal runs) ---
  for (int idx = tid; idx < TOTV; idx += R_THREADS) {
    int s = a.vslot[idx];
    int rank = slotinfo[2 * s], nv = slotinfo[2 * s + 1];
    int v0 = P->slot_voff[s];
    int k = idx - v0;
    if (rank < 0 || k >= nv) continue;
    for (int c = 0; c < ncopy; ++c) {
      const short2* pv = ivert + c * TOTV + v0;
      int k2 = (k + 1 == nv) ? 0 : k + 1;
      short2 p0 = pv[k], p1 = pv[k2];
      bool closing = (k == nv - 1);
      unsigned char fl;
      float dx = 0.0f;
      short hx = p1.x;
      bool horiz = (p0.y == p1.y);
      if (closing && p0.x == p1.x && p0.y == p1.y) fl = 0;   // last == first: no closing edge
      else if (!horiz) { fl = 1; dx = ((float)(p1.x - p0.x)) / (float)(p1.y - p0.y); }
      else {
        bool absorbed = false;
        if (k >= 1 && !closing) {
          short2 pp = pv[k - 1];
          if (pp.y == p0.y) absorbed = (p1.x > p0.x && p0.x > pp.x) || (p1.x < p0.x && p0.x < pp.x);
        }
        if (absorbed) fl = 0;
        else {
          fl = 2;
          // extend over the following absorbed horizontal edges (never the closing edge)
          int j = k + 1;
          short2 prev = p0, cur = p1;
          while (j <= nv - 2) {
            short2 nxt = pv[j + 1];
            bool ab = (cur.y == nxt.y) && (prev.y == cur.y) &&
                      ((nxt.x > cur.x && cur.x > prev.x) || (nxt.x < cur.x && cur.x < prev.x));
            if (!ab) break;
            hx = nxt.x; prev = cur; cur = nxt; ++j;
          }
        }
      }
      eflag[c * TOTV + idx] = fl;
      edx[c * TOTV + idx] = dx;
      hx1[c * TOTV + idx] = hx;
    }
  }

  // 16-pixel row segments owned by this thread, RGBX per pixel
  const int segs = (H * W) / 16;
  const unsigned bgx = ((unsigned)P->render.bg[0] & 255u) | (((unsigned)P->render.bg[1] & 255u) << 8) |
                       (((unsigned)P->render.bg[2] & 255u) << 16);
  unsigned px[R_MAXSEG][16];
#pragma unroll
  for (int k = 0; k < R_MAXSEG; ++k)
#pragma unroll
    for (int i = 0; i < 16; ++i) px[k][i] = bgx;

  for (int base = 0; base < total_items; base += chunk) {
    int nit = total_items - base;
    if (nit > chunk) nit = chunk;
    __syncthreads();
    // ---- 3: compact (item, row) work list: exclusive scan of clamped row counts ---------
    for (int i = tid; i < H * iwords; i += R_THREADS) rowitems[i] = 0u;
    if (tid < 64) {
      int run = 0;
      for (int i0 = 0; i0 < nit; i0 += 64) {
        int it = i0 + tid;
        int cnt = 0;
        if (it < nit) {
          int y0 = item_y[2 * (base + it)], y1 = item_y[2 * (base + it) + 1];
          if (y0 < 0) y0 = 0;
          if (y1 > H - 1) y1 = H - 1;   // rows >= H draw nothing (hline clips)
          cnt = (y1 >= y0) ? (y1 - y0 + 1) : 0;
        }
        int inc = cnt;
        for (int o = 1; o < 64; o <<= 1) {
          int t = __shfl_up(inc, o);
          if (tid >= o) inc += t;
        }
        if (it < nit) rowoff[it] = run + inc - cnt;
        run += __shfl(inc, 63);
      }
      if (tid == 0) rowoff[nit] = run;
    }
    __syncthreads();
    const int total_rows = rowoff[nit];
    // ---- 4: coverage masks, one thread per (item, row) ------------------------------------
    for (int w = tid; w < total_rows; w += R_THREADS) {
      int lo = 0, hi = nit - 1;   // last item with rowoff <= w
      while (lo < hi) {
        int mid = (lo + hi + 1) >> 1;
        if (rowoff[mid] <= w) lo = mid; else hi = mid - 1;
      }
      int it = lo, g = base + it;
      int ymin = item_y[2 * g], ymax = item_y[2 * g + 1];
      int ystart = ymin < 0 ? 0 : ymin;
      int y = ystart + (w - rowoff[it]);
      int pymax = ymax > H ? H : ymax;              // polygon_generic clamps ymax to ysize
      int sc = item_slot[g];
      int s = sc & 0xffff, c = sc >> 16;
      int v0 = c * TOTV + P->slot_voff[s];
      RPoly poly = {ivert + v0, edx + v0, hx1 + v0, eflag + v0, slotinfo[2 * s + 1]};
      RMask m = scanline_mask(poly, y, pymax, xxs + tid, W);
      masks[(size_t)w * words] = m.w0;
      if (words > 1) masks[(size_t)w * words + 1] = m.w1;
      if (m.w0 | m.w1) atomicOr(&rowitems[y * iwords + (it >> 5)], 1u << (it & 31));
    }
    __syncthreads();
    // ---- 5: compose, painter's order = item order -------------------------------------------
#pragma unroll
    for (int k = 0; k < R_MAXSEG; ++k) {
      int seg = tid + k * R_THREADS;
      if (seg < segs) {
        int y = (seg * 16) / W, x0 = (seg * 16) % W;
        for (int iw = 0; iw < iwords; ++iw) {
          unsigned bitsw = rowitems[y * iwords + iw];
          while (bitsw) {
            int b = __ffs((int)bitsw) - 1;
            bitsw &= bitsw - 1;
            int it = iw * 32 + b, g = base + it;
            int ymin = item_y[2 * g];
            int ystart = ymin < 0 ? 0 : ymin;
            size_t w = (size_t)rowoff[it] + (y - ystart);
            unsigned long long mw = masks[w * words + (x0 >> 6)];
            unsigned bits = (unsigned)(mw >> (x0 & 63)) & 0xFFFFu;
            if (!bits) continue;
            unsigned rgba = item_rgba[g];
            unsigned al = rgba >> 24;
            if (al == 255u) {
              unsigned fg = rgba & 0xFFFFFFu;
#pragma unroll
              for (int i = 0; i < 16; ++i) px[k][i] = (bits & (1u << i)) ? fg : px[k][i];
            } else {
              unsigned f0 = rgba & 255u, f1 = (rgba >> 8) & 255u, f2 = (rgba >> 16) & 255u;
#pragma unroll
              for (int i = 0; i < 16; ++i) {
                if (bits & (1u << i)) {
                  unsigned o = px[k][i];
                  px[k][i] = blend8(o & 255u, f0, al) | (blend8((o >> 8) & 255u, f1, al) << 8) |
                             (blend8((o >> 16) & 255u, f2, al) << 16);
                }
              }
            }
          }
        }
      }
    }
  }
  __syncthreads();
  // ---- 6: RGBX -> RGB, rows flipped (np.flipud), then 16-byte coalesced stores --------------
#pragma unroll
  for (int k = 0; k < R_MAXSEG; ++k) {
    int seg = tid + k * R_THREADS;
    if (seg < segs) {
      int y = (seg * 16) / W, x0 = (seg * 16) % W;
      unsigned* dst = reinterpret_cast<unsigned*>(frame + ((size_t)(H - 1 - y) * W + x0) * 3);
#pragma unroll
      for (int q = 0; q < 4; ++q) {   // 4 pixels (RGBX) -> 3 dwords (RGB)
        unsigned p0 = px[k][4 * q], p1 = px[k][4 * q + 1], p2 = px[k][4 * q + 2], p3 = px[k][4 * q + 3];
        dst[3 * q] = (p0 & 0xFFFFFFu) | (p1 << 24);
        dst[3 * q + 1] = ((p1 >> 8) & 0xFFFFu) | (p2 << 16);
        dst[3 * q + 2] = ((p2 >> 16) & 0xFFu) | (p3 << 8);
      }
    }
  }
  __syncthreads();
  const uint4* src = reinterpret_cast<const uint4*>(frame);
  uint4* out = reinterpret_cast<uint4*>(a.image + (size_t)env * H * W * 3);
  for (int i = tid; i < (H * W * 3) / 16; i += R_THREADS) out[i] = src[i];
}
