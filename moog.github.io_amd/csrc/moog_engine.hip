// moog_engine.hip -- kernels and C ABI of the MI355X batched MOOG step engine.
//
// Kernels (all gfx950, wave64):
//   moog_reset_kernel   one wavefront per env; device-side state initialisation
//                       (rejection sampler) for masked / auto-resetting envs
//   moog_step_kernel    one wavefront per env; rules -> action -> K physics
//                       substeps -> task, state record staged in LDS
//   moog_raster_kernel  one 256-thread workgroup per env; PIL-exact scanline
//                       polygon fill, painter's order composed in registers,
//                       coalesced 16-byte stores of the uint8 frame
// The C ABI at the bottom is what include/moog_engine.h declares.
#include <hip/hip_runtime.h>

#include <cmath>
#include <cstdio>
#include <cstring>
#include <string>
#include <vector>

#include "moog_device.h"

// =====================================================================================
// record staging: HBM <-> LDS, 16 bytes per lane, coalesced
// =====================================================================================
__device__ inline void load_record(const Env& e, const double* gf, const int32_t* gq) {
  const double2* src = reinterpret_cast<const double2*>(gf);
  double2* dst = reinterpret_cast<double2*>(e.f);
  for (int i = e.lane; i < e.L.f64_per_env / 2; i += 64) dst[i] = src[i];
  const int4* srci = reinterpret_cast<const int4*>(gq);
  int4* dsti = reinterpret_cast<int4*>(e.q);
  for (int i = e.lane; i < e.L.i32_per_env / 4; i += 64) dsti[i] = srci[i];
  wsync();
}

__device__ inline void store_record(const Env& e, double* gf, int32_t* gq) {
  wsync();
  double2* dst = reinterpret_cast<double2*>(gf);
  const double2* src = reinterpret_cast<const double2*>(e.f);
  for (int i = e.lane; i < e.L.f64_per_env / 2; i += 64) dst[i] = src[i];
  int4* dsti = reinterpret_cast<int4*>(gq);
  const int4* srci = reinterpret_cast<const int4*>(e.q);
  for (int i = e.lane; i < e.L.i32_per_env / 4; i += 64) dsti[i] = srci[i];
}

struct KArgs {
  const moog_program_t* P;
  moog_layout_t L;
  double* f64;
  int32_t* i32;
  const void* actions;
  const double* inj;
  int32_t inj_n;
  int32_t n_envs;
  uint64_t seed;
  int64_t env_index0;
  const uint8_t* mask;
  double* reward;
  double* discount;
  int32_t* step_type;
  int32_t mode;
  const int16_t* vslot;
};

enum { MODE_STEP = 0, MODE_PHYSICS = 1, MODE_RESET_MASK = 2, MODE_RESET_AUTO = 3 };

extern __shared__ __attribute__((aligned(16))) unsigned char moog_lds[];

__device__ inline void bind_env(Env& e, const KArgs& a, int env) {
  e.P = a.P;
  e.L = a.L;
  e.f = reinterpret_cast<double*>(moog_lds);
  e.q = reinterpret_cast<int32_t*>(moog_lds + (size_t)a.L.f64_per_env * 8);
  e.bb = reinterpret_cast<double*>(moog_lds + (size_t)a.L.f64_per_env * 8 + (size_t)a.L.i32_per_env * 4);
  e.xf = e.bb + 4 * a.L.S;
  e.vslot = a.vslot;
  e.inj = a.inj ? a.inj + (size_t)env * a.inj_n : nullptr;
  e.inj_n = a.inj_n;
  e.seed = a.seed;
  e.env_index = a.env_index0 + env;
  e.lane = threadIdx.x;
}

// reset_next word: 0 = running, 1 = reset on the next call (environment.py:100-101),
// 2 = was reset earlier in THIS call (the step kernel skips it and clears the mark).
__global__ __launch_bounds__(64) void moog_reset_kernel(KArgs a) {
  int env = blockIdx.x;
  if (env >= a.n_envs) return;
  int32_t* gq = a.i32 + (size_t)env * a.L.i32_per_env;
  bool want;
  if (a.mode == MODE_RESET_MASK) want = (a.mask == nullptr) || (a.mask[env] != 0);
  else want = (gq[a.L.o_reset_next] == 1);
  if (!want) return;
  Env e;
  bind_env(e, a, env);
  double* gf = a.f64 + (size_t)env * a.L.f64_per_env;
  load_record(e, gf, gq);
  if (e.inj && e.lane == 0) e.q[e.L.o_rng + 2] = 0;
  wsync();
  env_reset(e);
  wsync();
  if (e.lane == 0) {
    e.q[e.L.o_reset_next] = (a.mode == MODE_RESET_AUTO) ? 2 : 0;
    if (a.reward) a.reward[env] = __builtin_nan("");
    if (a.discount) a.discount[env] = __builtin_nan("");
    if (a.step_type) a.step_type[env] = 0;
  }
  store_record(e, gf, gq);
}

__global__ __launch_bounds__(64) void moog_step_kernel(KArgs a) {
  int env = blockIdx.x;
  if (env >= a.n_envs) return;
  int32_t* gq = a.i32 + (size_t)env * a.L.i32_per_env;
  if (a.mode == MODE_STEP && gq[a.L.o_reset_next] == 2) {  // reset earlier in this call
    if (threadIdx.x == 0) gq[a.L.o_reset_next] = 0;
    return;
  }
  Env e;
  bind_env(e, a, env);
  double* gf = a.f64 + (size_t)env * a.L.f64_per_env;
  load_record(e, gf, gq);
  if (e.inj && e.lane == 0) e.q[e.L.o_rng + 2] = 0;
  wsync();
  bbox_build_all(e);
  const moog_program_t* P = a.P;
  if (a.mode == MODE_PHYSICS) {
    for (int k = 0; k < P->updates_per_env_step; ++k) apply_physics(e);
    store_record(e, gf, gq);
    return;
  }
  // environment.py:98-126
  for (int r = 0; r < P->n_rules; ++r) rule_step(e, r);
  double ax = 0, ay = 0;
  int ga = 4;
  if (P->action.kind == MOOG_ACTION_GRID) ga = reinterpret_cast<const int32_t*>(a.actions)[env];
  else {
    ax = reinterpret_cast<const double*>(a.actions)[2 * env];
    ay = reinterpret_cast<const double*>(a.actions)[2 * env + 1];
  }
  action_step(e, ax, ay, ga);
  for (int k = 0; k < P->updates_per_env_step; ++k) apply_physics(e);
  int sc = e.q[e.L.o_step_count] + 1;
  wsync();
  if (e.lane == 0) e.q[e.L.o_step_count] = sc;
  wsync();
  int sr = 0;
  double r = task_reward(e, sc, &sr);
  wsync();
  if (e.lane == 0) {
    if (sr) e.q[e.L.o_reset_next] = 1;
    if (a.reward) a.reward[env] = r;
    if (a.discount) a.discount[env] = sr ? 0.0 : 1.0;
    if (a.step_type) a.step_type[env] = sr ? 2 : 1;
  }
  store_record(e, gf, gq);
}

// =====================================================================================
// rasteriser: Pillow ImageDraw.polygon (RGBA blend) + PILRenderer painter's loop
// (pil_renderer.py:88-120; Draw.c polygon_generic / add_edge / hline32rgba as
// restated and fuzz-validated in oracle/moog_oracle.c)
// =====================================================================================
#define R_THREADS 256
#define R_MAXE 32   // max edges per polygon (31-vertex cap for rendering)
#define R_XX 24     // max crossings per scanline

struct REdge { short x0, y0, x1, y1; float dx; };   // 12 bytes
struct RItem { int n_edges; int ymin, ymax; unsigned rgba; };

struct RArgs {
  const moog_program_t* P;
  moog_layout_t L;
  const double* f64;
  const int32_t* i32;
  uint8_t* image;
  int32_t n_envs;
  int32_t items_per_chunk;
  int32_t words_per_row;
};

__device__ inline int pil_round_up(float f) {
  return (int)((f >= 0.0f) ? floorf(f + 0.5f) : -floorf(fabsf(f) + 0.5f));
}
__device__ inline int pil_round_down(float f) {
  return (int)((f >= 0.0f) ? ceilf(f - 0.5f) : -ceilf(fabsf(f) - 0.5f));
}

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

// One thread builds the edge list of one draw item (ImagingDrawPolygon's loop,
// including the merge of consecutive collinear horizontal edges).
__device__ inline void build_item(const RArgs& a, const double* gf, const int32_t* gq, int slot,
                                  int copy, RItem* item, REdge* edges) {
  const moog_program_t* P = a.P;
  int W = P->render.width, H = P->render.height;
  int n = gq[a.L.o_nverts + slot];
  const double* v = gf + a.L.o_verts + 2 * P->slot_voff[slot];
  double ox = 0, oy = 0;
  bool torus = (P->render.polymod == MOOG_POLYMOD_TORUS);
  if (torus) { ox = (double)(copy / 3 - 1); oy = (double)(copy % 3 - 1); }
  unsigned r8, g8, b8;
  const double* col = gf + a.L.o_color + 3 * slot;
  if (P->render.cmap == MOOG_CMAP_HSV) hsv_to_rgb_u8(col[0], col[1], col[2], r8, g8, b8);
  else { r8 = (unsigned)(int)col[0] & 255u; g8 = (unsigned)(int)col[1] & 255u; b8 = (unsigned)(int)col[2] & 255u; }
  unsigned a8 = (unsigned)gq[a.L.o_opacity + slot] & 255u;
  item->rgba = r8 | (g8 << 8) | (b8 << 16) | (a8 << 24);
  int ne = 0, ymin = H - 1, ymax = 0;
  int fx = 0, fy = 0, px = 0, py = 0, ppx = 0, ppy = 0;
  for (int i = 0; i <= n; ++i) {
    int cx, cy;
    if (i < n) {
      double vx = v[2 * i], vy = v[2 * i + 1];
      if (torus) { vx = vx + ox; vy = vy + oy; }
      cx = (int)((double)W * vx);
      cy = (int)((double)H * vy);
    } else {  // closing edge if last != first
      if (px == fx && py == fy) break;
      cx = fx; cy = fy;
    }
    if (i == 0) { fx = cx; fy = cy; px = cx; py = cy; continue; }
    // edge (px,py) -> (cx,cy); the vertex before px is (ppx,ppy) when i >= 2
    bool merged = false;
    if (i < n + 0 && py == cy && i >= 2 && py == ppy && ne > 0) {
      // horizontal edge right after another horizontal edge (not the closing edge)
      if (cx > px && px > ppx) { edges[ne - 1].x1 = clamp16(cx); merged = true; }
      else if (cx < px && px < ppx) { edges[ne - 1].x1 = clamp16(cx); merged = true; }
    }
    if (!merged && ne < R_MAXE) {
      REdge ed;
      ed.x0 = clamp16(px); ed.y0 = clamp16(py); ed.x1 = clamp16(cx); ed.y1 = clamp16(cy);
      ed.dx = (py == cy) ? 0.0f : ((float)(cx - px)) / (float)(cy - py);
      edges[ne++] = ed;
      int lo = py < cy ? py : cy, hi = py < cy ? cy : py;
      if (ymin > lo) ymin = lo;
      if (ymax < hi) ymax = hi;
    }
    ppx = px; ppy = py; px = cx; py = cy;
  }
  if (ymin < 0) ymin = 0;
  if (ymax > H) ymax = H;
  item->n_edges = ne;
  item->ymin = ymin;
  item->ymax = ymax;
}

__device__ inline void mask_fill(unsigned long long* m, int words, int W, int x0, int x1) {
  if (x0 < 0) x0 = 0; else if (x0 >= W) return;
  if (x1 < 0) return; else if (x1 >= W) x1 = W - 1;
  if (x0 > x1) return;
  for (int w = x0 >> 6; w <= (x1 >> 6) && w < words; ++w) {
    int lo = (w == (x0 >> 6)) ? (x0 & 63) : 0;
    int hi = (w == (x1 >> 6)) ? (x1 & 63) : 63;
    unsigned long long bits = (hi - lo == 63) ? ~0ull : (((1ull << (hi - lo + 1)) - 1ull) << lo);
    m[w] |= bits;
  }
}

__device__ inline void draw_horizontal(const RItem& it, const REdge* ed, int y, int* x_pos,
                                       unsigned long long* m, int words, int W) {
  for (int i = 0; i < it.n_edges; ++i) {
    if (ed[i].y0 == y && ed[i].y1 == y) {
      int xa = ed[i].x0, xb = ed[i].x1;
      int xmin = xa < xb ? xa : xb, xmax = xa < xb ? xb : xa;
      if (*x_pos != -1 && *x_pos < xmin) continue;
      if (*x_pos > xmin) {
        xmin = *x_pos;
        if (xmax < xmin) continue;
      }
      mask_fill(m, words, W, xmin, xmax);
      *x_pos = xmax + 1;
    }
  }
}

// Coverage of one scanline of one polygon: polygon_generic(hasAlpha=1), one row.
__device__ inline void scanline_mask(const RItem& it, const REdge* ed, int y, int poly_ymax,
                                     float* xx, unsigned long long* m, int words, int W) {
  int j = 0;
  for (int i = 0; i < it.n_edges; ++i) {
    int y0 = ed[i].y0, y1 = ed[i].y1;
    if (y0 == y1) continue;
    int emin = y0 < y1 ? y0 : y1, emax = y0 < y1 ? y1 : y0;
    if (y < emin || y > emax) continue;
    float dx = ed[i].dx;
    float x = (float)(y - y0) * dx + (float)ed[i].x0;
    if (j < R_XX) xx[j] = x;
    int myj = j;
    ++j;
    if (y == emax && y < poly_ymax) {
      if (j < R_XX) xx[j] = x;
      ++j;
    } else if (dx != 0.0f) {
      // connect discontiguous corners (tip whose two edges lean the same way)
      int jj = 0;
      for (int k = 0; k < i; ++k) {
        int ky0 = ed[k].y0, ky1 = ed[k].y1;
        if (ky0 == ky1) continue;
        int kmin = ky0 < ky1 ? ky0 : ky1, kmax = ky0 < ky1 ? ky1 : ky0;
        if (y < kmin || y > kmax) continue;
        int kpos = jj;
        jj += (y == kmax && y < poly_ymax) ? 2 : 1;
        float kdx = ed[k].dx;
        if ((dx > 0 && kdx <= 0) || (dx < 0 && kdx >= 0)) continue;
        bool top = (y == emin && y == kmin), bot = (y == emax && y == kmax);
        if (!(top || bot)) continue;
        if (x != (float)(y - ky0) * kdx + (float)ed[k].x0) continue;
        int off = top ? 1 : -1;
        float adj = (float)(y + off - y0) * dx + (float)ed[i].x0;
        float adjo = (float)(y + off - ky0) * kdx + (float)ed[k].x0;
        if (adj > x && adjo > x) {
          float vv = (float)(pil_round_up(fminf(adj, adjo)) - 1);
          if (vv > x && kpos < R_XX) xx[kpos] = vv;
        } else if (adj < x && adjo < x) {
          float vv = (float)(pil_round_up(fmaxf(adj, adjo)) + 1);
          if (vv < x && kpos < R_XX) xx[kpos] = vv;
        }
        break;
      }
      (void)myj;
    }
  }
  if (j > R_XX) j = R_XX;
  for (int p = 1; p < j; ++p) {  // insertion sort (qsort with x_cmp)
    float key = xx[p];
    int q = p - 1;
    while (q >= 0 && xx[q] > key) { xx[q + 1] = xx[q]; --q; }
    xx[q + 1] = key;
  }
  int x_pos = (j == 0) ? -1 : 0;
  for (int i = 1; i < j; i += 2) {
    int x_end = pil_round_down(xx[i]);
    if (x_end < x_pos) continue;
    draw_horizontal(it, ed, y, &x_pos, m, words, W);
    if (x_end < x_pos) continue;
    int x_start = pil_round_up(xx[i - 1]);
    if (x_pos > x_start) {
      x_start = x_pos;
      if (x_end < x_start) continue;
    }
    mask_fill(m, words, W, x_start, x_end);
    x_pos = x_end + 1;
  }
  draw_horizontal(it, ed, y, &x_pos, m, words, W);
}

__device__ inline unsigned blend8(unsigned bg, unsigned fg, unsigned al) {
  unsigned t = bg * (255u - al) + fg * al + 128u;
  return ((t >> 8) + t) >> 8;
}

// LDS carve-up (dynamic): items | edges | xx scratch | coverage masks | frame staging
__global__ __launch_bounds__(R_THREADS) void moog_raster_kernel(RArgs a) {
  int env = blockIdx.x;
  if (env >= a.n_envs) return;
  const moog_program_t* P = a.P;
  const int W = P->render.width, H = P->render.height;
  const int S = P->n_slots;
  const int ncopy = (P->render.polymod == MOOG_POLYMOD_TORUS) ? 9 : 1;
  const int words = a.words_per_row;
  const int chunk = a.items_per_chunk;
  const double* gf = a.f64 + (size_t)env * a.L.f64_per_env;
  const int32_t* gq = a.i32 + (size_t)env * a.L.i32_per_env;
  const int tid = threadIdx.x;

  unsigned char* p = moog_lds;
  RItem* items = reinterpret_cast<RItem*>(p); p += (size_t)chunk * sizeof(RItem);
  REdge* edges = reinterpret_cast<REdge*>(p); p += (size_t)chunk * R_MAXE * sizeof(REdge);
  float* xxs = reinterpret_cast<float*>(p); p += (size_t)R_THREADS * R_XX * sizeof(float);
  p = reinterpret_cast<unsigned char*>(((uintptr_t)p + 15) & ~(uintptr_t)15);
  unsigned long long* masks = reinterpret_cast<unsigned long long*>(p);
  p += (size_t)chunk * H * words * 8;
  uint8_t* frame = p;  // H*W*3 bytes, flipped rows
  int* slot_list = reinterpret_cast<int*>(frame + (((size_t)H * W * 3 + 15) & ~(size_t)15));
  // live sprites in slot (= layer, list) order; count kept at slot_list[S]
  if (tid == 0) {
    int c = 0;
    for (int s = 0; s < S; ++s)
      if (gq[a.L.o_flags + s] & MOOG_F_ALIVE) slot_list[c++] = s;
    slot_list[S] = c;
  }
  // background (pil_renderer.py:100)
  const int segs = (H * W) / 16;   // 16-pixel row segments
  const unsigned bgr = (unsigned)P->render.bg[0] & 255u, bgg = (unsigned)P->render.bg[1] & 255u,
                 bgb = (unsigned)P->render.bg[2] & 255u;
  __syncthreads();
  const int total_items = slot_list[S] * ncopy;

  // each thread owns up to 4 row segments of 16 pixels, composed in registers
  unsigned pr[4][12];  // 16 px * 3 B = 48 B = 12 dwords per segment
  const int my_segs = (segs + R_THREADS - 1) / R_THREADS;
  for (int k = 0; k < 4; ++k) {
#pragma unroll
    for (int d = 0; d < 12; ++d) {
      // bytes: pixel i channel c at byte 3*i+c
      unsigned b0 = (d * 4 + 0) % 3, b1 = (d * 4 + 1) % 3, b2 = (d * 4 + 2) % 3, b3 = (d * 4 + 3) % 3;
      unsigned c0 = b0 == 0 ? bgr : (b0 == 1 ? bgg : bgb);
      unsigned c1 = b1 == 0 ? bgr : (b1 == 1 ? bgg : bgb);
      unsigned c2 = b2 == 0 ? bgr : (b2 == 1 ? bgg : bgb);
      unsigned c3 = b3 == 0 ? bgr : (b3 == 1 ? bgg : bgb);
      pr[k][d] = c0 | (c1 << 8) | (c2 << 16) | (c3 << 24);
    }
  }

  for (int base = 0; base < total_items; base += chunk) {
    int nit = total_items - base;
    if (nit > chunk) nit = chunk;
    __syncthreads();
    // phase A1: edge lists, one thread per item
    for (int it = tid; it < nit; it += R_THREADS) {
      int g = base + it;
      build_item(a, gf, gq, slot_list[g / ncopy], g % ncopy, &items[it], &edges[it * R_MAXE]);
    }
    __syncthreads();
    // phase A2: coverage masks, one thread per (item, row)
    for (int w = tid; w < nit * H; w += R_THREADS) {
      int it = w / H, y = w - it * H;
      unsigned long long m[4] = {0ull, 0ull, 0ull, 0ull};
      const RItem item = items[it];
      if (y >= item.ymin && y <= item.ymax && y < H)
        scanline_mask(item, &edges[it * R_MAXE], y, item.ymax, &xxs[tid * R_XX], m, words, W);
      for (int q = 0; q < words; ++q) masks[((size_t)it * H + y) * words + q] = m[q];
    }
    __syncthreads();
    // phase B: compose this chunk's items over the owned pixels, painter's order
#pragma unroll
    for (int k = 0; k < 4; ++k) {
      int seg = tid + k * R_THREADS;
      if (k < my_segs && seg < segs) {
        int y = (seg * 16) / W, x0 = (seg * 16) % W;
        for (int it = 0; it < nit; ++it) {
          unsigned long long mw = masks[((size_t)it * H + y) * words + (x0 >> 6)];
          unsigned bits = (unsigned)(mw >> (x0 & 63)) & 0xFFFFu;
          if (!bits) continue;
          unsigned rgba = items[it].rgba;
          unsigned fg[3] = {rgba & 255u, (rgba >> 8) & 255u, (rgba >> 16) & 255u};
          unsigned al = rgba >> 24;
#pragma unroll
          for (int b = 0; b < 48; ++b) {
            int px = b / 3, ch = b % 3;
            if (bits & (1u << px)) {
              unsigned sh = (b & 3) * 8;
              unsigned old = (pr[k][b >> 2] >> sh) & 255u;
              unsigned nw = blend8(old, fg[ch], al);
              pr[k][b >> 2] = (pr[k][b >> 2] & ~(255u << sh)) | (nw << sh);
            }
          }
        }
      }
    }
  }
  __syncthreads();
  // phase C: stage the frame in LDS with rows flipped (np.flipud), then stream out
#pragma unroll
  for (int k = 0; k < 4; ++k) {
    int seg = tid + k * R_THREADS;
    if (k < my_segs && seg < segs) {
      int y = (seg * 16) / W, x0 = (seg * 16) % W;
      unsigned* dst = reinterpret_cast<unsigned*>(frame + ((size_t)(H - 1 - y) * W + x0) * 3);
#pragma unroll
      for (int d = 0; d < 12; ++d) dst[d] = pr[k][d];
    }
  }
  __syncthreads();
  const uint4* src = reinterpret_cast<const uint4*>(frame);
  uint4* out = reinterpret_cast<uint4*>(a.image + (size_t)env * H * W * 3);
  for (int i = tid; i < (H * W * 3) / 16; i += R_THREADS) out[i] = src[i];
}

// =====================================================================================
// host side: engine object + C ABI
// =====================================================================================
static thread_local std::string g_err;
static int fail(int code, const std::string& msg) { g_err = msg; return code; }
#define HIPCHK(x)                                                                        \
  do {                                                                                   \
    hipError_t err_ = (x);                                                               \
    if (err_ != hipSuccess)                                                              \
      return fail(MOOG_E_HIP, std::string(#x) + ": " + hipGetErrorString(err_));         \
  } while (0)

struct TimedKernel {
  std::vector<std::pair<hipEvent_t, hipEvent_t>> pending;
  double total_ms = 0;
  int64_t launches = 0;
};

struct moog_engine {
  moog_program_t prog;
  moog_layout_t L;
  moog_program_t* d_prog = nullptr;
  int16_t* d_vslot = nullptr;
  int32_t n_envs = 0;
  int device = 0;
  uint64_t seed = 0;
  int64_t env_index0 = 0;
  moog_state_view_t view{nullptr, nullptr};
  size_t step_lds = 0, raster_lds = 0;
  int raster_chunk = 0, raster_words = 0;
  bool timing = false;
  TimedKernel timed[MOOG_K_COUNT];
};

extern "C" {

int moog_abi_version(void) { return MOOG_ABI_VERSION; }
const char* moog_last_error(void) { return g_err.c_str(); }
int64_t moog_program_sizeof(void) { return (int64_t)sizeof(moog_program_t); }

static int validate(const moog_program_t* p) {
  if (!p) return fail(MOOG_E_INVALID, "null program");
  if (p->abi_version != MOOG_ABI_VERSION) return fail(MOOG_E_INVALID, "program abi_version mismatch");
  if (p->n_slots < 0 || p->n_slots > MOOG_MAX_SLOTS) return fail(MOOG_E_INVALID, "n_slots out of range");
  if (p->n_layers < 0 || p->n_layers > MOOG_MAX_LAYERS) return fail(MOOG_E_INVALID, "n_layers out of range");
  if (p->updates_per_env_step < 1) return fail(MOOG_E_INVALID, "updates_per_env_step < 1");
  for (int s = 0; s < p->n_slots; ++s)
    if (p->slot_vcap[s] > R_MAXE - 1) return fail(MOOG_E_UNSUPPORTED, "sprites with more than 31 vertices");
  for (int l = 0; l < p->n_layers; ++l)
    if (p->layer_nslots[l] > 64 * 2) return fail(MOOG_E_UNSUPPORTED, "layer too large");
  if (p->render.width % 16 != 0 || p->render.width > 256 || p->render.height > 1024 ||
      (p->render.width * p->render.height) / 16 > 4 * R_THREADS)
    return fail(MOOG_E_UNSUPPORTED, "render size unsupported (width % 16 == 0, <= 128x128)");
  return MOOG_OK;
}

int moog_engine_create(const moog_program_t* prog, int32_t n_envs, int32_t device_id, uint64_t seed,
                       int64_t env_index0, moog_engine_t** out) {
  if (!out) return fail(MOOG_E_INVALID, "null out");
  int rc = validate(prog);
  if (rc) return rc;
  if (n_envs <= 0) return fail(MOOG_E_INVALID, "n_envs <= 0");
  HIPCHK(hipSetDevice(device_id));
  moog_engine* e = new moog_engine();
  e->prog = *prog;
  moog_layout(prog, &e->L);
  e->n_envs = n_envs;
  e->device = device_id;
  e->seed = seed;
  e->env_index0 = env_index0;
  hipError_t err = hipMalloc(&e->d_prog, sizeof(moog_program_t));
  if (err != hipSuccess) { delete e; return fail(MOOG_E_NOMEM, "hipMalloc(program) failed"); }
  err = hipMemcpy(e->d_prog, prog, sizeof(moog_program_t), hipMemcpyHostToDevice);
  if (err != hipSuccess) { hipFree(e->d_prog); delete e; return fail(MOOG_E_HIP, "hipMemcpy(program) failed"); }
  {
    std::vector<int16_t> vs((size_t)(prog->n_total_verts > 0 ? prog->n_total_verts : 1), 0);
    for (int sl = 0; sl < prog->n_slots; ++sl)
      for (int k = 0; k < prog->slot_vcap[sl]; ++k) vs[prog->slot_voff[sl] + k] = (int16_t)sl;
    err = hipMalloc(&e->d_vslot, vs.size() * sizeof(int16_t));
    if (err == hipSuccess)
      err = hipMemcpy(e->d_vslot, vs.data(), vs.size() * sizeof(int16_t), hipMemcpyHostToDevice);
    if (err != hipSuccess) { hipFree(e->d_prog); delete e; return fail(MOOG_E_NOMEM, "vertex table"); }
  }
  e->step_lds = (size_t)e->L.f64_per_env * 8 + (size_t)e->L.i32_per_env * 4 +
                (size_t)e->L.S * 12 * 8;
  if (e->step_lds > 160 * 1024) {
    hipFree(e->d_prog); delete e;
    return fail(MOOG_E_UNSUPPORTED, "state record does not fit in 160 KB of LDS");
  }
  // raster LDS plan
  int W = prog->render.width, H = prog->render.height;
  e->raster_words = (W + 63) / 64;
  int ncopy = prog->render.polymod == MOOG_POLYMOD_TORUS ? 9 : 1;
  int max_items = prog->n_slots * ncopy;
  size_t fixed = (size_t)R_THREADS * R_XX * 4 + 16 + (((size_t)H * W * 3 + 15) & ~(size_t)15) +
                 (size_t)prog->n_slots * 4 + 64;
  size_t per_item = sizeof(RItem) + (size_t)R_MAXE * sizeof(REdge) + (size_t)H * e->raster_words * 8;
  size_t budget = 150 * 1024;
  int chunk = (int)((budget - fixed) / per_item);
  if (chunk < 1) { hipFree(e->d_prog); delete e; return fail(MOOG_E_UNSUPPORTED, "raster LDS plan"); }
  if (chunk > max_items) chunk = max_items > 0 ? max_items : 1;
  // keep two workgroups per CU when everything fits in one chunk of <= 64 KB
  e->raster_chunk = chunk;
  e->raster_lds = fixed + per_item * chunk;
  err = hipFuncSetAttribute(reinterpret_cast<const void*>(moog_step_kernel),
                            hipFuncAttributeMaxDynamicSharedMemorySize, (int)e->step_lds);
  if (err == hipSuccess)
    err = hipFuncSetAttribute(reinterpret_cast<const void*>(moog_reset_kernel),
                              hipFuncAttributeMaxDynamicSharedMemorySize, (int)e->step_lds);
  if (err == hipSuccess)
    err = hipFuncSetAttribute(reinterpret_cast<const void*>(moog_raster_kernel),
                              hipFuncAttributeMaxDynamicSharedMemorySize, (int)e->raster_lds);
  if (err != hipSuccess) {
    hipFree(e->d_prog); delete e;
    return fail(MOOG_E_HIP, std::string("hipFuncSetAttribute: ") + hipGetErrorString(err));
  }
  *out = e;
  return MOOG_OK;
}

static void drain(TimedKernel& t) {
  for (auto& pr : t.pending) {
    hipEventSynchronize(pr.second);
    float ms = 0;
    if (hipEventElapsedTime(&ms, pr.first, pr.second) == hipSuccess) { t.total_ms += ms; t.launches++; }
    hipEventDestroy(pr.first);
    hipEventDestroy(pr.second);
  }
  t.pending.clear();
}

int moog_engine_destroy(moog_engine_t* e) {
  if (!e) return MOOG_OK;
  for (int k = 0; k < MOOG_K_COUNT; ++k) drain(e->timed[k]);
  if (e->d_prog) hipFree(e->d_prog);
  if (e->d_vslot) hipFree(e->d_vslot);
  delete e;
  return MOOG_OK;
}

int moog_engine_layout(const moog_engine_t* e, moog_layout_t* out) {
  if (!e || !out) return fail(MOOG_E_INVALID, "null argument");
  *out = e->L;
  return MOOG_OK;
}

int moog_engine_load_state(moog_engine_t* e, const moog_state_view_t* view) {
  if (!e || !view || !view->f64 || !view->i32) return fail(MOOG_E_INVALID, "null state view");
  if (((uintptr_t)view->f64 & 15) || ((uintptr_t)view->i32 & 15))
    return fail(MOOG_E_INVALID, "state buffers must be 16-byte aligned");
  e->view = *view;
  return MOOG_OK;
}

struct Bracket {
  moog_engine* e; int id; hipStream_t s; hipEvent_t a = nullptr, b = nullptr;
  Bracket(moog_engine* e_, int id_, hipStream_t s_) : e(e_), id(id_), s(s_) {
    if (e->timing) { hipEventCreate(&a); hipEventCreate(&b); hipEventRecord(a, s); }
  }
  ~Bracket() {
    if (e->timing) { hipEventRecord(b, s); e->timed[id].pending.emplace_back(a, b); }
  }
};

static KArgs make_args(moog_engine* e, const void* actions, const moog_inject_t* inj,
                       const moog_step_out_t* out, int mode, const uint8_t* mask) {
  KArgs a;
  a.P = e->d_prog; a.L = e->L; a.f64 = e->view.f64; a.i32 = e->view.i32;
  a.actions = actions;
  a.inj = (inj && inj->uniforms) ? inj->uniforms : nullptr;
  a.inj_n = (inj && inj->uniforms) ? inj->per_env : 0;
  a.n_envs = e->n_envs; a.seed = e->seed; a.env_index0 = e->env_index0;
  a.mask = mask;
  a.reward = out ? out->reward : nullptr;
  a.discount = out ? out->discount : nullptr;
  a.step_type = out ? out->step_type : nullptr;
  a.mode = mode;
  a.vslot = e->d_vslot;
  return a;
}

static int launch_raster(moog_engine* e, uint8_t* image, hipStream_t s) {
  RArgs r;
  r.P = e->d_prog; r.L = e->L; r.f64 = e->view.f64; r.i32 = e->view.i32; r.image = image;
  r.n_envs = e->n_envs; r.items_per_chunk = e->raster_chunk; r.words_per_row = e->raster_words;
  {
    Bracket br(e, MOOG_K_RASTER, s);
    hipLaunchKernelGGL(moog_raster_kernel, dim3(e->n_envs), dim3(R_THREADS), e->raster_lds, s, r);
  }
  HIPCHK(hipGetLastError());
  return MOOG_OK;
}

static int ready(moog_engine* e) {
  if (!e) return fail(MOOG_E_INVALID, "null engine");
  if (!e->view.f64) return fail(MOOG_E_INVALID, "moog_engine_load_state has not been called");
  HIPCHK(hipSetDevice(e->device));
  return MOOG_OK;
}

int moog_engine_reset(moog_engine_t* e, const uint8_t* env_mask_dev, const moog_inject_t* inject,
                      const moog_step_out_t* out, void* hip_stream) {
  int rc = ready(e);
  if (rc) return rc;
  hipStream_t s = (hipStream_t)hip_stream;
  KArgs a = make_args(e, nullptr, inject, out, MODE_RESET_MASK, env_mask_dev);
  {
    Bracket br(e, MOOG_K_RESET, s);
    hipLaunchKernelGGL(moog_reset_kernel, dim3(e->n_envs), dim3(64), e->step_lds, s, a);
  }
  HIPCHK(hipGetLastError());
  if (out && out->image) return launch_raster(e, out->image, s);
  return MOOG_OK;
}

int moog_engine_step(moog_engine_t* e, const void* actions_dev, const moog_inject_t* inject,
                     const moog_step_out_t* out, void* hip_stream) {
  int rc = ready(e);
  if (rc) return rc;
  if (!actions_dev) return fail(MOOG_E_INVALID, "null actions");
  hipStream_t s = (hipStream_t)hip_stream;
  KArgs a = make_args(e, actions_dev, inject, out, MODE_RESET_AUTO, nullptr);
  {
    Bracket br(e, MOOG_K_RESET, s);
    hipLaunchKernelGGL(moog_reset_kernel, dim3(e->n_envs), dim3(64), e->step_lds, s, a);
  }
  a.mode = MODE_STEP;
  {
    Bracket br(e, MOOG_K_STEP, s);
    hipLaunchKernelGGL(moog_step_kernel, dim3(e->n_envs), dim3(64), e->step_lds, s, a);
  }
  HIPCHK(hipGetLastError());
  if (out && out->image) return launch_raster(e, out->image, s);
  return MOOG_OK;
}

int moog_engine_physics_only(moog_engine_t* e, const moog_inject_t* inject, void* hip_stream) {
  int rc = ready(e);
  if (rc) return rc;
  hipStream_t s = (hipStream_t)hip_stream;
  KArgs a = make_args(e, nullptr, inject, nullptr, MODE_PHYSICS, nullptr);
  {
    Bracket br(e, MOOG_K_STEP, s);
    hipLaunchKernelGGL(moog_step_kernel, dim3(e->n_envs), dim3(64), e->step_lds, s, a);
  }
  HIPCHK(hipGetLastError());
  return MOOG_OK;
}

int moog_engine_render(moog_engine_t* e, uint8_t* image_dev, void* hip_stream) {
  int rc = ready(e);
  if (rc) return rc;
  if (!image_dev) return fail(MOOG_E_INVALID, "null image");
  return launch_raster(e, image_dev, (hipStream_t)hip_stream);
}

int moog_engine_set_timing(moog_engine_t* e, int32_t enabled) {
  if (!e) return fail(MOOG_E_INVALID, "null engine");
  e->timing = enabled != 0;
  return MOOG_OK;
}

int moog_engine_kernel_time(moog_engine_t* e, int32_t kernel_id, double* total_ms, int64_t* launches) {
  if (!e || kernel_id < 0 || kernel_id >= MOOG_K_COUNT) return fail(MOOG_E_INVALID, "bad kernel id");
  drain(e->timed[kernel_id]);
  if (total_ms) *total_ms = e->timed[kernel_id].total_ms;
  if (launches) *launches = e->timed[kernel_id].launches;
  e->timed[kernel_id].total_ms = 0;
  e->timed[kernel_id].launches = 0;
  return MOOG_OK;
}

}  // extern "C"
