// Host model of the mask rasteriser (moog.github.io_amd/csrc/moog_raster_mask_core.h): the kernel's own phase functions,
// run thread by thread on the CPU with the barriers as loop boundaries.  Test infrastructure (tests/test_raster_mask_model.py):
// lets the algorithm be checked against the oracle renderer and the Pillow corpus without a GPU.
#include <stdlib.h>
#include <vector>

#define RM_STATS 1

#include "../../moog.github.io_amd/csrc/moog_raster_mask_core.h"

// p3 .. p5 of one pass, thread by thread
template <int WORDS, bool COMPACT>
static void model_pass(const RmArgs& a, const RmCtx& c, int env, int base, int end, int total_rows, int s_lo, int T, int waves) {
  for (int t = 0; t < T; ++t) rm_p3<WORDS, COMPACT>(a, c, base, end, s_lo, t, T);
  { RmSortKey sk; for (int t = 0; t < T; ++t) rm_p4a(c, total_rows, t, T, sk); for (int t = 0; t < T; ++t) rm_p4b(c, total_rows, t, T, sk); }
  for (int t = 0; t < T; ++t) rm_p4<WORDS, COMPACT>(a, c, total_rows, t, T, c.xx + (t / 64) * a.plan.xx_stride);
  if (a.big) for (int t = 0; t < T; ++t) rm_p4_big<WORDS, COMPACT>(a, c, t, T, c.xx + (t / 64) * a.plan.xx_stride, reinterpret_cast<uint8_t*>(c.xx + waves * a.plan.xx_stride));
  for (int t = 0; t < T; ++t) rm_p5<WORDS>(a, c, env, base == 0, s_lo, t, T);
}

extern "C" {

// One polygon of n integer canvas points on a W x H canvas -> cov[H][W] (0 / 1).  mode 0: the fast row routine with the
// generic one as its fallback (the kernel's behaviour), 1: generic only.  stats[0] += rows, stats[1] += generic rows.
int rm_model_polygon(const int* xy, int n, int W, int H, uint8_t* cov, int mode, long long* stats) {
  if (n < 1 || n > RM_MAX_NV || W > 128) return -1;   // (long polygons: rm_model_frames)
  std::vector<uint32_t> pv(n);
  int ymin = 0x7fffffff, ymax = -0x7fffffff;
  for (int k = 0; k < n; ++k) {
    const int x = rm_clamp16(xy[2 * k]), y = rm_clamp16(xy[2 * k + 1]);
    pv[k] = (uint32_t)(uint16_t)x | ((uint32_t)(uint16_t)y << 16);
    if (y < ymin) ymin = y;
    if (y > ymax) ymax = y;
  }
  const int pymax = ymax > H ? H : ymax;
  std::vector<RmEdge> edges(n);
  std::vector<RmRow> rows(H > 0 ? H : 1);
  for (auto& r : rows) { r.act = r.heads = r.tipP = r.tipN = 0u; }
  std::vector<char> shallow(H > 0 ? H : 1, 0);
  for (int k = 0; k < n; ++k) {
    RmEdge E = {0u, 0u, 0u, 0u};
    const int kind = rm_build_edge(pv.data(), k, n, &E);
    edges[k] = E;
    if (!kind) continue;
    const uint32_t bit = 1u << k;
    const int y0 = rm_y0(E), y1 = rm_y1(E);
    if (kind == 2) { if (y0 >= 0 && y0 < H) rows[y0].heads |= bit; continue; }
    const int emin = y0 < y1 ? y0 : y1, emax = y0 < y1 ? y1 : y0;
    for (int y = emin < 0 ? 0 : emin; y <= (emax > H - 1 ? H - 1 : emax); ++y) rows[y].act |= bit;
    const float dx = rm_u2f(E.w1);
    if (dx != 0.0f) {
      if (emin >= 0 && emin < H) { (dx > 0 ? rows[emin].tipP : rows[emin].tipN) |= bit; if (fabsf(dx) >= 1.49f) shallow[emin] = 1; }
      if (emax == pymax && emax >= 0 && emax < H) { (dx > 0 ? rows[emax].tipP : rows[emax].tipN) |= bit; if (fabsf(dx) >= 1.49f) shallow[emax] = 1; }
    }
  }
  float xx[RM_XX];
  for (int y = (ymin < 0 ? 0 : ymin); y <= (ymax > H - 1 ? H - 1 : ymax); ++y) {
    RmMask<2> m;
    bool ok = false;
    if (mode == 0) ok = rm_row_fast<2>(edges.data(), rows[y], y, pymax, W, shallow[y] != 0, m);
    if (!ok) { m = rm_row_generic<2>(edges.data(), n, rows[y].heads, y, pymax, xx); if (stats) stats[1]++; }
    if (stats) stats[0]++;
    for (int x = 0; x < W; ++x) cov[(size_t)y * W + x] = rm_bit(m, x) ? 1 : 0;
  }
  return 0;
}

// Whole frames of n_envs state records; image: [n_envs][H][Wpad][3].  sref_*: the reference record of the static prefix
// (or n_static = 0).  stats: [0] rows, [1] generic rows, [2] passes.  The records go through the emitter (rm_emit on the
// records as the ABI lays them out: what the derive kernel runs, and -- on the record in LDS -- the step kernel) into draw
// records, and the frame's phases read those.  draw_out (or null): the draw records, [n_envs][*draw_stride] bytes.
int rm_model_frames(const moog_program_t* P, const double* f64, const int32_t* i32, int n_envs, uint8_t* image,
                    int threads, int cap_rows, int n_static, int nsv, const double* sref_f64, const int32_t* sref_i32,
                    const uint8_t* sbg, const uint32_t* rgb_override, long long* stats, int compact) {
  (void)nsv;
  moog_layout_t L;
  moog_layout(P, &L);
  RmArgs a;
  memset(&a, 0, sizeof a);
  RmEmit em;
  memset(&em, 0, sizeof em);
  a.image = image;
  for (int sl = 0; sl < P->n_slots; ++sl) {
    if (P->slot_vcap[sl] > RM_BIG_NV) return -2;
    if (P->slot_vcap[sl] > RM_MAX_NV) a.big = 1;
  }
  em.ncopy = P->render.polymod == MOOG_POLYMOD_TORUS ? 9 : 1;
  a.n_envs = n_envs; em.slots = P->n_slots; em.S = a.S = P->n_slots * em.ncopy;
  if (a.S > 256) return -4;
  const int cw = P->render.width, ch = P->render.height;
  a.W = (cw + 15) & ~15; a.H = ch; a.flip = 1;
  em.W = a.W; em.H = a.H; em.scale_w = cw;
  if (a.W > 128 || a.H > 128 || P->render.aa > 1) return -3;
  a.cap_rows = cap_rows < a.H ? a.H : cap_rows;
  a.iwords = (a.S + 31) / 32; if (a.iwords < 1) a.iwords = 1;
  em.cmap = P->render.cmap;
  em.first_person = P->render.polymod == MOOG_POLYMOD_FIRST_PERSON;
  if (em.first_person) { em.fp_slot0 = P->layer_slot0[P->render.polymod_layer]; em.fp_nslots = P->layer_nslots[P->render.polymod_layer]; }
  a.bg = ((uint32_t)P->render.bg[0] & 255u) | (((uint32_t)P->render.bg[1] & 255u) << 8) | (((uint32_t)P->render.bg[2] & 255u) << 16);
  a.threads = threads;
  em.n_static = em.ncopy > 1 ? 0 : n_static;
  a.n_static = em.n_static;
  if (em.n_static > 0) {
    em.sref_v = sref_f64 + L.o_verts; em.sref_col = sref_f64 + L.o_color;
    em.sref_flags = sref_i32 + L.o_flags; em.sref_nv = sref_i32 + L.o_nverts; em.sref_opa = sref_i32 + L.o_opacity;
    a.sbg = sbg;
  }
  em.rgb_override = rgb_override;
  em.lay = rm_draw_layout(em.S, L.TOTV * em.ncopy);
  std::vector<uint8_t> draw((size_t)n_envs * em.lay.stride, 0xCD);   // (whatever the buffer held before: the record says how much of it counts)
  em.out = draw.data();
  a.draw = draw.data(); a.lay = em.lay;
  const int T = threads, waves = T / 64;
  a.compact = compact;
  rm_plan(a.S, L.TOTV * em.ncopy, a.W, a.H, a.cap_rows, a.iwords, waves, a.big, &a.plan, compact);
  std::vector<unsigned char> lds(a.plan.total + 64);
  const RmCtx c = rm_ctx(a.plan, lds.data());
  std::vector<uint32_t> vinfo((size_t)(L.TOTV > 0 ? L.TOTV : 1), 0u);
  for (int sl = 0; sl < P->n_slots; ++sl)
    for (int k = 0; k < P->slot_vcap[sl]; ++k) vinfo[P->slot_voff[sl] + k] = (uint32_t)sl | ((uint32_t)k << 8);
  std::vector<long long> scratch((RM_EMIT_SCRATCH_WORDS(em.slots, em.S, em.ncopy) + 1) / 2);
  for (int env = 0; env < n_envs; ++env) {
    RmSrcRecord src;
    src.P = P; src.L = &L; src.f = f64 + (size_t)env * L.f64_per_env; src.q = i32 + (size_t)env * L.i32_per_env; src.vi = vinfo.data();
    for (auto& w : scratch) w = (long long)0xA5A5A5A5A5A5A5A5ull;   // (whatever the LDS held)
    RmEmitScratch sc;
    rm_emit_scratch(reinterpret_cast<int32_t*>(scratch.data()), em.slots, em.ncopy, &sc);
    rm_emit(em, src, env, -1, sc, L.TOTV);
    memset(lds.data(), 0xA5, lds.size());   // LDS is not zero when a workgroup starts
    for (int t = 0; t < T; ++t) rm_load(a, c, env, t, T);
    const int s_lo = rm_s_lo(a, c);
    for (int base = 0;;) {
      const int end = rm_pass_end(a, c, base);
      const int total_rows = c.rowoff[end] - c.rowoff[base];
      if (!c.misc[6]) for (int w = 0; w < waves; ++w) rm_p2_assign(a, c, base, end, s_lo, -1);
      if (a.W > 64) { if (compact) model_pass<2, true>(a, c, env, base, end, total_rows, s_lo, T, waves); else model_pass<2, false>(a, c, env, base, end, total_rows, s_lo, T, waves); }
      else { if (compact) model_pass<1, true>(a, c, env, base, end, total_rows, s_lo, T, waves); else model_pass<1, false>(a, c, env, base, end, total_rows, s_lo, T, waves); }
      if (stats) { stats[0] += total_rows; stats[2]++; for (int q = 1; q < 16; ++q) if (q != 2) { if (q == 9) { if (rm_stats[q] > stats[q]) stats[q] = rm_stats[q]; } else stats[q] += rm_stats[q]; rm_stats[q] = 0; } }
      if (end >= a.S) break;
      base = end;
      for (int t = 0; t < T; ++t) rm_next_pass(a, c, t, T);
    }
  }
  return 0;
}

void rm_model_hist(long long* out) { for (int i = 0; i < 16; ++i) { out[i] = rm_hist[i]; rm_hist[i] = 0; } }

}  // extern "C"
