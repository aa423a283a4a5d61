// moog_raster_wave.h -- the WAVE rasteriser: one wavefront renders one env's frame from the env's draw list
// (moog_drawlist.h), with no workgroup barrier anywhere.  Same Pillow semantics, bit for bit, as the workgroup
// rasteriser of moog_raster_kernel.h (ImagingDrawPolygon / polygon_generic(hasAlpha) / hline32rgba as restated in
// oracle/moog_oracle.c; reference moog/observers/pil_renderer.py:88-120) -- the per-edge, per-row and per-segment
// routines are the same functions -- but a different division of labour:
//
//   * input is the draw list: live vertices only, already integer canvas points, 64 per ROUND, a polygon never
//     straddling rounds.  The f64 record is read for the colours only.
//   * a round is processed start to end by the wave's 64 lanes (lane = vertex = the edge leaving it): item row
//     ranges, row records allocated for the round's polygons, edge records, corner fix-up partners, crossing pushes.
//     Everything a lane needs from its polygon's other vertices / edges is in LDS written by the same wave a moment
//     ago: the only synchronisation is the wave's own LDS ordering (wsync()).
//   * then rows (lane = (item, row): sorting network + Pillow's span loop -> coverage mask), the rare rows
//     (several heads / generic), and compose (lane = 16-pixel segment).
//   * frames whose polygons need more row or edge records than the LDS plan holds take several PASSES over item
//     ranges; the partially composed frame round-trips through the output (L2), as in the workgroup kernel.
//
// Eligible programs (moog_engine.hip): one tile (canvas <= 128 x 128), no anti-aliasing, no
// polygon modifier, <= 64 sprite slots, <= 32 vertices per sprite.  Everything else takes the workgroup kernel.
#ifndef MOOG_RASTER_WAVE_H_
#define MOOG_RASTER_WAVE_H_
#include "moog_raster_kernel.h"
#include "moog_drawlist.h"

// Generic scanline with the crossing list at stride STRIDE (scanline_mask_generic uses R_SLOW)
template <int STRIDE>
__device__ inline RMask scanline_mask_generic_s(const RPoly& p, int y, int poly_ymax, float* xx, int W, const int R_XX) {
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
    if (j < R_XX) xx[j * STRIDE] = x;
    ++j;
    if (y == emax && y < poly_ymax) {
      if (j < R_XX) xx[j * STRIDE] = x;
      ++j;
    } else if (dx != 0.0f && (y == emin || y == emax)) {
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
        if (kpos < R_XX) xx[kpos * STRIDE] = (float)vv;
      }
    }
  }
  if (j > R_XX) j = R_XX;
  for (int q = 1; q < j; ++q) {
    float key = xx[q * STRIDE];
    int r = q - 1;
    while (r >= 0 && xx[r * STRIDE] > key) { xx[(r + 1) * STRIDE] = xx[r * STRIDE]; --r; }
    xx[(r + 1) * STRIDE] = key;
  }
  int x_pos = (j == 0) ? -1 : 0;
  for (int i = 1; i < j; i += 2) {
    int x_end = pil_round_down(xx[i * STRIDE]);
    if (x_end < x_pos) continue;
    draw_horizontal(p, y, &x_pos, m, W, 0);
    if (x_end < x_pos) continue;
    int x_start = pil_round_up(xx[(i - 1) * STRIDE]);
    if (x_pos > x_start) {
      x_start = x_pos;
      if (x_end < x_start) continue;
    }
    mask_fill(m, W, 0, x_start, x_end);
    x_pos = x_end + 1;
  }
  draw_horizontal(p, y, &x_pos, m, W, 0);
  return m;
}

// inclusive prefix sum over the wave's 64 lanes (DPP row shifts + the two row broadcasts: no LDS round trip)
__device__ __forceinline__ int wave_incl_scan(int v) {
  v += __builtin_amdgcn_update_dpp(0, v, 0x111, 0xf, 0xf, false);   // row_shr:1
  v += __builtin_amdgcn_update_dpp(0, v, 0x112, 0xf, 0xf, false);   // row_shr:2
  v += __builtin_amdgcn_update_dpp(0, v, 0x114, 0xf, 0xf, false);   // row_shr:4
  v += __builtin_amdgcn_update_dpp(0, v, 0x118, 0xf, 0xf, false);   // row_shr:8
  v += __builtin_amdgcn_update_dpp(0, v, 0x142, 0xa, 0xf, false);   // row_bcast:15 -> rows 1, 3
  v += __builtin_amdgcn_update_dpp(0, v, 0x143, 0xc, 0xf, false);   // row_bcast:31 -> rows 2, 3
  return v;
}

// One crossing of table edge E with row y (push_crossing of moog_raster_kernel.h without the queues: a row that needs the
// generic routine only gets its flag, the row stage finds it).
__device__ __forceinline__ void push_row(RRow* rows, int rb, const REdge& E, int emin, int emax, int pymax, short vtop, short vbot, int y) {
  float x = (float)(y - (int)E.y0) * E.dx + (float)E.x0;
  const bool bot = (y == emax);
  const bool dup = bot && (y < pymax);
  const short vv = (y == emin) ? vtop : ((bot && !dup) ? vbot : R_NONE);
  const unsigned n = dup ? 2u : 1u;
  if (vv != R_NONE) x = (float)vv;
  const unsigned key = (unsigned)(pil_round_up(x) + pil_round_down(x) + R_KEY_BIAS);
  RRow* r = rows + (rb + y);
  const unsigned pos = atomicAdd(&r->cnt, n) & R_CNT_MASK;
  if (pos < R_CAP) r->key[pos] = (unsigned short)key;
  if (n == 2u && pos + 1 < R_CAP) r->key[pos + 1] = (unsigned short)key;
  bool gen = !(fabsf(x) <= R_XLIM) || (pos <= R_CAP && pos + n > R_CAP);
  if (vv != R_NONE) {
    const int tipx = (y == E.y0) ? E.x0 : E.x1;
    const unsigned bit = R_FIX_ONE << (tipx & 7);
    gen = gen || (atomicOr(&r->cnt, bit) & bit);
  }
  if (gen) atomicOr(&r->cnt, R_GENERIC);
}

__device__ __forceinline__ void rw_barrier() {
  if (RW_WAVES > 1) __syncthreads(); else wsync();
}

template <int WORDS>
__device__ __forceinline__ void raster_wave(const RWArgs& a, const int env) {
  PProg P = as_const_prog(a.P);
  const int W = a.W, H = a.H, nseg = W >> 4;
  const int tid = (int)threadIdx.x, lane = tid & 63;
  const int wv = uni(tid >> 6);
  const uint32_t* __restrict__ dl = a.dl + (size_t)env * a.dl_stride;
  const double* gf = a.f64 + (size_t)env * a.L.f64_per_env;
  const int32_t* gq = a.i32 + (size_t)env * a.L.i32_per_env;
  const RWPlan& pl = a.plan;
  const int e_cap = pl.e_cap, r_cap = pl.r_cap;
  const int E_ROUNDS = e_cap >> 6;
  REdge* edges = reinterpret_cast<REdge*>(moog_lds + pl.o_edge);
  RRow* rows = reinterpret_cast<RRow*>(moog_lds + pl.o_rows);
  int* item_y = reinterpret_cast<int*>(moog_lds + pl.o_item_y);
  unsigned* item_rgba = reinterpret_cast<unsigned*>(moog_lds + pl.o_rgba);
  int* rowbase = reinterpret_cast<int*>(moog_lds + pl.o_rowbase);
  unsigned* iinfo = reinterpret_cast<unsigned*>(moog_lds + pl.o_iinfo);
  unsigned long long* rowitems = reinterpret_cast<unsigned long long*>(moog_lds + pl.o_rowitems);
  uint8_t* rowitem = reinterpret_cast<uint8_t*>(moog_lds + pl.o_rowitem);
  unsigned short* pend = reinterpret_cast<unsigned short*>(moog_lds + pl.o_pend) + RW_PEND * wv;   // this wave's rare rows
  const int long_words = ((RW_LONG + RW_VLONG) > a.xxcap * RW_SLOW ? (RW_LONG + RW_VLONG) : a.xxcap * RW_SLOW);
  unsigned* longlist = reinterpret_cast<unsigned*>(moog_lds + pl.o_long) + ((long_words + 3) & ~3) * wv;   // this wave's long edges (the others push their rows themselves)
  float* xxs = reinterpret_cast<float*>(longlist);
  int* misc = reinterpret_cast<int*>(moog_lds + pl.o_misc);      // shared: [5] / [6] the prefix's entries / colours differ
  int* wmisc = misc + 8 + 8 * wv;                                // per wave: [1] long list, [4] very long edges, [7] spare
  unsigned short* dummy = reinterpret_cast<unsigned short*>(moog_lds + pl.o_dummy) + 64 * wv + lane;   // where masked-off key stores go

  // ---- header, colours, tables -----------------------------------------------------------------------------------
  const int n_rounds = uni((int)dl[0]);
  const int n_items = uni((int)dl[1]);
  // lanes used / first item of round r: held by lane r, fetched with readlane (r is wave uniform)
  const int my_cnt = lane < DL_MAX_ROUNDS ? (int)reinterpret_cast<const uint8_t*>(dl + 4)[lane] : 0;
  const int my_first = lane < DL_MAX_ROUNDS ? (int)reinterpret_cast<const uint8_t*>(dl + 8)[lane] : 255;
  auto round_lanes = [&](int r) -> int { return __builtin_amdgcn_readlane(my_cnt, r & (DL_MAX_ROUNDS - 1)); };
  auto round_first = [&](int r) -> int {   // first item of round r (n_items behind the last round)
    return r >= n_rounds ? n_items : __builtin_amdgcn_readlane(my_first, r & (DL_MAX_ROUNDS - 1));
  };
  struct Entry { uint4 e; unsigned info; };   // an edge record and its info word
  auto load_entry = [&](int r) -> Entry {
    Entry en;
    en.e = make_uint4(0u, 0u, 0u, 0u); en.info = 0u;
    if (r < n_rounds && lane < round_lanes(r)) { en.e = dl_edges(dl, r)[lane]; en.info = dl_infos(dl, r)[lane]; }
    return en;
  };
  // the first rounds of this wave: on their way while the tables are set up
  Entry en_a = load_entry(wv), en_b = load_entry(wv + RW_WAVES), en_c = load_entry(wv + 2 * RW_WAVES);
  const unsigned iyw = lane < n_items ? dl[28 + lane] : 0u;
  const int NS = a.n_static;
  if (wv == RW_WAVES - 1) {   // colours: lane = sprite slot
    bool st_bad = false;
    const int s = lane;
    if (s < a.L.S) {
      const unsigned g = reinterpret_cast<const uint8_t*>(dl + 12)[s];
      // (every load of the slot goes out at once: one trip to HBM, not one per dependent step)
      const int opa = gq[a.L.o_opacity + s];
      const double* col = gf + a.L.o_color + 3 * s;
      const double c0 = col[0], c1 = col[1], c2 = col[2];
      if (s < NS) {
        const double* rc = a.sref_col + 3 * s;
        st_bad = g != (unsigned)s || opa != a.sref_opa[s] || __double_as_longlong(c0) != __double_as_longlong(rc[0]) ||
                 __double_as_longlong(c1) != __double_as_longlong(rc[1]) || __double_as_longlong(c2) != __double_as_longlong(rc[2]);
      }
      if (g != 255u) {
        unsigned r8, g8, b8;
        if (P->render.cmap == MOOG_CMAP_HSV) hsv_to_rgb_u8(c0, c1, c2, r8, g8, b8);
        else { r8 = (unsigned)(int)c0 & 255u; g8 = (unsigned)(int)c1 & 255u; b8 = (unsigned)(int)c2 & 255u; }
        item_rgba[g] = r8 | (g8 << 8) | (b8 << 16) | (((unsigned)opa & 255u) << 24);
      }
    }
    const bool any_bad = __any(st_bad);
    if (lane == 0) misc[6] = any_bad ? 1 : 0;
  }
  if (wv == 0) {   // row ranges of the items (from the list), the prefix's entries against the reference's
    if (lane < n_items) { item_y[2 * lane] = (short)(iyw & 0xffffu); item_y[2 * lane + 1] = (short)(iyw >> 16); }
    bool bad = false;
    if (NS > 0 && lane < a.nsl) {
      const uint4 re = dl_edges(a.sref_dl, 0)[lane];
      const unsigned ri = dl_infos(a.sref_dl, 0)[lane];
      bad = lane >= round_lanes(0) || re.x != en_a.e.x || re.y != en_a.e.y || re.z != en_a.e.z || re.w != en_a.e.w || ri != en_a.info;
    }
    const bool any_bad = __any(bad);
    if (lane == 0) misc[5] = any_bad ? 1 : 0;
  }
  for (int i = tid; i < r_cap; i += RW_THREADS) {
    uint4* r = reinterpret_cast<uint4*>(rows + i);
    r[0] = make_uint4(~0u, ~0u, ~0u, ~0u);
    r[1] = make_uint4(~0u, ~0u, 0u, 0u);
  }
  for (int y = tid; y < H; y += RW_THREADS) rowitems[y] = 0ull;
  if (lane < 8) wmisc[lane] = 0;
  rw_barrier();
  const bool prefix_ok = NS > 0 && n_items >= NS && n_rounds > 0 && misc[5] == 0 && misc[6] == 0;
  const unsigned bgx = ((unsigned)P->render.bg[0] & 255u) | (((unsigned)P->render.bg[1] & 255u) << 8) |
                       (((unsigned)P->render.bg[2] & 255u) << 16);
  uint8_t* out = a.image + (size_t)env * H * W * 3;
  const int segs = H * nseg;

  int g_lo = prefix_ok ? NS : 0;      // items below this one are drawn (earlier passes) or in the cached picture
  const bool from_cache = prefix_ok;
  bool first_pass = true;
  for (;;) {   // ---- passes (one, unless the frame needs more row / edge records than the plan holds) --------------
    // ---- row records of the pass's items (lane = item; both waves compute the same) ---------------------------
    int r_lo = 0;
    while (r_lo + 1 < n_rounds && round_first(r_lo + 1) <= g_lo) ++r_lo;
    if (!first_pass || r_lo != 0) { en_a = load_entry(r_lo + wv); en_b = load_entry(r_lo + wv + RW_WAVES); en_c = load_entry(r_lo + wv + 2 * RW_WAVES); }
    const int g_edge = round_first(r_lo + E_ROUNDS);   // the items behind it have no edge records in this pass
    int g_hi, total_rows;
    {
      const int g = lane;
      int cnt = 0, ystart = 0;
      const bool mine = g >= g_lo && g < n_items;
      if (mine) {
        const int iy0 = item_y[2 * g], iy1 = item_y[2 * g + 1];
        const int y0 = iy0 < 0 ? 0 : iy0, y1 = iy1 > H - 1 ? H - 1 : iy1;   // rows >= H draw nothing (hline clips)
        cnt = y1 >= y0 ? y1 - y0 + 1 : 0;
        ystart = y0;
      }
      const int inc = wave_incl_scan(cnt);
      const unsigned long long stop = __ballot(mine && (inc > r_cap || g >= g_edge));   // (an item alone always fits: r_cap >= H)
      g_hi = stop ? __ffsll((long long)stop) - 1 : n_items;
      total_rows = g_hi > g_lo ? __builtin_amdgcn_readlane(inc, g_hi > 0 ? g_hi - 1 : 0) : 0;
      if (mine && g < g_hi) {
        rowbase[g] = inc - cnt - ystart;
        if (wv == 0)   // the item of every row record of the pass
          for (int w = inc - cnt; w < inc; ++w) rowitem[w] = (uint8_t)g;
      }
    }
    wsync();
    int r_hi = r_lo;
    while (r_hi + 1 < n_rounds && round_first(r_hi + 1) < g_hi) ++r_hi;

    // ---- the wave's rounds: the edges of the list mark their heads' rows and push their crossings --------------------
    if (a.debug_stop != 2 && g_hi > g_lo)
    for (int r = r_lo + wv; r <= r_hi; r += RW_WAVES) {
      const Entry en = en_a;
      en_a = en_b; en_b = en_c; en_c = load_entry(r + 3 * RW_WAVES);   // (three rounds ahead)
      const bool valid = lane < round_lanes(r);
      const int g = valid ? (int)(en.info & 255u) : 255, k = (int)((en.info >> 8) & 255u), nv = valid ? (int)((en.info >> 16) & 127u) : 0;
      const bool active = valid && g >= g_lo && g < g_hi;
      const int ebase = (r - r_lo) * 64;
      REdge E;
      E.x0 = (short)(en.e.x & 0xffffu); E.y0 = (short)(en.e.x >> 16); E.x1 = (short)(en.e.y & 0xffffu); E.y1 = (short)(en.e.y >> 16);
      E.dx = __uint_as_float(en.e.z); E.vtop = (short)(en.e.w & 0xffffu); E.vbot = (short)(en.e.w >> 16);
      const int x0 = E.x0, y0 = E.y0, x1 = E.x1, y1 = E.y1;
      const bool tbl = active && y0 != y1;
      const bool head = active && (en.info & DL_INFO_HEAD) != 0u;
      const int iy1 = active ? item_y[2 * g + 1] : 0;
      const int rb = active ? rowbase[g] : 0;
      if (active) {
        *reinterpret_cast<uint4*>(edges + ebase + lane) = en.e;
        if (k == 0) iinfo[g] = (unsigned)(ebase + lane) | ((unsigned)nv << 16);
      }
      if (a.debug_stop == 3) continue;
      // ---- horizontal heads mark their row ----------------------------------------------------------------------
      if (__any(head)) {
        if (head && y0 >= 0 && y0 < H) atomicOr(&rows[rb + y0].hbits, 1u << (k & 31));
      }
      // ---- the crossings of the edge's first four rows -----------------------------------------------------------
      const int pymax = iy1 > H ? H : iy1;    // polygon_generic clamps ymax to ysize
      const int emin = y0 < y1 ? y0 : y1, emax = y0 < y1 ? y1 : y0;
      const short vtop = E.vtop, vbot = E.vbot;
      const int ya = emin < 0 ? 0 : emin, yb = emax > H - 1 ? H - 1 : emax;
      unsigned pos[4], key[4], nn[4];
      bool on[4], fix[4], gen[4];
#pragma unroll
      for (int j = 0; j < 4; ++j) {
        const int y = ya + j;
        on[j] = tbl && y <= yb;
        float x = (float)(y - y0) * E.dx + (float)x0;
        const bool bot = (y == emax);
        const bool dup = bot && (y < pymax);      // polygon_generic: an edge's last row counts twice
        const short vv = (y == emin) ? vtop : ((bot && !dup) ? vbot : R_NONE);
        nn[j] = dup ? 2u : 1u;
        if (vv != R_NONE) x = (float)vv;
        key[j] = (unsigned)(pil_round_up(x) + pil_round_down(x) + R_KEY_BIAS);
        fix[j] = on[j] && vv != R_NONE;
        gen[j] = on[j] && !(fabsf(x) <= R_XLIM);
        pos[j] = atomicAdd(on[j] ? &rows[rb + y].cnt : reinterpret_cast<unsigned*>(wmisc + 7), on[j] ? nn[j] : 0u);
      }
      bool any_fix = false, any_gen = false;
#pragma unroll
      for (int j = 0; j < 4; ++j) {
        const unsigned p = pos[j] & R_CNT_MASK;
        unsigned short* kp = on[j] ? rows[rb + ya + j].key : dummy;
        const unsigned pa = on[j] ? (p < R_CAP - 1 ? p : R_CAP - 1) : 0u;
        const unsigned pb = on[j] ? (p + nn[j] - 1u < R_CAP - 1 ? p + nn[j] - 1u : R_CAP - 1) : 0u;
        kp[pa] = (unsigned short)key[j];   // (a row with more than R_CAP keys is drawn by the generic routine: its keys do not matter)
        kp[pb] = (unsigned short)key[j];
        gen[j] = gen[j] || (on[j] && p <= R_CAP && p + nn[j] > R_CAP);
        any_fix = any_fix || fix[j];
        any_gen = any_gen || gen[j];
      }
      if (__any(any_fix)) {
        // Two fix-ups on one row are independent unless they belong to the same tip point (then the reference overwrites
        // one partner entry twice): the row remembers the tip columns mod 8.  (Only an edge's first and last row have one.)
#pragma unroll
        for (int j = 0; j < 4; ++j) {
          if (fix[j]) {
            const int y = ya + j;
            const int tipx = (y == y0) ? x0 : x1;
            const unsigned bit = R_FIX_ONE << (tipx & 7);
            gen[j] = gen[j] || (atomicOr(&rows[rb + y].cnt, bit) & bit);
            any_gen = any_gen || gen[j];
          }
        }
      }
      if (__any(any_gen)) {
#pragma unroll
        for (int j = 0; j < 4; ++j)
          if (gen[j]) atomicOr(&rows[rb + ya + j].cnt, R_GENERIC);
      }
      if (tbl && yb - ya >= 4) {   // rows 5+: listed for lane groups (below), or pushed here when the list is full
        const unsigned entry = (unsigned)(ebase + lane) | ((unsigned)g << 16);
        const bool vlong = yb - ya >= 12;
        const int li = atomicAdd(vlong ? &wmisc[4] : &wmisc[1], 1);
        if (li < (vlong ? RW_VLONG : RW_LONG)) longlist[vlong ? RW_LONG + li : li] = entry;
        else for (int y = ya + 4; y <= yb; ++y) push_row(rows, rb, E, emin, emax, pymax, vtop, vbot, y);
      }
    }
    wsync();
    // ---- the remaining rows of the wave's long edges: eight lanes per edge with up to 12 rows, the whole wave per longer edge
    if (a.debug_stop != 2 && a.debug_stop != 3 && a.debug_stop != 32) {
      const int nlong = wmisc[1] < RW_LONG ? wmisc[1] : RW_LONG, nvlong = wmisc[4] < RW_VLONG ? wmisc[4] : RW_VLONG, sub = lane & 7;
      const int ngroups = (nlong + 7) >> 3;
      for (int u = 0; u < ngroups + nvlong; ++u) {
        const bool vl = (u >= ngroups);
        const int li = vl ? u - ngroups : u * 8 + (lane >> 3);
        if (!vl && li >= nlong) continue;
        const unsigned entry = vl ? longlist[RW_LONG + li] : longlist[li];
        const int g = (int)(entry >> 16);
        const REdge E = edges[entry & 0xffffu];
        const int rb = rowbase[g];
        const int iymax = item_y[2 * g + 1];
        const int pymax = iymax > H ? H : iymax;
        const int emin = E.y0 < E.y1 ? E.y0 : E.y1, emax = E.y0 < E.y1 ? E.y1 : E.y0;
        const int ya = emin < 0 ? 0 : emin, yb = emax > H - 1 ? H - 1 : emax;
        if (vl) {
          for (int y = ya + 4 + lane; y <= yb; y += 64) push_row(rows, rb, E, emin, emax, pymax, E.vtop, E.vbot, y);
        } else if (ya + 4 + sub <= yb) {
          push_row(rows, rb, E, emin, emax, pymax, E.vtop, E.vbot, ya + 4 + sub);
        }
      }
    }
    rw_barrier();

    // ---- coverage masks, one lane per row record; rows with several horizontal heads and rows for the generic scanline
    //      (more than 12 crossings, colliding fix-ups, far off-canvas crossings) are set aside in the wave's lists and
    //      drawn together (once a list holds more than a round's worth, and at the end) -------------------------------
    if (a.debug_stop == 0 || a.debug_stop == 5 || a.debug_stop == 31) {
      unsigned short* pend_m = pend;
      unsigned short* pend_s = pend + RW_PEND / 2;
      int n_multi = 0, n_slow = 0;
      for (int w0 = 64 * wv;;) {
        const bool main = w0 < total_rows;
        if (main) {
          const int w = w0 + lane;
          unsigned k[16];
          int cnt = 0, g = 0, hxmin = 0, hxmax = 0;
          unsigned hbits = 0u;
          bool head = false, slow = false, multi = false;
          const bool live = w < total_rows;
          if (live) {
            const uint4* rr = reinterpret_cast<const uint4*>(rows + w);
            const uint4 q0 = rr[0], q1 = rr[1];
            g = rowitem[w];
            k[0] = q0.x & 0xffffu; k[1] = q0.x >> 16; k[2] = q0.y & 0xffffu; k[3] = q0.y >> 16;
            k[4] = q0.z & 0xffffu; k[5] = q0.z >> 16; k[6] = q0.w & 0xffffu; k[7] = q0.w >> 16;
            k[8] = q1.x & 0xffffu; k[9] = q1.x >> 16; k[10] = q1.y & 0xffffu; k[11] = q1.y >> 16;
            hbits = q1.z;
            const unsigned cw = q1.w;
            cnt = cw & R_CNT_MASK;
            slow = (cw & R_GENERIC) != 0u;
            head = !slow && hbits != 0u && (hbits & (hbits - 1u)) == 0u;
            multi = !slow && hbits != 0u && !head;
          } else {
#pragma unroll
            for (int q = 0; q < R_CAP; ++q) k[q] = 0xffffu;
          }
#pragma unroll
          for (int q = R_CAP; q < 16; ++q) k[q] = 0xffffffffu;
          if (slow || multi) cnt = 0;
          const int y = live ? w - rowbase[g] : 0;
          if (__any(head)) {
            if (head) {   // the row's one head: its record
              const unsigned xb = (unsigned)__float_as_int(edges[(iinfo[g] & 0xffffu) + __ffs((int)hbits) - 1].dx);
              hxmin = (short)(xb & 0xffffu); hxmax = (short)(xb >> 16);
            }
          }
          RMask m;
          if (__any(cnt > 8)) {
            sort_network<16>(k);
            m = span_loop<R_CAP / 2, 16, WORDS>(k, cnt, head, hxmin, hxmax, W, 0);
          } else {
            unsigned k8[8] = {k[0], k[1], k[2], k[3], k[4], k[5], k[6], k[7]};
            sort_network<8>(k8);
            m = span_loop<4, 8, WORDS>(k8, cnt, head, hxmin, hxmax, W, 0);
          }
          if (live && !slow && !multi) {
            unsigned long long* mp = reinterpret_cast<unsigned long long*>(rows + w);
            mp[0] = m.w0;
            if (WORDS > 1) mp[1] = m.w1;
            if ((m.w0 | (WORDS > 1 ? m.w1 : 0ull)) != 0ull) atomicOr(&rowitems[y], 1ull << g);
          }
          const unsigned long long mm = __ballot(multi), ms = __ballot(slow);
          if (mm | ms) {
            const unsigned long long below = (1ull << lane) - 1ull;
            if (multi) pend_m[n_multi + __popcll(mm & below)] = (unsigned short)w;
            if (slow) pend_s[n_slow + __popcll(ms & below)] = (unsigned short)w;
            n_multi += __popcll(mm); n_slow += __popcll(ms);
            wsync();
          }
          w0 += RW_THREADS;
        }
        const bool last = w0 >= total_rows;
        if (n_multi > 0 && (last || n_multi > RW_PEND / 2 - 64)) {   // rows with several heads: same keys, heads through the row's head bits
          for (int qi = lane; qi < n_multi; qi += 64) {
            const int w = pend_m[qi];
            const uint4* rr = reinterpret_cast<const uint4*>(rows + w);
            const uint4 q0 = rr[0], q1 = rr[1];
            unsigned k[16];
            k[0] = q0.x & 0xffffu; k[1] = q0.x >> 16; k[2] = q0.y & 0xffffu; k[3] = q0.y >> 16;
            k[4] = q0.z & 0xffffu; k[5] = q0.z >> 16; k[6] = q0.w & 0xffffu; k[7] = q0.w >> 16;
            k[8] = q1.x & 0xffffu; k[9] = q1.x >> 16; k[10] = q1.y & 0xffffu; k[11] = q1.y >> 16;
#pragma unroll
            for (int q = R_CAP; q < 16; ++q) k[q] = 0xffffffffu;
            const int cnt = q1.w & R_CNT_MASK, g = rowitem[w];
            const int y = w - rowbase[g];
            const unsigned hbits = q1.z;   // the heads on this row (the polygon has <= 32 edges)
            const REdge* pe = edges + (iinfo[g] & 0xffffu);
            RMask m;
            if (__any(cnt > 8)) {
              sort_network<16>(k);
              m = span_loop_pending<R_CAP / 2, 16, WORDS>(k, cnt, pe, hbits, W, 0);
            } else {
              unsigned k8[8] = {k[0], k[1], k[2], k[3], k[4], k[5], k[6], k[7]};
              sort_network<8>(k8);
              m = span_loop_pending<4, 8, WORDS>(k8, cnt, pe, hbits, W, 0);
            }
            unsigned long long* mp = reinterpret_cast<unsigned long long*>(rows + w);
            mp[0] = m.w0;
            if (WORDS > 1) mp[1] = m.w1;
            if ((m.w0 | (WORDS > 1 ? m.w1 : 0ull)) != 0ull) atomicOr(&rowitems[y], 1ull << g);
          }
          n_multi = 0;
          wsync();
        }
        if (n_slow > 0 && (last || n_slow > RW_PEND / 2 - 64)) {   // (the wave's long-edge list is dead: its words are the crossing list)
          for (int qi = lane < RW_SLOW ? lane : n_slow; qi < n_slow; qi += RW_SLOW) {
            const int w = pend_s[qi];
            const int g = rowitem[w];
            const int y = w - rowbase[g];
            const int iymax = item_y[2 * g + 1];
            const int pymax = iymax > H ? H : iymax;
            const unsigned ii = iinfo[g];
            RPoly poly = {edges + (ii & 0xffffu), (int)(ii >> 16), nullptr, 1, rows[w].hbits};
            const RMask m = scanline_mask_generic_s<RW_SLOW>(poly, y, pymax, xxs + lane, W, a.xxcap);
            unsigned long long* mp = reinterpret_cast<unsigned long long*>(rows + w);
            mp[0] = m.w0;
            if (WORDS > 1) mp[1] = m.w1;
            if ((m.w0 | (WORDS > 1 ? m.w1 : 0ull)) != 0ull) atomicOr(&rowitems[y], 1ull << g);
          }
          n_slow = 0;
          wsync();
        }
        if (last) break;
      }
    }
    rw_barrier();

    // ---- compose (painter's order = item order), pack RGB, store flipped.  A 16-pixel segment of a row without any
    //      item is written straight from the cached picture / the background colour; the others are listed (the edge
    //      records are dead: their words hold the wave's list) and composed densely --------------------------------------
    if (a.debug_stop == 0 || a.debug_stop == 31) {
      const int tl_elems = (e_cap * 8 > segs * 2 ? e_cap * 8 : segs * 2) / RW_WAVES;   // (raster_wave_plan: the larger of the two uses)
      unsigned short* tlist = reinterpret_cast<unsigned short*>(moog_lds + pl.o_edge) + (size_t)tl_elems * wv;
      const unsigned bg0 = (bgx & 0xFFFFFFu) | (bgx << 24), bg1 = ((bgx >> 8) & 0xFFFFu) | (bgx << 16), bg2 = ((bgx >> 16) & 0xFFu) | (bgx << 8);
      int nt = 0;
      for (int s0 = 64 * wv; s0 < segs; s0 += RW_THREADS) {
        const int seg = s0 + lane;
        bool touched = false;
        if (seg < segs) {
          const int y = div_small(seg, nseg), sg = seg - y * nseg, x0 = sg * 16;
          touched = rowitems[y] != 0ull;
          if (!touched && first_pass) {
            uint4* dst = reinterpret_cast<uint4*>(out + ((size_t)(H - 1 - y) * W + x0) * 3);
            uint4 c0 = make_uint4(bg0, bg1, bg2, bg0), c1 = make_uint4(bg1, bg2, bg0, bg1), c2 = make_uint4(bg2, bg0, bg1, bg2);
            if (from_cache) {
              const uint4* src = reinterpret_cast<const uint4*>(a.sbg + ((size_t)(H - 1 - y) * W + x0) * 3);
              c0 = src[0]; c1 = src[1]; c2 = src[2];
            }
            dst[0] = c0; dst[1] = c1; dst[2] = c2;
          }
        }
        const unsigned long long tm = __ballot(touched);
        if (touched) tlist[nt + __popcll(tm & ((1ull << lane) - 1ull))] = (unsigned short)seg;
        nt += __popcll(tm);
      }
      wsync();
      for (int t0 = 0; t0 < nt; t0 += 64) {
        if (t0 + lane >= nt) continue;
        const int seg = tlist[t0 + lane];
        const int y = div_small(seg, nseg), sg = seg - y * nseg, x0 = sg * 16;
        uint4* dst = reinterpret_cast<uint4*>(out + ((size_t)(H - 1 - y) * W + x0) * 3);
        unsigned long long bitsw = rowitems[y];
        unsigned px[16];
        if (first_pass && !from_cache) {
#pragma unroll
          for (int i = 0; i < 16; ++i) px[i] = bgx;
        } else {  // continue from the previous pass, or from the cached picture of the static prefix
          const uint4* src = first_pass ? reinterpret_cast<const uint4*>(a.sbg + ((size_t)(H - 1 - y) * W + x0) * 3) : dst;
          const uint4 q0 = src[0], q1 = src[1], q2 = src[2];
          const unsigned d[12] = {q0.x, q0.y, q0.z, q0.w, q1.x, q1.y, q1.z, q1.w, q2.x, q2.y, q2.z, q2.w};
#pragma unroll
          for (int q = 0; q < 4; ++q) {
            px[4 * q] = d[3 * q] & 0xFFFFFFu;
            px[4 * q + 1] = (d[3 * q] >> 24) | ((d[3 * q + 1] & 0xFFFFu) << 8);
            px[4 * q + 2] = (d[3 * q + 1] >> 16) | ((d[3 * q + 2] & 0xFFu) << 16);
            px[4 * q + 3] = d[3 * q + 2] >> 8;
          }
        }
        while (bitsw) {
          const int g = __ffsll((long long)bitsw) - 1;
          bitsw &= bitsw - 1ull;
          const unsigned* mrow = reinterpret_cast<const unsigned*>(rows + (rowbase[g] + y));
          const unsigned bits = (mrow[x0 >> 5] >> (x0 & 31)) & 0xFFFFu;
          if (bits == 0u) continue;
          const unsigned rgba = item_rgba[g];
          const unsigned al = rgba >> 24;
          if (al == 255u) {
            const unsigned fg = rgba & 0xFFFFFFu;
#pragma unroll
            for (int i = 0; i < 16; ++i) px[i] = (bits & (1u << i)) ? fg : px[i];
          } else {
            const unsigned f0 = rgba & 255u, f1 = (rgba >> 8) & 255u, f2 = (rgba >> 16) & 255u;
#pragma unroll
            for (int i = 0; i < 16; ++i) {
              if (bits & (1u << i)) {
                const unsigned o = px[i];
                px[i] = blend8(o & 255u, f0, al) | (blend8((o >> 8) & 255u, f1, al) << 8) |
                        (blend8((o >> 16) & 255u, f2, al) << 16);
              }
            }
          }
        }
        unsigned d[12];
#pragma unroll
        for (int q = 0; q < 4; ++q) {   // 4 pixels (RGBX) -> 3 dwords (RGB)
          const unsigned p0 = px[4 * q], p1 = px[4 * q + 1], p2 = px[4 * q + 2], p3 = px[4 * q + 3];
          d[3 * q] = (p0 & 0xFFFFFFu) | (p1 << 24);
          d[3 * q + 1] = ((p1 >> 8) & 0xFFFFu) | (p2 << 16);
          d[3 * q + 2] = ((p2 >> 16) & 0xFFu) | (p3 << 8);
        }
        dst[0] = make_uint4(d[0], d[1], d[2], d[3]);
        dst[1] = make_uint4(d[4], d[5], d[6], d[7]);
        dst[2] = make_uint4(d[8], d[9], d[10], d[11]);
      }
    }
    if (g_hi >= n_items) break;
    g_lo = g_hi;
    first_pass = false;
    // the next pass starts from clean tables and reads this pass's pixels back
    __builtin_amdgcn_fence(__ATOMIC_SEQ_CST, "workgroup");
    rw_barrier();
    for (int i = tid; i < r_cap; i += RW_THREADS) {
      uint4* r = reinterpret_cast<uint4*>(rows + i);
      r[0] = make_uint4(~0u, ~0u, ~0u, ~0u);
      r[1] = make_uint4(~0u, ~0u, 0u, 0u);
    }
    for (int y = tid; y < H; y += RW_THREADS) rowitems[y] = 0ull;
    if (lane < 8) wmisc[lane] = 0;
    rw_barrier();
  }
}

#endif  // MOOG_RASTER_WAVE_H_
