// moog_raster.h -- HIP scanline polygon rasteriser (gfx950), bit-exact with
// Pillow's ImageDraw.polygon in RGBA blend mode as used by PILRenderer
// (reference moog/observers/pil_renderer.py:88-120; Pillow Draw.c
// ImagingDrawPolygon / polygon_generic(hasAlpha=1) / hline32rgba as restated and
// fuzz-validated against Pillow 12.2.0 in oracle/moog_oracle.c).
//
// One 256-thread workgroup renders one env's frame.  The kernel is bound by VALU
// issue and by the divergence of per-row loops, so the work is cut into units that
// fill the lanes evenly:
//   1  thread per vertex: (int)(W*x) canvas points, per-item row ranges (LDS atomics)
//   2  thread per vertex: the edge leaving it (ImagingDrawPolygon: add_edge + merging
//      of horizontal runs); table edges and horizontal heads go to a compact list
//   3  thread per listed edge: polygon_generic's corner fix-up partners for the
//      edge's two end rows, then the edge PUSHES one crossing per row it spans into
//      that row's record (slot by LDS atomic); rows beyond the first four of an edge
//      are handled eight lanes per edge.  A crossing is stored as a 16-bit key
//      ROUND_UP(x) + ROUND_DOWN(x): both roundings are monotone in x, so sorting the
//      keys sorts the crossings, and the span loop only needs the two roundings.
//   4  thread per (item, row): load <= 12 keys, sorting network in registers,
//      Pillow's span / x_pos / horizontal-line logic -> 64/128-bit coverage mask
//   5  thread per 16-pixel row segment: compose the items covering that SEGMENT in
//      painter's order (one blend per covered pixel), pack RGB, store the flipped row
// Rows with more than 12 crossings, two fix-ups, two horizontal heads or far off-canvas
// crossings (< 1 %)
// take a generic routine that keeps the crossing list in LDS.  Each output byte is
// written exactly once; algorithmic bytes = H*W*3 + live vertices * 16.
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>

#include "../../include/moog_engine.h"
#include "moog_raster_mask_core.h"

#define R_THREADS 256
#define R_SLOW 8            // lanes that run the generic scanline concurrently (bounds its LDS scratch)
#define R_CAP 12            // crossing keys per row record
#define R_KEY_BIAS 32768
#define R_XLIM 16000.0f     // |x| beyond this does not fit a key: generic routine
#define R_NONE ((short)-32768)
#define R_CNT_MASK 0x1ffu   // row word: bits 0-8 crossings, 9-16 fix-up tip columns, 17 generic, 18-31 item + 1
#define R_FIX_ONE 0x200u
#define R_GENERIC 0x20000u
#define R_ITEM_SHIFT 18

struct RPlan {   // LDS carve-up (byte offsets), computed once on the host
  unsigned o_pbase, o_edge, o_ivert, o_long, o_list, o_xx, o_item_y, o_item_rgba, o_rowbase, o_rowoff, o_head,
      o_rows, o_seg, o_queue, o_misc, total;
};

// What the mask rasteriser (moog_raster_mask_core.h) needs beyond RArgs: fixed per engine.  ok: the program's frames are
// one tile, its polygons have <= 32 vertices and no polygon modifier copies them -- moog_raster_launch then draws
// ordinary frames (not the prefix pictures, not draw-list or per-env-prefix frames) with that kernel.
struct RmSetup {
  int32_t ok;
  int32_t S, slots, ncopy, big, compact, cap_rows, iwords, cmap, first_person, fp_slot0, fp_nslots;   // compact: 4-byte edge records (RmEdgesCompact)
  uint32_t bg;
  RmPlan plan;
  uint32_t lds;
  int32_t persist_slots;   // > 0: workgroups resident on the device at once; a launch of more frames is one round of workgroups that draw several frames each
};

struct RArgs {
  RmSetup ms;
  // (mask rasteriser) the draw records it reads (moog_draw_record.h).  draw_ready: the step kernel wrote them for exactly these
  // records (moog_engine_step); else moog_raster_launch derives them from f64 / i32 first.  env0: the engine's index of env 0
  // of this launch (launches over a chunk of the envs).
  RmEmit em;
  int32_t draw_ready, env0;
  int32_t* rows_seen;     // (mask rasteriser) host-mapped word for frames that want more row records, or null
  const moog_program_t* P;
  moog_layout_t L;
  const double* f64;
  const int32_t* i32;
  uint8_t* image;
  const uint32_t* vinfo;  // per vertex slot: sprite slot | index within the sprite << 8
  int32_t n_envs;
  int32_t chunk;       // row records per pass
  int32_t words;       // 64-bit words per row mask (of a tile)
  int32_t tile_w;      // columns per workgroup tile (<= 128, a multiple of 16 that divides the width)
  int32_t band_h;      // rows per workgroup tile
  int32_t tiles_x;     // tiles per canvas row
  int32_t bands;       // tile rows per canvas
  int32_t canvas_w, canvas_h;   // the canvas in memory: anti_aliasing x the observation size (pil_renderer.py:65-66), the width
                                // rounded up to a multiple of 16 (the rasteriser works in 16-pixel segments)
  int32_t scale_w;     // the canvas width the vertices are scaled by (= canvas_w unless that was rounded up: the extra
                       // columns are drawn like Pillow would draw a wider image and cropped by the caller)
  int32_t flip;        // 1: rows are written bottom-up (np.flipud, pil_renderer.py:118); 0 for a canvas that is down-sampled next
  int32_t iwords;      // 32-bit words of a segment's item bitmask
  int32_t hwords;      // 32-bit words of an item's head bitmask
  int32_t debug_stop;  // >0: return after that phase (profiling aid)
  RPlan plan;
  int32_t xxcap;       // crossing-list capacity of the generic scanline = 2 * max vertices per sprite
  // Static prefix: sprites that a reset always creates the same way and that nothing moves (the
  // border walls of most configs) come first in painter's order.  The engine renders them once,
  // on top of the background colour, into `sbg`; the kernel compares every frame's prefix with
  // the reference record bit for bit (vertices, colours, flags) and, when equal, skips those
  // sprites and composes on top of `sbg`.  A frame whose prefix differs takes the ordinary path.
  int32_t n_static;    // slots in the prefix (0: none)
  int32_t nsv;         // vertex slots of the prefix (they are the first nsv of the record)
  int32_t build;       // 1: render ONLY the prefix of env 0 (the launch that fills `sbg`)
  const double* sref_v;     // reference world vertices [nsv][2]
  const double* sref_col;   // reference colours [n_static][3]
  const int32_t* sref_flags; // reference flags / vertex counts / opacities [n_static] each
  const int32_t* sref_nv;
  const int32_t* sref_opa;
  const uint8_t* sbg;       // background + prefix, [H][W][3] in output (flipped) order
  // Per-env prefix (programs whose leading sprites never move but differ from env to env and episode to episode -- a
  // maze's walls): every env has a picture of its own, sbg + env * sbg_env_stride, and a snapshot of the record it was
  // drawn from; a check launch (moog_prefix_check_launch) compares the live prefix with the snapshot before the frames are
  // drawn and names the envs whose picture a `build` launch has to draw again (env_build).  The frame launch then skips
  // the prefix of every env without comparing anything.
  size_t sbg_env_stride;    // bytes between the envs' pictures; 0: one picture for all (the comparison is the frame kernel's)
  const uint32_t* rgb_override;   // [n_envs][S] r | g << 8 | b << 16 per sprite slot instead of the colour map (moog_engine_set_color_override), or null
  const int32_t* env_build; // build launches: [n_envs] 1 = draw this env's picture, 0 = the workgroup has nothing to do; null: every env
};

// Edge record, 16 bytes, one per vertex slot (the edge from vertex k to k + 1).
// Table (non-horizontal) edges carry dx and the fix-up replacement values of their top
// and bottom rows; horizontal heads keep xmin | xmax << 16 in the dx bits.
struct REdge { short x0, y0, x1, y1; float dx; short vtop, vbot; };

// Row record, 32 bytes: the keys of the row's crossings (later its coverage mask), which
// edges of the polygon are horizontal heads on this row (bit k = edge k), counters + owning item.
struct RRow { unsigned short key[R_CAP]; unsigned hbits; unsigned cnt; };

__host__ __device__ inline size_t r_align(size_t x) { return (x + 15) & ~(size_t)15; }

__host__ __device__ inline void raster_plan(int S, int TOTV, int ncopy, int W, int H, int cap_rows,
                                            int iwords, int hwords, int xxcap, RPlan* p) {
  size_t o = 0;
  size_t nv = (size_t)TOTV * ncopy, items = (size_t)S * ncopy;
  p->o_edge = o; o = r_align(o + nv * sizeof(REdge));
  // union: integer vertices (phases 1-3a, read two at a time) / segment item masks (phases 3b-5)
  {
    size_t v1 = nv * 4 + 8, v2 = (size_t)H * (W / 16) * iwords * 4;
    p->o_ivert = o; p->o_seg = o; o = r_align(o + (v1 > v2 ? v1 : v2));
  }
  // union: long-edge lists (phase 3) / generic crossing lists (phase 4)
  size_t l1 = nv * 4, l2 = (size_t)xxcap * R_SLOW * 4;
  p->o_long = o; p->o_xx = o; o = r_align(o + (l1 > l2 ? l1 : l2));
  p->o_list = o; o = r_align(o + nv * 4);             // compact edge list (phases 2-3, every pass)
  p->o_pbase = o; o = r_align(o + (size_t)S * 4);     // first edge record of each sprite slot
  p->o_item_y = o; o = r_align(o + items * 8);        // ymin, ymax (ints, atomics)
  p->o_item_rgba = o; o = r_align(o + items * 4);
  p->o_rowbase = o; o = r_align(o + items * 4);       // first row record - first row
  p->o_rowoff = o; o = r_align(o + (items + 1) * 4);
  p->o_head = o; o = r_align(o + items * hwords * 4); // which edges of the item are heads
  p->o_rows = o; o = r_align(o + (size_t)cap_rows * sizeof(RRow));
  p->o_queue = o; o = r_align(o + (size_t)cap_rows * 4);   // generic rows from the front, multi-head rows from the back
  p->o_misc = o; o = r_align(o + 64);
  p->total = o;
}

// moog_raster.hip: the kernel's own translation unit
// Image.resize(LANCZOS) of a batch of canvases [n][ch][cw][3] -> observations [n][oh][ow][3], flipped; tmp: [n][ch][ow][3]
struct RResize { int32_t cw, ch, ow, oh, kh, kv; const int32_t* bh; const int32_t* bv; const int32_t* ch_coef; const int32_t* cv_coef;
                 int32_t hspan; /* bytes of a canvas row that 256 consecutive output columns read, at most (+ alignment slack) */
                 int32_t cstride, tstride; /* pixels per canvas row / per row of the intermediate picture in memory (multiples of 4) */ };
// Rows of `row_bytes` bytes out of rows `in_stride` bytes apart (frames whose width is not a multiple of 16 are drawn wider).
void moog_crop_launch(const uint8_t* in, uint8_t* out, size_t rows, int in_stride, int row_bytes, hipStream_t stream);
void moog_resize_launch(const RResize& r, const uint8_t* canvas, uint8_t* tmp, uint8_t* out, int n, hipStream_t stream);
int moog_raster_configure_mask(size_t lds_bytes);   // the same for the mask rasteriser's kernels
int moog_raster_configure(size_t lds_bytes);   // hipFuncSetAttribute(max dynamic LDS); returns a hipError_t
void moog_raster_launch(const RArgs& a, size_t lds_bytes, hipStream_t stream);
// Per-env prefix check: one wavefront per env compares the first n_static slots of the live record (alive bit, vertex count,
// opacity, colour, live vertices) with the env's snapshot.  Different, or no picture yet (valid[env] == 0): the record is
// copied to the snapshot, build[env] = 1, valid[env] = 1; equal: build[env] = 0.  A change in the middle of an episode
// (step_count != 0) of an env that had a picture lowers *min_changed (host-visible) to the first slot that changed: the
// engine shortens the prefix to the slots that really stay put.
struct PCArgs {
  const moog_program_t* P; moog_layout_t L;
  const double* f64; const int32_t* i32;   // live records
  double* s_f64; int32_t* s_i32;           // snapshots, same layout
  int32_t* valid; int32_t* build; int32_t* min_changed;
  int32_t n_envs, n_static;
};
void moog_prefix_check_launch(const PCArgs& a, hipStream_t stream);
