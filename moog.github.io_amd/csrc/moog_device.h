// moog_device.h -- device-side MOOG step logic for gfx950 (CDNA4).
//
// Execution model: ONE WAVEFRONT (64 lanes) OWNS ONE ENV.  The env's whole state
// record (include/moog_engine.h layout) is staged in LDS; the reference's
// sequential, order-dependent pair loop (physics.py:103-108) is kept as a
// wave-uniform scalar loop, and the lanes parallelise *inside* each step:
//   - broad phase: lanes = candidate partner sprites, __ballot -> candidate mask
//   - Path.intersects_path: lanes = segment pairs / vertices, __ballot any/all
//   - contact search: lanes = contained vertices (rows of the crossing matrix),
//     argmin per lane, numpy-ordered argmax by lane scan
//   - path translate / rotate: lanes = vertices
// No MFMA: the work is O(S^2) geometry with data-dependent branches.
// All arithmetic is IEEE fp64 without contraction (-ffp-contract=off) so that
// discrete decisions (argmin/argmax, >, isclose) match numpy.
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>

#include "../../include/moog_engine.h"

// The lowered config is read-only device memory.  Reading it through the constant
// address space lets wave-uniform accesses compile to scalar loads (s_load, scalar
// cache) instead of per-lane vector loads.
#define MOOG_CONST __attribute__((address_space(4)))
typedef const MOOG_CONST moog_program_t* PProg;
typedef const MOOG_CONST moog_force_t* PForce;
typedef const MOOG_CONST moog_corrective_t* PCorr;
typedef const MOOG_CONST moog_dinstr_t* PDinstr;
typedef const MOOG_CONST moog_rule_t* PRule;
typedef const MOOG_CONST moog_task_t* PTask;
typedef const MOOG_CONST moog_action_t* PAction;
typedef const MOOG_CONST moog_shape_t* PShape;
// One entry of the flattened force list: a (force, layer a, layer b) combination of physics.py:96-108 with everything its
// loop header needs in 64 contiguous bytes (one scalar load), built once per engine on the host (moog_flatten_forces in
// moog_kernels.h).  The nested loops over program.forces read ~10 dependent scalars per combination -- force kind, list
// lengths, layer ids, slot ranges, the Collision parameters -- ten times per env-step: 6 % of the contact-heavy envs'
// cycles and 12 % of the typical env's sat in those headers (profiles/r04_step_sections.txt).
struct FOp {
  int32_t fi, kind;          // index into program.forces (the rarely used kinds still read their record), MOOG_FORCE_*
  int32_t a0, a1, b0, b1;    // slot ranges of the two layers (b0 = b1 = 0 for a one-layer force)
  int32_t symmetric, i0, i1; // as in moog_force_t
  int32_t n_b;               // 0: one-layer force
  double p0, p1;
  int32_t pad[2];
};
typedef const MOOG_CONST FOp* PFOp;
typedef const MOOG_CONST moog_genop_t* PGenop;
typedef const MOOG_CONST moog_factor_t* PFactor;
#ifdef MOOG_SPEC_PROGRAM_INC
// A program-specialised build (moog_step_spec.hip, moog/_spec.py): the lowered config is a constant of the translation unit
// (`static const moog_program_t MOOG_SPEC_PROGRAM = {...};` generated from the program's bytes), so every load of it folds
// and what the program does not use is not compiled: the headline workload's step kernel is 34 k VALU instructions instead
// of 49 k and 7 % faster (profiles/r05_step_spec.txt).  The engine checks the embedded program against its own, byte for byte.
#include MOOG_SPEC_PROGRAM_INC
__device__ __forceinline__ PProg as_const_prog(const moog_program_t*) { return (PProg)&MOOG_SPEC_PROGRAM; }
#else
__device__ __forceinline__ PProg as_const_prog(const moog_program_t* p) {
  return (PProg)(unsigned long long)p;
}
#endif

#define EPS_INTERP 1e-8  // sprite.py:35
#ifdef MOOG_PROFILE
#define PROF_T0 const long long prof_t0_ = clock64()
#define PROF_ADD(e_, k) const_cast<Env&>(e_).prof[k] += clock64() - prof_t0_
#else
#define PROF_T0
#define PROF_ADD(e_, k)
#endif
// Section announcements for the sampling watcher (moog_kernels.h, -DMOOG_WATCH builds only): one LDS store, no wait.
#ifdef MOOG_WATCH
#define SEC(e_, id) do { if ((e_).secw && (e_).lane == 0) __hip_atomic_store((e_).secw, (id), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP); } while (0)
#else
#define SEC(e_, id) do { } while (0)
#endif
enum { SEC_PROLOGUE = 0, SEC_RULES = 1, SEC_FORCES = 2, SEC_BROAD = 3, SEC_LIST = 4, SEC_BATCH_FORM = 5, SEC_BATCH = 6, SEC_PATH = 7,
       SEC_SEARCH_CONTAIN = 8, SEC_SEARCH_MATRIX = 9, SEC_SEARCH_ROWS = 10, SEC_SEARCH_FINISH = 11, SEC_RESOLVE = 12, SEC_DISJOINT = 13,
       SEC_RETEST = 14, SEC_PAIR_SCAN = 15, SEC_PAIR_CONSUME = 16, SEC_INT_POSE = 17, SEC_INT_LONG = 18, SEC_INT_VERTS = 19,
       SEC_INT_BOXES = 20, SEC_TASK = 21, SEC_STORE = 22, SEC_STEP_CONTROL = 23, SEC_CONSUME = 24, SEC_SEARCH_SELECT = 25 };
#define EPS_COLL 1e-2    // collisions.py:46
#define MOOG_F_TMP 0x100 // scratch flag bit (vanish marks)
#define DINF (__builtin_inf())

struct Env {
  double* f;               // LDS f64 record (hot fields only, see HotLayout)
  int32_t* q;              // LDS i32 record (hot fields only)
  PProg P;                 // lowered config (constant address space)
  PFOp fops;               // flattened force list (constant address space)
  int n_fops;
  moog_layout_t L;         // layout of the LDS records
  double* gcol;            // this env's colours / opacity / shape ids in HBM: fields the step
  int32_t* gopa;           //   path never reads are not staged in LDS (LDS per env sets how many
  int32_t* gshape;         //   envs a CU holds, and the kernels scale with resident waves)
  int32_t* gtele;          // Portal bookkeeping bits, likewise
  const double* inj;
  int inj_n;
  uint64_t seed;
  int64_t env_index;
  int lane;
  float* bb;               // LDS scratch [S][8]: conservative 8-DOP (lo x, y, x+y, x-y; hi x, y, x+y, x-y)
  const int16_t* vslot;    // global [TOTV]: vertex index -> slot
  unsigned cur_fmask;      // float32 factors of the sprite being created
  int cell_i, cell_j;      // maze cell (row, column) of the sprite being created (MOOG_CELL_* ops)
  int cur_slot;            // slot of the sprite being created (a computed shape is staged in its vertex area)
  int restart;             // set by a generation op: the initializer starts over (red_green.py:155,203)
  int cell_tab_n, cell_nw; // rank -> cell table of the episode's maze in LDS (cells; wall cells), 0: none
  double xs_centroid[2], xs_inertia[2];   // centroid / inertia per unit area of a computed shape (MOOG_DIST_EXPR_SHAPE)
  int xs_n;
  unsigned fac_f32, expr_f32;   // float32 samples among the staged factors (MOOG_X_FACTOR); computed factors that came out float32
  unsigned flat_f32;            // bit k: factor k of the sample just drawn is a float32 Continuous sample
  uint8_t* lst;            // LDS scratch [128]: compacted edge index lists
  int32_t* voff;           // LDS copy of program.slot_voff [S]
  uint16_t* cand;          // LDS scratch [CAND_CAP]: broad-phase survivors (s0 << 8 | s1)
  unsigned long long* rowm; // LDS scratch [64]: candidate bit matrix of a layer against itself
  double* xstack;          // LDS [program.xstack_depth][64] or null: per-lane value stacks of eval_expr_t<true>
  int n_path, n_resp, n_disj;   // profiling counters (path tests, contact searches, make_disjoint calls)
#ifdef MOOG_PROFILE
  long long prof[16];      // cycles per section (tools/step_profile.sh builds with -DMOOG_PROFILE)
#endif
#ifdef MOOG_WATCH
  int32_t* secw;           // LDS word the watcher wavefront samples (moog_engine_read_watch), or null
#endif
  int dbg;                 // profiling aid: bit0 skip collisions, bit1 skip integrate, bit2 skip narrow phase
  int32_t* layer_hw;       // global [2 * MOOG_MAX_LAYERS] or null: per dynamic layer, the most sprites an append ever wanted
                           // room for (over all envs and calls), then the number of appends dropped because the layer was full
  double xarg;             // the argument of the function being evaluated (MOOG_X_ARG)
};

// How the step path reads the env's program and (hot) layout.  Generic kernels: members of the env's descriptor (registers; scratch
// memory in the functions the every-component variants leave out of line).  Program-specialised builds: the constants themselves --
// also inside those out-of-line functions, which took them by reference through the descriptor until round 6.
#define CAND_CAP 128   // entries of the candidate list (collision_same_layer / collision_layer_pair); a full list is consumed before the scan continues
// Likewise the places of the record and of the scratch areas in the wavefront's LDS: pointers in the descriptor for the generic
// kernels; in a specialised build -- the step kernel only, one wavefront per workgroup, the record at the start of its LDS --
// addresses the compiler knows (an LDS access then carries its offset as an immediate and the pointers cost no registers).
#ifdef MOOG_SPEC_PROGRAM_INC
__device__ __forceinline__ moog_layout_t moog_spec_hot_layout_fn();   // (moog_kernels.h: hot_layout(moog_layout(&MOOG_SPEC_PROGRAM)).L, folded at compile time)
extern __shared__ __attribute__((aligned(16))) unsigned char moog_lds[];
#define EL(e_) (moog_spec_hot_layout_fn())
#define EP(e_) ((PProg)&MOOG_SPEC_PROGRAM)
#define MOOG_SPEC_LDS_Q ((size_t)EL(0).f64_per_env * 8)
#define MOOG_SPEC_LDS_BB (MOOG_SPEC_LDS_Q + (size_t)EL(0).i32_per_env * 4)
#define MOOG_SPEC_LDS_VOFF (MOOG_SPEC_LDS_BB + (size_t)EL(0).S * 32)
#define MOOG_SPEC_LDS_CAND (MOOG_SPEC_LDS_VOFF + (size_t)((EL(0).S + 3) & ~3) * 4)
#define EF(e_) (reinterpret_cast<double*>(moog_lds))
#define EQ(e_) (reinterpret_cast<int32_t*>(moog_lds + MOOG_SPEC_LDS_Q))
#define EBB(e_) (reinterpret_cast<float*>(moog_lds + MOOG_SPEC_LDS_BB))
#define EVOFF(e_) (reinterpret_cast<int32_t*>(moog_lds + MOOG_SPEC_LDS_VOFF))
#define ECAND(e_) (reinterpret_cast<uint16_t*>(moog_lds + MOOG_SPEC_LDS_CAND))
#define ELST(e_) (reinterpret_cast<uint8_t*>(moog_lds + MOOG_SPEC_LDS_CAND + CAND_CAP * 2))
#define EROWM(e_) (reinterpret_cast<unsigned long long*>(moog_lds + MOOG_SPEC_LDS_CAND + CAND_CAP * 2 + 128))
#else
#define EL(e_) ((e_).L)
#define EP(e_) ((e_).P)
#define EF(e_) ((e_).f)
#define EQ(e_) ((e_).q)
#define EBB(e_) ((e_).bb)
#define EVOFF(e_) ((e_).voff)
#define ECAND(e_) ((e_).cand)
#define ELST(e_) ((e_).lst)
#define EROWM(e_) ((e_).rowm)
#endif
#define PX(s) (EF(e)[EL(e).o_pos + 2 * (s)])
#define PY(s) (EF(e)[EL(e).o_pos + 2 * (s) + 1])
#define VELX(s) (EF(e)[EL(e).o_vel + 2 * (s)])
#define VELY(s) (EF(e)[EL(e).o_vel + 2 * (s) + 1])
#define ANG(s) (EF(e)[EL(e).o_angle + (s)])
#define ANGV(s) (EF(e)[EL(e).o_angvel + (s)])
#define MASS(s) (EF(e)[EL(e).o_mass + (s)])
// Colours, opacities, shape ids and Portal bits may live in HBM (fields the step path hardly ever touches): they are READ
// through COL / OPAC / SHAPEID / TELE and WRITTEN through the *_SET forms.
#define COL(s, c) (static_cast<const double*>(e.gcol)[3 * (s) + (c)])
#define COL_SET(s, c, v) (e.gcol[3 * (s) + (c)] = (v))
#define OPAC_SET(s, v) (e.gopa[(s)] = (v))
#define SHAPEID_SET(s, v) (e.gshape[(s)] = (v))
#define TELE_SET(s, v) (e.gtele[(s)] = (v))
#define INER(s, c) (EF(e)[EL(e).o_inertia + 2 * (s) + (c)])
#define MAXR(s) (EF(e)[EL(e).o_maxr + (s)])
#define FLAGS(s) (EQ(e)[EL(e).o_flags + (s)])
#define NV(s) (EQ(e)[EL(e).o_nverts + (s)])
#define OPAC(s) (static_cast<const int32_t*>(e.gopa)[(s)])
#define SHAPEID(s) (static_cast<const int32_t*>(e.gshape)[(s)])
#define TELE(s) (static_cast<const int32_t*>(e.gtele)[(s)])
#define VERT(s) (&EF(e)[EL(e).o_verts + 2 * EVOFF(e)[s]])
#define ALIVE(s) (FLAGS(s) & MOOG_F_ALIVE)
#define VALIAS(s) (EQ(e)[EL(e).o_valias + (s)])
#define SCALE(s) (EF(e)[EL(e).o_scale + (s)])
#define ASPECT(s) (EF(e)[EL(e).o_aspect + (s)])
#define FMASK(s) (EQ(e)[EL(e).o_fmask + (s)])

__device__ __forceinline__ int uni(int x) { return __builtin_amdgcn_readfirstlane(x); }

// Single-wave workgroup: LDS operations of one wave execute in order, so
// cross-lane visibility only needs the compiler/waitcnt fence.
__device__ __forceinline__ void wsync() {
  // LDS only ("local"): do not also drain outstanding global loads (vmcnt) here.
  // (Round 4 tried a wavefront-scope fence instead -- compile-time ordering only, no s_waitcnt; the LDS unit executes one
  //  wavefront's operations in order -- and measured nothing: 652 vs 655 us; nine waits in ten are needed by the load that
  //  follows anyway.  The waiting form stays.)
  __builtin_amdgcn_fence(__ATOMIC_SEQ_CST, "workgroup", "local");
  __builtin_amdgcn_wave_barrier();
}

// Fields that live in HBM (colours, opacities, shape ids, Portal bits of records too large to stage whole) are written by
// one lane and read by the others of the SAME wavefront later in the launch: workgroup scope is all that needs (the stores
// acknowledged before later loads issue; one compute unit, one L1).  __threadfence() -- agent scope -- writes this XCD's
// whole L2 back and invalidates it on gfx950 (buffer_wbl2 sc1 + buffer_inv sc1), per rule, per env, per step.
__device__ __forceinline__ void wave_global_fence() {
  __builtin_amdgcn_fence(__ATOMIC_SEQ_CST, "workgroup");
  __builtin_amdgcn_wave_barrier();
}

__device__ __forceinline__ double f32r(double x) { return (double)(float)x; }
__device__ __forceinline__ double norm2(double x, double y) { return sqrt(x * x + y * y); }   // norms along an axis: plain sums
// numpy as installed where the fixtures were recorded (OpenBLAS ddot accumulates with a fused multiply-add): np.dot of two
// float64 2-vectors = fma(a1, b1, a0 * b0); np.linalg.norm of a 1-D float64 array = sqrt(x.dot(x)).  See the oracle
// (npdot2 / npnorm in oracle/moog_oracle.c) and tests/golden/npdot.npz.
__device__ __forceinline__ double npdot2(double a0, double a1, double b0, double b1) { return fma(a1, b1, a0 * b0); }
__device__ __forceinline__ double npnorm(double x, double y) { return sqrt(fma(y, y, x * x)); }

// sin / cos of the small per-substep rotation angles (|th| = |angle_vel| / K, typically
// < 0.01 rad).  For |th| < 2^-5 the Taylor series through th^9 / th^8 is accurate to
// below one ulp (next terms < 1e-22), i.e. as good as the library routines without
// their argument reduction; larger angles use the library.
__device__ __forceinline__ void sincos_small(double th, double* s, double* c) {
  if (fabs(th) < 0.03125) {
    double x2 = th * th;
    *s = th * (1.0 + x2 * (-1.0 / 6 + x2 * (1.0 / 120 + x2 * (-1.0 / 5040 + x2 * (1.0 / 362880)))));
    *c = 1.0 + x2 * (-0.5 + x2 * (1.0 / 24 + x2 * (-1.0 / 720 + x2 * (1.0 / 40320))));
  } else {
    *s = sin(th); *c = cos(th);
  }
}

// exact n / d for 0 <= n < 4096, 1 <= d <= 128 without the integer-division sequence
__device__ __forceinline__ int div_small(int n, int d) {
  unsigned m = ((1u << 19) + (unsigned)d - 1u) / (unsigned)d;   // d is wave-uniform: scalar
  return (int)(((unsigned)n * m) >> 19);
}

__device__ __forceinline__ double shfl_d(double v, int src) {
  int lo = __shfl(__double2loint(v), src);
  int hi = __shfl(__double2hiint(v), src);
  return __hiloint2double(hi, lo);
}

// min over each row of 16 lanes, in every lane of the row, by DPP exchanges (xor 1, xor 2 inside
// the quads, then the two mirrors): VALU moves instead of four round trips through ds_bpermute.
template <int CTRL>
__device__ __forceinline__ double dpp_d(double v) {
  int lo = __builtin_amdgcn_update_dpp(0, __double2loint(v), CTRL, 0xf, 0xf, false);
  int hi = __builtin_amdgcn_update_dpp(0, __double2hiint(v), CTRL, 0xf, 0xf, false);
  return __hiloint2double(hi, lo);
}
__device__ __forceinline__ double row16_min(double v) {
  v = fmin(v, dpp_d<0xB1>(v));    // quad_perm [1,0,3,2]
  v = fmin(v, dpp_d<0x4E>(v));    // quad_perm [2,3,0,1]
  v = fmin(v, dpp_d<0x141>(v));   // row_half_mirror
  v = fmin(v, dpp_d<0x140>(v));   // row_mirror
  return v;
}

// ---- RNG: Philox4x32-10, bit-identical to oracle/moog_oracle.c ---------------
__device__ inline void philox4x32(uint32_t c[4], uint32_t k0, uint32_t k1) {
#pragma unroll
  for (int r = 0; r < 10; ++r) {
    uint64_t p0 = (uint64_t)0xD2511F53u * c[0];
    uint64_t p1 = (uint64_t)0xCD9E8D57u * c[2];
    uint32_t n0 = (uint32_t)(p1 >> 32) ^ c[1] ^ k0;
    uint32_t n1 = (uint32_t)p1;
    uint32_t n2 = (uint32_t)(p0 >> 32) ^ c[3] ^ k1;
    uint32_t n3 = (uint32_t)p0;
    c[0] = n0; c[1] = n1; c[2] = n2; c[3] = n3;
    k0 += 0x9E3779B9u; k1 += 0xBB67AE85u;
  }
}

// Wave-uniform draw: every lane computes the same value; lane 0 commits the counter.
__device__ inline double next_uniform(Env& e) {
  int32_t* r = &EQ(e)[EL(e).o_rng];
  double out;
  if (e.inj) {
    int cur = r[2];
    if (cur >= e.inj_n) {
      wsync();
      if (e.lane == 0) EQ(e)[EL(e).o_fault] |= MOOG_FAULT_INJECT_UNDERRUN;
      wsync();
      return 0.0;
    }
    out = e.inj[cur];
    wsync();
    if (e.lane == 0) r[2] = cur + 1;
    wsync();
    return out;
  }
  uint64_t ctr = (uint32_t)r[0] | ((uint64_t)(uint32_t)r[1] << 32);
  uint32_t c[4] = {(uint32_t)ctr, (uint32_t)(ctr >> 32), (uint32_t)e.env_index,
                   (uint32_t)((uint64_t)e.env_index >> 32)};
  philox4x32(c, (uint32_t)e.seed, (uint32_t)(e.seed >> 32));
  ctr += 1;
  wsync();
  if (e.lane == 0) { r[0] = (int32_t)(uint32_t)ctr; r[1] = (int32_t)(uint32_t)(ctr >> 32); }
  wsync();
  return ((double)(c[0] >> 5) * 67108864.0 + (double)(c[1] >> 6)) / 9007199254740992.0;
}

// The next n <= 64 draws at once: lane i returns draw i (the same values, in the same order, that n calls
// of next_uniform would give); the counter advances by n.  One Philox latency instead of n.
__device__ inline double next_uniforms_lanes(Env& e, int n) {
  int32_t* r = &EQ(e)[EL(e).o_rng];
  if (e.inj) {
    const int cur = r[2];
    const int have = e.inj_n - cur < n ? (e.inj_n - cur < 0 ? 0 : e.inj_n - cur) : n;
    const double out = (e.lane < have) ? e.inj[cur + e.lane] : 0.0;
    wsync();
    if (e.lane == 0) {
      r[2] = cur + have;
      if (have < n) EQ(e)[EL(e).o_fault] |= MOOG_FAULT_INJECT_UNDERRUN;
    }
    wsync();
    return out;
  }
  const uint64_t base = (uint32_t)r[0] | ((uint64_t)(uint32_t)r[1] << 32);
  const uint64_t ctr = base + (uint64_t)e.lane;
  uint32_t c[4] = {(uint32_t)ctr, (uint32_t)(ctr >> 32), (uint32_t)e.env_index,
                   (uint32_t)((uint64_t)e.env_index >> 32)};
  philox4x32(c, (uint32_t)e.seed, (uint32_t)(e.seed >> 32));
  const uint64_t next = base + (uint64_t)n;
  wsync();
  if (e.lane == 0) { r[0] = (int32_t)(uint32_t)next; r[1] = (int32_t)(uint32_t)(next >> 32); }
  wsync();
  return ((double)(c[0] >> 5) * 67108864.0 + (double)(c[1] >> 6)) / 9007199254740992.0;
}

// ---- matplotlib _path restatements (see oracle for the citations) ------------
__device__ __forceinline__ bool mpl_isclose(double a, double b) {
  return fabs(a - b) <= fmax(1e-10 * fmax(fabs(a), fabs(b)), 1e-13);
}

// 0: no intersection, the segments are not parallel; 1: collinear segments that overlap; 2: a crossing of two non-parallel
// segments; 4: parallel segments, no intersection.  Answers 0 and 2 are the same with the two segments exchanged: den, n1, n2
// become -den, -n2, -n1 exactly (products commute, a - b = -(b - a)), so u1 and u2 trade places and the test is symmetric in
// them; the parallel branch measures from the first segment's end points and is not symmetric in general (answers 1, 4).
__device__ inline int segments_intersect_kind(double x1, double y1, double x2, double y2, double x3,
                                              double y3, double x4, double y4) {
  double den = ((y4 - y3) * (x2 - x1)) - ((x4 - x3) * (y2 - y1));
  if (mpl_isclose(den, 0.0)) {
    double t_area = (x2 * y3 - x3 * y2) - x1 * (y3 - y2) + y1 * (x3 - x2);
    if (mpl_isclose(t_area, 0.0)) {
      if (x1 == x2 && x2 == x3) {
        return ((fmin(y1, y2) <= fmin(y3, y4) && fmin(y3, y4) <= fmax(y1, y2)) ||
                (fmin(y3, y4) <= fmin(y1, y2) && fmin(y1, y2) <= fmax(y3, y4))) ? 1 : 4;
      }
      return ((fmin(x1, x2) <= fmin(x3, x4) && fmin(x3, x4) <= fmax(x1, x2)) ||
              (fmin(x3, x4) <= fmin(x1, x2) && fmin(x1, x2) <= fmax(x3, x4))) ? 1 : 4;
    }
    return 4;
  }
  double n1 = ((x4 - x3) * (y1 - y3)) - ((y4 - y3) * (x1 - x3));
  double n2 = ((x2 - x1) * (y1 - y3)) - ((y2 - y1) * (x1 - x3));
  double u1 = n1 / den, u2 = n2 / den;
  return (((u1 > 0.0) | mpl_isclose(u1, 0.0)) & ((u1 < 1.0) | mpl_isclose(u1, 1.0)) &
          ((u2 > 0.0) | mpl_isclose(u2, 0.0)) & ((u2 < 1.0) | mpl_isclose(u2, 1.0))) ? 2 : 0;
}
__device__ inline bool segments_intersect(double x1, double y1, double x2, double y2, double x3,
                                          double y3, double x4, double y4) {
  return (segments_intersect_kind(x1, y1, x2, y2, x3, y3, x4, y4) & 3) != 0;
}

// even-odd crossing test, one lane, polygon vertices in LDS
__device__ inline bool point_in_poly(const double* v, int n, double tx, double ty) {
  if (!(isfinite(tx) && isfinite(ty))) return false;
  int inside = 0;
  double x0 = v[0], y0 = v[1];
  for (int i = 0; i < n; ++i) {
    int j = (i + 1 == n) ? 0 : i + 1;
    double x1 = v[2 * j], y1 = v[2 * j + 1];
    const int f0 = (y0 >= ty), f1 = (y1 >= ty);
    inside ^= (int)(f0 != f1) & (int)(((y1 - ty) * (x0 - x1) >= (x1 - tx) * (y0 - y1)) == (bool)f1);
    x0 = x1; y0 = y1;
  }
  return inside != 0;
}

// Path.intersects_path(a, b, filled=True).  Wave-uniform result.
// Exact shortcuts (never change the result, see BB_MARGIN below): only edges of a
// whose own extent reaches b's hull along all four axes (and vice versa) can intersect
// anything, so those are compacted first and the lanes enumerate just the surviving
// edge pairs; the "every vertex of b inside a" test needs b's box inside a's box.
#define BB_MARGIN 1e-5
// (Straight-line code on purpose, here and in the other culls: `a || b` over double comparisons compiles to a branch per
//  term -- exec-mask save, s_cbranch_execz, restore -- and fmin / fmax of loaded values to a canonicalising v_max first; a
//  lone wavefront pays ~10-30 cycles per branch.  `both end points beyond the bound` is the same cull without min / max; a
//  NaN coordinate compares false and keeps the edge, which is conservative: such an edge intersects nothing anyway.)
// the 8-DOP in registers (lo = x, y, x+y, x-y minima; hi = the maxima)
__device__ __forceinline__ bool seg_outside_dop_r(double x1, double y1, double x2, double y2, const float4& lo,
                                                  const float4& hi) {
  const double p1 = x1 + y1, p2 = x2 + y2, m1 = x1 - y1, m2 = x2 - y2;
  const double hx = (double)hi.x + BB_MARGIN, hy = (double)hi.y + BB_MARGIN, hp = (double)hi.z + BB_MARGIN, hm = (double)hi.w + BB_MARGIN;
  const double lx = (double)lo.x - BB_MARGIN, ly = (double)lo.y - BB_MARGIN, lp = (double)lo.z - BB_MARGIN, lm = (double)lo.w - BB_MARGIN;
  return ((x1 > hx) & (x2 > hx)) | ((x1 < lx) & (x2 < lx)) | ((y1 > hy) & (y2 > hy)) | ((y1 < ly) & (y2 < ly)) |
         ((p1 > hp) & (p2 > hp)) | ((p1 < lp) & (p2 < lp)) | ((m1 > hm) & (m2 > hm)) | ((m1 < lm) & (m2 < lm));
}
__device__ __forceinline__ bool seg_outside_dop(double x1, double y1, double x2, double y2, const float* d) {
  return seg_outside_dop_r(x1, y1, x2, y2, *reinterpret_cast<const float4*>(d), *reinterpret_cast<const float4*>(d + 4));
}
// a point against an 8-DOP
__device__ __forceinline__ bool point_outside_dop(double x, double y, const float* d) {
  const float4 lo = *reinterpret_cast<const float4*>(d), hi = *reinterpret_cast<const float4*>(d + 4);
  const double p = x + y, m = x - y;
  return (x > (double)hi.x + BB_MARGIN) | (x < (double)lo.x - BB_MARGIN) | (y > (double)hi.y + BB_MARGIN) | (y < (double)lo.y - BB_MARGIN) |
         (p > (double)hi.z + BB_MARGIN) | (p < (double)lo.z - BB_MARGIN) | (m > (double)hi.w + BB_MARGIN) | (m < (double)lo.w - BB_MARGIN);
}
// two segments whose axis-aligned boxes are more than BB_MARGIN apart
__device__ __forceinline__ bool segs_apart(double x11, double y11, double x12, double y12, double x21, double y21, double x22, double y22) {
  const double M = BB_MARGIN;
  return ((x11 > x21 + M) & (x11 > x22 + M) & (x12 > x21 + M) & (x12 > x22 + M)) |
         ((x21 > x11 + M) & (x21 > x12 + M) & (x22 > x11 + M) & (x22 > x12 + M)) |
         ((y11 > y21 + M) & (y11 > y22 + M) & (y12 > y21 + M) & (y12 > y22 + M)) |
         ((y21 > y11 + M) & (y21 > y12 + M) & (y22 > y11 + M) & (y22 > y12 + M));
}

// *proper (when given): the answer is the same with a and b exchanged -- true because two non-parallel edges cross
// (segments_intersect_kind 2), or false without a pair of parallel edges among those tested (the culls and the two
// containment tests are symmetric in a and b as a set)
__device__ inline bool paths_intersect_filled(const Env& e, const double* va, int na,
                                              const double* vb, int nb, const float* da,
                                              const float* db, bool* proper = nullptr) {
  if (proper) *proper = false;
  bool parallel = false;
  // da / db: conservative 8-DOPs of a and b
  // A polygon without a finite vertex (a sprite whose position / angle went NaN or inf) is an EMPTY path for matplotlib
  // (PathNanRemover), and path_in_path of an empty path is vacuously true: it overlaps everything (see the oracle).
  // (not gated on the boxes being NaN: the branch costs the step kernel 600 more SGPR spills and 8 % of its time)
  {
    bool fin_a = false, fin_b = false;
    if (e.lane < na) fin_a = isfinite(va[2 * e.lane]) && isfinite(va[2 * e.lane + 1]);
    if (e.lane < nb) fin_b = isfinite(vb[2 * e.lane]) && isfinite(vb[2 * e.lane + 1]);
    for (int k = 64 + e.lane; k < na; k += 64) fin_a = fin_a || (isfinite(va[2 * k]) && isfinite(va[2 * k + 1]));
    for (int k = 64 + e.lane; k < nb; k += 64) fin_b = fin_b || (isfinite(vb[2 * k]) && isfinite(vb[2 * k + 1]));
    const bool fa = __ballot(fin_a) != 0ull, fb = __ballot(fin_b) != 0ull;
    if (!fa || !fb) return (na + 1 >= 3 && !fb) || (nb + 1 >= 3 && !fa);
  }
  bool ka = false, kb = false;
  if (e.lane < na) {
    int i2 = (e.lane + 1 == na) ? 0 : e.lane + 1;
    // NaN coordinates: comparisons are false -> edge kept (conservative)
    ka = !seg_outside_dop(va[2 * e.lane], va[2 * e.lane + 1], va[2 * i2], va[2 * i2 + 1], db);
  }
  if (e.lane < nb) {
    int j2 = (e.lane + 1 == nb) ? 0 : e.lane + 1;
    kb = !seg_outside_dop(vb[2 * e.lane], vb[2 * e.lane + 1], vb[2 * j2], vb[2 * j2 + 1], da);
  }
  unsigned long long ma = __ballot(ka), mb = __ballot(kb);
  int ca = __popcll(ma), cb = __popcll(mb);
  int total = ca * cb;
  if (total > 0) {
    unsigned long long below = (1ull << e.lane) - 1ull;
    if (ka) ELST(e)[__popcll(ma & below)] = (uint8_t)e.lane;
    if (kb) ELST(e)[64 + __popcll(mb & below)] = (uint8_t)e.lane;
    wsync();
    for (int base = 0; base < total; base += 64) {
      int idx = base + e.lane;
      int hit = 0;
      if (idx < total) {
        int ia = (total <= 4096) ? div_small(idx, cb) : idx / cb, ib = idx - ia * cb;
        int i = ELST(e)[ia], j = ELST(e)[64 + ib];
        int i2 = (i + 1 == na) ? 0 : i + 1, j2 = (j + 1 == nb) ? 0 : j + 1;
        double x11 = va[2 * i], y11 = va[2 * i + 1], x12 = va[2 * i2], y12 = va[2 * i2 + 1];
        double x21 = vb[2 * j], y21 = vb[2 * j + 1], x22 = vb[2 * j2], y22 = vb[2 * j2 + 1];
        const bool apart = segs_apart(x11, y11, x12, y12, x21, y21, x22, y22);
        if (!apart) {
          bool dega = mpl_isclose((x11 - x12) * (x11 - x12) + (y11 - y12) * (y11 - y12), 0);
          bool degb = mpl_isclose((x21 - x22) * (x21 - x22) + (y21 - y22) * (y21 - y22), 0);
          if (!dega && !degb) hit = segments_intersect_kind(x11, y11, x12, y12, x21, y21, x22, y22);
        }
      }
      if (__ballot((hit & 3) != 0) != 0ull) {
        if (proper) *proper = __ballot(hit == 2) != 0ull;
        wsync();
        return true;
      }
      parallel = parallel || __ballot(hit == 4) != 0ull;
    }
    wsync();
  }
  // path_in_path: all vertices of b inside a (possible only if box(b) within box(a))
  if ((na + 1 >= 3) & (nb > 0) &
      !((db[0] < da[0] - BB_MARGIN) | (db[1] < da[1] - BB_MARGIN) | (db[4] > da[4] + BB_MARGIN) | (db[5] > da[5] + BB_MARGIN))) {
    bool out = false;
    if (e.lane < nb) out = !point_in_poly(va, na, vb[2 * e.lane], vb[2 * e.lane + 1]);
    if (__ballot(out) == 0ull) return true;
  }
  if ((nb + 1 >= 3) & (na > 0) &
      !((da[0] < db[0] - BB_MARGIN) | (da[1] < db[1] - BB_MARGIN) | (da[4] > db[4] + BB_MARGIN) | (da[5] > db[5] + BB_MARGIN))) {
    bool out = false;
    if (e.lane < na) out = !point_in_poly(vb, nb, va[2 * e.lane], va[2 * e.lane + 1]);
    if (__ballot(out) == 0ull) return true;
  }
  if (proper) *proper = !parallel;
  return false;
}

// ---- conservative bounding volumes (engine-side broad phase) ----------------------------
// Path.intersects_path(filled) can only be true when the two polygons come within
// matplotlib's isclose tolerances (rtol 1e-10, atol 1e-13) of each other, so rejecting
// pairs that are more than BB_MARGIN apart along any axis never changes a result.  Each
// sprite carries an 8-DOP: its extent along x, y, x+y and x-y (float32 is enough: the
// rounding, < 1e-6 over the ten translations of a step, stays inside BB_MARGIN = 1e-5).
// It is built from the cached vertices at kernel start and after every rotation, and
// follows translations.  On the headline workload the two diagonal axes reject 44 % of
// the pairs whose axis-aligned boxes overlap.
#define BB(s, c) (EBB(e)[8 * (s) + (c)])

// extents of n vertices (one lane scans them); NaN vertices are ignored by fmin / fmax
__device__ inline void dop_scan(const double* v, int n, float* d) {
  double l0 = DINF, l1 = DINF, l2 = DINF, l3 = DINF, h0 = -DINF, h1 = -DINF, h2 = -DINF, h3 = -DINF;
  for (int k = 0; k < n; ++k) {
    double x = v[2 * k], y = v[2 * k + 1], p = x + y, m = x - y;
    l0 = fmin(l0, x); h0 = fmax(h0, x); l1 = fmin(l1, y); h1 = fmax(h1, y);
    l2 = fmin(l2, p); h2 = fmax(h2, p); l3 = fmin(l3, m); h3 = fmax(h3, m);
  }
  // no finite vertex at all (fmin / fmax skipped the NaNs, inf made the box infinite): the polygon overlaps everything
  // for the reference (paths_intersect_filled), so its box must never reject a pair -- NaN compares false
  if (!(l0 <= h0 && l1 <= h1 && h0 < DINF && l0 > -DINF && h1 < DINF && l1 > -DINF)) {
    bool any = false;
    for (int k = 0; k < n; ++k) any = any || (isfinite(v[2 * k]) && isfinite(v[2 * k + 1]));
    if (!any) l0 = l1 = l2 = l3 = h0 = h1 = h2 = h3 = __builtin_nan("");
  }
  d[0] = (float)l0; d[1] = (float)l1; d[2] = (float)l2; d[3] = (float)l3;
  d[4] = (float)h0; d[5] = (float)h1; d[6] = (float)h2; d[7] = (float)h3;
}

__device__ __forceinline__ void dop_translate(float* d, double dx, double dy) {
  double dp = dx + dy, dm = dx - dy;
  d[0] = (float)((double)d[0] + dx); d[4] = (float)((double)d[4] + dx);
  d[1] = (float)((double)d[1] + dy); d[5] = (float)((double)d[5] + dy);
  d[2] = (float)((double)d[2] + dp); d[6] = (float)((double)d[6] + dp);
  d[3] = (float)((double)d[3] + dm); d[7] = (float)((double)d[7] + dm);
}

// after (re)creation of sprite s: one lane scans (reset path only)
__device__ inline void bbox_exact_wave(Env& e, int s) {
  wsync();
  if (e.lane == 0) dop_scan(VERT(s), NV(s), &BB(s, 0));
  wsync();
}

// lanes = sprites; each lane scans its own vertex list (kernel prologue)
__device__ inline void bbox_build_all(Env& e) {
  PProg P = EP(e);
  for (int s = e.lane; s < P->n_slots; s += 64) dop_scan(VERT(s), NV(s), &BB(s, 0));
  wsync();
}

__device__ __forceinline__ bool bbox_apart(const Env& e, int s0, int s1) {
  const float* a = &BB(s0, 0);
  const float* b = &BB(s1, 0);
  const float M = (float)BB_MARGIN;
  return (a[0] > b[4] + M) | (b[0] > a[4] + M) | (a[1] > b[5] + M) | (b[1] > a[5] + M) |
         (a[2] > b[6] + M) | (b[2] > a[6] + M) | (a[3] > b[7] + M) | (b[3] > a[7] + M);
}

// `np.linalg.norm(p0 - p1) > r0 + r1` (sprite.py:464-466).  The square root is only
// taken when the squared distance is within 1e-9 (relative) of the threshold, where
// its rounding could matter; elsewhere the comparison of squares decides identically.
__device__ __forceinline__ bool circles_apart(const Env& e, int s0, int s1) {
  double dx = PX(s0) - PX(s1), dy = PY(s0) - PY(s1);
  double d2 = fma(dy, dy, dx * dx), r = MAXR(s0) + MAXR(s1), r2 = r * r;   // (sprite.py:464: a 1-D norm)
  const bool far = (d2 > r2 * (1.0 + 1e-9)) & (r >= 0), near = d2 < r2 * (1.0 - 1e-9);
  if (far | near) return far;
  return sqrt(d2) > r;
}

// sprite.py:462-484.  `prechecked`: the caller has already evaluated the bounding
// circle / box rejects on the current state (broad phase).
__device__ inline bool overlaps(const Env& e, int s0, int s1, bool prechecked = false, bool* proper = nullptr) {
  if (proper) *proper = false;
  if (!prechecked) {
    if (circles_apart(e, s0, s1)) return false;
    if (bbox_apart(e, s0, s1)) return false;
  }
  if (e.dbg & 8) return false;
  if (e.dbg & 128) const_cast<Env&>(e).n_path++;
  PROF_T0;
  SEC(e, SEC_PATH);
  const bool hit = paths_intersect_filled(e, VERT(s0), NV(s0), VERT(s1), NV(s1), &BB(s0, 0), &BB(s1, 0), proper);
  PROF_ADD(e, 0);
  SEC(e, SEC_STEP_CONTROL);
  return hit;
}

// Does sprite s overlap any live sprite of slots [t0, t1)?  Lanes = the other sprites for the bounding
// circle / box rejects (the rejection sampler's inner loop: dozens of earlier sprites per try), then the
// few survivors take the path test one at a time.  Same predicate as a loop over overlaps().
__device__ inline bool overlaps_any(const Env& e, int s, int t0, int t1) {
  for (int base = t0; base < t1; base += 64) {
    const int t = base + e.lane;
    bool cand = false;
    if (t < t1 && t != s && ALIVE(t)) cand = !circles_apart(e, s, t) && !bbox_apart(e, s, t);
    unsigned long long m = __ballot(cand);
    while (m) {
      const int tt = base + __ffsll((long long)m) - 1;
      m &= m - 1ull;
      if (overlaps(e, s, tt, true)) return true;
    }
  }
  return false;
}

// The live sprites of slots [t0, t1) that sprite s may overlap (bounding circle and box not apart; s itself when it is in the
// range), visited in slot order: fn(t) runs the exact test.  Lanes = the other sprites for the rejects -- what a loop over
// overlaps() does one sprite per wave pass (rules and tasks that test one sprite against whole layers).
template <class Fn>
__device__ inline void for_each_near(const Env& e, int s, int t0, int t1, Fn fn) {
  for (int base = t0; base < t1; base += 64) {
    const int t = base + e.lane;
    bool cand = false;
    if (t < t1 && ALIVE(t)) cand = (t == s) || (!circles_apart(e, s, t) && !bbox_apart(e, s, t));
    unsigned long long m = __ballot(cand);
    while (m) {
      const int tt = base + __ffsll((long long)m) - 1;
      m &= m - 1ull;
      fn(tt);
    }
  }
}

// Narrow phase of up to four broad-phase candidates at once, sixteen lanes each.  About nine
// candidates in ten do not overlap, and a lone Path.intersects_path keeps ~10 lanes busy, so the
// ordered pair loop first asks how many of its next candidates are decided "no overlap" by the very
// tests of paths_intersect_filled (same culls, same segment arithmetic); those are skipped together.
// The first candidate that hits, or that needs what sixteen lanes cannot do (a polygon of more than
// 16 vertices, more than 16 surviving edge pairs, a containment test), ends the prefix and takes the
// ordinary path (without repeating the test when its edges were seen to cross).  The state cannot
// change in between: only a hit moves sprites.
// k-th set bit (k = 0 first) of a 16-bit mask, k < popcount: a binary descent on popcounts, in registers
__device__ __forceinline__ int nth_set_bit16(unsigned m, int k) {
  int pos = 0;
  int c = __popc(m & 0xffu);
  if (k >= c) { k -= c; pos = 8; m >>= 8; }
  c = __popc(m & 0xfu);
  if (k >= c) { k -= c; pos += 4; m >>= 4; }
  c = __popc(m & 0x3u);
  if (k >= c) { k -= c; pos += 2; m >>= 2; }
  if (k >= (int)(m & 1u)) pos += 1;
  return pos;
}

// k-th set bit of a 32-bit mask, k < popcount
__device__ __forceinline__ int nth_set_bit32(unsigned m, int k) {
  const int c = __popc(m & 0xffffu);
  const bool hi = k >= c;
  return nth_set_bit16(hi ? m >> 16 : m & 0xffffu, hi ? k - c : k) + (hi ? 16 : 0);
}

#define CAND_SKIP 0xFFFF   // a candidate entry struck from the list (collision_same_layer): counts as rejected
// G lanes per candidate: 16 (four candidates a batch), or -- with -DMOOG_WIDE_BATCH -- 32 (two) for polygons of 17 - 32
// vertices: falling_balls_64's 30-gons never fit sixteen lanes and each of its candidates takes the whole wave's path test.
// Exact either way (the parity suite passes with it); not faster, see narrow_reject_prefix below.
template <int G>
__device__ inline int narrow_reject_prefix_g(const Env& e, int c, int n) {
  constexpr int SH = G == 16 ? 4 : 5;
  constexpr unsigned long long GM = G == 16 ? 0xffffull : 0xffffffffull;
  const int grp = e.lane >> SH, gl = e.lane & (G - 1);
  const int pr0 = grp < n ? (int)ECAND(e)[c + grp] : CAND_SKIP;
  const bool active = pr0 != CAND_SKIP;
  const int pr = active ? pr0 : 0;
  const int s0 = pr >> 8, t = pr & 255;
  // (all the loads that depend only on the pair go out together: this routine is a chain of LDS round
  //  trips on the critical path of a contact-heavy env)
  const int na_ = NV(s0), nb_ = NV(t);
  const double* va = VERT(s0);
  const double* vb = VERT(t);
  const float* da = &BB(s0, 0);
  const float* db = &BB(t, 0);
  const float4 al = *reinterpret_cast<const float4*>(da), ah = *reinterpret_cast<const float4*>(da + 4);
  const float4 bl = *reinterpret_cast<const float4*>(db), bh = *reinterpret_cast<const float4*>(db + 4);
  const int na = active ? na_ : 0, nb = active ? nb_ : 0;
  // this lane's own edge of either polygon (vertex gl and the next one)
  const int ga1 = (gl < na) ? gl : 0, ga2 = (gl + 1 >= na) ? 0 : gl + 1;
  const int gb1 = (gl < nb) ? gl : 0, gb2 = (gl + 1 >= nb) ? 0 : gl + 1;
  const double2 a1 = *reinterpret_cast<const double2*>(va + 2 * ga1), a2 = *reinterpret_cast<const double2*>(va + 2 * ga2);
  const double2 b1 = *reinterpret_cast<const double2*>(vb + 2 * gb1), b2 = *reinterpret_cast<const double2*>(vb + 2 * gb2);
  const bool big = (na > G) | (nb > G);
  // the "all vertices of one inside the other" tests would run (box within box): not here
  bool slow = (na + 1 >= 3) & (nb > 0) &
              !((bl.x < al.x - BB_MARGIN) | (bl.y < al.y - BB_MARGIN) | (bh.x > ah.x + BB_MARGIN) | (bh.y > ah.y + BB_MARGIN));
  slow |= (nb + 1 >= 3) & (na > 0) &
          !((al.x < bl.x - BB_MARGIN) | (al.y < bl.y - BB_MARGIN) | (ah.x > bh.x + BB_MARGIN) | (ah.y > bh.y + BB_MARGIN));
  slow &= active;
  if (__ballot(slow) & GM) return 0;   // the first candidate takes the ordinary path anyway
  if (G == 16) {   // the first candidate is too long for sixteen lanes (and for nothing else): thirty-two, if they do
    const unsigned long long bg = __ballot(big & active), b32 = __ballot(active & ((na > 32) | (nb > 32)));
    if (bg & GM) return (b32 & GM) ? 0 : -1;
  } else if (__ballot(big & active) & GM) return 0;
  slow |= big & active;
  bool ka = false, kb = false;
  if (active && !slow) {
    if (gl < na) ka = !seg_outside_dop_r(a1.x, a1.y, a2.x, a2.y, bl, bh);
    if (gl < nb) kb = !seg_outside_dop_r(b1.x, b1.y, b2.x, b2.y, al, ah);
  }
  const unsigned long long ma = __ballot(ka), mb = __ballot(kb);
  const unsigned ga = (unsigned)((ma >> (G * grp)) & GM), gb = (unsigned)((mb >> (G * grp)) & GM);
  const int ca = __popc(ga), cb = __popc(gb), total = ca * cb;
  slow = slow || total > G;
  int hit = 0;
  if (active && !slow && gl < total) {
    // gl / cb for gl < G, 1 <= cb <= G (the quotient of a half-integer is never near an integer)
    const int ia = (int)(((float)gl + 0.5f) / (float)cb), ib = gl - ia * cb;
    const int i = G == 16 ? nth_set_bit16(ga, ia) : nth_set_bit32(ga, ia);   // the surviving edges, in order
    const int j = G == 16 ? nth_set_bit16(gb, ib) : nth_set_bit32(gb, ib);
    const int i2 = (i + 1 == na) ? 0 : i + 1, j2 = (j + 1 == nb) ? 0 : j + 1;
    const double2 p11 = *reinterpret_cast<const double2*>(va + 2 * i), p12 = *reinterpret_cast<const double2*>(va + 2 * i2);
    const double2 p21 = *reinterpret_cast<const double2*>(vb + 2 * j), p22 = *reinterpret_cast<const double2*>(vb + 2 * j2);
    const double x11 = p11.x, y11 = p11.y, x12 = p12.x, y12 = p12.y;
    const double x21 = p21.x, y21 = p21.y, x22 = p22.x, y22 = p22.y;
    const bool apart = segs_apart(x11, y11, x12, y12, x21, y21, x22, y22);
    if (!apart) {
      bool dega = mpl_isclose((x11 - x12) * (x11 - x12) + (y11 - y12) * (y11 - y12), 0);
      bool degb = mpl_isclose((x21 - x22) * (x21 - x22) + (y21 - y22) * (y21 - y22), 0);
      if (!dega && !degb) hit = segments_intersect_kind(x11, y11, x12, y12, x21, y21, x22, y22);
    }
  }
  const unsigned long long stop = __ballot((hit & 3) != 0 || slow), slows = __ballot(slow), props = __ballot(hit == 2);
  int r = 0;
  while (r < n && ((stop >> (G * r)) & GM) == 0ull) ++r;
  // bit 8: the candidate that ended the prefix is a proven overlap (its edges cross); bit 9: two non-parallel edges do
  if (r == n) return r | 1024;   // bit 10: every candidate looked at is rejected (n of them: four, or two of the long ones)
  if (((slows >> (G * r)) & GM) == 0ull) {
    if ((props >> (G * r)) & GM) r |= 512;
    r |= 256;
  }
  return r;
}

__device__ inline int narrow_reject_prefix(const Env& e, int c, int n) {
  const int r = narrow_reject_prefix_g<16>(e, c, n);
#ifdef MOOG_WIDE_BATCH   // measured (profiles/r05_step_experiments.txt): falling_balls_64 +0.7 %, colliding_predators_32 -2 %: off
  if (r < 0) return narrow_reject_prefix_g<32>(e, c, n < 2 ? n : 2);
#endif
  return r < 0 ? 0 : r;
}

// sprite.py:442-460 (one point, one lane)
__device__ inline bool contains_points1(const Env& e, int s, double x, double y) {
  if (FLAGS(s) & MOOG_F_SYM_CIRCLE) return norm2(x - PX(s), y - PY(s)) <= MAXR(s);
  return point_in_poly(VERT(s), NV(s), x, y);
}
// sprite.py:432-440
__device__ inline bool contains_point(const Env& e, int s, double x, double y) {
  if (FLAGS(s) & MOOG_F_SYM_CIRCLE) return npnorm(x - PX(s), y - PY(s)) < MAXR(s);   // sprite.py:436 (1-D)
  return point_in_poly(VERT(s), NV(s), x, y);
}

// sprite.py:616-633 position setter; lanes = vertices
__device__ inline void set_position(Env& e, int s, double nx, double ny) {
  double dx = nx - PX(s), dy = ny - PY(s);
  double* v = VERT(s);
  int n = NV(s);
  wsync();
  for (int k = e.lane; k < n; k += 64) {   // > 64 vertices: annuli (shapes.py:170-188)
    v[2 * k] = v[2 * k] + dx;
    v[2 * k + 1] = v[2 * k + 1] + dy;
  }
  if (e.lane == 0) {
    PX(s) = nx; PY(s) = ny;
    dop_translate(&BB(s, 0), dx, dy);
  }
  wsync();
}

// sprite.py:426-430 update_pos_from_vel for EVERY live sprite (physics.py:114-117),
// including the reference's float32 propagation (see oracle).  Phase 1, lanes =
// sprites: new position, translation delta, rotation coefficients (the per-sprite
// sin/cos run in parallel).  Phase 2, lanes = (sprite, part): L = 4, 2 or 1 lanes per sprite (64 / L >= S), each
// walking every L-th vertex of its sprite's list: translate (position setter, sprite.py:616-633) then rotate about
// the new position (angle setter, :531-540, matplotlib rotate_around), and the exact 8-DOP of a sprite that rotated
// from the very vertices it has just written.  Per-vertex arithmetic is exactly the reference's; sprites are
// independent so the order does not matter.  (Round 3 walked the vertex ARRAY 64 slots at a time -- capacity, not live
// vertices: 8 rounds on the headline workload, each fetching its sprite's transform through 14 ds_bpermute and its
// slot through a global load, then a third phase re-reading the vertices for the boxes: 13.2 k cycles per call for a
// lone wavefront, a fifth of the mean env's step.)
#define LONG_NV_PER_PART 4   // integrate_all: a vertex list of more than 4 x this is "long"
__device__ __forceinline__ double rdlane_d(double v, int lane) {   // lane: wave uniform
  return __hiloint2double(__builtin_amdgcn_readlane(__double2hiint(v), lane), __builtin_amdgcn_readlane(__double2loint(v), lane));
}
__device__ inline void integrate_all(Env& e, double dt) {
  PProg P = EP(e);
  const int S = P->n_slots;
  wsync();
  for (int sbase = 0; sbase < S; sbase += 64) {
    // lanes per sprite in this chunk of (at most) 64 sprites
    const int nchunk = S - sbase < 64 ? S - sbase : 64;
    const int Lp = nchunk <= 16 ? 4 : (nchunk <= 32 ? 2 : 1);
    const int Sp = 64 / Lp;                       // sprites per pass = lane stride between the parts of one sprite
    // ---- phase 1: lane = sprite sbase + lane
    SEC(e, SEC_INT_POSE);
    double r0 = 0, r1 = 0, r2 = 1, r3 = 0, r4 = 0, r5 = 0;   // ddx, ddy, a, b, tx, ty
    int rmode = 0;                                           // 0 dead, 1 translate, 2 translate + rotate
    {
      const int s = sbase + e.lane;
      if (e.lane < nchunk) {
        int fl = FLAGS(s);
        if (fl & MOOG_F_ALIVE) {
          double dx, dy;
          if (fl & MOOG_F_VEL_F32) {
            float dtf = (float)dt;
            dx = (double)(dtf * (float)VELX(s));
            dy = (double)(dtf * (float)VELY(s));
          } else {
            dx = dt * VELX(s);
            dy = dt * VELY(s);
          }
          double ox = PX(s), oy = PY(s);
          double nx = ox + dx, ny = oy + dy;
          double ddx = nx - ox, ddy = ny - oy;
          double w = ANGV(s);
          double a = 1, b = 0, tx = 0, ty = 0;
          rmode = 1;
          if (w != 0.0) {  // `if self._angle_vel:` (NaN is truthy)
            double dth;
            if (fl & MOOG_F_ANGVEL_F32) {
              float t = (float)dt * (float)w;
              float a_old = (float)ANG(s);
              float a_new = a_old + t;
              dth = (double)(a_new - a_old);
              ANG(s) = (double)a_new;
            } else {
              double a_old = ANG(s);
              double a_new = a_old + dt * w;
              dth = a_new - a_old;
              ANG(s) = a_new;
            }
            sincos_small(dth, &b, &a);
            tx = (a * (-nx) - b * (-ny)) + nx;
            ty = (b * (-nx) + a * (-ny)) + ny;
            rmode = 2;
          }
          r0 = ddx; r1 = ddy; r2 = a; r3 = b; r4 = tx; r5 = ty;
          PX(s) = nx; PY(s) = ny;
          if (rmode != 2) dop_translate(&BB(s, 0), ddx, ddy);
        }
      }
    }
    // ---- phase 2a: a few sprites with long vertex lists (the agent's 30-gon among 10-gons) would set the trip count of
    //      the per-sprite loops below: the whole wave walks such a list, lanes = vertices (transform by readlane)
    SEC(e, SEC_INT_LONG);
    unsigned long long longm = __ballot(e.lane < nchunk && rmode != 0 && NV(sbase + (e.lane < nchunk ? e.lane : 0)) > 4 * LONG_NV_PER_PART);
    if (__popcll(longm) > 4) longm = 0ull;   // (many of them -- falling_balls' discs: the per-sprite loops are the better shape)
    const unsigned long long coopm = longm;
    while (longm) {
      const int l = __ffsll((long long)longm) - 1;
      longm &= longm - 1ull;
      const int s = sbase + l;
      const double x0 = rdlane_d(r0, l), x1 = rdlane_d(r1, l), x2 = rdlane_d(r2, l), x3 = rdlane_d(r3, l);
      const double x4 = rdlane_d(r4, l), x5 = rdlane_d(r5, l);
      const bool rot = __builtin_amdgcn_readlane(rmode, l) == 2;
      double* v = VERT(s);
      const int n = NV(s);
      for (int k = e.lane; k < n; k += 64) {
        double2 p = *reinterpret_cast<const double2*>(v + 2 * k);
        double vx = p.x + x0, vy = p.y + x1;
        if (rot) {
          const double rx = (x2 * vx + (-x3) * vy) + x4;
          const double ry = (x3 * vx + x2 * vy) + x5;
          vx = rx; vy = ry;
        }
        p.x = vx; p.y = vy;
        *reinterpret_cast<double2*>(v + 2 * k) = p;
      }
      if (rot) {   // (rare: one lane scans, as at kernel start)
        wsync();
        if (e.lane == 0) dop_scan(v, n, &BB(s, 0));
      }
    }
    // ---- phase 2b: lane = (sprite sl, part h); the transform comes from lane sl
    SEC(e, SEC_INT_VERTS);
    const int sl = e.lane & (Sp - 1), h = e.lane / Sp;
    double x0 = r0, x1 = r1, x2 = r2, x3 = r3, x4 = r4, x5 = r5;
    int mode = rmode;
    if (Lp > 1) {
      x0 = shfl_d(r0, sl); x1 = shfl_d(r1, sl); x2 = shfl_d(r2, sl); x3 = shfl_d(r3, sl);
      x4 = shfl_d(r4, sl); x5 = shfl_d(r5, sl); mode = __shfl(rmode, sl);
    }
    const int s = sbase + sl;
    const bool on = sl < nchunk && mode != 0 && !((coopm >> sl) & 1ull);
    const int n = on ? NV(s) : 0;
    double* v = on ? VERT(s) : &EF(e)[EL(e).o_verts];
    const float FINF = __builtin_inff();
    float l0 = FINF, l1 = FINF, l2 = FINF, l3 = FINF, h0 = -FINF, h1 = -FINF, h2 = -FINF, h3 = -FINF;
    int fin = 0;
    const bool rot = mode == 2;
    // (branch-free body, the next vertex in flight while this one is transformed; the box in float32: rounding is monotonic,
    //  so the extremes of the rounded coordinates are the rounded extremes dop_scan computes; fminf / fmaxf skip NaNs as there)
    int k = h;
    double2 p = make_double2(0, 0);
    if (k < n) p = *reinterpret_cast<const double2*>(v + 2 * k);
    while (k < n) {
      const int kn = k + Lp;
      double2 pn = p;
      if (kn < n) pn = *reinterpret_cast<const double2*>(v + 2 * kn);
      double vx = p.x + x0, vy = p.y + x1;
      const double rx = (x2 * vx + (-x3) * vy) + x4;
      const double ry = (x3 * vx + x2 * vy) + x5;
      vx = rot ? rx : vx; vy = rot ? ry : vy;
      *reinterpret_cast<double2*>(v + 2 * k) = make_double2(vx, vy);
      const float fx = (float)vx, fy = (float)vy, fp = (float)(vx + vy), fm = (float)(vx - vy);
      l0 = fminf(l0, fx); h0 = fmaxf(h0, fx); l1 = fminf(l1, fy); h1 = fmaxf(h1, fy);
      l2 = fminf(l2, fp); h2 = fmaxf(h2, fp); l3 = fminf(l3, fm); h3 = fmaxf(h3, fm);
      fin |= (int)isfinite(vx) & (int)isfinite(vy);
      p = pn; k = kn;
    }
    // ---- exact boxes of the sprites that rotated: combine the parts, part 0 writes (dop_scan's rule for a polygon
    //      without a finite vertex: a NaN box never rejects a pair)
    SEC(e, SEC_INT_BOXES);
    if (Lp > 1) {
      for (int o = Sp; o < 64; o <<= 1) {
        l0 = fminf(l0, __shfl_xor(l0, o)); h0 = fmaxf(h0, __shfl_xor(h0, o));
        l1 = fminf(l1, __shfl_xor(l1, o)); h1 = fmaxf(h1, __shfl_xor(h1, o));
        l2 = fminf(l2, __shfl_xor(l2, o)); h2 = fmaxf(h2, __shfl_xor(h2, o));
        l3 = fminf(l3, __shfl_xor(l3, o)); h3 = fmaxf(h3, __shfl_xor(h3, o));
        fin |= __shfl_xor(fin, o);
      }
    }
    if (on && rot && h == 0) {
      if (!(l0 <= h0 && l1 <= h1 && h0 < FINF && l0 > -FINF && h1 < FINF && l1 > -FINF) && !fin)
        l0 = l1 = l2 = l3 = h0 = h1 = h2 = h3 = __builtin_nanf("");
      float* d = &BB(s, 0);
      *reinterpret_cast<float4*>(d) = make_float4(l0, l1, l2, l3);
      *reinterpret_cast<float4*>(d + 4) = make_float4(h0, h1, h2, h3);
    }
  }
  wsync();
}

// Tether(update_angle_vel=False) hands every tethered sprite the SAME ndarray
// (tether_physics.py:90): an in-place update through one sprite is seen through all of
// them until a velocity is assigned afresh.  VALIAS(s) names the sharing group; after an
// in-place update the value is copied to the other members.  Programs without such a
// tether (vel_alias == 0, a scalar branch) have no o_valias words at all.
__device__ inline void vel_share(Env& e, int s) {
  if (!EP(e)->vel_alias) return;
  const int g = VALIAS(s);
  if (!g) return;
  const double vx = VELX(s), vy = VELY(s);
  const int S = EP(e)->n_slots;
  for (int t = e.lane; t < S; t += 64)
    if (t != s && ALIVE(t) && VALIAS(t) == g) { VELX(t) = vx; VELY(t) = vy; }
  wsync();
}
__device__ inline void vel_unshare(Env& e, int s) {   // caller: lane 0
  if (EP(e)->vel_alias) VALIAS(s) = 0;
}

__device__ inline void vel_iadd(Env& e, int s, double dx, double dy) {
  double vx = VELX(s), vy = VELY(s);
  if (FLAGS(s) & MOOG_F_VEL_F32) { vx = f32r(vx + dx); vy = f32r(vy + dy); }
  else { vx = vx + dx; vy = vy + dy; }
  wsync();
  if (e.lane == 0) { VELX(s) = vx; VELY(s) = vy; }
  wsync();
  vel_share(e, s);
}

__device__ inline void angvel_iadd(Env& e, int s, double dw) {
  double w = ANGV(s);
  if (FLAGS(s) & MOOG_F_ANGVEL_F32) w = f32r(w + dw);
  else w = w + dw;
  wsync();
  if (e.lane == 0) ANGV(s) = w;
  wsync();
}

// ---- collisions -----------------------------------------------------------------
enum { CV_NONE = 0, CV_OK = 1, CV_FUTURE = 2 };
struct CVec { int status; double px, py, nx, ny, sx, sy, qx, qy; };

__device__ inline void compose(double o[6], const double s[6], const double f[6]) {
  double r0 = s[0] * f[0] + s[1] * f[3];
  double r1 = s[0] * f[1] + s[1] * f[4];
  double r2 = (s[0] * f[2] + s[1] * f[5]) + s[2];
  double r3 = s[3] * f[0] + s[4] * f[3];
  double r4 = s[3] * f[1] + s[4] * f[4];
  double r5 = (s[3] * f[2] + s[4] * f[5]) + s[5];
  o[0] = r0; o[1] = r1; o[2] = r2; o[3] = r3; o[4] = r4; o[5] = r5;
}
__device__ inline void rot_around(double m[6], double x, double y, double th) {
  double a, b;
  sincos_small(th, &b, &a);
  m[0] = a; m[1] = -b; m[2] = (a * (-x) - b * (-y)) + x;
  m[3] = b; m[4] = a; m[5] = (b * (-x) + a * (-y)) + y;
}

// collisions.py:62-98
__device__ inline void relative_motion_matrix(const Env& e, int ps, int as, double dt, double m[6]) {
  double th0, tx0, ty0, th1, tx1, ty1;
  float dtf = (float)dt;
  int fp = FLAGS(ps), fa = FLAGS(as);
  if (fp & MOOG_F_ANGVEL_F32) th0 = (double)((float)(-ANGV(ps)) * dtf);
  else th0 = (-1 * ANGV(ps)) * dt;
  if (fp & MOOG_F_VEL_F32) {
    tx0 = (double)((float)(-VELX(ps)) * dtf); ty0 = (double)((float)(-VELY(ps)) * dtf);
  } else { tx0 = (-1 * VELX(ps)) * dt; ty0 = (-1 * VELY(ps)) * dt; }
  if (fa & MOOG_F_ANGVEL_F32) th1 = (double)((float)ANGV(as) * dtf);
  else th1 = ANGV(as) * dt;
  if (fa & MOOG_F_VEL_F32) {
    tx1 = (double)((float)VELX(as) * dtf); ty1 = (double)((float)VELY(as) * dtf);
  } else { tx1 = VELX(as) * dt; ty1 = VELY(as) * dt; }
  double A[6], B[6] = {1, 0, tx0, 0, 1, ty0}, C[6], D[6] = {1, 0, tx1, 0, 1, ty1};
  rot_around(A, PX(ps), PY(ps), th0);
  rot_around(C, PX(as), PY(as), th1);
  compose(m, B, A);
  compose(m, C, m);
  compose(m, D, m);
}

// collisions.py:101-232.  Lane i < n0 owns vertex i of s0 (a row of the
// crossing-coefficient matrix when contained in s1).
__device__ inline void directed_collision_vectors(const Env& e, int s0, int s1, double dt, CVec& out) {
  out.status = CV_NONE;
  const double* v0 = VERT(s0);
  const double* v1 = VERT(s1);
  int n0 = NV(s0), n1 = NV(s1);
  // lanes j < n1 hold edge j of s1 for both the containment tests and the row scan below
  const int j = e.lane;
  double e1x = 0, e1y = 0, e2x = 0, e2y = 0;
  if (j < n1) {
    int j2 = (j + 1 == n1) ? 0 : j + 1;
    e1x = v1[2 * j]; e1y = v1[2 * j + 1]; e2x = v1[2 * j2]; e2y = v1[2 * j2 + 1];
  }
  // sprite_1.contains_points(vertices_0) (collisions.py:139): lanes = vertices of s0
  double cx = 0, cy = 0;
  bool contained = false, maybe = false;
  const bool disc = (FLAGS(s1) & MOOG_F_SYM_CIRCLE) != 0;
  if (e.lane < n0) {
    cx = v0[2 * e.lane]; cy = v0[2 * e.lane + 1];
    if (disc) contained = norm2(cx - PX(s1), cy - PY(s1)) <= MAXR(s1);   // sprite.py:453-456
    else  // a point outside s1's vertex box (or non-finite) cannot be inside the polygon
      maybe = (int)isfinite(cx) & (int)isfinite(cy) & (int)!point_outside_dop(cx, cy, &BB(s1, 0));
  }
  // even-odd test of the few remaining points, one at a time, lanes = edges of s1
  // (matplotlib point_in_path_impl: parity of the edge toggles, order independent)
  uint64_t mb = __ballot(maybe);
  while (mb) {
    int l = __ffsll((long long)mb) - 1;
    mb &= mb - 1;
    double tx = shfl_d(cx, l), ty = shfl_d(cy, l);
    bool toggle = false;
    if (j < n1) {
      bool f0 = (e1y >= ty), f1 = (e2y >= ty);
      if (f0 != f1) toggle = (((e2y - ty) * (e1x - e2x) >= (e2x - tx) * (e1y - e2y)) == f1);
    }
    int par = __popcll(__ballot(toggle)) & 1;
    if (e.lane == l) contained = (par != 0);
  }
  uint64_t cmask = __ballot(contained);
  if (cmask == 0ull) return;
  if (e.dbg & 32) return;
  double m[6];
  relative_motion_matrix(e, s0, s1, dt, m);
  // Rows of the crossing-coefficient matrix = contained vertices, visited in vertex order by
  // a wave-uniform loop; within a row the lanes are the edges of s1 (sprite.py:108-163).  Per
  // row: np.argmin(|1 - cross_a|) with non-crossings at -inf (first index, NaN wins), the
  // crossing point and its distance; across rows: np.argmax (first max, NaN wins).
  const double ds1x = e2x - e1x, ds1y = e2y - e1y;
  bool anycross = false;
  int ci = -1, e1 = 0;
  double bv = 0, ca = 0, bpx = 0, bpy = 0, bsx = 0, bsy = 0;
  bool cnan = false;
  uint64_t mm = cmask;
  while (mm) {
    int l = __ffsll((long long)mm) - 1;
    mm &= mm - 1;
    double rcx = shfl_d(cx, l), rcy = shfl_d(cy, l);
    double pvx = (m[0] * rcx + m[1] * rcy) + m[2];
    double pvy = (m[3] * rcx + m[4] * rcy) + m[5];
    double ds0x = rcx - pvx, ds0y = rcy - pvy;
    bool crossing = false;
    double cav = -DINF;   // cross_a only matters where the edge is crossed (collisions.py:185)
    if (j < n1) {
      double den = (ds0x * ds1y - ds0y * ds1x) + EPS_INTERP;
      double mx = e1x - pvx, my = e1y - pvy;
      double B = (mx * ds0y - my * ds0x) / den;
      crossing = (B >= 0) && (B <= 1);
      if (crossing) cav = (mx * ds1y - my * ds1x) / den;
    }
    double ab = fabs(1. - cav);
    uint64_t crossm = __ballot(crossing);
    anycross = anycross || (crossm != 0ull);
    // argmin over the lanes j < n1
    uint64_t inrow = (n1 >= 64) ? ~0ull : ((1ull << n1) - 1ull);
    uint64_t nanm = __ballot(isnan(ab)) & inrow;
    int best;
    if (nanm) best = __ffsll((long long)nanm) - 1;
    else {
      double v = (j < n1) ? ab : DINF, mn = v;
      mn = row16_min(mn);
      mn = fmin(mn, shfl_d(mn, e.lane ^ 16));
      mn = fmin(mn, shfl_d(mn, e.lane ^ 32));
      uint64_t eq = __ballot(v == mn) & inrow;
      best = __ffsll((long long)eq) - 1;
    }
    double bca = shfl_d(cav, best);
    double cpx = pvx + bca * (rcx - pvx), cpy = pvy + bca * (rcy - pvy);
    double dfx = rcx - cpx, dfy = rcy - cpy;
    double dist = norm2(dfx, dfy);
    if (dist == DINF) dist = 0;
    bool take;
    if (ci < 0) { take = true; cnan = isnan(dist); }
    else if (cnan) take = false;
    else if (isnan(dist)) { take = true; cnan = true; }
    else take = dist > bv;
    if (take) { ci = l; bv = dist; e1 = best; ca = bca; bpx = cpx; bpy = cpy; bsx = dfx; bsy = dfy; }
  }
  if (!anycross) return;   // `if not np.any(crossings)` (collisions.py:177-179)
  if (e.dbg & 64) return;
  out.px = bpx; out.py = bpy;
  out.sx = bsx; out.sy = bsy;
  out.nx = out.ny = out.qx = out.qy = 0;
  if (ca > 1) { out.status = CV_FUTURE; return; }
  int e2 = (e1 + 1 == n1) ? 0 : e1 + 1;
  double dvx = v1[2 * e2] - v1[2 * e1], dvy = v1[2 * e2 + 1] - v1[2 * e1 + 1];
  double nx = dvy, ny = -1 * dvx;
  double nn = npnorm(nx, ny);
  out.nx = nx / nn; out.ny = ny / nn;
  double sc = npdot2(out.sx, out.sy, dvx, dvy) / npdot2(dvx, dvy, dvx, dvy);
  out.qx = out.sx - dvx * sc;
  out.qy = out.sy - dvy * sc;
  out.status = CV_OK;
}

// Both directed searches of get_collision_vectors at once, 32 lanes each (polygons of <= 32
// vertices): lanes 0-31 run directed(s1, s0), lanes 32-63 directed(s0, s1), with the very
// arithmetic of directed_collision_vectors; the two dependent chains overlap instead of adding up.
__device__ inline void directed_collision_vectors_pair(const Env& e, int sa, int sb, double dt, CVec& ra, CVec& rb) {
#ifdef MOOG_PROFILE
  long long prof_t_ = clock64();
#define PROF_LAP(k) { const long long n_ = clock64(); const_cast<Env&>(e).prof[k] += n_ - prof_t_; prof_t_ = n_; }
#else
#define PROF_LAP(k)
#endif
  SEC(e, SEC_SEARCH_CONTAIN);
  const int h = e.lane >> 5, j = e.lane & 31, hb = h << 5;
  const int s0 = h ? sa : sb, s1 = h ? sb : sa;
  const double* v0 = VERT(s0);
  const double* v1 = VERT(s1);
  const int n0 = NV(s0), n1 = NV(s1);
  double e1x = 0, e1y = 0, e2x = 0, e2y = 0;
  if (j < n1) {
    int j2 = (j + 1 == n1) ? 0 : j + 1;
    e1x = v1[2 * j]; e1y = v1[2 * j + 1]; e2x = v1[2 * j2]; e2y = v1[2 * j2 + 1];
  }
  double cx = 0, cy = 0;
  bool contained = false, maybe = false;
  const bool disc = (FLAGS(s1) & MOOG_F_SYM_CIRCLE) != 0;
  if (j < n0) {
    cx = v0[2 * j]; cy = v0[2 * j + 1];
    if (disc) contained = norm2(cx - PX(s1), cy - PY(s1)) <= MAXR(s1);
    else maybe = (int)isfinite(cx) & (int)isfinite(cy) & (int)!point_outside_dop(cx, cy, &BB(s1, 0));
  }
  unsigned mbh = (unsigned)(__ballot(maybe) >> hb);
  while (__any(mbh != 0u)) {
    const bool on = mbh != 0u;
    const int l = on ? __ffs((int)mbh) - 1 : 0;
    if (on) mbh &= mbh - 1u;
    double tx = shfl_d(cx, hb + l), ty = shfl_d(cy, hb + l);
    const bool f0 = (e1y >= ty), f1 = (e2y >= ty);
    const bool toggle = (int)on & (int)(j < n1) & (int)(f0 != f1) & (int)(((e2y - ty) * (e1x - e2x) >= (e2x - tx) * (e1y - e2y)) == f1);
    const int par = __popc((unsigned)(__ballot(toggle) >> hb)) & 1;
    if (on && j == l) contained = (par != 0);
  }
  const unsigned long long call = __ballot(contained);
  const unsigned cmh = (unsigned)(call >> hb);
  ra.status = CV_NONE; rb.status = CV_NONE;
  PROF_LAP(12);
  if (call == 0ull) return;
  SEC(e, SEC_SEARCH_MATRIX);
  double m[6];
  relative_motion_matrix(e, s0, s1, dt, m);
  PROF_LAP(13);
  SEC(e, SEC_SEARCH_ROWS);
  const double ds1x = e2x - e1x, ds1y = e2y - e1y;
  bool anycross = false;
  int ci = -1, e1 = 0;
  double bv = 0, ca = 0, bpx = 0, bpy = 0, bsx = 0, bsy = 0;
  bool cnan = false;
  unsigned mm = cmh;
  const unsigned inrow = (n1 >= 32) ? ~0u : ((1u << n1) - 1u);
  while (__any(mm != 0u)) {
    const bool on = mm != 0u;
    const int l = on ? __ffs((int)mm) - 1 : 0;
    if (on) mm &= mm - 1u;
    double rcx = shfl_d(cx, hb + l), rcy = shfl_d(cy, hb + l);
    double pvx = (m[0] * rcx + m[1] * rcy) + m[2];
    double pvy = (m[3] * rcx + m[4] * rcy) + m[5];
    double ds0x = rcx - pvx, ds0y = rcy - pvy;
    // (evaluated in every lane: the two quotients cost less than the branches around them)
    const double den = (ds0x * ds1y - ds0y * ds1x) + EPS_INTERP;
    const double mx = e1x - pvx, my = e1y - pvy;
    const double B = (mx * ds0y - my * ds0x) / den;
    const bool crossing = (int)on & (int)(j < n1) & (int)(B >= 0) & (int)(B <= 1);
    const double cav = crossing ? (mx * ds1y - my * ds1x) / den : -DINF;   // cross_a only matters where the edge is crossed
    double ab = fabs(1. - cav);
    const unsigned crossm = (unsigned)(__ballot(crossing) >> hb);
    if (on) anycross = anycross || (crossm != 0u);
    const unsigned nanm = (unsigned)(__ballot(isnan(ab)) >> hb) & inrow;
    double v = (j < n1) ? ab : DINF, mn = v;
    mn = row16_min(mn);
    mn = fmin(mn, shfl_d(mn, e.lane ^ 16));
    const unsigned eq = (unsigned)(__ballot(v == mn) >> hb) & inrow;
    int best = nanm ? __ffs((int)nanm) - 1 : __ffs((int)eq) - 1;
    if (!on || best < 0) best = 0;
    double bca = shfl_d(cav, hb + best);
    double cpx = pvx + bca * (rcx - pvx), cpy = pvy + bca * (rcy - pvy);
    double dfx = rcx - cpx, dfy = rcy - cpy;
    double dist = norm2(dfx, dfy);
    if (dist == DINF) dist = 0;
    {   // np.argmax over the rows so far: first maximum, NaN wins
      const bool first = ci < 0, dn = isnan(dist);
      const bool take = (int)on & ((int)first | ((int)!cnan & ((int)dn | (int)(dist > bv))));
      cnan = on ? (first ? dn : (cnan | dn)) : cnan;
      ci = take ? l : ci; bv = take ? dist : bv; e1 = take ? best : e1; ca = take ? bca : ca;
      bpx = take ? cpx : bpx; bpy = take ? cpy : bpy; bsx = take ? dfx : bsx; bsy = take ? dfy : bsy;
    }
  }
  PROF_LAP(14);
  SEC(e, SEC_SEARCH_FINISH);
  // this half's result (uniform within the half), then one lane of each half speaks for it
  int st = CV_NONE;
  double nx = 0, ny = 0, qx = 0, qy = 0;
  if (cmh != 0u && anycross) {
    if (ca > 1) st = CV_FUTURE;
    else {
      int e2 = (e1 + 1 == n1) ? 0 : e1 + 1;
      double dvx = v1[2 * e2] - v1[2 * e1], dvy = v1[2 * e2 + 1] - v1[2 * e1 + 1];
      double tnx = dvy, tny = -1 * dvx;
      double nn = npnorm(tnx, tny);
      nx = tnx / nn; ny = tny / nn;
      double sc = npdot2(bsx, bsy, dvx, dvy) / npdot2(dvx, dvy, dvx, dvy);
      qx = bsx - dvx * sc;
      qy = bsy - dvy * sc;
      st = CV_OK;
    }
  }
  ra.status = __shfl(st, 0); rb.status = __shfl(st, 32);
  ra.px = shfl_d(bpx, 0); rb.px = shfl_d(bpx, 32);
  ra.py = shfl_d(bpy, 0); rb.py = shfl_d(bpy, 32);
  ra.sx = shfl_d(bsx, 0); rb.sx = shfl_d(bsx, 32);
  ra.sy = shfl_d(bsy, 0); rb.sy = shfl_d(bsy, 32);
  ra.nx = shfl_d(nx, 0); rb.nx = shfl_d(nx, 32);
  ra.ny = shfl_d(ny, 0); rb.ny = shfl_d(ny, 32);
  ra.qx = shfl_d(qx, 0); rb.qx = shfl_d(qx, 32);
  ra.qy = shfl_d(qy, 0); rb.qy = shfl_d(qy, 32);
  PROF_LAP(15);
}

// collisions.py:235-289
// *mirror_status (when given): the status the call with s0 and s1 exchanged would return on the same state.  That call
// runs the same two directed searches and picks between them by the same two norms, compared the other way round.
__device__ inline void get_collision_vectors(const Env& e, int s0, int s1, double dt, CVec& out, int* mirror_status = nullptr) {
  CVec r0, r1;
  if (NV(s0) <= 32 && NV(s1) <= 32 && !(e.dbg & 64)) {
    directed_collision_vectors_pair(e, s0, s1, dt, r0, r1);
  } else {
    directed_collision_vectors(e, s1, s0, dt, r0);
    directed_collision_vectors(e, s0, s1, dt, r1);
  }
  SEC(e, SEC_SEARCH_SELECT);
  double a0x = 0, a0y = 0, a1x = 0, a1y = 0;
  if (r0.status != CV_NONE) {
    r0.nx = -1. * r0.nx; r0.ny = -1. * r0.ny;
    r0.sx = -1. * r0.sx; r0.sy = -1. * r0.sy;
    a0x = r0.sx; a0y = r0.sy;
  }
  if (r1.status != CV_NONE) { a1x = r1.sx; a1y = r1.sy; }
  const double n0 = npnorm(a0x, a0y), n1 = npnorm(a1x, a1y);
  if (mirror_status) *mirror_status = (n1 > n0) ? r1.status : r0.status;
  if (n0 > n1) out = r0;
  else out = r1;
}

// collisions.py:292-350
__device__ inline void collide_without_update_angle_vel(Env& e, int s0, int s1, const CVec& c,
                                                        double elasticity, int symmetric) {
  double nx = c.nx, ny = c.ny;
  double nn = npnorm(nx, ny);
  if (!(fabs(nn - 1.) <= 1e-4 + 1e-5 * 1.)) {
    wsync();
    if (e.lane == 0) EQ(e)[EL(e).o_fault] |= MOOG_FAULT_BAD_NORMAL;
    wsync();
    return;
  }
  double m0 = MASS(s0), m1 = MASS(s1);
  double d0 = npdot2(VELX(s0), VELY(s0), nx, ny);
  double d1 = npdot2(VELX(s1), VELY(s1), nx, ny);
  double v0nx = d0 * nx, v0ny = d0 * ny, v1nx = d1 * nx, v1ny = d1 * ny;
  double cmx, cmy;
  if (symmetric) {
    cmx = (v0nx * m0 + v1nx * m1) / (m0 + m1);
    cmy = (v0ny * m0 + v1ny * m1) / (m0 + m1);
  } else { cmx = v1nx; cmy = v1ny; }
  double f = 1 + elasticity;
  double a0 = f * (cmx - v0nx), a1 = f * (cmy - v0ny), b0 = f * (cmx - v1nx), b1 = f * (cmy - v1ny);
  vel_iadd(e, s0, a0, a1);
  vel_iadd(e, s1, b0, b1);
}

// collisions.py:353-454
__device__ inline void collide_with_update_angle_vel(Env& e, int s0, int s1, const CVec& c,
                                                     double elasticity, int symmetric) {
  double nx = c.nx, ny = c.ny;
  double m0 = MASS(s0), m1 = MASS(s1), w0 = ANGV(s0), w1 = ANGV(s1);
  double i0 = (0 + m0 * INER(s0, 0)) + m0 * INER(s0, 1);
  double i1 = (0 + m1 * INER(s1, 0)) + m1 * INER(s1, 1);
  double v0 = npdot2(VELX(s0), VELY(s0), nx, ny);
  double v1 = npdot2(VELX(s1), VELY(s1), nx, ny);
  double c0x = c.px - PX(s0), c0y = c.py - PY(s0);
  double c1x = c.px - PX(s1), c1y = c.py - PY(s1);
  double r0 = npnorm(c0x, c0y), r1 = npnorm(c1x, c1y);
  double sin0 = (c0x * ny - c0y * nx) / r0, sin1 = (c1x * ny - c1y * nx) / r1;
  double S0 = r0 * sin0, S1 = r1 * sin1;
  double a = m0 + m1 + m0 * m1 * ((S0 * S0 / i0) + (S1 * S1 / i1));
  double b = (1 + elasticity) * (v0 - v1 + w0 * S0 - w1 * S1);
  double dv0, dv1;
  if (symmetric) { dv0 = -1 * m1 * b / a; dv1 = m0 * b / a; }
  else { dv0 = -1 * m1 * b / (a - m0); dv1 = 0.; }
  double dw0 = m0 * dv0 * S0 / i0, dw1 = m1 * dv1 * S1 / i1;
  vel_iadd(e, s0, dv0 * nx, dv0 * ny);
  vel_iadd(e, s1, dv1 * nx, dv1 * ny);
  angvel_iadd(e, s0, dw0);
  angvel_iadd(e, s1, dw1);
}

// collisions.py:658-748 _position_correction, given the crossing point closest
// to sA's centre (p0), its edge index on sA (fwd) and on sB (b0).  Wave-uniform.
__device__ inline void position_correction(const Env& e, double p0x, double p0y, int sA, int fwd,
                                           int sB, int b0, double out[2]) {
  const double* va = VERT(sA);
  const double* vb = VERT(sB);
  int nA = NV(sA), nB = NV(sB);
  int b1 = ((b0 - 1) % nB + nB) % nB;
  double q0x = vb[2 * b1], q0y = vb[2 * b1 + 1];
  double bx = vb[2 * b0] - q0x, by = vb[2 * b0 + 1] - q0y;
  double bn = npnorm(bx, by);
  bx /= bn; by /= bn;
  double sg = npdot2(PX(sB) - q0x, PY(sB) - q0y, bx, by);
  double sgn = isnan(sg) ? sg : (sg > 0 ? 1. : (sg < 0 ? -1. : 0.));
  double nvx = bx * -1 * sgn, nvy = by * -1 * sgn;
  int bwd = ((fwd - 1) % nA + nA) % nA;
  int parity, curi;
  if (npdot2(va[2 * fwd] - p0x, va[2 * fwd + 1] - p0y, nvx, nvy) > 0) { parity = 1; curi = fwd; }
  else if (npdot2(va[2 * bwd] - p0x, va[2 * bwd + 1] - p0y, nvx, nvy) > 0) { parity = -1; curi = bwd; }
  else { out[0] = DINF; out[1] = DINF; return; }
  double worst = 0;
  for (int it = 0; it < 4 * nA; ++it) {
    double pen = npdot2(va[2 * curi] - p0x, va[2 * curi + 1] - p0y, nvx, nvy);
    if (!(pen > 0)) break;
    if (pen > worst) worst = pen;
    curi = ((curi + parity) % nA + nA) % nA;
  }
  out[0] = worst * nvx; out[1] = worst * nvy;
}

// collisions.py:586-655.  Lanes = edge pairs; only the crossing closest to each
// sprite's centre (lowest row-major index on ties) and the crossing count are needed.
// Returns false when it left the state alone (fewer than two crossings, collisions.py:618-619): the caller then knows that
// no sprite moved (a pair that keeps overlapping that way -- two tips crossing -- comes back twice in every sub-step).
__device__ inline bool make_disjoint(Env& e, int s0, int s1, int symmetric) {
  const double* va = VERT(s0);
  const double* vb = VERT(s1);
  int n0 = NV(s0), n1 = NV(s1);
  int total = n0 * n1;
  int cnt = 0;
  double bdA = DINF, bdB = DINF, bxA = 0, byA = 0, bxB = 0, byB = 0;
  int biA = -1, biB = -1;
  double ax = PX(s0), ay = PY(s0), bxc = PX(s1), byc = PY(s1);
  for (int base = 0; base < total; base += 64) {
    int idx = base + e.lane;
    if (idx < total) {
      int i = idx / n1, j = idx - i * n1;
      int i2 = (i + 1 == n0) ? 0 : i + 1, j2 = (j + 1 == n1) ? 0 : j + 1;
      double ds0x = va[2 * i2] - va[2 * i], ds0y = va[2 * i2 + 1] - va[2 * i + 1];
      double ds1x = vb[2 * j2] - vb[2 * j], ds1y = vb[2 * j2 + 1] - vb[2 * j + 1];
      double den = (ds0x * ds1y - ds0y * ds1x) + EPS_INTERP;
      double mx = vb[2 * j] - va[2 * i], my = vb[2 * j + 1] - va[2 * i + 1];
      double A = (mx * ds1y - my * ds1x) / den;
      double B = (mx * ds0y - my * ds0x) / den;
      if ((A > 0) & (A < 1) & (B > 0) & (B < 1)) {
        double cpx = va[2 * i] + A * ds0x, cpy = va[2 * i + 1] + A * ds0y;
        ++cnt;
        double dA = norm2(cpx - ax, cpy - ay), dB = norm2(cpx - bxc, cpy - byc);
        if (biA < 0 || dA < bdA) { bdA = dA; biA = idx; bxA = cpx; byA = cpy; }
        if (biB < 0 || dB < bdB) { bdB = dB; biB = idx; bxB = cpx; byB = cpy; }
      }
    }
  }
  // wave reductions: total count; argmin (distance, then index)
  int tot = cnt;
  for (int o = 32; o > 0; o >>= 1) tot += __shfl_xor(tot, o);
  if (tot <= 1) return false;
  for (int o = 32; o > 0; o >>= 1) {
    double od = shfl_d(bdA, e.lane ^ o), ox = shfl_d(bxA, e.lane ^ o), oy = shfl_d(byA, e.lane ^ o);
    int oi = __shfl_xor(biA, o);
    bool take = (oi >= 0) && (biA < 0 || od < bdA || (od == bdA && oi < biA));
    if (take) { bdA = od; biA = oi; bxA = ox; byA = oy; }
    od = shfl_d(bdB, e.lane ^ o); ox = shfl_d(bxB, e.lane ^ o); oy = shfl_d(byB, e.lane ^ o);
    oi = __shfl_xor(biB, o);
    take = (oi >= 0) && (biB < 0 || od < bdB || (od == bdB && oi < biB));
    if (take) { bdB = od; biB = oi; bxB = ox; byB = oy; }
  }
  biA = uni(biA); biB = uni(biB);
  double c0[2], c1[2];
  position_correction(e, bxA, byA, s0, biA / n1, s1, biA % n1, c0);
  position_correction(e, bxB, byB, s1, biB % n1, s0, biB / n1, c1);
  double cx, cy;
  if (npnorm(c0[0], c0[1]) > npnorm(c1[0], c1[1])) {
    cx = -1 * (1 + EPS_COLL) * c0[0]; cy = -1 * (1 + EPS_COLL) * c0[1];
  } else {
    cx = (1 + EPS_COLL) * c0[0]; cy = (1 + EPS_COLL) * c0[1];
  }
  if (!(isfinite(cx) && isfinite(cy))) { cx = 0; cy = 0; }
  if (symmetric) {
    set_position(e, s0, PX(s0) + 0.5 * cx, PY(s0) + 0.5 * cy);
    set_position(e, s1, PX(s1) - 0.5 * cx, PY(s1) - 0.5 * cy);
  } else {
    set_position(e, s0, PX(s0) + cx, PY(s0) + cy);
  }
  return true;
}

// collisions.py:548-576: position pop-out by `perpendicular` followed by the impulse
// (_collide_with / _collide_without_update_angle_vel, :292-454).  The reference does this
// as six in-place updates; here every new value is computed from registers first (same
// arithmetic, same order) and the state is written once: lanes = vertices for the two
// path translations, lane 0 for the scalars.
__device__ inline void resolve_contact(Env& e, double elasticity, int s0, int s1, const CVec& c,
                                       int symmetric, int upd) {
  const int f0 = FLAGS(s0), f1 = FLAGS(s1);
  // --- pop-out (collisions.py:548-555)
  double p0x = PX(s0), p0y = PY(s0), p1x = PX(s1), p1y = PY(s1);
  double n0x, n0y, n1x = p1x, n1y = p1y;
  if (symmetric) {
    n0x = p0x - (0.5 + EPS_COLL) * c.qx; n0y = p0y - (0.5 + EPS_COLL) * c.qy;
    n1x = p1x + (0.5 + EPS_COLL) * c.qx; n1y = p1y + (0.5 + EPS_COLL) * c.qy;
  } else {
    n0x = p0x - (1. + EPS_COLL) * c.qx; n0y = p0y - (1. + EPS_COLL) * c.qy;
  }
  const double d0x = n0x - p0x, d0y = n0y - p0y, d1x = n1x - p1x, d1y = n1y - p1y;
  // --- impulse, evaluated on the displaced positions
  const double nx = c.nx, ny = c.ny;
  double v0x = VELX(s0), v0y = VELY(s0), v1x = VELX(s1), v1y = VELY(s1);
  double w0 = ANGV(s0), w1 = ANGV(s1);
  const double m0 = MASS(s0), m1 = MASS(s1);
  double a0x, a0y, a1x, a1y, dw0 = 0, dw1 = 0;
  bool fault = false;
  if (upd) {  // collisions.py:353-454
    double i0 = (0 + m0 * INER(s0, 0)) + m0 * INER(s0, 1);
    double i1 = (0 + m1 * INER(s1, 0)) + m1 * INER(s1, 1);
    double v0 = npdot2(v0x, v0y, nx, ny);
    double v1 = npdot2(v1x, v1y, nx, ny);
    double c0x = c.px - n0x, c0y = c.py - n0y;
    double c1x = c.px - n1x, c1y = c.py - n1y;
    double r0 = npnorm(c0x, c0y), r1 = npnorm(c1x, c1y);
    double sin0 = (c0x * ny - c0y * nx) / r0, sin1 = (c1x * ny - c1y * nx) / r1;
    double S0 = r0 * sin0, S1 = r1 * sin1;
    double a = m0 + m1 + m0 * m1 * ((S0 * S0 / i0) + (S1 * S1 / i1));
    double b = (1 + elasticity) * (v0 - v1 + w0 * S0 - w1 * S1);
    double dv0, dv1;
    if (symmetric) { dv0 = -1 * m1 * b / a; dv1 = m0 * b / a; }
    else { dv0 = -1 * m1 * b / (a - m0); dv1 = 0.; }
    dw0 = m0 * dv0 * S0 / i0; dw1 = m1 * dv1 * S1 / i1;
    a0x = dv0 * nx; a0y = dv0 * ny; a1x = dv1 * nx; a1y = dv1 * ny;
  } else {    // collisions.py:292-350
    double nn = npnorm(nx, ny);
    fault = !(fabs(nn - 1.) <= 1e-4 + 1e-5 * 1.);   // np.isclose(norm, 1., atol=1e-4) -> ValueError
    double q0 = npdot2(v0x, v0y, nx, ny);
    double q1 = npdot2(v1x, v1y, nx, ny);
    double v0nx = q0 * nx, v0ny = q0 * ny, v1nx = q1 * nx, v1ny = q1 * ny;
    double cmx, cmy;
    if (symmetric) {
      cmx = (v0nx * m0 + v1nx * m1) / (m0 + m1);
      cmy = (v0ny * m0 + v1ny * m1) / (m0 + m1);
    } else { cmx = v1nx; cmy = v1ny; }
    double f = 1 + elasticity;
    a0x = f * (cmx - v0nx); a0y = f * (cmy - v0ny); a1x = f * (cmx - v1nx); a1y = f * (cmy - v1ny);
  }
  if (!fault) {   // in-place adds with the reference's float32 rounding (see vel_iadd)
    if (f0 & MOOG_F_VEL_F32) { v0x = f32r(v0x + a0x); v0y = f32r(v0y + a0y); } else { v0x = v0x + a0x; v0y = v0y + a0y; }
    if (EP(e)->vel_alias && VALIAS(s0) && VALIAS(s0) == VALIAS(s1)) { v1x = v0x; v1y = v0y; }   // one shared ndarray
    if (f1 & MOOG_F_VEL_F32) { v1x = f32r(v1x + a1x); v1y = f32r(v1y + a1y); } else { v1x = v1x + a1x; v1y = v1y + a1y; }
    if (upd) {
      w0 = (f0 & MOOG_F_ANGVEL_F32) ? f32r(w0 + dw0) : w0 + dw0;
      w1 = (f1 & MOOG_F_ANGVEL_F32) ? f32r(w1 + dw1) : w1 + dw1;
    }
  }
  // --- commit
  double* va = VERT(s0);
  double* vb = VERT(s1);
  const int na = NV(s0), nb = NV(s1);
  wsync();
  if (e.lane < na) { va[2 * e.lane] = va[2 * e.lane] + d0x; va[2 * e.lane + 1] = va[2 * e.lane + 1] + d0y; }
  if (symmetric && e.lane < nb) { vb[2 * e.lane] = vb[2 * e.lane] + d1x; vb[2 * e.lane + 1] = vb[2 * e.lane + 1] + d1y; }
  if (e.lane == 0) {
    PX(s0) = n0x; PY(s0) = n0y;
    dop_translate(&BB(s0, 0), d0x, d0y);
    if (symmetric) { PX(s1) = n1x; PY(s1) = n1y; dop_translate(&BB(s1, 0), d1x, d1y); }
    if (fault) EQ(e)[EL(e).o_fault] |= MOOG_FAULT_BAD_NORMAL;
    else {
      VELX(s0) = v0x; VELY(s0) = v0y; VELX(s1) = v1x; VELY(s1) = v1y;
      if (upd) { ANGV(s0) = w0; ANGV(s1) = w1; }
    }
  }
  wsync();
  if (EP(e)->vel_alias && !fault) {
    if (VALIAS(s0) != VALIAS(s1)) vel_share(e, s0);
    vel_share(e, s1);
  }
}

// collisions.py:494-584.  Returns true when a sprite position changed (the broad
// phase must then be redone for the following pairs); velocity-only outcomes and
// "future contact" no-ops return false.
// `known_hit`: the caller has already seen the two paths intersect on the current state
// `mirror_noop` (when given): set when this call changed nothing AND the call with s0 and s1 exchanged would change
// nothing either as long as neither sprite is touched: either the paths do not overlap, by tests that are symmetric in the
// two sprites (paths_intersect_filled's *proper), or they cross in a pair of non-parallel edges (so overlaps_sprite is true
// both ways round, segments_intersect_kind) and the exchanged search ends "future contact" too.
struct CollP { int symmetric, upd, maxdepth; double elasticity; };   // Collision(...) parameters (wave uniform)
__device__ inline bool collision_step(Env& e, const CollP& F, int s0, int s1, int K, bool known_hit = false, bool proper_hit = false,
                                      bool* mirror_noop = nullptr) {
  const int symmetric = F.symmetric, upd = F.upd, maxdepth = F.maxdepth;
  s0 = uni(s0); s1 = uni(s1);
  bool moved = false;
  if (mirror_noop) *mirror_noop = false;
  for (int depth = 0; depth <= maxdepth; ++depth) {
    SEC(e, SEC_STEP_CONTROL);
    if (s0 == s1) return moved;
    bool proper = proper_hit;
    if (!(known_hit && depth == 0) && !overlaps(e, s0, s1, depth == 0, &proper)) {
      if (mirror_noop && depth == 0) *mirror_noop = proper;   // no overlap, and none the other way round
      return moved;
    }
    if (e.dbg & 16) return moved;
    if (e.dbg & 128) e.n_resp++;
    double dt = 1. / K;
    CVec c;
    int mirror = CV_NONE;
    { PROF_T0; get_collision_vectors(e, s0, s1, dt, c, &mirror); PROF_ADD(e, 1); }
    if (c.status == CV_NONE) {
#ifndef MOOG_COUNT_PREFIX
      if (e.dbg & 128) e.n_disj++;
#endif
      SEC(e, SEC_DISJOINT);
      PROF_T0; const bool dm = make_disjoint(e, s0, s1, symmetric); PROF_ADD(e, 2);
      moved = moved || dm;
      if (!dm) return moved;   // (state untouched: the next depth would repeat this very call)
    } else if (c.status == CV_FUTURE) {
      if (mirror_noop && depth == 0) *mirror_noop = proper && mirror == CV_FUTURE;
      return moved;
    } else {
      SEC(e, SEC_RESOLVE);
      PROF_T0; resolve_contact(e, F.elasticity, s0, s1, c, symmetric, upd); PROF_ADD(e, 3);
      moved = true;
    }
  }
  return moved;
}

// ---- Newtonian forces --------------------------------------------------------------
__device__ inline void newton_apply(Env& e, int s, double fx, double fy, int K) {
  double m = MASS(s);
  if (!isfinite(m)) return;
  double den = m * (double)K;
  vel_iadd(e, s, fx / den, fy / den);
}

// New velocity of sprite s under a single-sprite force other than RandomForce (which draws from the
// env's RNG stream in sprite order); false: the velocity is left alone (infinite mass).  One lane's work.
__device__ inline bool force_single_newvel(const Env& e, PForce F, int s, int K, double* ovx, double* ovy) {
  double fx, fy;
  const double m = MASS(s);
  switch (F->kind) {
    case MOOG_FORCE_DRAG: {
      if (FLAGS(s) & MOOG_F_VEL_F32) {
        if (!isfinite(m)) return false;
        float c = (float)(-1 * F->p0), mf = (float)m, den = (float)(m * (double)K);
        float ffx = (c * (float)VELX(s)) * mf, ffy = (c * (float)VELY(s)) * mf;
        *ovx = (double)((float)VELX(s) + ffx / den);
        *ovy = (double)((float)VELY(s) + ffy / den);
        return true;
      }
      double c = -1 * F->p0;
      fx = (c * VELX(s)) * m; fy = (c * VELY(s)) * m;
      break;
    }
    case MOOG_FORCE_KINETIC_FRICTION: {
      double vn = sqrt(VELX(s) * VELX(s) + VELY(s) * VELY(s));
      double nx = 0, ny = 0;
      if (vn != 0) { nx = VELX(s) / vn; ny = VELY(s) / vn; }
      double c = -1 * F->p0;
      fx = (c * nx) * MASS(s); fy = (c * ny) * MASS(s);
      break;
    }
    case MOOG_FORCE_DOWN_GRAVITY: {
      double gm = F->p0 * MASS(s);
      fx = gm * 0; fy = gm * 1;
      break;
    }
    default: return false;
  }
  // newton_apply + vel_iadd
  if (!isfinite(m)) return false;
  const double den = m * (double)K;
  const double dx = fx / den, dy = fy / den;
  double vx = VELX(s), vy = VELY(s);
  if (FLAGS(s) & MOOG_F_VEL_F32) { vx = f32r(vx + dx); vy = f32r(vy + dy); }
  else { vx = vx + dx; vy = vy + dy; }
  *ovx = vx; *ovy = vy;
  return true;
}

__device__ inline void maze_walk_step(Env& e, PForce F, int s, int K);
__device__ inline void maze_walk_det_step(Env& e, PForce F, int s, int K);

__device__ inline void force_single(Env& e, PForce F, int s, int K) {
  if (F->kind == MOOG_FORCE_RANDOM) {
    double r = 0 + (F->p0 - 0) * next_uniform(e);
    double th = 0 + (2 * 3.14159265358979323846 - 0) * next_uniform(e);
    newton_apply(e, s, r * cos(th), r * sin(th), K);
    return;
  }
  double vx = 0, vy = 0;
  const bool ch = force_single_newvel(e, F, s, K, &vx, &vy);
  if (!ch) return;   // (wave uniform: every lane evaluates the same sprite)
  wsync();
  if (e.lane == 0) { VELX(s) = vx; VELY(s) = vy; }
  wsync();
  vel_share(e, s);
}

// The same for every live sprite of slots [a0, a1) at once, lanes = sprites: without shared velocity
// arrays (program.vel_alias) the sprites of a layer are independent under these forces.
__device__ inline void force_single_layer(Env& e, PForce F, int a0, int a1, int K) {
  wsync();
  for (int s = a0 + e.lane; s < a1; s += 64) {
    double vx = 0, vy = 0;
    if (ALIVE(s) && force_single_newvel(e, F, s, K, &vx, &vy)) { VELX(s) = vx; VELY(s) = vy; }
  }
  wsync();
}

struct XStores;
__device__ inline double eval_expr(Env& e, int off, int s0, int s1, int* out_tag, XStores* st);
template <bool DYN>
__device__ inline void force_pair_newton(Env& e, PForce F, int s0, int s1, int K) {
  double dx = PX(s1) - PX(s0), dy = PY(s1) - PY(s0);
  double dist = npnorm(dx, dy);
  double f0x = 0, f0y = 0, f1x = 0, f1y = 0;
  if (dist != 0.) {
    double ux = dx / dist, uy = dy / dist, mag = 0;
    if (F->kind == MOOG_FORCE_GRAVITY) {
      mag = F->p0 * MASS(s0) * MASS(s1) * dist;
    } else if (F->kind == MOOG_FORCE_DISTANCE_LINEAR) {
      double eh = -1. * F->p0 / F->p1;
      mag = F->p0 + F->p1 * dist;
      if (!F->i0 && dist > eh) mag = 0;
      if (!F->i1 && dist < eh) mag = 0;
    } else if (F->kind == MOOG_FORCE_DISTANCE_SPRING) {
      mag = -1. * F->p0 * (dist - F->p1);
    } else if (F->kind == MOOG_FORCE_DISTANCE_EXPR) {   // any force_fn(distance), traced (the kernels with the expression VM)
#ifndef MOOG_NO_DISTANCE_EXPR   // (A/B builds)
      if constexpr (DYN) { e.xarg = dist; mag = eval_expr(e, F->i0, s0, s1, nullptr, nullptr); }
#endif
    }
    f1x = mag * ux; f1y = mag * uy;
    if (F->symmetric) { f0x = -1 * f1x; f0y = -1 * f1y; }
  }
  newton_apply(e, s0, f0x, f0y, K);
  newton_apply(e, s1, f1x, f1y, K);
}

// ---- rigid tethers (tether_physics.py:16-201) ----------------------------------------------
// The i-th live sprite of layer l (list order), or -1.
__device__ inline int nth_alive(Env& e, int l, int i) {
  PProg P = EP(e);
  for (int s = P->layer_slot0[l]; s < P->layer_slot0[l] + P->layer_nslots[l]; ++s)
    if (ALIVE(s) && i-- == 0) return s;
  return -1;
}

// Visits the members of one tether group in the reference's order: every live sprite of
// the layers (Tether, :129-131) or the zi-th live sprite of each layer (zipped, :199).
template <class F>
__device__ inline void tether_members(Env& e, PCorr C, int zi, F f) {
  PProg P = EP(e);
  if (C->kind == MOOG_CORR_TETHER) {
    for (int a = 0; a < C->n_layers; ++a) {
      int l = C->layers[a];
      for (int s = P->layer_slot0[l]; s < P->layer_slot0[l] + P->layer_nslots[l]; ++s)
        if (ALIVE(s)) f(s);
    }
  } else {
    for (int a = 0; a < C->n_layers; ++a) f(nth_alive(e, C->layers[a], zi));
  }
}

// _tether_sprites (:43-91) for one group, run by lane 0.  Masses are Python floats (the
// lowering rejects sampled masses); the velocity sums follow numpy's float32 / float64
// promotion exactly as the oracle restates it.
__device__ __forceinline__ void tether_group(Env& e, PCorr C, int zi, int K, int group) {
  int n = 0;
  double total_mass = 0;
  tether_members(e, C, zi, [&](int s) { ++n; total_mass = total_mass + MASS(s); });
  if (n == 0 || isinf(total_mass)) return;
  double cx = 0, cy = 0, mx = 0, my = 0;
  bool f32 = true;
  int i = 0;
  tether_members(e, C, zi, [&](int s) {
    cx = cx + MASS(s) * PX(s); cy = cy + MASS(s) * PY(s);
    const bool tf32 = (FLAGS(s) & MOOG_F_VEL_F32) != 0;
    double tx, ty;
    if (tf32) { tx = (double)((float)MASS(s) * (float)VELX(s)); ty = (double)((float)MASS(s) * (float)VELY(s)); }
    else { tx = MASS(s) * VELX(s); ty = MASS(s) * VELY(s); }
    if (i == 0) { mx = 0 + tx; my = 0 + ty; f32 = tf32; }
    else if (f32 && tf32) { mx = (double)((float)mx + (float)tx); my = (double)((float)my + (float)ty); }
    else { mx = mx + tx; my = my + ty; f32 = false; }
    ++i;
  });
  cx = cx / total_mass; cy = cy / total_mass;
  double tvx, tvy;
  if (f32) { tvx = (double)((float)mx / (float)total_mass); tvy = (double)((float)my / (float)total_mass); }
  else { tvx = mx / total_mass; tvy = my / total_mass; }
  if (C->has_anchor) { cx = C->anchor[0]; cy = C->anchor[1]; tvx = 0; tvy = 0; f32 = false; }
  if (C->update_angle_vel) {
    // _change_rotation_coordinates (:16-40): radius and perpendicular about the origin
    auto arm = [&](int s, double& radius, double& perpx, double& perpy) {
      double hx, hy;
      if (FLAGS(s) & MOOG_F_VEL_F32) {
        float dxf = (float)VELX(s) / (float)K, dyf = (float)VELY(s) / (float)K;
        hx = (double)(0.5f * dxf); hy = (double)(0.5f * dyf);
      } else { hx = 0.5 * (VELX(s) / (double)K); hy = 0.5 * (VELY(s) / (double)K); }
      double parx = (PX(s) + hx) - cx, pary = (PY(s) + hy) - cy;
      radius = npnorm(parx, pary);
      parx = parx / radius; pary = pary / radius;
      perpx = 0.0 * parx + (-1.0) * pary; perpy = 1.0 * parx + 0.0 * pary;
    };
    double Ltot = 0, Itot = 0;
    tether_members(e, C, zi, [&](int s) {
      double radius, perpx, perpy;
      arm(s, radius, perpx, perpy);
      double rx, ry;
      if ((FLAGS(s) & MOOG_F_VEL_F32) && f32) { rx = (double)((float)VELX(s) - (float)tvx); ry = (double)((float)VELY(s) - (float)tvy); }
      else { rx = VELX(s) - tvx; ry = VELY(s) - tvy; }
      double perp_vel = npdot2(rx, ry, perpx, perpy);
      double moi = 0 + MASS(s) * INER(s, 0);
      moi = moi + MASS(s) * INER(s, 1);
      double L = perp_vel * MASS(s) * radius;
      L += ANGV(s) * moi;
      double I = moi + MASS(s) * radius * radius;
      Ltot = Ltot + L; Itot = Itot + I;
    });
    const double w = Ltot / Itot;
    tether_members(e, C, zi, [&](int s) {
      double radius, perpx, perpy;
      arm(s, radius, perpx, perpy);   // position and this sprite's velocity are still the old ones
      VELX(s) = tvx + radius * perpx * w;
      VELY(s) = tvy + radius * perpy * w;
      ANGV(s) = w;
      FLAGS(s) &= ~(MOOG_F_VEL_F32 | MOOG_F_ANGVEL_F32);
      vel_unshare(e, s);
    });
  } else {
    tether_members(e, C, zi, [&](int s) {
      VELX(s) = tvx; VELY(s) = tvy;
      ANGV(s) = 0.;
      int fl = FLAGS(s) & ~(MOOG_F_VEL_F32 | MOOG_F_ANGVEL_F32);
      FLAGS(s) = f32 ? (fl | MOOG_F_VEL_F32) : fl;
      VALIAS(s) = group;
    });
  }
}

// Tether.apply_physics (:122-136), TetherZippedLayers.apply_physics (:176-201)
__device__ inline void tether(Env& e, PCorr C, int ci) {
  PProg P = EP(e);
  const int K = P->updates_per_env_step;
  wsync();
  if (e.lane == 0) {
    if (C->kind == MOOG_CORR_TETHER) {
      tether_group(e, C, 0, K, 1 + ci * MOOG_MAX_SLOTS);
    } else if (C->n_layers > 0) {
      int cnt0 = 0;
      bool same = true;
      for (int a = 0; a < C->n_layers; ++a) {
        int l = C->layers[a], cnt = 0;
        for (int s = P->layer_slot0[l]; s < P->layer_slot0[l] + P->layer_nslots[l]; ++s) cnt += ALIVE(s) ? 1 : 0;
        if (a == 0) cnt0 = cnt; else same = same && (cnt == cnt0);
      }
      if (!same) EQ(e)[EL(e).o_fault] |= MOOG_FAULT_TETHER_ZIP;
      else for (int i = 0; i < cnt0; ++i) tether_group(e, C, i, K, 1 + ci * MOOG_MAX_SLOTS + i);
    }
  }
  wsync();
}

// constant_speed.py:34-46
__device__ inline void constant_speed(Env& e, PCorr C) {
  PProg P = EP(e);
  for (int a = 0; a < C->n_layers; ++a) {
    int l = C->layers[a];
    int s0 = P->layer_slot0[l], s1 = s0 + P->layer_nslots[l];
    for (int s = s0; s < s1; ++s) {
      if (!ALIVE(s)) continue;
      double ox = VELX(s), oy = VELY(s);
      bool wr = false;
      if (FLAGS(s) & MOOG_F_VEL_F32) {
        float vx = (float)ox, vy = (float)oy;
        float n = sqrtf(vx * vx + vy * vy);
        if (n != 0.0f) {
          float sp = (float)C->speed;
          ox = (double)((sp * vx) / n); oy = (double)((sp * vy) / n); wr = true;
        }
      } else {
        double n = npnorm(ox, oy);
        if (n != 0.0) { ox = (C->speed * ox) / n; oy = (C->speed * oy) / n; wr = true; }
      }
      wsync();
      if (wr && e.lane == 0) { VELX(s) = ox; VELY(s) = oy; vel_unshare(e, s); }
      wsync();
    }
  }
}

// Broad phase of one ordered pair: both sprites alive, bounding circles (sprite.py:464-466) and conservative
// 8-DOPs not apart.  Symmetric in (s0, t).  Every load goes out before anything is tested.
__device__ __forceinline__ bool broad_pair(const Env& e, int s0, int t, bool in) {
  const int fl0 = FLAGS(s0), fl1 = FLAGS(t);
  const float4 al = *reinterpret_cast<const float4*>(&BB(s0, 0)), ah = *reinterpret_cast<const float4*>(&BB(s0, 4));
  const float4 bl = *reinterpret_cast<const float4*>(&BB(t, 0)), bh = *reinterpret_cast<const float4*>(&BB(t, 4));
  const double2 p0 = *reinterpret_cast<const double2*>(&PX(s0)), p1 = *reinterpret_cast<const double2*>(&PX(t));
  const double r0 = MAXR(s0), r1 = MAXR(t);
  const float M = (float)BB_MARGIN;
  const bool apart = (al.x > bh.x + M) | (bl.x > ah.x + M) | (al.y > bh.y + M) | (bl.y > ah.y + M) |
                     (al.z > bh.z + M) | (bl.z > ah.z + M) | (al.w > bh.w + M) | (bl.w > ah.w + M);
  // A sprite without a finite vertex (NaN box: dop_scan / dop_translate) "overlaps" everything for the reference, but a
  // collision with it changes nothing: neither polygon has a vertex inside the other or a crossing, get_collision_vectors
  // finds no contact, _make_disjoint (collisions.py:586-655) sees fewer than two crossings and returns -- at every depth
  // of the recursion.  Such pairs are left out of the candidates (falling_balls_64: 1 % of the envs hold such a ball after
  // 120 steps, and each paired it with all 63 others in every substep).
  const bool nan_box = !(al.x == al.x) | !(bl.x == bl.x);
  bool cand = (int)in & (int)(s0 != t) & (int)((fl0 & fl1 & MOOG_F_ALIVE) != 0) & (int)!apart & (int)!nan_box;
  // circles_apart, on the values already loaded (computed for every lane: cheaper than a branch around it)
  const double dx = p0.x - p1.x, dy = p0.y - p1.y;
  const double d2 = fma(dy, dy, dx * dx), r = r0 + r1, r2 = r * r;
  const bool far = (d2 > r2 * (1.0 + 1e-9)) & (r >= 0), near = d2 < r2 * (1.0 - 1e-9);
  bool ca = far;
  if (cand & !(far | near)) ca = sqrt(d2) > r;   // (the square root only where its rounding could matter)
  return cand & !ca;
}

// ---- mazes: maze_lib/maze.py Maze, physics/maze_walk.py RandomMazeWalk, physics/maze_physics.py MazePhysics.
//      Scalar logic per sprite: every lane evaluates it redundantly (wave uniform), lane 0 commits; the one
//      vector step is the rotation of a sprite that turned (lanes = vertices).
#ifndef MOOG_WITH_MAZE   // 0: a translation unit whose kernel leaves the maze components out (their local arrays
#define MOOG_WITH_MAZE 1  //    enlarge the scratch frame of every program that shares the kernel)
#endif
#define MAZE_EPS 1e-5   // maze_physics.py:15, maze_walk.py:14

// maze.py:107-112 open_vertex(i, j): inside the matrix and not a wall (maze[j, i])
// row j of the maze: a program constant, or (a maze drawn per reset) part of the env's record
__device__ __forceinline__ uint32_t maze_row(const Env& e, int j) {
  return EP(e)->maze.random ? (uint32_t)EQ(e)[EL(e).o_maze + j] : EP(e)->maze.rows[j];
}
__device__ __forceinline__ int maze_open(const Env& e, long i, long j) {
  const int n = EP(e)->maze.size;
  if (i < 0 || j < 0 || i >= n || j >= n) return 0;
  return !((maze_row(e, (int)j) >> i) & 1u);
}
// maze.py:114-120 valid_directions: [[open(i-1, j), open(i+1, j)], [open(i, j-1), open(i, j+1)]]
__device__ inline void maze_valid_directions(const Env& e, long i, long j, double v[2][2]) {
  v[0][0] = maze_open(e, i - 1, j); v[0][1] = maze_open(e, i + 1, j);
  v[1][0] = maze_open(e, i, j - 1); v[1][1] = maze_open(e, i, j + 1);
}
// numpy floor_divide on doubles (npy_divmod)
__device__ inline double np_floor_divide(double a, double b) {
  if (b == 0) return a / b;
  double mod = fmod(a, b), div = (a - mod) / b;
  if (mod != 0) { if ((b < 0) != (mod < 0)) div -= 1.0; }
  double fl;
  if (div != 0) { fl = floor(div); if (div - fl > 0.5) fl += 1.0; }
  else fl = copysign(0.0, a / b);
  return fl;
}
__device__ __forceinline__ double np_sign(double x) { return isnan(x) ? x : (x > 0 ? 1. : (x < 0 ? -1. : 0.)); }
__device__ __forceinline__ long np_rint_l(double x) { return (long)rint(x); }
__device__ __forceinline__ int argmax_abs2(const double v[2]) {   // np.argmax(np.abs(v)): NaN wins, first on ties
  if (isnan(v[0])) return 0;
  if (isnan(v[1])) return 1;
  return fabs(v[1]) > fabs(v[0]) ? 1 : 0;
}

// velocity := (vx, vy) as a fresh float64 array
__device__ inline void maze_set_velocity(Env& e, int s, double vx, double vy) {
  wsync();
  if (e.lane == 0) { VELX(s) = vx; VELY(s) = vy; FLAGS(s) &= ~MOOG_F_VEL_F32; vel_unshare(e, s); }
  wsync();
}

// maze_walk.py:149-193 RandomMazeWalk._step_sprite (with :52-79 _get_pos_vel and :81-93 _get_nearest_point)
__device__ inline void maze_walk_step(Env& e, PForce F, int s, int K) {
  PProg P = EP(e);
  if (isinf(MASS(s))) return;
  const double speed = F->p0, gs = 1. / P->maze.size, half = 0.5 * gs;
  const double px = PX(s), py = PY(s);
  double vel[2] = {speed * np_sign(VELX(s)), speed * np_sign(VELY(s))};
  const double nx = px + vel[0] / K, ny = py + vel[1] / K;
  const long n0 = np_rint_l(px / gs - 0.5), n1 = np_rint_l(py / gs - 0.5);
  const double ix = gs * n0 + half, iy = gs * n1 + half;
  const double d_next_cur = (0 + fabs(nx - px)) + fabs(ny - py);
  const double d_int_next = (0 + fabs(nx - ix)) + fabs(ny - iy);
  const double d_int_cur = (0 + fabs(nx - ix)) + fabs(ny - iy);   // (the reference measures from next_position here too)
  const bool entering = d_next_cur > d_int_cur && d_next_cur > d_int_next;
  double valid[2][2];
  if (entering) {
    maze_valid_directions(e, n0, n1, valid);
    if (F->i0 & 1) {   // :121-147 _update_valid_directions
      const int axis = argmax_abs2(vel);
      const double direction = np_sign(vel[axis]);
      if (direction != 0) {
        const int fwd = (int)(0.5 * (1 + direction)), back = (int)(0.5 * (1 - direction));
        const bool can_continue = valid[axis][fwd] != 0;
        if (!can_continue && (F->i0 & 2)) { /* every open direction stays valid */ }
        else if (can_continue && (F->i0 & 4)) {
          valid[0][0] = valid[0][1] = valid[1][0] = valid[1][1] = 0;
          valid[axis][fwd] = 1;
        } else valid[axis][back] = 0;
      }
    }
  } else if (vel[0] == 0. && vel[1] == 0.) {
    const double rx = half + n0 * gs, ry = half + n1 * gs;
    const bool on0 = fabs(rx - px) < MAZE_EPS, on1 = fabs(ry - py) < MAZE_EPS;
    if (on0 && on1) maze_valid_directions(e, n0, n1, valid);
    else {
      valid[0][0] = valid[0][1] = valid[1][0] = valid[1][1] = 0;
      const int row = 1 - (on0 ? 0 : (on1 ? 1 : 0));   // 1 - np.argmax(on_grid)
      valid[row][0] = valid[row][1] = 1;
    }
  } else {
    maze_set_velocity(e, s, vel[0], vel[1]);
    return;
  }
  // :183-192: sample = valid_directions * np.random.rand(2, 2); argmax of the flattened sample (first maximum, NaN wins)
  double sample[4];
  for (int k = 0; k < 4; ++k) sample[k] = valid[k >> 1][k & 1] * next_uniform(e);
  int ind = 0;
  for (int k = 1; k < 4; ++k)
    if (!isnan(sample[ind]) && (isnan(sample[k]) || sample[k] > sample[ind])) ind = k;
  const double nv = (1 + MAZE_EPS) * speed * (2 * (ind & 1) - 1);
  if (ind >> 1) vel[1] = nv; else vel[0] = nv;
  maze_set_velocity(e, s, vel[0], vel[1]);
}

// maze_walk.py:225-243 DeterministicMazeWalk._step_sprite (see the oracle's maze_walk_det_step): the prescribed velocities
// are read front to back by whichever sprite asks next and never rewound; the read position is the scalar of a
// MOOG_RULE_STATE_SLOT entry.  A prescribed velocity with a different sign pattern sets the velocity to
// np.clip(..., -speed, -speed) = (-speed, -speed) (NaN where the sum is NaN); otherwise the velocity is left alone.
__device__ inline void maze_walk_det_step(Env& e, PForce F, int s, int K) {
  PProg P = EP(e);
  const double speed = F->p0, gs = 1. / P->maze.size, half = 0.5 * gs;
  const double px = PX(s), py = PY(s);
  const double vel[2] = {speed * np_sign(VELX(s)), speed * np_sign(VELY(s))};
  const double nx = px + vel[0] / K, ny = py + vel[1] / K;
  const long n0 = np_rint_l(px / gs - 0.5), n1 = np_rint_l(py / gs - 0.5);
  const double ix = gs * n0 + half, iy = gs * n1 + half;
  const double d_next_cur = (0 + fabs(nx - px)) + fabs(ny - py);
  const double d_int_next = (0 + fabs(nx - ix)) + fabs(ny - iy);
  const double d_int_cur = (0 + fabs(nx - ix)) + fabs(ny - iy);   // (measured from next_position, as in the reference)
  const bool entering = d_next_cur > d_int_cur && d_next_cur > d_int_next;
  if (!(entering || (vel[0] == 0. && vel[1] == 0.))) return;
  const int slot = EL(e).o_rule + uni(F->symmetric);
  const int k = (int)EF(e)[slot];
  if (k >= F->i1) return;   // `if len(self._step_velocities) > 0`
  const double new0 = P->cand[F->i0 + 2 * k], new1 = P->cand[F->i0 + 2 * k + 1];
  wsync();
  if (e.lane == 0) EF(e)[slot] = (double)(k + 1);
  wsync();
  if (np_sign(new0) != np_sign(vel[0]) || np_sign(new1) != np_sign(vel[1])) {   // np.any(np.sign(new) != np.sign(velocity))
    const double t0 = (1 - MAZE_EPS) * vel[0] + new0, t1 = (1 - MAZE_EPS) * vel[1] + new1;
    maze_set_velocity(e, s, isnan(t0) ? t0 : -speed, isnan(t1) ? t1 : -speed);
  }
}

// maze_physics.py:48-109 _get_position_affordances (the position itself is returned unchanged)
__device__ inline bool maze_affordances(Env& e, const double pos[2], double aff[2][2]) {
  PProg P = EP(e);
  const double gs = 1. / P->maze.size, half = 0.5 * gs;
  long nearest[2], inds[2];
  bool on[2];
  for (int a = 0; a < 2; ++a) {
    nearest[a] = np_rint_l(pos[a] / gs - 0.5);
    const double rounded = half + nearest[a] * gs;
    on[a] = fabs(rounded - pos[a]) < MAZE_EPS;
    inds[a] = (long)np_floor_divide(pos[a] - half, gs);
    if (on[a]) inds[a] = nearest[a];
  }
  aff[0][0] = aff[0][1] = aff[1][0] = aff[1][1] = 0;
  if (!on[0] && !on[1]) {
    wsync();
    if (e.lane == 0) EQ(e)[EL(e).o_fault] |= MOOG_FAULT_OFF_GRID;
    wsync();
    return false;
  }
  if (on[0] && on[1]) {
    double v[2][2];
    maze_valid_directions(e, inds[0], inds[1], v);
    for (int a = 0; a < 2; ++a) { aff[a][0] = v[a][0] * gs * -1.; aff[a][1] = v[a][1] * gs * 1.; }
  } else {
    const int i = 1 - (on[0] ? 0 : 1);
    aff[i][0] = inds[i] * gs + half - pos[i];
    aff[i][1] = (inds[i] + 1) * gs + half - pos[i];
  }
  return true;
}

// maze_physics.py:111-163 _get_new_velocity: recursive in the reference (through a vertex the remainder of the step
// is solved from the vertex's affordances); D bounds the depth (a step crosses at most a few vertices)
template <int D>
__device__ inline bool maze_new_velocity(Env& e, double pos[2], double v[2], double aff[2][2], int axis, double out[2]) {
  if constexpr (D == 0) {
    wsync();
    if (e.lane == 0) EQ(e)[EL(e).o_fault] |= MOOG_FAULT_OFF_GRID;
    wsync();
    out[0] = out[1] = 0;
    return false;
  } else {
    if (axis < 0) axis = argmax_abs2(v);
    if (aff[axis][0] <= v[axis] && v[axis] <= aff[axis][1]) {
      v[1 - axis] = 0;
      out[0] = v[0]; out[1] = v[1];
      return true;
    }
    int direction = (int)(0.5 + 0.5 * np_sign(v[axis]));
    if (aff[axis][direction] == 0) {
      axis = 1 - axis;
      direction = (int)(0.5 + 0.5 * np_sign(v[axis]));
      if (aff[axis][direction] == 0 || v[axis] == 0) { out[0] = out[1] = 0; return true; }
      return maze_new_velocity<D - 1>(e, pos, v, aff, axis, out);
    }
    pos[axis] += aff[axis][direction];
    double vaff[2][2];
    if (!maze_affordances(e, pos, vaff)) { out[0] = out[1] = 0; return false; }
    const double scaling = aff[axis][direction] / v[axis];
    double rem[2] = {(1. - scaling) * v[0], (1. - scaling) * v[1]};
    double post[2];
    if (!maze_new_velocity<D - 1>(e, pos, rem, vaff, -1, post)) { out[0] = out[1] = 0; return false; }
    v[0] *= scaling; v[1] *= scaling;
    v[1 - axis] = 0;
    v[0] += post[0]; v[1] += post[1];
    out[0] = v[0]; out[1] = v[1];
    return true;
  }
}

// sprite.py:531-540 angle setter: the cached path is rotated about the position (lanes = vertices)
__device__ inline void rotate_path(Env& e, int s, double d_theta) {
  const double a = cos(d_theta), b = sin(d_theta);
  const double x = PX(s), y = PY(s);
  const double tx = (a * (-x) - b * (-y)) + x;
  const double ty = (b * (-x) + a * (-y)) + y;
  double* v = VERT(s);
  const int n = NV(s);
  wsync();
  for (int k = e.lane; k < n; k += 64) {
    const double vx = v[2 * k], vy = v[2 * k + 1];
    v[2 * k] = (a * vx + (-b) * vy) + tx;
    v[2 * k + 1] = (b * vx + a * vy) + ty;
  }
  wsync();
  bbox_exact_wave(e, s);
}

// maze_physics.py:186-203 _update_sprite_in_maze + :165-184 _update_sprite_angle
__device__ inline void maze_update_sprite(Env& e, PCorr C, int s) {
  double v[2] = {VELX(s), VELY(s)};
  if ((v[0] == 0 && v[1] == 0) || isnan(v[0]) || isnan(v[1])) return;
  const double max_speed = C->anchor[0], cspeed = C->speed;
  if (!isnan(max_speed))
    for (int a = 0; a < 2; ++a) v[a] = v[a] < -max_speed ? -max_speed : (v[a] > max_speed ? max_speed : v[a]);
  if (!isnan(cspeed))
    for (int a = 0; a < 2; ++a) { v[a] += np_sign(v[a]); v[a] *= cspeed; }
  double pos[2] = {PX(s), PY(s)}, aff[2][2], nv[2];
  if (!maze_affordances(e, pos, aff)) return;
  set_position(e, s, pos[0], pos[1]);   // sprite.position = np.copy(position): a translation by zero
  if (!maze_new_velocity<6>(e, pos, v, aff, -1, nv)) return;
  double new_angle;
  if (nv[0] == 0 && nv[1] == 0) new_angle = __builtin_nan("");
  else if (nv[1] == 0) new_angle = -0.5 * np_sign(nv[0]) * 3.14159265358979323846;
  else if (np_sign(nv[1]) > 0) new_angle = atan(-nv[0] / nv[1]);
  else new_angle = 3.14159265358979323846 + atan(-nv[0] / nv[1]);
  const double old_angle = ANG(s);
  if (!isnan(new_angle) && fabs(new_angle - old_angle) > MAZE_EPS) {
    rotate_path(e, s, new_angle - old_angle);
    if (e.lane == 0) ANG(s) = new_angle;
    wsync();
  }
  maze_set_velocity(e, s, nv[0], nv[1]);
}

// maze_physics.py:205-211 MazePhysics.apply_physics
__device__ inline void maze_physics(Env& e, PCorr C) {
  PProg P = EP(e);
  for (int a = 0; a < C->n_layers; ++a) {
    const int l = C->layers[a];
    for (int s = P->layer_slot0[l]; s < P->layer_slot0[l] + P->layer_nslots[l]; ++s)
      if (ALIVE(s)) maze_update_sprite(e, C, s);
  }
}

// Collision force over (layer la) x (layer lb): the reference visits ordered pairs
// (s0, s1), s0 outer, sequentially (physics.py:103-108) and every resolved contact
// moves sprites, so later pairs must see the new state.  Broad phase: the flattened
// pair index space is scanned 64 pairs per wave instruction (bounding circles as in
// overlaps_sprite + conservative boxes) and the survivors are appended, in
// reference order, to a candidate list in LDS.  The list is consumed sequentially by
// the narrow phase; as soon as a pair actually overlapped (state may have changed),
// the list is discarded and rebuilt from the next pair on.
__device__ inline void collision_layer_pair(Env& e, const CollP& F, int a0, int a1, int b0,
                                            int b1, int K) {
  const int nB = b1 - b0, total = (a1 - a0) * nB;
  int start = 0;
  while (start < total) {
    // ---- build: candidates with flattened index >= start --------------------------------
    // (two chunks of 64 pairs per round and a shorter look-ahead were tried: no gain)
    int count = 0, scanned = start;
    wsync();
    PROF_T0;
    SEC(e, SEC_PAIR_SCAN);
    while (scanned < total && count <= CAND_CAP - 64) {
      // Every load of the round goes out before anything is tested (a lane's pair is known from its index
      // alone), so a round costs one LDS round trip instead of a chain of dependent ones: this loop is
      // the largest single piece of an env's critical path.
      const int idx = scanned + e.lane;
      const bool in = idx < total;
      const int idc = in ? idx : 0;
      const int i = (total <= 4096 && nB <= 128) ? div_small(idc, nB) : idc / nB;
      const int s0 = a0 + i, t = b0 + (idc - i * nB);
      const bool cand = broad_pair(e, s0, t, in);
      uint64_t m = __ballot(cand);
      if (cand) ECAND(e)[count + __popcll(m & ((1ull << e.lane) - 1ull))] = (uint16_t)((s0 << 8) | t);
      count += __popcll(m);
      scanned += 64;
    }
    wsync();
    PROF_ADD(e, 4);
    if (scanned > total) scanned = total;
    // ---- consume ----------------------------------------------------------------------------
    bool rebuilt = false;
    for (int c = 0; c < count; ++c) {
      bool known_hit = false;
      SEC(e, SEC_PAIR_CONSUME);
      if (count - c >= 2 && !(e.dbg & (4 | 32))) {   // skip the leading candidates that do not overlap
        PROF_T0;
        SEC(e, SEC_BATCH);
        const int n = count - c < 4 ? count - c : 4;
        const int rr = uni(narrow_reject_prefix(e, c, n)), r = rr & 255;
#ifdef MOOG_COUNT_PREFIX
        if (e.dbg & 128) { e.n_disj += 1 + 1000 * n; }
#endif
        PROF_ADD(e, 8);
        SEC(e, SEC_PAIR_CONSUME);
        known_hit = (rr & 256) != 0;
        c += r;
        if (rr & 1024) { --c; continue; }   // all of them: on to the next batch
      }
      int pr = uni((int)ECAND(e)[c]);
      int s0 = pr >> 8, t = pr & 255;
      if (!(e.dbg & 4) && collision_step(e, F, s0, t, K, known_hit)) {
        start = (s0 - a0) * nB + (t - b0) + 1;
        rebuilt = true;
        break;
      }
    }
    if (!rebuilt) start = scanned;
  }
}

// The same for a layer against itself (n <= 64 sprites), where the broad-phase test is symmetric: the
// candidate pairs live in a bit matrix (row i = partners of sprite a0 + i), filled from the n (n - 1) / 2
// unordered pairs (half the rounds of the ordered scan), and after a contact only the rows / columns of
// the sprites it moved are re-tested (one round) instead of re-scanning every later pair.  The ordered
// candidate list the narrow phase consumes is written from the matrix, rows in order, bits in order:
// exactly the list the ordered scan would build.
// (Tried in round 4 and dropped, both measured with tools/fn_bench.py: the matrix for two different layers -- the list
//  builder's prefix sums cost the three small scans of the headline workload more than re-scanning after a contact,
//  20.9 k cycles per sub-step against 18.9 k; and filling the matrix with lane = row sprite walking its partners -- fewer
//  loads, but a dependent LDS round trip per partner instead of one per 64 pairs: 27.5 k.)
__device__ inline void collision_same_layer(Env& e, const CollP& F, int a0, int a1, int K) {
  const int n = a1 - a0, total = n * n;
  const int symmetric = F.symmetric;
  unsigned long long* rowm = EROWM(e);
  wsync();
  PROF_T0;
  SEC(e, SEC_BROAD);
  if (e.lane < n) rowm[e.lane] = 0ull;
  wsync();
  {   // unordered pairs by rounds of a round-robin: row r pairs column c with c + r + 1 (mod n)
    const int h = (n - 1) >> 1;                        // full rows
    const int npairs = h * n + ((n & 1) ? 0 : n >> 1);   // (an even n has a last half row)
    for (int base = 0; base < npairs; base += 64) {
      const int idx = base + e.lane;
      const bool in = idx < npairs;
      const int idc = in ? idx : 0;
      const int r = div_small(idc, n), c = idc - r * n;
      int j = c + r + 1;
      if (j >= n) j -= n;
      const bool cand = broad_pair(e, a0 + c, a0 + j, in);
      if (cand) { atomicOr(&rowm[c], 1ull << j); atomicOr(&rowm[j], 1ull << c); }
    }
  }
  wsync();
  PROF_ADD(e, 4);
  // Lane r: the columns c for which the ordered pair (a0 + r, a0 + c) is known to change nothing -- its mirror image
  // (a0 + c, a0 + r) was visited earlier in this sub-step, overlapped in a pair of non-parallel edges and ended "future
  // contact" both ways round (collision_step's mirror_noop), and neither sprite has been touched since.  The reference
  // visits both orders (physics.py:103-108); in the contact-heavy envs that set the kernel's duration three searches in
  // four end "future contact", and a pair that overlaps without colliding does so twice in every sub-step.
  unsigned long long skipbits = 0ull;
  const bool use_skip = !uni(EP(e)->vel_alias);   // (velocity arrays shared across sprites: a contact elsewhere may touch the pair)
  int start = 0;
  while (start < total) {
    // ---- the ordered list of candidates with flattened index >= start, as many whole rows as fit --------
    PROF_T0;
    SEC(e, SEC_LIST);
    const int srow = div_small(start, n), scol = start - srow * n;
    unsigned long long bits = 0ull;
    if (e.lane < n && e.lane >= srow) {
      bits = rowm[e.lane] & ~skipbits;
      if (e.lane == srow) bits &= ~((1ull << scol) - 1ull);
    }
    const int cnt = __popcll(bits);
    int inc = cnt;
    for (int o = 1; o < 64; o <<= 1) {
      const int tt = __shfl_up(inc, o);
      if (e.lane >= o) inc += tt;
    }
    const unsigned long long over = __ballot(inc > CAND_CAP);
    const int rows_end = over ? (__ffsll((long long)over) - 1) : 64;   // rows >= rows_end do not fit: next pass
    int count = 0;
    {
      const int last = rows_end - 1 < 63 ? rows_end - 1 : 63;
      count = rows_end > 0 ? __shfl(inc, last) : 0;
      if (rows_end <= srow) {   // (a single row with more than CAND_CAP partners cannot happen: n <= 64 < CAND_CAP)
        count = 0;
      }
    }
    if (e.lane < rows_end) {
      int pos = inc - cnt;
      unsigned long long b = bits;
      while (b) {
        const int j = __ffsll((long long)b) - 1;
        b &= b - 1ull;
        ECAND(e)[pos++] = (uint16_t)(((a0 + e.lane) << 8) | (a0 + j));
      }
    }
    const int scanned = rows_end >= n ? total : rows_end * n;
    wsync();
    PROF_ADD(e, 4);
    // ---- consume ----------------------------------------------------------------------------
    bool rebuilt = false;
    // The mirror image (j, i) of a pair (i, j), i < j, that turned out to be a no-op both ways round is struck from the list
    // (when the list reaches that far; `skipbits` keeps it out of later lists); a struck entry counts as rejected.
    auto mark_mirror = [&](int s0k, int tk) {   // (s0k, tk): wave uniform, tk > s0k
      const int j = tk - a0, i = s0k - a0;
      if (e.lane == j) skipbits |= 1ull << i;
      // struck from the list in hand, wherever it is: the lanes look at the (at most CAND_CAP = 128) entries together -- keeping
      // the list builder's per-row offsets alive for this instead cost the kernels that carry the expression VM 200 VGPR spills
      const int key = (tk << 8) | s0k;
      for (int k = e.lane; k < count; k += 64)
        if ((int)ECAND(e)[k] == key) ECAND(e)[k] = (uint16_t)CAND_SKIP;
    };
    for (int c = 0; c < count; ++c) {
      bool known_hit = false, proper_hit = false;
      SEC(e, SEC_CONSUME);
      if (count - c >= 2 && !(e.dbg & (4 | 32))) {   // pass over the leading candidates that do not overlap
        PROF_T0;
        SEC(e, SEC_BATCH);
        const int nn = count - c < 4 ? count - c : 4;
        const int rr = uni(narrow_reject_prefix(e, c, nn)), r = rr & 255;
#ifdef MOOG_COUNT_PREFIX   // (analysis builds: batches in the make_disjoint counter, tools/heavy_bench.py)
        if (e.dbg & 128) { e.n_disj += 1 + 1000 * nn; }
#endif
        SEC(e, SEC_CONSUME);
        PROF_ADD(e, 8);
        known_hit = (rr & 256) != 0;
        proper_hit = (rr & 512) != 0;
        c += r;
        if (rr & 1024) { --c; continue; }   // all of them: on to the next batch
      }
      const int pr = uni((int)ECAND(e)[c]);
      if (pr == CAND_SKIP) continue;
      const int s0 = pr >> 8, t = pr & 255;
      if (e.dbg & 4) continue;
      bool noop = false;
      if (collision_step(e, F, s0, t, K, known_hit, proper_hit, &noop)) {
        start = (s0 - a0) * n + (t - a0) + 1;
        // re-test the pairs of the sprites the contact moved (s0; t as well when symmetric)
        PROF_T0;
        SEC(e, SEC_RETEST);
        for (int w = 0; w < (symmetric ? 2 : 1); ++w) {
          const int m = (w ? t : s0) - a0;
          wsync();
          const bool cand = broad_pair(e, a0 + m, a0 + (e.lane < n ? e.lane : 0), e.lane < n);
          const unsigned long long cm = __ballot(cand);
          if (e.lane < n) {
            unsigned long long rw = rowm[e.lane] & ~(1ull << m);
            if (cand) rw |= 1ull << m;
            rowm[e.lane] = (e.lane == m) ? cm : rw;
          }
          wsync();
        }
        // what was known about pairs of s0 or t is stale (position, velocity or angular velocity changed)
        skipbits &= ~((1ull << (s0 - a0)) | (1ull << (t - a0)));
        if (e.lane == s0 - a0 || e.lane == t - a0) skipbits = 0ull;
        PROF_ADD(e, 4);
        rebuilt = true;
        break;
      }
      SEC(e, SEC_CONSUME);
      // nothing changed, and nothing would with the sprites exchanged: the mirror image (t, s0) comes later in this sub-step
      if (noop && use_skip && t > s0 && !(e.dbg & 512)) { mark_mirror(s0, t); wsync(); }
    }
    if (!rebuilt) start = scanned;
  }
}

// physics.py:88-117 (one substep).  DYN: the kernel variant that carries the rarely used components (here the maze
// walk / MazePhysics, whose scratch frame must not weigh on the plain step kernel).
template <bool DYN>
__device__ __forceinline__ void apply_physics(Env& e) {   // (forced: see the note at moog_step_kernel)
  PProg P = EP(e);
  const int K = uni(P->updates_per_env_step);
  const int n_fops = e.n_fops;
  for (int k = 0; k < n_fops; ++k) {
    SEC(e, SEC_FORCES);
    PFOp op = &e.fops[k];
    const int kind = uni(op->kind), n_b = uni(op->n_b);
    const int a0 = uni(op->a0), a1 = uni(op->a1), b0 = uni(op->b0), b1 = uni(op->b1);
    if (kind == MOOG_FORCE_COLLISION) {
      if (!(e.dbg & 1)) {
        PROF_T0;
        CollP cp;
        cp.symmetric = uni(op->symmetric); cp.upd = uni(op->i0); cp.maxdepth = uni(op->i1); cp.elasticity = op->p0;
        if (a0 == b0 && a1 == b1 && a1 - a0 <= 64 && a1 - a0 >= 2) collision_same_layer(e, cp, a0, a1, K);
        else collision_layer_pair(e, cp, a0, a1, b0, b1, K);
        PROF_ADD(e, 7);
      }
      continue;
    }
    PForce F = &P->forces[uni(op->fi)];
    if (n_b == 0) {
      if (kind != MOOG_FORCE_RANDOM && kind != MOOG_FORCE_MAZE_WALK && kind != MOOG_FORCE_MAZE_WALK_DET && !uni(P->vel_alias)) {
        force_single_layer(e, F, a0, a1, K);
      } else {
        for (int s = a0; s < a1; ++s)
          if (ALIVE(s)) {
            if constexpr (DYN && MOOG_WITH_MAZE) {
              if (kind == MOOG_FORCE_MAZE_WALK) { maze_walk_step(e, F, s, K); continue; }
              if (kind == MOOG_FORCE_MAZE_WALK_DET) { maze_walk_det_step(e, F, s, K); continue; }
            }
            force_single(e, F, s, K);
          }
      }
    } else {
      for (int s0 = a0; s0 < a1; ++s0) {
        if (!ALIVE(s0)) continue;
        for (int s1 = b0; s1 < b1; ++s1)
          if (ALIVE(s1)) force_pair_newton<DYN>(e, F, s0, s1, K);
      }
    }
  }
  const int n_corr = uni(P->n_corrective);
  for (int c = 0; c < n_corr; ++c) {
    if (P->corrective[c].kind == MOOG_CORR_CONSTANT_SPEED) constant_speed(e, &P->corrective[c]);
    else if (P->corrective[c].kind == MOOG_CORR_MAZE) { if constexpr (DYN && MOOG_WITH_MAZE) maze_physics(e, &P->corrective[c]); }
    else tether(e, &P->corrective[c], c);
  }
  { PROF_T0; if (!(e.dbg & 2)) integrate_all(e, 1. / K); PROF_ADD(e, 5); }
}

// ---- game rules ------------------------------------------------------------------------
__device__ inline double np_remainder1(double a) {
  double m = fmod(a, 1.0);
  if (m != 0) { if (m < 0) m += 1.0; }
  else m = 0.0;
  return m;
}

// ---- sprite expressions (MOOG_X_*): config callables traced at build time by
//      moog/_symbolic.py, evaluated wave-uniformly with numpy 2 scalar promotion.  The value
//      stack (16 doubles) and the pending attribute writes (13 doubles) live in the LDS words
//      of the collision candidate list, which is idle outside apply_physics.
struct XStores { unsigned mask; unsigned tags; };   // tags: 2 bits per attribute

__device__ inline double xattr(Env& e, int s, int a, int& tag) {
  const bool has = EP(e)->sprite_factors != 0;
  const int fm = has ? FMASK(s) : 0;
  const int fl = FLAGS(s);
  switch (a) {
    case MOOG_XA_X: tag = 2; return PX(s);
    case MOOG_XA_Y: tag = 2; return PY(s);
    case MOOG_XA_XVEL: tag = (fl & MOOG_F_VEL_F32) ? 1 : 2; return VELX(s);
    case MOOG_XA_YVEL: tag = (fl & MOOG_F_VEL_F32) ? 1 : 2; return VELY(s);
    case MOOG_XA_ANGLE: tag = ((fm >> MOOG_FAC_ANGLE) & 1) ? 1 : 2; return ANG(s);
    case MOOG_XA_ANGVEL: tag = (fl & MOOG_F_ANGVEL_F32) ? 1 : 2; return ANGV(s);
    case MOOG_XA_MASS: tag = (fm >> MOOG_FAC_MASS) & 1; return MASS(s);
    case MOOG_XA_C0: tag = (fm >> MOOG_FAC_C0) & 1; return COL(s, 0);
    case MOOG_XA_C1: tag = (fm >> MOOG_FAC_C1) & 1; return COL(s, 1);
    case MOOG_XA_C2: tag = (fm >> MOOG_FAC_C2) & 1; return COL(s, 2);
    case MOOG_XA_OPACITY: tag = 0; return (double)OPAC(s);
    case MOOG_XA_SCALE: tag = (fm >> MOOG_FAC_SCALE) & 1; return has ? SCALE(s) : __builtin_nan("");
    case MOOG_XA_ASPECT: tag = (fm >> MOOG_FAC_ASPECT) & 1; return has ? ASPECT(s) : __builtin_nan("");
    default: tag = 0; return __builtin_nan("");
  }
}

__device__ inline double np_rem(double a, double b) {   // numpy remainder (npy_divmod)
  double mod = fmod(a, b);
  if (b == 0) return mod;
  if (mod != 0) { if ((b < 0) != (mod < 0)) mod += b; }
  else mod = copysign(0.0, b);
  return mod;
}
__device__ inline float np_remf(float a, float b) {
  float mod = fmodf(a, b);
  if (b == 0) return mod;
  if (mod != 0) { if ((b < 0) != (mod < 0)) mod += b; }
  else mod = copysignf(0.0f, b);
  return mod;
}

// LANES: every lane evaluates the expression for ITS OWN sprite (s0 / s1 differ per lane; the code is still fetched once per
// wave): filters over whole layers, 64 sprites per pass instead of one.  The value stack is then per lane, in e.xstack
// (program.xstack_depth entries per lane); only pure one-sprite expressions take this path (MOOG_FILTER_EXPR_LANES).
template <bool LANES>
__device__ inline double eval_expr_t(Env& e, int off, int s0, int s1, int* out_tag, XStores* st) {
  PProg P = EP(e);
  double* vbase = LANES ? e.xstack + e.lane : reinterpret_cast<double*>(ECAND(e));   // [MOOG_X_STACK] (x 64 lanes, lane-minor)
  double* sv = reinterpret_cast<double*>(ECAND(e)) + MOOG_X_STACK;   // pending writes [13] (never with LANES)
  constexpr int VS = LANES ? 64 : 1;
#define v(i) vbase[(i) * VS]
  unsigned tags = 0;                                       // 2 bits per stack entry
  int n = 0;
  wsync();
#define XTAG(i) ((int)((tags >> (2 * (i))) & 3u))
#define XSETTAG(i, t) tags = (tags & ~(3u << (2 * (i)))) | ((unsigned)(t) << (2 * (i)))
  for (int pc = uni(off);; ++pc) {   // (the code pointer is wave uniform: scalar fetches, not one vector load per instruction)
    PDinstr I = &P->dcode[uni(pc)];
    const int op = uni(I->op);
    if (op == MOOG_X_END) break;
    if (op == MOOG_X_CONST) { v(n) = I->x; XSETTAG(n, I->b ? 2 : 0); ++n; continue; }
    if (op == MOOG_X_ATTR) { int t; v(n) = xattr(e, I->b ? s1 : s0, I->a, t); XSETTAG(n, t); ++n; continue; }
    if (op == MOOG_X_RULE_STATE) { v(n) = EF(e)[EL(e).o_rule + I->a]; XSETTAG(n, 0); ++n; continue; }
    if (op == MOOG_X_SLOT_CONST) { v(n) = P->cand[I->a + (I->b ? s1 : s0)]; XSETTAG(n, 0); ++n; continue; }   // sprite.metadata[key]
    if (op == MOOG_X_RULE_STATE2) { v(n) = EF(e)[EL(e).o_rule2 + I->a]; XSETTAG(n, 0); ++n; continue; }
    if (op == MOOG_X_ARG) { v(n) = e.xarg; XSETTAG(n, 2); ++n; continue; }   // (np.linalg.norm gives a float64)
    // (values an initializer left in the record, read by rules and tasks while stepping: in every kernel, so that a program
    //  whose INITIALIZER needs the kernels that carry every component can still be stepped by the others, moog_kernels.h
    //  "late reset")
    if (op == MOOG_X_HDRAW_T) { v(n) = EF(e)[EL(e).o_hdraw + I->a]; XSETTAG(n, (int)EF(e)[EL(e).o_hdraw + I->a + 1]); ++n; continue; }
    if (op == MOOG_X_ZIP_ATTR) {   // the sprite at s0's list position in layer b (zip(state[A], state[B]) in a config-local rule)
      const int partner = P->layer_slot0[I->b] + (s0 - P->layer_slot0[P->slot_layer[s0]]);
      int t; v(n) = xattr(e, partner, I->a, t); XSETTAG(n, t); ++n; continue;
    }
    if (op == MOOG_X_FMA) {   // np.dot / 1-D np.linalg.norm of float64 2-vectors: one rounding (npdot2); float32: two
      n -= 2;
      const double a = v(n - 1), b = v(n), c = v(n + 1);
      const int ta = XTAG(n - 1), tb = XTAG(n), tc = XTAG(n + 1);
      const bool any2 = ta == 2 || tb == 2 || tc == 2, any1 = ta == 1 || tb == 1 || tc == 1;
      if (any1 && !any2) { const float p = (float)a * (float)b; v(n - 1) = (double)(p + (float)c); }
      else v(n - 1) = fma(a, b, c);
      XSETTAG(n - 1, any2 ? 2 : (any1 ? 1 : 0));
      continue;
    }
    if (op == MOOG_X_HDRAW) { v(n) = EF(e)[EL(e).o_hdraw + I->a]; XSETTAG(n, 0); ++n; continue; }   // (a Python float)
    if (op == MOOG_X_SLOT_ATTR) { int t; v(n) = xattr(e, I->b, I->a, t); XSETTAG(n, t); ++n; continue; }
    if constexpr (MOOG_WITH_MAZE != 0) {   // expressions about the sprite being created: only in the kernels that carry every component
      if (op == MOOG_X_STORE_VERT) {   // raw shape coordinate -> the vertex area of the slot being created
        --n;
        if (e.lane == 0) VERT(e.cur_slot)[I->a] = v(n);
        continue;
      }
      if (op == MOOG_X_FACTOR) {   // a factor of the sprite being created, staged by sample_factors
        v(n) = reinterpret_cast<const double*>(ELST(e))[I->a];
        XSETTAG(n, ((e.fac_f32 >> I->a) & 1u) ? 1 : 0);
        ++n;
        continue;
      }
    }
    if (op == MOOG_X_OVERLAPS_SLOTS) {   // two fixed sprites (an initializer's look-ahead, state-level task functions)
      const bool ov = ALIVE(I->a) && ALIVE(I->b) && overlaps(e, I->a, I->b);
      v(n) = ov ? 1.0 : 0.0; XSETTAG(n, 0); ++n;
      continue;
    }
    if (op == MOOG_X_OVERLAPS_FIRST) {   // sprite.overlaps_sprite(state[L][0])
      const int sp = I->b ? s1 : s0;
      int first = -1;
      for (int q = P->layer_slot0[I->a]; q < P->layer_slot0[I->a] + P->layer_nslots[I->a] && first < 0; ++q)
        if (ALIVE(q)) first = q;
      const bool ov = first >= 0 && overlaps(e, sp, first);
      v(n) = ov ? 1.0 : 0.0; XSETTAG(n, 0); ++n;
      continue;
    }
    if (op == MOOG_X_STORE) {
      --n;
      st->mask |= 1u << I->a;
      st->tags = (st->tags & ~(3u << (2 * I->a))) | ((unsigned)XTAG(n) << (2 * I->a));
      sv[I->a] = v(n);
      continue;
    }
    if (op == MOOG_X_SELECT) {
      n -= 2;
      const bool c = v(n - 1) != 0;
      const int t = c ? XTAG(n) : XTAG(n + 1);
      v(n - 1) = c ? v(n) : v(n + 1);
      XSETTAG(n - 1, t);
      continue;
    }
    if (op >= MOOG_X_NEG && op <= MOOG_X_SIGN) {
      const double a = v(n - 1);
      const bool f32 = XTAG(n - 1) == 1;
      double r = a;
      switch (op) {
        case MOOG_X_NEG: r = -a; break;
        case MOOG_X_ABS: r = fabs(a); break;
        case MOOG_X_SQRT: r = f32 ? (double)sqrtf((float)a) : sqrt(a); break;
        case MOOG_X_SIN: r = f32 ? (double)(float)sin(a) : sin(a); break;
        case MOOG_X_COS: r = f32 ? (double)(float)cos(a) : cos(a); break;
        case MOOG_X_FLOOR: r = floor(a); break;
        case MOOG_X_NOT: r = (a != 0) ? 0.0 : 1.0; XSETTAG(n - 1, 0); break;
        case MOOG_X_SIGN: r = (a > 0) ? 1.0 : ((a < 0) ? -1.0 : a); break;
        default: break;
      }
      v(n - 1) = r;
      continue;
    }
    --n;
    const double a = v(n - 1), b = v(n);
    const int ta = XTAG(n - 1), tb = XTAG(n);
    const bool f32 = (ta == 1 || tb == 1) && ta != 2 && tb != 2;
    int rt = (ta == 2 || tb == 2) ? 2 : ((ta == 1 || tb == 1) ? 1 : 0);
    const float af = (float)a, bf = (float)b;
    double r = 0;
    switch (op) {
      case MOOG_X_ADD: r = f32 ? (double)(af + bf) : a + b; break;
      case MOOG_X_SUB: r = f32 ? (double)(af - bf) : a - b; break;
      case MOOG_X_MUL: r = f32 ? (double)(af * bf) : a * b; break;
      case MOOG_X_DIV: r = f32 ? (double)(af / bf) : a / b; break;
      case MOOG_X_REM: r = f32 ? (double)np_remf(af, bf) : np_rem(a, b); break;
      case MOOG_X_MIN: r = f32 ? (double)((af < bf || isnan(af)) ? af : bf) : ((a < b || isnan(a)) ? a : b); break;
      case MOOG_X_MAX: r = f32 ? (double)((af > bf || isnan(af)) ? af : bf) : ((a > b || isnan(a)) ? a : b); break;
      case MOOG_X_LT: r = f32 ? (af < bf) : (a < b); rt = 0; break;
      case MOOG_X_LE: r = f32 ? (af <= bf) : (a <= b); rt = 0; break;
      case MOOG_X_GT: r = f32 ? (af > bf) : (a > b); rt = 0; break;
      case MOOG_X_GE: r = f32 ? (af >= bf) : (a >= b); rt = 0; break;
      case MOOG_X_EQ: r = f32 ? (af == bf) : (a == b); rt = 0; break;
      case MOOG_X_NE: r = f32 ? (af != bf) : (a != b); rt = 0; break;
      case MOOG_X_AND: r = (a != 0) && (b != 0); rt = 0; break;
      case MOOG_X_OR: r = (a != 0) || (b != 0); rt = 0; break;
      default: break;
    }
    v(n - 1) = r;
    XSETTAG(n - 1, rt);
  }
  const double top = n > 0 ? v(n - 1) : 0.0;
  if (out_tag) *out_tag = n > 0 ? XTAG(n - 1) : 0;
#undef XTAG
#undef XSETTAG
#undef v
  wsync();
  return top;
}

__device__ inline double eval_expr(Env& e, int off, int s0, int s1, int* out_tag, XStores* st) {
  return eval_expr_t<false>(e, off, s0, s1, out_tag, st);
}

// the attribute writes of a modifier (sprite.py setters :540-664)
// the stores of a modifier that do not move vertices (velocities, mass, colours, opacity), by one lane for sprite s
__device__ __forceinline__ void apply_light_stores(Env& e, int s, const XStores& st, double vx, double vy, double w, double m,
                                                   double c0, double c1, double c2, double op) {
  auto tag = [&](int a) { return (int)((st.tags >> (2 * a)) & 3u); };
  auto has = [&](int a) { return ((st.mask >> a) & 1u) != 0; };
  const bool sf = EP(e)->sprite_factors != 0;
  int fl = FLAGS(s);
  if (has(MOOG_XA_XVEL) || has(MOOG_XA_YVEL)) {   // a fresh ndarray
    VELX(s) = vx; VELY(s) = vy;
    fl &= ~MOOG_F_VEL_F32;
    if (tag(MOOG_XA_XVEL) == 1 && tag(MOOG_XA_YVEL) == 1) fl |= MOOG_F_VEL_F32;
    vel_unshare(e, s);
  }
  if (has(MOOG_XA_ANGVEL)) {
    ANGV(s) = w;
    fl &= ~MOOG_F_ANGVEL_F32;
    if (tag(MOOG_XA_ANGVEL) == 1) fl |= MOOG_F_ANGVEL_F32;
  }
  FLAGS(s) = fl;
  int fm = sf ? FMASK(s) : 0;
  if (has(MOOG_XA_MASS)) { MASS(s) = m; fm = (fm & ~(1 << MOOG_FAC_MASS)) | ((tag(MOOG_XA_MASS) == 1) << MOOG_FAC_MASS); }
  if (has(MOOG_XA_C0)) { COL_SET(s, 0, c0); fm = (fm & ~(1 << MOOG_FAC_C0)) | ((tag(MOOG_XA_C0) == 1) << MOOG_FAC_C0); }
  if (has(MOOG_XA_C1)) { COL_SET(s, 1, c1); fm = (fm & ~(1 << MOOG_FAC_C1)) | ((tag(MOOG_XA_C1) == 1) << MOOG_FAC_C1); }
  if (has(MOOG_XA_C2)) { COL_SET(s, 2, c2); fm = (fm & ~(1 << MOOG_FAC_C2)) | ((tag(MOOG_XA_C2) == 1) << MOOG_FAC_C2); }
  if (sf) FMASK(s) = fm;
  if (has(MOOG_XA_OPACITY)) OPAC_SET(s, (int32_t)op);
}

__device__ inline void run_modifier(Env& e, int xmod, int s) {
  XStores st = {0u, 0u};
  eval_expr(e, xmod, s, s, nullptr, &st);
  const double* sv = reinterpret_cast<const double*>(ECAND(e)) + MOOG_X_STACK;
  auto tag = [&](int a) { return (int)((st.tags >> (2 * a)) & 3u); };
  auto has = [&](int a) { return ((st.mask >> a) & 1u) != 0; };
  const double vx = sv[MOOG_XA_XVEL], vy = sv[MOOG_XA_YVEL], w = sv[MOOG_XA_ANGVEL];
  const double m = sv[MOOG_XA_MASS], c0 = sv[MOOG_XA_C0], c1 = sv[MOOG_XA_C1], c2 = sv[MOOG_XA_C2];
  const double op = sv[MOOG_XA_OPACITY];
  const double nx = has(MOOG_XA_X) ? sv[MOOG_XA_X] : PX(s), ny = has(MOOG_XA_Y) ? sv[MOOG_XA_Y] : PY(s);
  wsync();
  if (has(MOOG_XA_X) || has(MOOG_XA_Y)) set_position(e, s, nx, ny);
  wsync();
  if constexpr (MOOG_WITH_MAZE != 0) {   // sprite.py:531-540 (in the kernels that carry every component)
    if (has(MOOG_XA_ANGLE)) {
      const double na = sv[MOOG_XA_ANGLE];
      rotate_path(e, s, na - ANG(s));
      if (e.lane == 0) ANG(s) = na;
      wsync();
    }
  }
  if (e.lane == 0) apply_light_stores(e, s, st, vx, vy, w, m, c0, c1, c2, op);
  wave_global_fence();   // colours / opacity live in HBM: later rules of this launch may read them
  wsync();
}

// ---- layers that rules append to / pop from (CreateSprites, ChangeLayer): the reference's
//      Python lists.  Live sprites stay packed at the front of the layer's slots, in list order.
__device__ inline void move_slot(Env& e, int dst, int src) {
  wsync();
  const moog_layout_t& L = EL(e);
  const int n2 = 2 * NV(src);
  double* vd = VERT(dst);
  const double* vs = VERT(src);
  for (int k = e.lane; k < n2; k += 64) vd[k] = vs[k];
  if (e.lane < 8) BB(dst, e.lane) = BB(src, e.lane);
  if (e.lane == 0) {
    for (int c = 0; c < 2; ++c) {
      EF(e)[L.o_pos + 2 * dst + c] = EF(e)[L.o_pos + 2 * src + c];
      EF(e)[L.o_vel + 2 * dst + c] = EF(e)[L.o_vel + 2 * src + c];
      EF(e)[L.o_inertia + 2 * dst + c] = EF(e)[L.o_inertia + 2 * src + c];
    }
    for (int c = 0; c < 3; ++c) COL_SET(dst, c, COL(src, c));
    ANG(dst) = ANG(src); ANGV(dst) = ANGV(src); MASS(dst) = MASS(src); MAXR(dst) = MAXR(src);
    FLAGS(dst) = FLAGS(src); NV(dst) = NV(src); OPAC_SET(dst, OPAC(src)); SHAPEID_SET(dst, SHAPEID(src));
    TELE_SET(dst, TELE(src));
    if (EP(e)->vel_alias) VALIAS(dst) = VALIAS(src);
    if (EP(e)->sprite_factors) { SCALE(dst) = SCALE(src); ASPECT(dst) = ASPECT(src); FMASK(dst) = FMASK(src); }
    FLAGS(src) = 0; NV(src) = 0;
  }
  wave_global_fence();
  wsync();
}

__device__ inline void layer_compact(Env& e, int l) {   // list.pop() of the vanished entries
  PProg P = EP(e);
  if (!P->layer_dynamic[l]) return;
  const int s0 = P->layer_slot0[l], s1 = s0 + P->layer_nslots[l];
  // nothing to do when the live sprites already are a prefix of the layer
  {
    bool gap = false;
    int live = 0;
    for (int base = s0; base < s1; base += 64) {
      const int t = base + e.lane;
      const unsigned long long m = __ballot(t < s1 && ALIVE(t));
      const int n_here = __popcll(m), span = (s1 - base < 64) ? s1 - base : 64;
      (void)span;
      if (m != ((n_here >= 64) ? ~0ull : ((1ull << n_here) - 1ull)) || (n_here > 0 && live != base - s0)) gap = true;
      live += n_here;
    }
    if (!gap) return;
  }
  // 1. vertices and boxes, sprite by sprite in list order (64 lanes per sprite): a sprite only ever moves down, into
  //    the vertex area of a slot whose own sprite has moved on or is gone
  {
    int j = s0;
    for (int s = s0; s < s1; ++s) {
      if (!ALIVE(s)) continue;
      if (s != j) {
        const int n2 = 2 * NV(s);
        double* vd = VERT(j);
        const double* vs = VERT(s);
        for (int k = e.lane; k < n2; k += 64) vd[k] = vs[k];
        if (e.lane < 8) BB(j, e.lane) = BB(s, e.lane);
        wsync();
      }
      ++j;
    }
  }
  // 2. the per-sprite fields, lanes = sprites, 64 slots at a time in list order: every lane reads its sprite (some
  //    fields live in HBM: one round trip for all of them instead of one per field per sprite), the moved-from slots
  //    are cleared, then every lane writes its sprite to its new slot
  const moog_layout_t& L = EL(e);
  const bool alias = EP(e)->vel_alias != 0, facs = EP(e)->sprite_factors != 0;
  int j0 = s0;
  for (int base = s0; base < s1; base += 64) {
    const int t = base + e.lane;
    const bool live = t < s1 && ALIVE(t);
    const unsigned long long m = __ballot(live);
    const int dst = j0 + __popcll(m & ((1ull << e.lane) - 1ull));
    const bool mv = live && dst != t;
    double px = 0, py = 0, vx = 0, vy = 0, ix = 0, iy = 0, c0 = 0, c1 = 0, c2 = 0, an = 0, av = 0, ms = 0, mr = 0, sc = 0, as = 0;
    int fl = 0, nv = 0, op = 0, sh = 0, te = 0, va = 0, fm = 0;
    if (mv) {
      px = EF(e)[L.o_pos + 2 * t]; py = EF(e)[L.o_pos + 2 * t + 1];
      vx = EF(e)[L.o_vel + 2 * t]; vy = EF(e)[L.o_vel + 2 * t + 1];
      ix = EF(e)[L.o_inertia + 2 * t]; iy = EF(e)[L.o_inertia + 2 * t + 1];
      c0 = COL(t, 0); c1 = COL(t, 1); c2 = COL(t, 2);
      an = ANG(t); av = ANGV(t); ms = MASS(t); mr = MAXR(t);
      fl = FLAGS(t); nv = NV(t); op = OPAC(t); sh = SHAPEID(t); te = TELE(t);
      if (alias) va = VALIAS(t);
      if (facs) { sc = SCALE(t); as = ASPECT(t); fm = FMASK(t); }
    }
    wave_global_fence();
    wsync();
    if (mv) { FLAGS(t) = 0; NV(t) = 0; }
    wsync();
    if (mv) {
      EF(e)[L.o_pos + 2 * dst] = px; EF(e)[L.o_pos + 2 * dst + 1] = py;
      EF(e)[L.o_vel + 2 * dst] = vx; EF(e)[L.o_vel + 2 * dst + 1] = vy;
      EF(e)[L.o_inertia + 2 * dst] = ix; EF(e)[L.o_inertia + 2 * dst + 1] = iy;
      COL_SET(dst, 0, c0); COL_SET(dst, 1, c1); COL_SET(dst, 2, c2);
      ANG(dst) = an; ANGV(dst) = av; MASS(dst) = ms; MAXR(dst) = mr;
      FLAGS(dst) = fl; NV(dst) = nv; OPAC_SET(dst, op); SHAPEID_SET(dst, sh); TELE_SET(dst, te);
      if (alias) VALIAS(dst) = va;
      if (facs) { SCALE(dst) = sc; ASPECT(dst) = as; FMASK(dst) = fm; }
    }
    wave_global_fence();
    wsync();
    j0 += __popcll(m);
  }
}

__device__ inline int layer_append_slot(Env& e, int l) {   // list.append(): the slot after the last entry
  PProg P = EP(e);
  int n = 0;
  for (int s = P->layer_slot0[l]; s < P->layer_slot0[l] + P->layer_nslots[l]; ++s) n += ALIVE(s) ? 1 : 0;
  if (e.layer_hw && e.lane == 0)   // sizing hint for layer_capacity (the reference's lists are unbounded)
    __hip_atomic_fetch_max(&e.layer_hw[l], n + 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
  if (n >= P->layer_nslots[l]) {
    wsync();
    if (e.lane == 0) {
      EQ(e)[EL(e).o_fault] |= MOOG_FAULT_LAYER_FULL;
      if (e.layer_hw) __hip_atomic_fetch_add(&e.layer_hw[MOOG_MAX_LAYERS + l], 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    }
    wsync();
    return -1;
  }
  return P->layer_slot0[l] + n;
}

__device__ inline bool sprite_filter_x(Env& e, int kind, int xoff, int s) {
  if (kind == MOOG_FILTER_ALWAYS) return true;
  return eval_expr(e, xoff, s, s, nullptr, nullptr) != 0;
}
__device__ inline bool sprite_filter(Env& e, PRule R, int s) {
  return sprite_filter_x(e, R->filter, R->xfilter, s);
}

template <bool X>
__device__ inline void create_sprite(Env& e, int s, const double* fac, int vel_f32, int angvel_f32);
template <bool X>
__device__ inline void sample_op_factors(Env& e, PGenop op, double* fac, int& vel_f32, int& angvel_f32);
__device__ inline int genop_count(Env& e, PGenop op);

// DYN: the program has rules that create / move sprites at run time; the plain step kernel is
// compiled without them so that the common configs do not carry the sampler.
template <bool DYN>
__device__ inline void rule_leaf_step(Env& e, int ri) {
  PProg P = EP(e);
  PRule R = &P->rules[ri];
  if constexpr (DYN) {
    if (R->kind == MOOG_RULE_DRAWS) {   // the np.random calls at the top of a config-local rule's step
      const double u0 = next_uniform(e);
      const double u1 = R->i0 > 1 ? next_uniform(e) : 0.0;
      wsync();
      if (e.lane == 0) { EF(e)[EL(e).o_rule + ri] = u0; EF(e)[EL(e).o_rule2 + ri] = u1; }
      wsync();
      return;
    }
    if (R->kind == MOOG_RULE_VANISH_BY_FILTER) {   // vanish.py:31-39,58-61
      const int a0 = P->layer_slot0[R->l0], a1 = a0 + P->layer_nslots[R->l0];
      if (R->filter == MOOG_FILTER_EXPR_LANES && e.xstack) {   // lanes = the layer's sprites
        for (int base = a0; base < a1; base += 64) {
          const int t = base + e.lane;
          const bool live = t < a1 && ALIVE(t);
          const bool hit = eval_expr_t<true>(e, R->xfilter, live ? t : a0, live ? t : a0, nullptr, nullptr) != 0;
          if (live && hit) FLAGS(t) &= ~MOOG_F_ALIVE;
          wsync();
        }
      } else
      for (int s = a0; s < a1; ++s) {
        if (!ALIVE(s) || !sprite_filter(e, R, s)) continue;
        wsync();
        if (e.lane == 0) FLAGS(s) &= ~MOOG_F_ALIVE;
        wsync();
      }
      layer_compact(e, R->l0);
      return;
    }
    if (R->kind == MOOG_RULE_CHANGE_LAYER) {       // change_layer.py:36-46
      const int a0 = P->layer_slot0[R->l0], a1 = a0 + P->layer_nslots[R->l0];
      if (R->filter == MOOG_FILTER_EXPR_LANES && e.xstack) {
        for (int base = a0; base < a1; base += 64) {
          const int t = base + e.lane;
          const bool live = t < a1 && ALIVE(t);
          const bool hit = eval_expr_t<true>(e, R->xfilter, live ? t : a0, live ? t : a0, nullptr, nullptr) != 0;
          if (live && hit) FLAGS(t) |= MOOG_F_TMP;
          wsync();
        }
      } else
      for (int s = a0; s < a1; ++s) {
        if (!ALIVE(s) || !sprite_filter(e, R, s)) continue;
        wsync();
        if (e.lane == 0) FLAGS(s) |= MOOG_F_TMP;
        wsync();
      }
      for (int s = a0; s < a1; ++s) {
        if (!(FLAGS(s) & MOOG_F_TMP)) continue;
        int dst = layer_append_slot(e, R->l1);
        wsync();
        if (e.lane == 0) FLAGS(s) &= ~MOOG_F_TMP;
        wsync();
        if (dst < 0) continue;
        move_slot(e, dst, s);
      }
      layer_compact(e, R->l0);
      return;
    }
    if (MOOG_WITH_MAZE != 0 && R->kind == MOOG_RULE_MODIFY_SPRITES && (R->i0 & 16)) {   // (in the kernels that carry every component)
      // every live sprite of the layers gets the same constants (mass, colours, opacity, velocities): one
      // evaluation, then lanes = sprites (pacman's `unglue` rule touches ~50 sprites every step)
      int first = -1;
      for (int a = 0; a < R->n_layers && first < 0; ++a) {
        const int l = R->layers[a];
        for (int s = P->layer_slot0[l]; s < P->layer_slot0[l] + P->layer_nslots[l] && first < 0; ++s)
          if (ALIVE(s)) first = s;
      }
      if (first < 0) return;
      XStores st = {0u, 0u};
      eval_expr(e, R->xmod, first, first, nullptr, &st);
      const double* sv = reinterpret_cast<const double*>(ECAND(e)) + MOOG_X_STACK;
      const double vx = sv[MOOG_XA_XVEL], vy = sv[MOOG_XA_YVEL], w = sv[MOOG_XA_ANGVEL], m = sv[MOOG_XA_MASS];
      const double c0 = sv[MOOG_XA_C0], c1 = sv[MOOG_XA_C1], c2 = sv[MOOG_XA_C2], op = sv[MOOG_XA_OPACITY];
      wsync();
      for (int a = 0; a < R->n_layers; ++a) {
        const int l = R->layers[a];
        for (int s = P->layer_slot0[l] + e.lane; s < P->layer_slot0[l] + P->layer_nslots[l]; s += 64)
          if (ALIVE(s)) apply_light_stores(e, s, st, vx, vy, w, m, c0, c1, c2, op);
      }
      wave_global_fence();
      wsync();
      return;
    }
    if (R->kind == MOOG_RULE_MODIFY_SPRITES) {     // modify_sprites.py:35-52
      // the sprites passing the filter are marked first (filters see the unmodified state)
      int n = 0;
      for (int a = 0; a < R->n_layers; ++a) {
        const int l = R->layers[a];
        if (R->filter == MOOG_FILTER_EXPR_LANES && e.xstack) {
          const int l0 = P->layer_slot0[l], l1 = l0 + P->layer_nslots[l];
          for (int base = l0; base < l1; base += 64) {
            const int t = base + e.lane;
            const bool live = t < l1 && ALIVE(t);
            const bool hit = eval_expr_t<true>(e, R->xfilter, live ? t : l0, live ? t : l0, nullptr, nullptr) != 0;
            if (live && hit) FLAGS(t) |= MOOG_F_TMP;
            wsync();
            n += __popcll(__ballot(live && hit));
          }
          continue;
        }
        for (int s = P->layer_slot0[l]; s < P->layer_slot0[l] + P->layer_nslots[l]; ++s) {
          if (!ALIVE(s) || !sprite_filter(e, R, s)) continue;
          wsync();
          if (e.lane == 0) FLAGS(s) |= MOOG_F_TMP;
          wsync();
          ++n;
        }
      }
      if (n == 0) return;
      int pick = (R->i0 & 8) ? 0 : -1;   // bit 3: a config-local rule on `state[L][0]`
      if (R->i0 & 1) {   // sample_one: np.random.choice(sprites_to_modify)
        pick = 0;
        if (n > 1) { pick = (int)(next_uniform(e) * n); if (pick >= n) pick = n - 1; }
      }
      int k = 0;
      for (int a = 0; a < R->n_layers; ++a) {
        const int l = R->layers[a];
        for (int s = P->layer_slot0[l]; s < P->layer_slot0[l] + P->layer_nslots[l]; ++s) {
          if (!(FLAGS(s) & MOOG_F_TMP)) continue;
          wsync();
          if (e.lane == 0) FLAGS(s) &= ~MOOG_F_TMP;
          wsync();
          if (pick < 0 || k == pick) run_modifier(e, R->xmod, s);
          ++k;
        }
      }
      return;
    }
    if (R->kind == MOOG_RULE_MODIFY_ON_CONTACT) {  // contact_rules.py:112-141
      for (int side = 0; side < 2; ++side) {
        const int xmod = side ? R->xmod1 : R->xmod;
        if (xmod < 0) continue;
        const int na = side ? R->n_layers1 : R->n_layers, nb = side ? R->n_layers : R->n_layers1;
        for (int a = 0; a < na; ++a) {
          const int la = side ? R->layers1[a] : R->layers[a];
          for (int s = P->layer_slot0[la]; s < P->layer_slot0[la] + P->layer_nslots[la]; ++s) {
            if (!ALIVE(s)) continue;
            if (!sprite_filter_x(e, side ? R->filter1 : R->filter, side ? R->xfilter1 : R->xfilter, s)) continue;
            bool any = false;
            for (int b = 0; b < nb && !any; ++b) {
              const int lb = side ? R->layers[b] : R->layers1[b];
              any = overlaps_any(e, s, P->layer_slot0[lb], P->layer_slot0[lb] + P->layer_nslots[lb]);   // (never s itself)
            }
            if (any) run_modifier(e, xmod, s);
          }
        }
      }
      return;
    }
    if (R->kind == MOOG_RULE_CREATE_SPRITES) {     // create_sprites.py:31-37 + sprite_generators.py:77-105
      PGenop op = &P->ops[R->op];
      const int n = genop_count(e, op);
      int first = -1;
      bool gave_up = false;
      for (int k = 0; k < n; ++k) {
        const int s = layer_append_slot(e, R->l0);
        if (s < 0) break;
        if (first < 0) first = s;
        int count = 0;
        for (;;) {
          double fac[MOOG_NUM_FACTORS];
          int vel_f32, angvel_f32;
          e.cur_slot = s;
          sample_op_factors<MOOG_WITH_MAZE != 0>(e, op, fac, vel_f32, angvel_f32);
          create_sprite<MOOG_WITH_MAZE != 0>(e, s, fac, vel_f32, angvel_f32);
          bool ov = false;
          for (int a = 0; a < R->n_layers && !ov; ++a) {
            int l = R->layers[a];
            // (the sprites this call has made so far join the layer only afterwards, create_sprites.py:34-37)
            ov = overlaps_any(e, s, P->layer_slot0[l], l == R->l0 ? first : P->layer_slot0[l] + P->layer_nslots[l]);
          }
          if (op->disjoint && !ov) ov = overlaps_any(e, s, first, s);
          if (!ov) break;
          if (count > op->max_tries) {
            wsync();
            if (e.lane == 0) {
              if (op->fail_gracefully) { FLAGS(s) = 0; NV(s) = 0; }   // the generator returns what it has (:93-95)
              else EQ(e)[EL(e).o_fault] |= MOOG_FAULT_SAMPLER_EXHAUSTED;
            }
            wsync();
            gave_up = op->fail_gracefully != 0;
            break;
          }
          ++count;
        }
        if (gave_up) break;
        wsync();
        if (e.lane == 0) FLAGS(s) |= MOOG_F_ALIVE;
        wsync();
      }
      wave_global_fence();   // the new sprites' colours / opacity / shape ids are in HBM
      return;
    }
  }
  switch (R->kind) {
    case MOOG_RULE_VANISH_ON_CONTACT: {
      int a0 = P->layer_slot0[R->l0], a1 = a0 + P->layer_nslots[R->l0];
      int b0 = P->layer_slot0[R->l1], b1 = b0 + P->layer_nslots[R->l1];
      for (int s = a0; s < a1; ++s) {
        if (!ALIVE(s)) continue;
        bool kill = false;
        for_each_near(e, s, b0, b1, [&](int t) { if (!kill && overlaps(e, s, t, true)) kill = true; });
        if (kill) {
          wsync();
          if (e.lane == 0) FLAGS(s) |= MOOG_F_TMP;
          wsync();
        }
      }
      wsync();
      for (int s = a0 + e.lane; s < a1; s += 64)
        if (FLAGS(s) & MOOG_F_TMP) FLAGS(s) &= ~(MOOG_F_TMP | MOOG_F_ALIVE);
      wsync();
      if constexpr (DYN) layer_compact(e, R->l0);
      break;
    }
    case MOOG_RULE_FIXATION: {   // fixation.py:48-58
      int a = -1, t = -1;
      for (int s = P->layer_slot0[R->l0]; s < P->layer_slot0[R->l0] + P->layer_nslots[R->l0] && a < 0; ++s)
        if (ALIVE(s)) a = s;
      for (int s = P->layer_slot0[R->l1]; s < P->layer_slot0[R->l1] + P->layer_nslots[R->l1] && t < 0; ++s)
        if (ALIVE(s)) t = s;
      if (a < 0 || t < 0) break;   // (state[layer][0] of an empty layer raises IndexError in the reference)
      const double dx = PX(a) - PX(t), dy = PY(a) - PY(t);
      const double dist = npnorm(dx, dy);   // np.linalg.norm (1-D)
      const double cnt = EF(e)[EL(e).o_rule + ri];
      wsync();
      if (e.lane == 0) EF(e)[EL(e).o_rule + ri] = dist < R->p0 ? cnt + 1 : 0;
      wsync();
      break;
    }
    case MOOG_RULE_KEEP_NEAR_CENTER: {   // re_center.py:48-58
      int agent = -1;
      for (int s = P->layer_slot0[R->l0]; s < P->layer_slot0[R->l0] + P->layer_nslots[R->l0] && agent < 0; ++s)
        if (ALIVE(s)) agent = s;
      if (agent < 0) break;
      const double ax = PX(agent) - 0.5, ay = PY(agent) - 0.5;
      const double dx = -1. * R->p0 * (double)(ax > R->p0) + R->p0 * (double)(ax < -1. * R->p0);
      const double dy = -1. * R->p1 * (double)(ay > R->p1) + R->p1 * (double)(ay < -1. * R->p1);
      if (dx != 0 || dy != 0)
        for (int a = 0; a < R->n_layers; ++a) {
          const int l = R->layers[a];
          for (int s = P->layer_slot0[l]; s < P->layer_slot0[l] + P->layer_nslots[l]; ++s)
            if (ALIVE(s)) set_position(e, s, PX(s) + dx, PY(s) + dy);
        }
      break;
    }
    case MOOG_RULE_TORUS_WRAP: {
      for (int a = 0; a < R->n_layers; ++a) {
        int l = R->layers[a];
        int s0 = P->layer_slot0[l], s1 = s0 + P->layer_nslots[l];
        for (int s = s0; s < s1; ++s)
          if (ALIVE(s)) set_position(e, s, np_remainder1(PX(s)), np_remainder1(PY(s)));
      }
      break;
    }
    case MOOG_RULE_PORTAL: {
      int p0 = P->layer_slot0[R->l1], p1 = p0 + P->layer_nslots[R->l1];
      int np_ = 0;
      for (int s = p0; s < p1; ++s) if (ALIVE(s)) ++np_;
      if (np_ % 2 != 0) {
        wsync();
        if (e.lane == 0) EQ(e)[EL(e).o_fault] |= MOOG_FAULT_ODD_PORTALS;
        wsync();
        break;
      }
      int t0 = P->layer_slot0[R->l0], t1 = t0 + P->layer_nslots[R->l0];
      for (int s = t0; s < t1; ++s) {
        if (!ALIVE(s)) continue;
        int entry = -1, k = 0, entry_slot = -1;
        for (int p = p0; p < p1; ++p) {
          if (!ALIVE(p)) continue;
          if (entry < 0 && contains_point(e, p, PX(s), PY(s))) { entry = k; entry_slot = p; }
          ++k;
        }
        (void)entry_slot;
        int tele = TELE(s);
        if (entry < 0) {
          wsync();
          if (e.lane == 0) TELE_SET(s, tele & ~(1 << ri));
          wave_global_fence();   // the Portal bits live in HBM
          wsync();
          continue;
        }
        if (tele & (1 << ri)) continue;
        int ex = (entry % 2) ? entry - 1 : entry + 1;
        int exs = -1;
        k = 0;
        for (int p = p0; p < p1; ++p) {
          if (!ALIVE(p)) continue;
          if (k == ex) exs = p;
          ++k;
        }
        set_position(e, s, PX(exs), PY(exs));
        if (e.lane == 0) TELE_SET(s, tele | (1 << ri));
        wave_global_fence();
        wsync();
      }
      break;
    }
    case MOOG_RULE_BOOSTER: {
      double cnt = EF(e)[EL(e).o_rule + ri] - 1;
      int a0 = P->layer_slot0[R->l0], a1 = a0 + P->layer_nslots[R->l0];
      int agent = -1;
      for (int s = a0; s < a1 && agent < 0; ++s) if (ALIVE(s)) agent = s;
      if (agent >= 0) {
        if (cnt == DINF) {
          int b0 = P->layer_slot0[R->l1], b1 = b0 + P->layer_nslots[R->l1];
          bool any = false;
          for_each_near(e, agent, b0, b1, [&](int t) { if (!any && overlaps(e, agent, t, true)) any = true; });
          if (any) {
            double m = MASS(agent) * R->p0;
            double c2 = 1. - (1. - COL(agent, 2)) * R->p1;
            wsync();
            if (e.lane == 0) { MASS(agent) = m; COL_SET(agent, 2, c2); }
          wave_global_fence();
            wave_global_fence();
            cnt = R->p2;
          }
        } else if (cnt <= 0) {
          double m = MASS(agent) / R->p0;
          double c2 = 1. - (1. - COL(agent, 2)) / R->p1;
          wsync();
          if (e.lane == 0) { MASS(agent) = m; COL_SET(agent, 2, c2); }
          wave_global_fence();
          cnt = DINF;
        }
      }
      wsync();
      if (e.lane == 0) EF(e)[EL(e).o_rule + ri] = cnt;
      wsync();
      break;
    }
    default: break;
  }
}

// rule.reset() of one table entry, without the random duration of a PHASE (rule_reset_tree draws it, in the reference's order)
__device__ inline void rule_reset(Env& e, int ri) {
  PRule R = &EP(e)->rules[ri];
  wsync();
  if (R->kind == MOOG_RULE_PORTAL)
    for (int s = e.lane; s < EP(e)->n_slots; s += 64) TELE_SET(s, TELE(s) & ~(1 << ri));
  wave_global_fence();
  if (e.lane == 0 && R->kind != MOOG_RULE_STATE_SLOT)   // (a state slot lives as long as the environment)
    EF(e)[EL(e).o_rule + ri] = (R->kind == MOOG_RULE_TIMED) ? R->p0 :   // timing.py:47
        ((R->kind == MOOG_RULE_PHASE || R->kind == MOOG_RULE_PHASE_SEQUENCE || R->kind == MOOG_RULE_FIXATION) ? 0.0 : DINF);
  wsync();
  if (R->kind == MOOG_RULE_TIMED && R->op != 0) {   // a callable interval: drawn here, before the children are reset (timing.py:47)
    const int lo = (int)(R->op == 2 ? R->p1 : R->p0), hi = (int)R->p2;
    int k = (int)(next_uniform(e) * (hi - lo));   // np.random.randint(lo, hi)
    if (k >= hi - lo) k = hi - lo - 1;
    double width = (R->op == 1) ? R->p1 : (double)(lo + k) - R->p0;
    if (R->op == 3) {   // DelayedRule(start = <callable>, duration = <callable>), timing.py:84-86: the start above, then the duration
      const int lo2 = (int)R->p1, hi2 = R->i0;
      int k2 = (int)(next_uniform(e) * (hi2 - lo2));
      if (k2 >= hi2 - lo2) k2 = hi2 - lo2 - 1;
      width = (double)(lo2 + k2);
    }
    wsync();
    if (e.lane == 0) {
      EF(e)[EL(e).o_rule + ri] = (R->op == 2) ? R->p0 : (double)(lo + k);
      EF(e)[EL(e).o_rule2 + ri] = width;
    }
    wsync();
  }
}

// task_phases.py:69-75 Phase.reset: the one-time and continual rules are reset first, `self._current_duration =
// self._duration()` comes last -- so a Phase with a random duration draws AFTER everything in its subtree has (a nested
// random-duration Phase draws before the one that contains it).
__device__ inline void rule_draw_duration(Env& e, int ri) {
  PRule R = &EP(e)->rules[ri];
  if (!(R->kind == MOOG_RULE_PHASE && R->op == 1)) return;
  const int lo = (int)R->p0, hi = (int)R->p2;
  int k = (int)(next_uniform(e) * (hi - lo));   // np.random.randint(lo, hi)
  if (k >= hi - lo) k = hi - lo - 1;
  wsync();
  if (e.lane == 0) EF(e)[EL(e).o_rule2 + ri] = (double)(lo + k);
  wsync();
}

// TimedRule.reset / ConditionalRule.reset / Phase.reset (timing.py:46-49, conditional.py:56-58, task_phases.py:69-75): the
// whole subtree of a top-level rule is the contiguous run of entries that follows it (pre-order).  Scalars are reset in
// that order; a subtree's duration is drawn when its last entry has been reset (post-order), innermost first.
__device__ inline void rule_reset_tree(Env& e, int ri) {
  PProg P = EP(e);
  int end = ri + 1;
  while (end < P->n_rules && P->rules[end].parent >= ri) ++end;
  for (int c = ri; c < end; ++c) {
    rule_reset(e, c);
    const int next_parent = (c + 1 < end) ? P->rules[c + 1].parent : ri - 1;   // subtrees that do not contain entry c + 1 end here
    for (int a = c; a >= ri && a > next_parent; a = P->rules[a].parent) {
      rule_draw_duration(e, a);
      if (P->rules[a].parent < ri) break;
    }
  }
}

// all(pred(s) ...) / any(pred(s) ...) over a layer, or expr(layer[0]) (conditions traced by
// moog/_symbolic.py trace_state_condition; MOOG_COND_* and MOOG_RCOND_* share the numbering)
__device__ inline double layer_condition(Env& e, int kind, int layer, int xoff) {
  PProg P = EP(e);
  if (kind == MOOG_COND_STATE_EXPR) return eval_expr(e, xoff, 0, 0, nullptr, nullptr);   // (no sprite of its own: s0 / s1 unused)
  const int a0 = P->layer_slot0[layer], a1 = a0 + P->layer_nslots[layer];
  if (kind == MOOG_COND_FIRST_EXPR) {
    for (int s = a0; s < a1; ++s)
      if (ALIVE(s)) return eval_expr(e, xoff, s, s, nullptr, nullptr);
    return 0;
  }
  bool all = true, any = false;
  for (int s = a0; s < a1; ++s) {
    if (!ALIVE(s)) continue;
    const bool v = eval_expr(e, xoff, s, s, nullptr, nullptr) != 0;
    all = all && v; any = any || v;
  }
  return (kind == MOOG_COND_ALL_EXPR) ? (all ? 1 : 0) : (any ? 1 : 0);
}

// value of a rule's state condition (ConditionalRule repeat count, Phase end condition)
template <bool DYN>
__device__ inline int rule_condition(Env& e, PRule R, double p_bernoulli) {
  if (R->cond == MOOG_RCOND_BERNOULLI) return next_uniform(e) < p_bernoulli ? 1 : 0;
  if constexpr (!DYN) return 0;   // the other conditions make the host pick the DYN kernel
  if (R->cond == MOOG_RCOND_CONTACT_COUNT) {   // contact_rules.py:28-56
    PProg P = EP(e);
    int n = 0;
    for (int s = P->layer_slot0[R->l0]; s < P->layer_slot0[R->l0] + P->layer_nslots[R->l0]; ++s)
      if (ALIVE(s))
        for_each_near(e, s, P->layer_slot0[R->l1], P->layer_slot0[R->l1] + P->layer_nslots[R->l1],
                      [&](int t) { if (overlaps(e, s, t, true)) ++n; });
    return n;
  }
  if ((R->cond >= MOOG_RCOND_ALL_EXPR && R->cond <= MOOG_RCOND_FIRST_EXPR) || R->cond == MOOG_RCOND_STATE_EXPR)
    return (int)layer_condition(e, R->cond, R->l0, R->xfilter);
  if (R->cond == MOOG_RCOND_COUNT_EXPR) {   // accumulated per-sprite terms
    PProg P = EP(e);
    double acc = 0;
    for (int s = P->layer_slot0[R->l0]; s < P->layer_slot0[R->l0] + P->layer_nslots[R->l0]; ++s)
      if (ALIVE(s)) acc += eval_expr(e, R->xfilter, s, s, nullptr, nullptr);
    return (int)acc;
  }
  return 0;
}

__device__ __forceinline__ bool is_combinator(int k) {
  return k == MOOG_RULE_TIMED || k == MOOG_RULE_CONDITIONAL || k == MOOG_RULE_PHASE ||
         k == MOOG_RULE_PHASE_SEQUENCE;
}

// What a combinator does with its children in this call (timing.py:51-59, conditional.py:60-63,
// task_phases.py:68-82,129-141): run them n times; a Phase runs its first i0 children (the
// one-time rules) only on its first step; a PhaseSequence runs only child number `only`.
struct RuleGate { int n, only, i0; bool first; };

__device__ inline int n_children(Env& e, int ri) {
  PProg P = EP(e);
  int n = 0;
  for (int c = ri + 1; c < P->n_rules && P->rules[c].parent >= ri; ++c) n += (P->rules[c].parent == ri) ? 1 : 0;
  return n;
}

template <bool DYN>
__device__ inline RuleGate rule_open(Env& e, int ri) {
  PRule R = &EP(e)->rules[ri];
  RuleGate g = {0, -1, 0, true};
  const double st = EF(e)[EL(e).o_rule + ri];
  if (R->kind == MOOG_RULE_TIMED) {
    const double width = R->op != 0 ? EF(e)[EL(e).o_rule2 + ri] : (R->p1 - R->p0);
    g.n = (st <= 0 && st + width > 0) ? 1 : 0;
    // the countdown happens after the children in the reference; they never read it
    wsync();
    if (e.lane == 0) EF(e)[EL(e).o_rule + ri] = st - 1;
    wsync();
  } else if (R->kind == MOOG_RULE_CONDITIONAL) {
    g.n = rule_condition<DYN>(e, R, R->p0);
  } else if (R->kind == MOOG_RULE_PHASE) {
    g.n = (st < 0) ? 0 : 1;   // should_end
    g.i0 = R->i0;
    g.first = (st == 0);
  } else {   // PHASE_SEQUENCE
    g.only = (int)st;
    g.n = 1;
    if (g.only >= n_children(e, ri)) {
      g.n = 0;
      wsync();
      if (e.lane == 0) EQ(e)[EL(e).o_fault] |= MOOG_FAULT_PHASE_END;
      wsync();
    }
  }
  return g;
}

template <bool DYN>
__device__ inline void rule_close(Env& e, int ri, const RuleGate& g) {
  PProg P = EP(e);
  PRule R = &P->rules[ri];
  if (g.n == 0) return;
  if (R->kind == MOOG_RULE_PHASE) {
    double st = EF(e)[EL(e).o_rule + ri] + 1;
    const double duration = (R->op == 1) ? EF(e)[EL(e).o_rule2 + ri] : R->p0;
    if (st >= duration || (R->cond && rule_condition<DYN>(e, R, R->p1) != 0)) st = -1;
    wsync();
    if (e.lane == 0) EF(e)[EL(e).o_rule + ri] = st;
    wsync();
  } else if (R->kind == MOOG_RULE_PHASE_SEQUENCE) {
    int k = 0, cur = -1;
    for (int c = ri + 1; c < P->n_rules && P->rules[c].parent >= ri; ++c) {
      if (P->rules[c].parent != ri) continue;
      if (k == g.only) cur = c;
      ++k;
    }
    if (cur >= 0 && EF(e)[EL(e).o_rule + cur] < 0) {   // the current phase ended: move on
      wsync();
      if (e.lane == 0) {
        EF(e)[EL(e).o_rule + ri] = (double)(g.only + 1);
        if (g.only + 1 >= k) EQ(e)[EL(e).o_fault] |= MOOG_FAULT_PHASE_END;   // self._phases[ind]: IndexError
      }
      wsync();
    }
  }
}

__device__ __forceinline__ bool child_selected(const RuleGate& g, int parent_kind, int k) {
  if (g.only >= 0) return k == g.only;
  if (parent_kind == MOOG_RULE_PHASE) return k >= g.i0 || g.first;
  return true;
}

// rule.step() of a top-level rule.  Nesting is at most two combinators deep, so the walk is
// three explicit levels (no device recursion).
template <bool DYN>
__device__ inline void rule_step(Env& e, int ri) {
  PProg P = EP(e);
  const int k0 = P->rules[ri].kind;
  if (!is_combinator(k0)) { rule_leaf_step<DYN>(e, ri); return; }
  const int nr = P->n_rules;
  const RuleGate g0 = rule_open<DYN>(e, ri);
  for (int i0 = 0; i0 < g0.n; ++i0) {
    int kc = 0;
    for (int c = ri + 1; c < nr && P->rules[c].parent >= ri; ++c) {
      if (P->rules[c].parent != ri) continue;
      const bool sel = child_selected(g0, k0, kc);
      ++kc;
      if (!sel) continue;
      const int k1 = P->rules[c].kind;
      if (!is_combinator(k1)) { rule_leaf_step<DYN>(e, c); continue; }
      const RuleGate g1 = rule_open<DYN>(e, c);
      for (int i1 = 0; i1 < g1.n; ++i1) {
        int kd = 0;
        for (int d = c + 1; d < nr && P->rules[d].parent >= c; ++d) {
          if (P->rules[d].parent != c) continue;
          const bool sel1 = child_selected(g1, k1, kd);
          ++kd;
          if (sel1) rule_leaf_step<DYN>(e, d);
        }
      }
      rule_close<DYN>(e, c, g1);
    }
  }
  rule_close<DYN>(e, ri, g0);
}

// ---- tasks ---------------------------------------------------------------------------------
__device__ inline bool task_condition(const Env& e, PTask T) {
  PProg P = EP(e);
  int l = T->cond_layer;
  int a0 = P->layer_slot0[l], a1 = a0 + P->layer_nslots[l];
  if (T->cond == MOOG_COND_LAYER_EMPTY) {
    for (int s = a0; s < a1; ++s) if (ALIVE(s)) return false;
    return true;
  }
  if (T->cond == MOOG_COND_ALL_Y_LT) {
    for (int s = a0; s < a1; ++s) if (ALIVE(s) && !(PY(s) < T->cond_value)) return false;
    return true;
  }
  return false;
}
// conditions that run the expression evaluator (DYN kernels only)
__device__ inline bool task_condition_x(Env& e, PTask T) {
  if ((T->cond >= MOOG_COND_ALL_EXPR && T->cond <= MOOG_COND_FIRST_EXPR) || T->cond == MOOG_COND_STATE_EXPR)
    return layer_condition(e, T->cond, T->cond_layer, T->xcond) != 0;
  return task_condition(e, T);
}

template <bool DYN>
__device__ inline double task_reward(Env& e, int step_count, int* should_reset) {
  PProg P = EP(e);
  double reward = 0;
  int reward_t = 0;   // numpy dtype tag of the running sum (composite_task.py:36-40)
  int sr = ((double)step_count >= P->timeout_steps);
  for (int ti = 0; ti < P->n_tasks; ++ti) {
    PTask T = &P->tasks[ti];
    double cnt = EF(e)[EL(e).o_task + ti];
    double r = 0;
    int tsr = 0, rt = 0;
    if (T->kind == MOOG_TASK_CONTACT_REWARD) {
      for (int a = 0; a < T->n0; ++a) {
        int la = T->layers0[a];
        int a0 = P->layer_slot0[la], a1 = a0 + P->layer_nslots[la];
        for (int s0 = a0; s0 < a1; ++s0) {
          if (!ALIVE(s0)) continue;
          for (int b = 0; b < T->n1; ++b) {
            int lb = T->layers1[b];
            int b0 = P->layer_slot0[lb], b1 = b0 + P->layer_nslots[lb];
            // (the pair condition is a pure expression: asking it only of the pairs whose boxes touch changes nothing)
            for_each_near(e, s0, b0, b1, [&](int s1) {
              if constexpr (DYN) {
                if (T->xcond >= 0 && eval_expr(e, T->xcond, s0, s1, nullptr, nullptr) == 0) return;
              }
              if (overlaps(e, s0, s1, true)) {
                r = T->p0; rt = 0;
                if constexpr (DYN) {
                  if (T->xreward >= 0) r = eval_expr(e, T->xreward, s0, s1, &rt, nullptr);
                }
                if (cnt == DINF) cnt = T->p1;
              }
            });
          }
        }
      }
      cnt -= 1;
      tsr = (cnt < 0);
    } else if (T->kind == MOOG_TASK_RESET) {
      bool hit = false;
      if (cnt == DINF) {
        if constexpr (DYN) hit = task_condition_x(e, T); else hit = task_condition(e, T);
      }
      if (hit) {   // reset.py:55-58: reward_fn(state) when the condition first holds
        r = T->p0;
        if constexpr (DYN) { if (T->xreward >= 0) { int t2 = 0; r = eval_expr(e, T->xreward, 0, 0, &t2, nullptr); } }
        cnt = T->p1;
      }
      else r = 0.;
      cnt -= 1;
      tsr = (cnt < 0);
    } else if (T->kind == MOOG_TASK_STAY_ALIVE) {
      r = ((step_count + 1) % T->i0 == 0) ? T->p0 : 0;
    }
    wsync();
    if (e.lane == 0) EF(e)[EL(e).o_task + ti] = cnt;
    wsync();
    if (reward_t == 2 || rt == 2) { reward = reward + r; reward_t = 2; }
    else if (reward_t == 1 || rt == 1) { reward = (double)((float)reward + (float)r); reward_t = 1; }
    else reward = reward + r;
    sr = sr || tsr;
  }
  *should_reset = sr;
  return reward;
}

// ---- action spaces ---------------------------------------------------------------------------
// f32: the caller's actions are float32 (joystick.py:42-43): `scaling_factor * action` is then a float32 product
__device__ inline void action_step(Env& e, int k, double ax_in, double ay_in, int grid_action, bool f32 = false) {
  PProg P = EP(e);
  PAction A = (k == 0) ? &P->action : &P->more_actions[k - 1];
  const int om = EL(e).o_action + 2 * k;
  if (A->kind == MOOG_ACTION_SET_POSITION) {   // set_position.py:48-58 (`momentum` = inertia)
    for (int a = 0; a < A->n_layers; ++a) {
      const int l = A->layers[a];
      for (int s = P->layer_slot0[l]; s < P->layer_slot0[l] + P->layer_nslots[l]; ++s)
        if (ALIVE(s))
          set_position(e, s, A->momentum * PX(s) + (1 - A->momentum) * ax_in,
                       A->momentum * PY(s) + (1 - A->momentum) * ay_in);
    }
    return;
  }
  double m0 = EF(e)[om], m1 = EF(e)[om + 1];
  if (A->kind == MOOG_ACTION_JOYSTICK) {
    double ax = ax_in, ay = A->constrained_lr ? 0. : ay_in;
    m0 *= A->momentum; m1 *= A->momentum;
    if (f32) { m0 += (double)((float)A->scaling_factor * (float)ax); m1 += (double)((float)A->scaling_factor * (float)ay); }
    else { m0 += A->scaling_factor * ax; m1 += A->scaling_factor * ay; }
  } else {
    double mx = (grid_action == 0) ? -1. : (grid_action == 1 ? 1. : 0.);
    double my = (grid_action == 2) ? -1. : (grid_action == 3 ? 1. : 0.);
    m0 *= A->momentum; m1 *= A->momentum;
    m0 += mx; m1 += my;
  }
  double sc = A->scaling_factor;
  if (m0 < -sc) m0 = -sc;
  if (m0 > sc) m0 = sc;
  if (m1 < -sc) m1 = -sc;
  if (m1 > sc) m1 = sc;
  wsync();
  if (e.lane == 0) { EF(e)[om] = m0; EF(e)[om + 1] = m1; }
  wsync();
  for (int a = 0; a < A->n_layers; ++a) {
    int l = A->layers[a];
    int s0 = P->layer_slot0[l], s1 = s0 + P->layer_nslots[l];
    for (int s = s0; s < s1; ++s) {
      if (!ALIVE(s)) continue;
      double m = MASS(s);
      if (A->control_velocity) {
        wsync();
        if (e.lane == 0) { VELX(s) = m0 / m; VELY(s) = m1 / m; FLAGS(s) &= ~MOOG_F_VEL_F32; vel_unshare(e, s); }
        wsync();
      } else {
        vel_iadd(e, s, m0 / m, m1 / m);
      }
    }
  }
}

// ---- reset path (sprite.py:261-424, distributions.py, sprite_generators.py:77-105) ----------
template <bool X>   // X: computed shapes possible (rare-components kernels only)
__device__ inline void create_sprite(Env& e, int s, const double* fac, int vel_f32, int angvel_f32) {
  PProg P = EP(e);
  const int sid = (int)fac[MOOG_FAC_SHAPE];
  const bool computed = X && sid < 0;   // MOOG_DIST_EXPR_SHAPE: the centred path is staged in the slot's vertex area
  PShape sh = &P->shapes[computed ? 0 : sid];
  double x = fac[MOOG_FAC_X], y = fac[MOOG_FAC_Y];
  double angle = fac[MOOG_FAC_ANGLE], scale = fac[MOOG_FAC_SCALE], aspect = fac[MOOG_FAC_ASPECT];
  double sx = scale, sy = scale * aspect;
  double c = cos(angle), sn = sin(angle);
  double m00 = c * sx, m01 = (-sn) * sy, m10 = sn * sx, m11 = c * sy;
  int n = computed ? e.xs_n : sh->nverts;
  if (n > P->slot_vcap[s]) n = P->slot_vcap[s];
  double* v = VERT(s);
  double r = -1.0;
  const double cen0 = computed ? e.xs_centroid[0] : sh->centroid[0], cen1 = computed ? e.xs_centroid[1] : sh->centroid[1];
  const double ine0 = computed ? e.xs_inertia[0] : sh->inertia[0], ine1 = computed ? e.xs_inertia[1] : sh->inertia[1];
  wsync();
  for (int k = e.lane; k < n; k += 64) {
    double ux, uy;
    if (computed) { ux = v[2 * k]; uy = v[2 * k + 1]; }   // (n <= 64: every lane reads its own vertex before writing it)
    else { ux = P->shape_verts[sh->voff + k][0]; uy = P->shape_verts[sh->voff + k][1]; }
    double vx = (m00 * ux + m01 * uy) + x;
    double vy = (m10 * ux + m11 * uy) + y;
    v[2 * k] = vx; v[2 * k + 1] = vy;
    r = fmax(r, norm2(vx - x, vy - y));
  }
  // np.max over the vertex radii (NaN-free in practice; fmax is order independent)
  for (int o = 32; o > 0; o >>= 1) r = fmax(r, shfl_d(r, e.lane ^ o));
  if (e.lane == 0) {
    NV(s) = n;
    SHAPEID_SET(s, sid);
    MAXR(s) = r;
    INER(s, 0) = ine0 * (sx * sx);
    INER(s, 1) = ine1 * (sy * sy);
    PX(s) = x; PY(s) = y;
    ANG(s) = angle;
    VELX(s) = fac[MOOG_FAC_XVEL]; VELY(s) = fac[MOOG_FAC_YVEL];
    ANGV(s) = fac[MOOG_FAC_ANGVEL];
    MASS(s) = fac[MOOG_FAC_MASS];
    COL_SET(s, 0, fac[MOOG_FAC_C0]); COL_SET(s, 1, fac[MOOG_FAC_C1]); COL_SET(s, 2, fac[MOOG_FAC_C2]);
    OPAC_SET(s, (int32_t)fac[MOOG_FAC_OPACITY]);
    TELE_SET(s, 0);
    vel_unshare(e, s);
    int fl = 0;
    if (!computed && sh->is_circle && aspect == 1) fl |= MOOG_F_SYM_CIRCLE;
    if (vel_f32) fl |= MOOG_F_VEL_F32;
    if (angvel_f32) fl |= MOOG_F_ANGVEL_F32;
    FLAGS(s) = fl;
    if (P->sprite_factors) {   // sprite.py:307-316: angle / scale / aspect are float()-ed
      SCALE(s) = scale; ASPECT(s) = aspect;
      FMASK(s) = (int32_t)(e.cur_fmask & ((1u << MOOG_FAC_C0) | (1u << MOOG_FAC_C1) | (1u << MOOG_FAC_C2) |
                                          (1u << MOOG_FAC_MASS)));
    }
  }
  wsync();
  set_position(e, s, x + cen0, y + cen1);
  bbox_exact_wave(e, s);
}

// sprite.py:360-401 for a polygon computed per episode: signed area, centroid and inertia by the triangle fan from
// the origin (sequential sums, every lane redundantly: the order is part of the result); a clockwise polygon is
// reversed; the path is centred on the centroid (1 * x + 0 * y + t); inertia about the centroid, per unit area.
// In: raw vertices in the slot's vertex area.  Out: the centred path there, e.xs_* for create_sprite.
__device__ inline void shape_record(Env& e, int s, int n) {
  double* raw = VERT(s);
  wsync();
  double in0 = 0, in1 = 0, area = 0, c0 = 0, c1 = 0;
  for (int i = 0; i < n; ++i) {
    const int j = (i + 1 == n) ? 0 : i + 1;
    const double ax = raw[2 * i], ay = raw[2 * i + 1], bx = raw[2 * j], by = raw[2 * j + 1];
    const double cr = ax * by - ay * bx;
    const double w = (1. / 12.) * cr;
    in0 += w * ((ax * ax + bx * bx) + ax * bx);
    in1 += w * ((ay * ay + by * by) + ay * by);
    const double tri = cr / 2.;
    area += tri;
    c0 += ((ax + bx) / 3.) * tri;
    c1 += ((ay + by) / 3.) * tri;
  }
  c0 /= area; c1 /= area;
  const bool rev = area < 0;
  if (rev) { in0 *= -1.; in1 *= -1.; area *= -1.; }
  const double n0 = -1 * c0, n1 = -1 * c1;
  double px = 0, py = 0;
  if (e.lane < n) { const int src = rev ? n - 1 - e.lane : e.lane; px = raw[2 * src]; py = raw[2 * src + 1]; }
  wsync();
  if (e.lane < n) { raw[2 * e.lane] = (1.0 * px + 0.0 * py) + n0; raw[2 * e.lane + 1] = (0.0 * px + 1.0 * py) + n1; }
  wsync();
  in0 -= area * (c0 * c0); in1 -= area * (c1 * c1);
  e.xs_n = n;
  e.xs_centroid[0] = c0; e.xs_centroid[1] = c1;
  e.xs_inertia[0] = in0 / area; e.xs_inertia[1] = in1 / area;
}

// X: the program may carry reset-time expressions (direct np.random draws of the initializer and factors / shapes
// computed from them); only the rare-components kernels compile that in.
template <bool X>
__device__ inline void sample_factors(Env& e, PGenop op, double* fac) {
  // Lanes = factors: lane k reads factor k's record (one round trip for all fourteen instead of a chain of
  // dependent scalar loads per sampled factor), takes its uniform from the op's batch of draws (draw_pos: the
  // reference's sample order, fixed when the config was lowered) and computes its value; the values then go to
  // every lane.  This sits inside the rejection loop of the sampler, i.e. on the critical path of every reset.
  const int kf = e.lane < MOOG_NUM_FACTORS ? e.lane : 0;
  PFactor F = &op->factors[kf];
  const int kind = F->kind, f32 = F->f32, n_cand = F->n_cand, cand_off = F->cand_off, dpos = F->draw_pos;
  const double fa = F->a, fb = F->b;
  const int ndraw = op->n_draws;
  const double draws = ndraw > 0 ? next_uniforms_lanes(e, ndraw) : 0.0;
  const double u = shfl_d(draws, dpos >= 0 ? dpos : 0);
  double val = fa;
  if (kind == MOOG_DIST_CONTINUOUS) {
    val = fa + (fb - fa) * u;
    if (f32) val = f32r(val);
  } else if (kind == MOOG_DIST_DISCRETE) {
    int idx = (int)(u * n_cand);
    if (idx >= n_cand) idx = n_cand - 1;
    val = EP(e)->cand[cand_off + idx];
  } else if (kind == MOOG_DIST_MAZE_COORD) {   // factors read off the maze cell the sprite sits on (pacman.py:47-65, maze.py:98-111)
    val = EP(e)->cand[cand_off + (n_cand ? e.cell_j : e.cell_i)];
  } else if (kind == MOOG_DIST_MAZE_SHAPE) {
    val = fa + (double)(e.cell_j * EP(e)->maze.size + e.cell_i);
  }
#pragma unroll
  for (int q = 0; q < MOOG_NUM_FACTORS; ++q) fac[q] = shfl_d(val, q);
  if constexpr (X) {   // direct np.random draws of the initializer taken between the factor draws: their uniforms to the record
    if (op->n_sampled > 0 && ndraw > 0) {
      int kd = 0;
      for (int k = 0; k < op->n_sampled; ++k) {
        const int fi = op->sample_order[k];
        if (fi >= MOOG_NUM_FACTORS) {
          const double uh = shfl_d(draws, kd);
          wsync();
          if (e.lane == 0) EF(e)[EL(e).o_hdraw + fi - MOOG_NUM_FACTORS] = uh;
          wsync();
          ++kd;
        } else {
          const int kk = op->factors[fi].kind;
          kd += (kk == MOOG_DIST_CONTINUOUS || kk == MOOG_DIST_DISCRETE) ? 1 : 0;
        }
      }
    }
  }
  // which factors are float32 samples (lanes = factors)
  e.flat_f32 = (unsigned)(__ballot(e.lane < MOOG_NUM_FACTORS && kind == MOOG_DIST_CONTINUOUS && f32 != 0) & 0x3fffull);
  bool any_expr = false;
  if constexpr (X) {
    any_expr = __any(e.lane < MOOG_NUM_FACTORS && (kind == MOOG_DIST_EXPR || kind == MOOG_DIST_EXPR_SHAPE));
    e.expr_f32 = 0u;
  }
  if (X && any_expr) {   // factors the initializer computed from its draws (after all draws of the op are in)
    // a DependentDistribution's function reads the sampled factors themselves: staged in LDS for MOOG_X_FACTOR
    unsigned f32m = 0u;
    wsync();
#pragma unroll
    for (int q = 0; q < MOOG_NUM_FACTORS; ++q) {
      if (e.lane == 0) reinterpret_cast<double*>(ELST(e))[q] = fac[q];
    }
    f32m = e.flat_f32;
    wsync();
    e.fac_f32 = f32m; e.expr_f32 = 0u;
#pragma unroll
    for (int q = 0; q < MOOG_NUM_FACTORS; ++q) {
      PFactor Fq = &op->factors[q];
      if (Fq->kind == MOOG_DIST_EXPR) {
        int tag = 0;
        fac[q] = eval_expr(e, Fq->cand_off, 0, 0, &tag, nullptr);
        if (tag == 1) e.expr_f32 |= 1u << q;
      }
      else if (Fq->kind == MOOG_DIST_EXPR_SHAPE) {
        eval_expr(e, Fq->cand_off, 0, 0, nullptr, nullptr);   // raw vertices -> VERT(cur_slot)
        shape_record(e, e.cur_slot, Fq->n_cand);
        fac[q] = -1.0;
      }
    }
  } else {
    e.expr_f32 = 0u;
  }
}

// ---- distribution programs (moog_dinstr_t): Mixture / Intersection / SetMinus / Selection /
//      Discrete(probs) sampling in the reference's draw order (distributions.py:159-405).
//      Wave-uniform: every lane runs the same program on the same uniforms; `fac` stays in
//      registers (static indexing through select chains).

__device__ inline double fac_get(const double* fac, int a) {
  double v = 0;
#pragma unroll
  for (int q = 0; q < MOOG_NUM_FACTORS; ++q) if (q == a) v = fac[q];
  return v;
}
__device__ inline void fac_set(double* fac, int a, double v) {
#pragma unroll
  for (int q = 0; q < MOOG_NUM_FACTORS; ++q) if (q == a) fac[q] = v;
}

// numpy legacy choice(n, p): searchsorted(cumsum(p) / cumsum(p)[-1], u, side='right')
__device__ inline int choice_p(Env& e, int poff, int n, double u) {
  PProg P = EP(e);
  double last = 0;
  for (int i = 0; i < n; ++i) last = (i == 0) ? P->cand[poff] : last + P->cand[poff + i];
  double acc = 0;
  int idx = 0;
  for (int i = 0; i < n; ++i) {
    acc = (i == 0) ? P->cand[poff] : acc + P->cand[poff + i];
    if (acc / last <= u) idx = i + 1;
  }
  return idx >= n ? n - 1 : idx;
}

__device__ inline int dist_pred(Env& e, int off, int len, const double* fac, unsigned f32mask) {
  PProg P = EP(e);
  unsigned stack = 0;
  for (int pc = off; pc < off + len; ++pc) {
    PDinstr I = &P->dcode[pc];
    const int op = I->op, a = I->a, b = I->b;
    if (op == MOOG_P_RANGE) {
      double v = fac_get(fac, a);
      bool in = ((f32mask >> a) & 1u) ? ((float)v >= (float)I->x && (float)v < (float)I->y)
                                      : (v >= I->x && v < I->y);
      stack = (stack << 1) | (in ? 1u : 0u);
    } else if (op == MOOG_P_SET) {
      double v = fac_get(fac, a);
      bool in = false;
      for (int k = 0; k < b; ++k) {
        double c = P->cand[I->c + k];
        in = in || (((f32mask >> a) & 1u) ? ((float)c == (float)v) : (c == v));
      }
      stack = (stack << 1) | (in ? 1u : 0u);
    } else if (op == MOOG_P_AND || op == MOOG_P_OR) {
      unsigned m = (1u << b) - 1u, top = stack & m;
      bool v = (op == MOOG_P_AND) ? (top == m) : (top != 0u);
      stack = ((stack >> b) << 1) | (v ? 1u : 0u);
    } else if (op == MOOG_P_NOT) {
      stack ^= 1u;
    }
  }
  return (int)(stack & 1u);
}

__device__ inline void run_dist_program(Env& e, int pc, double* fac, unsigned& f32mask) {
  PProg P = EP(e);
  int tries0 = 0, tries1 = 0;
  for (;;) {
    PDinstr I = &P->dcode[pc];
    const int op = I->op, a = I->a, b = I->b;
    if (op == MOOG_D_CONT) {
      double u = next_uniform(e);
      double v = I->x + (I->y - I->x) * u;
      if (b) { v = f32r(v); f32mask |= 1u << a; } else f32mask &= ~(1u << a);
      fac_set(fac, a, v); ++pc;
    } else if (op == MOOG_D_DISC) {
      double u = next_uniform(e);
      int idx = (int)(u * b);
      if (idx >= b) idx = b - 1;
      fac_set(fac, a, P->cand[I->c + idx]); f32mask &= ~(1u << a); ++pc;
    } else if (op == MOOG_D_DISCP) {
      int idx = choice_p(e, I->d, b, next_uniform(e));
      fac_set(fac, a, P->cand[I->c + idx]); f32mask &= ~(1u << a); ++pc;
    } else if (op == MOOG_D_CONST) {
      fac_set(fac, a, I->x); f32mask &= ~(1u << a); ++pc;
    } else if (op == MOOG_D_CHOICE) {
      int idx = choice_p(e, I->d, b, next_uniform(e));
      pc = P->dcode[pc + 1 + idx].a;
    } else if (op == MOOG_D_JUMP) {
      pc = a;
    } else if (op == MOOG_D_LOOP) {
      if (a == 0) tries0 = 0; else tries1 = 0;
      ++pc;
    } else if (op == MOOG_D_TEST) {
      int t = (a == 0) ? ++tries0 : ++tries1;
      if (dist_pred(e, I->c, b, fac, f32mask) == I->d) { ++pc; }
      else if (t >= MOOG_DIST_MAX_TRIES) {
        if (e.lane == 0) EQ(e)[EL(e).o_fault] |= MOOG_FAULT_DIST_EXHAUSTED;
        ++pc;
      } else pc = (int)I->x;
    } else {
      return;   // MOOG_D_END
    }
  }
}

// n ~ randint(count_min, count_max + 1) when the generator's num_sprites is a range
__device__ inline int genop_count(Env& e, PGenop op) {
  int n = op->count_max;
  if (op->count_min < op->count_max) {
    double u = next_uniform(e);
    int span = op->count_max + 1 - op->count_min;
    int k = (int)(u * span);
    if (k >= span) k = span - 1;
    n = op->count_min + k;
  }
  return n;
}

// one `factor_dist.sample()`: the factors and which of them are float32 samples
template <bool X>
__device__ inline void sample_op_factors(Env& e, PGenop op, double* fac, int& vel_f32, int& angvel_f32) {
  sample_factors<X>(e, op, fac);
  unsigned m = e.flat_f32;
  {
    const unsigned vb = (1u << MOOG_FAC_XVEL) | (1u << MOOG_FAC_YVEL);
    vel_f32 = (m & vb) == vb;
    angvel_f32 = (m >> MOOG_FAC_ANGVEL) & 1u;
  }
  if (op->code_off >= 0) {   // which factors are float32 samples depends on the branch taken
    run_dist_program(e, op->code_off, fac, m);
    const unsigned vb = (1u << MOOG_FAC_XVEL) | (1u << MOOG_FAC_YVEL);
    vel_f32 = (m & vb) == vb;
    angvel_f32 = (m >> MOOG_FAC_ANGVEL) & 1u;
  }
  if (X && e.expr_f32) {   // computed factors that came out float32 (a float32 sample under weak scalars)
    m |= e.expr_f32;
    const unsigned vb = (1u << MOOG_FAC_XVEL) | (1u << MOOG_FAC_YVEL);
    vel_f32 = (m & vb) == vb;
    angvel_f32 = (m >> MOOG_FAC_ANGVEL) & 1u;
  }
  e.cur_fmask = m;
}

// ---- the per-episode random maze of pacman.py:39-65 (see the oracle for the line-by-line citations) --------
// Wave-uniform scalar code: every lane runs it redundantly on the same LDS scratch (the bit-matrix words of the
// same-layer collision scan, free during a reset); a wave's LDS operations complete in order, so plain stores
// followed by loads need no barrier.  Draws are fetched up to 64 at a time (one Philox latency per batch).
// maze_generators.py:96-171 generate_random_maze_matrix + np.flip(axis=0) -> rows in the record.
__device__ inline void maze_generate(Env& e) {
  PProg P = EP(e);
  const int n = P->maze.gen_size, N = P->maze.size;
  uint32_t* m = reinterpret_cast<uint32_t*>(EROWM(e));   // [16] row masks of the size x size matrix (bit b of row a = wall)
  uint32_t* inl = m + MOOG_MAX_MAZE_GEN;               // [16] "is in closed_neighbors"
  uint8_t* list = reinterpret_cast<uint8_t*>(inl + MOOG_MAX_MAZE_GEN);   // [256] closed_neighbors as a << 4 | b
  const uint32_t full = (n >= 32) ? 0xffffffffu : ((1u << n) - 1u);
  for (;;) {   // a maze without open cells is drawn again (:154-156)
    int nlist = 0;
    for (int a = 0; a < MOOG_MAX_MAZE_GEN; ++a) { m[a] = full; inl[a] = 0u; }
    const double d2 = next_uniforms_lanes(e, 2);   // randint(0, size, size=(2,)) (:143)
    int pi = (int)(shfl_d(d2, 0) * n), pj = (int)(shfl_d(d2, 1) * n);
    if (pi >= n) pi = n - 1;
    if (pj >= n) pj = n - 1;
    for (;;) {
      // _open_point (:118-123): closed neighbours join the list once (order: up, down, left, right), then the point opens
#pragma unroll
      for (int d = 0; d < 4; ++d) {
        const int a = pi + (d == 0 ? -1 : (d == 1 ? 1 : 0)), b = pj + (d == 2 ? -1 : (d == 3 ? 1 : 0));
        if (a < 0 || b < 0 || a >= n || b >= n) continue;
        if (((m[a] >> b) & 1u) && !((inl[a] >> b) & 1u)) { inl[a] |= 1u << b; list[nlist++] = (uint8_t)(a << 4 | b); }
      }
      m[pi] &= ~(1u << pj);
      // _find_and_open_new_point (:125-140): np.random.shuffle(list) = for i = n - 1 .. 1: j = int(u * (i + 1)); swap
      for (int hi = nlist - 1; hi > 0; hi -= 64) {
        const int cnt = hi < 64 ? hi : 64;
        const double draws = next_uniforms_lanes(e, cnt);
        for (int t = 0; t < cnt; ++t) {
          const int i = hi - t;
          int j = (int)(shfl_d(draws, t) * (i + 1));
          if (j > i) j = i;
          const uint8_t vi = list[i], vj = list[j];
          list[i] = vj; list[j] = vi;
        }
      }
      bool found = false;
      for (int k = 0; k < nlist && !found; ++k) {
        const int a = list[k] >> 4, b = list[k] & 15;
        if (!((m[a] >> b) & 1u)) continue;
        // would opening (a, b) complete an open 2 x 2 block?  rows a-1, a, a+1 around columns b-1 .. b+1 (:45-57)
        const uint32_t up = a > 0 ? m[a - 1] : full, mid = m[a], dn = a + 1 < n ? m[a + 1] : full;
        bool will = false;
#pragma unroll
        for (int q = 0; q < 4; ++q) {
          const int ba = a - 1 + (q >> 1), bb = b - 1 + (q & 1);
          if (ba < 0 || bb < 0 || ba + 1 >= n || bb + 1 >= n) continue;
          const uint32_t r0 = (q >> 1) ? mid : up, r1 = (q >> 1) ? dn : mid;
          const int sum = (int)((r0 >> bb) & 1u) + (int)((r0 >> (bb + 1)) & 1u) + (int)((r1 >> bb) & 1u) +
                          (int)((r1 >> (bb + 1)) & 1u);
          will = will || sum <= 1;
        }
        if (!will) { pi = a; pj = b; found = true; }
      }
      if (!found) break;
    }
    // _remove_dead_ends (:62-93): close the first open cell (row-major) with fewer than two open neighbours, repeat
    for (bool again = true; again;) {
      again = false;
      for (int a = 0; a < n && !again; ++a) {
        const uint32_t up = a > 0 ? m[a - 1] : full, mid = m[a], dn = a + 1 < n ? m[a + 1] : full;
        for (int b = 0; b < n && !again; ++b) {
          if ((mid >> b) & 1u) continue;
          int open = (int)(!((up >> b) & 1u)) + (int)(!((dn >> b) & 1u));
          if (b > 0) open += (int)(!((mid >> (b - 1)) & 1u));
          if (b + 1 < n) open += (int)(!((mid >> (b + 1)) & 1u));
          if (open < 2) { m[a] = mid | (1u << b); again = true; }
        }
      }
    }
    uint32_t any_open = 0u;
    for (int a = 0; a < n; ++a) any_open |= ~m[a] & full;
    if (any_open) break;
  }
  // wall border (:159-167), flip, rows to the record
  const int start = N > n ? (N - n) / 2 : 0;
  const uint32_t fullN = (N >= 32) ? 0xffffffffu : ((1u << N) - 1u);
  for (int r = 0; r < N; ++r) {
    const int a = (P->maze.flip ? N - 1 - r : r) - start;
    uint32_t bits = fullN;
    if (a >= 0 && a < n) bits = (fullN & ~(full << start)) | ((m[a] & full) << start);
    if (e.lane == 0) EQ(e)[EL(e).o_maze + r] = (int32_t)bits;
  }
  wsync();
  // rank -> cell tables for the one-sprite ops that follow (each used to re-scan the matrix: a quarter of a pacman
  // reset): wall cells in Maze.to_sprites order (columns outer) from the front, open cells in np.argwhere order (rows
  // outer) behind them, as i << 8 | j, in the scratch the generator no longer needs.  Lanes = cells.
  e.cell_tab_n = 0;
  if (N * N <= 256) {
    unsigned short* tab = reinterpret_cast<unsigned short*>(EROWM(e));
    int nw = 0;
    for (int c0 = 0; c0 < N * N; c0 += 64) {
      const int c = c0 + e.lane;
      const int j = c / N, i = c - j * N;   // column-major
      const bool wall = c < N * N && ((maze_row(e, i) >> j) & 1u);
      const unsigned long long m = __ballot(wall);
      if (wall) tab[nw + __popcll(m & ((1ull << e.lane) - 1ull))] = (unsigned short)(i << 8 | j);
      nw += __popcll(m);
    }
    int no = 0;
    for (int c0 = 0; c0 < N * N; c0 += 64) {
      const int c = c0 + e.lane;
      const int i = c / N, j = c - i * N;   // row-major
      const bool open = c < N * N && !((maze_row(e, i) >> j) & 1u);
      const unsigned long long m = __ballot(open);
      if (open) tab[nw + no + __popcll(m & ((1ull << e.lane) - 1ull))] = (unsigned short)(i << 8 | j);
      no += __popcll(m);
    }
    wsync();
    e.cell_tab_n = N * N; e.cell_nw = nw;
  }
}

// k-th (0-based) open cell of the maze in np.argwhere order (rows outer); -1 when there are fewer
__device__ inline int maze_open_cell(const Env& e, int k) {
  const int N = EP(e)->maze.size;
  const uint32_t fullN = (N >= 32) ? 0xffffffffu : ((1u << N) - 1u);
  for (int i = 0; i < N; ++i) {
    uint32_t open = ~maze_row(e, i) & fullN;
    const int c = __popc(open);
    if (k < c) {
      for (; k > 0; --k) open &= open - 1u;
      return i << 8 | (__ffs((int)open) - 1);
    }
    k -= c;
  }
  return -1;
}

// maze.py:200-214 sample_distinct_open_points(k): the first k steps of a forward Fisher-Yates shuffle of the open
// cells' ranks (make_golden.py _choice): the permutation is tracked as the few displaced entries only
__device__ inline void maze_sample_points(Env& e, int k) {
  const int N = EP(e)->maze.size;
  const uint32_t fullN = (N >= 32) ? 0xffffffffu : ((1u << N) - 1u);
  int n = 0;
  for (int i = 0; i < N; ++i) n += __popc(~maze_row(e, i) & fullN);
  const int kk = k < MOOG_MAX_MAZE_POINTS ? k : MOOG_MAX_MAZE_POINTS;
  const int nd = kk < n ? kk : n;
  const double draws = nd > 0 ? next_uniforms_lanes(e, nd) : 0.0;
  int moved_pos[MOOG_MAX_MAZE_POINTS], moved_val[MOOG_MAX_MAZE_POINTS];   // perm[pos] = val where it differs from pos
#pragma unroll
  for (int t = 0; t < MOOG_MAX_MAZE_POINTS; ++t) {
    if (t >= kk) break;
    int point = -1;
    if (t < n) {
      int j = t + (int)(shfl_d(draws, t) * (n - t));
      if (j >= n) j = n - 1;
      // value at j (after the earlier swaps), value at t
      int vj = j, vt = t;
#pragma unroll
      for (int q = 0; q < MOOG_MAX_MAZE_POINTS; ++q)
        if (q < t) { if (moved_pos[q] == j) vj = moved_val[q]; if (moved_pos[q] == t) vt = moved_val[q]; }
      // perm[t] <-> perm[j]: position t is never read again; position j now holds vt
      bool upd = false;
#pragma unroll
      for (int q = 0; q < MOOG_MAX_MAZE_POINTS; ++q)
        if (q < t && moved_pos[q] == j) { moved_val[q] = vt; upd = true; }
      moved_pos[t] = upd ? -1 : j; moved_val[t] = vt;
      point = maze_open_cell(e, vj);
    } else { moved_pos[t] = -1; moved_val[t] = 0; }
    if (e.lane == 0) EQ(e)[EL(e).o_maze + MOOG_MAX_MAZE + t] = point;
  }
  wsync();
}

// the cell a one-sprite op sits on; false when the maze has no such cell (the slot stays dead)
__device__ inline bool maze_select_cell(Env& e, int sel, int arg) {
  int p = -1;
  if (sel == MOOG_CELL_SAMPLED) p = EQ(e)[EL(e).o_maze + MOOG_MAX_MAZE + arg];
  else if (e.cell_tab_n > 0) {   // the tables maze_generate left in LDS
    const unsigned short* tab = reinterpret_cast<const unsigned short*>(EROWM(e));
    if (sel == MOOG_CELL_WALL_RANK) p = arg < e.cell_nw ? (int)tab[arg] : -1;
    else p = arg < e.cell_tab_n - e.cell_nw ? (int)tab[e.cell_nw + arg] : -1;
  }
  else if (sel == MOOG_CELL_OPEN_RANK) p = maze_open_cell(e, arg);   // np.argwhere(maze.maze == 0), pacman.py:62
  else {   // Maze.to_sprites: x (column) outer, y (row) inner, maze.py:101-103
    const int N = EP(e)->maze.size;
    int seen = 0;
    for (int j = 0; j < N && p < 0; ++j) {
      uint32_t col = 0u;
      for (int i = 0; i < N; ++i) col |= ((maze_row(e, i) >> j) & 1u) << i;
      const int c = __popc(col);
      if (arg - seen < c) {
        for (int k = arg - seen; k > 0; --k) col &= col - 1u;
        p = (__ffs((int)col) - 1) << 8 | j;
      }
      seen += c;
    }
  }
  if (p < 0) return false;
  e.cell_i = p >> 8; e.cell_j = p & 255;
  return true;
}

template <bool DYN>
__device__ inline void run_genop(Env& e, int oi) {
  PProg P = EP(e);
  PGenop op = &P->ops[oi];
  if (op->runtime) return;   // CreateSprites generators run at rule time
  constexpr bool FULL = DYN && (MOOG_WITH_MAZE != 0);   // maze ops and reset-time expressions: the m3 / m4 kernels only
  if constexpr (FULL) {
    // an alternative of a sample_generator runs only when it was the one picked (sprite_generators.py:131-154)
    if (op->cond_hdraw > 0 && (int)EF(e)[EL(e).o_hdraw + op->cond_hdraw - 1] != op->cond_value) return;
    if (op->cell_sel == MOOG_CELL_CHOICE) {   // np.random.choice(generators, p=p)
      const int n = op->count_max, off = op->factors[0].cand_off;
      int idx = 0;
      if (off >= 0) {   // cdf = cumsum(p) / sum(p); searchsorted(cdf, u, side='right')
        const double u = next_uniform(e);
        while (idx < n && P->cand[off + idx] <= u) ++idx;
        if (idx >= n) idx = n - 1;
      } else if (n > 1) {
        idx = (int)(next_uniform(e) * n);
        if (idx >= n) idx = n - 1;
      }
      wsync();
      if (e.lane == 0) EF(e)[EL(e).o_hdraw + op->cell_arg] = (double)idx;
      wsync();
      return;
    }
    if (op->cell_sel == MOOG_CELL_GENERATE) { PROF_T0; maze_generate(e); PROF_ADD(e, 13); return; }
    if (op->cell_sel == MOOG_CELL_SAMPLE) { maze_sample_points(e, op->cell_arg); return; }
  }
  if constexpr (FULL) {
    if (op->cell_sel == MOOG_CELL_HDRAW) {   // a direct np.random draw of the initializer
      for (int tries = 0;; ++tries) {
        const double u = next_uniform(e);
        wsync();
        if (e.lane == 0) EF(e)[EL(e).o_hdraw + op->cell_arg] = u;
        wsync();
        if (op->code_off < 0) break;   // (no rejection loop around it)
        if (eval_expr(e, op->code_off, 0, 0, nullptr, nullptr) != 0.0) break;
        if (tries >= MOOG_DIST_MAX_TRIES) { if (e.lane == 0) EQ(e)[EL(e).o_fault] |= MOOG_FAULT_SAMPLER_EXHAUSTED; break; }
      }
      return;
    }
    if (op->cell_sel == MOOG_CELL_SIMULATE) {   // the initializer's look-ahead: physics steps until one of its exits holds
      bbox_build_all(e);
      const int K = uni(EP(e)->updates_per_env_step);
      int exit_k = 0;
      for (int it = 0;; ++it) {
        wsync();
        if (e.lane == 0) EF(e)[EL(e).o_hdraw + op->cell_arg + 1] = (double)it;   // `for step in range(n)`: the loop counter
        wsync();
        exit_k = (int)eval_expr(e, op->code_off, 0, 0, nullptr, nullptr);
        if (exit_k != 0) break;
        if (it >= op->count_max) { if (e.lane == 0) EQ(e)[EL(e).o_fault] |= MOOG_FAULT_SAMPLER_EXHAUSTED; break; }
        for (int k = 0; k < K; ++k) apply_physics<DYN>(e);
      }
      wsync();
      if (e.lane == 0) EF(e)[EL(e).o_hdraw + op->cell_arg] = (double)exit_k;
      wsync();
      if (exit_k > 0 && exit_k < 31 && ((op->max_tries >> exit_k) & 1)) e.restart = 1;   // `return state_initializer()`
      return;
    }
    if (op->cell_sel == MOOG_CELL_STORE) {   // `sprite.position = ...` / `.velocity = ...` on a built sprite
      bbox_build_all(e);
      run_modifier(e, op->code_off, op->cell_arg);
      return;
    }
    if (op->cell_sel == MOOG_CELL_PSTATE) {   // a number the initializer keeps across episodes (never cleared by resets)
      const int ri = op->cell_arg;
      const bool first = EF(e)[EL(e).o_rule2 + ri] == 0.0;
      const double v = first ? op->factors[0].a : eval_expr(e, op->code_off, 0, 0, nullptr, nullptr);
      wsync();
      if (e.lane == 0) { EF(e)[EL(e).o_rule + ri] = v; EF(e)[EL(e).o_rule2 + ri] = 1.0; }
      wsync();
      return;
    }
    if (op->cell_sel == MOOG_CELL_HEXPR) {   // a value computed from the draws, kept for several readers
      int tag = 0;
      const double v = eval_expr(e, op->code_off, 0, 0, &tag, nullptr);
      wsync();
      if (e.lane == 0) {
        EF(e)[EL(e).o_hdraw + op->cell_arg] = v;
        if (op->count_min) EF(e)[EL(e).o_hdraw + op->cell_arg + 1] = (double)tag;   // (np.copy keeps the dtype)
      }
      wsync();
      return;
    }
  }
  if constexpr (FULL) {
    if (op->cell_sel == MOOG_CELL_SHUFFLE) {   // np.random.shuffle(order) + reindexing = the same swaps on the list itself
      int n = 0;
      while (n < op->cell_arg && ALIVE(op->slot0 + n)) ++n;
      const int tmp = op->slot0 + op->cell_arg;
      for (int i = n - 1; i > 0; --i) {
        int j = (int)(next_uniform(e) * (i + 1));
        if (j > i) j = i;
        if (j == i) continue;
        move_slot(e, tmp, op->slot0 + i);
        move_slot(e, op->slot0 + i, op->slot0 + j);
        move_slot(e, op->slot0 + j, tmp);
      }
      return;
    }
  }
  int n = genop_count(e, op);
  if constexpr (FULL) {
    PROF_T0;
    if (op->cell_sel != MOOG_CELL_NONE && op->cell_sel != MOOG_CELL_HDRAW &&
        !maze_select_cell(e, op->cell_sel, op->cell_arg)) n = 0;
    PROF_ADD(e, 14);
  }
  // without_overlapping: the slot ranges of the ops to avoid, adjacent ones merged, read once per op (the rejection loop
  // below runs dozens of times per reset: no constant-memory chains inside it).  More than four ranges: the plain loop.
  int av_lo0 = 0, av_hi0 = 0, av_lo1 = 0, av_hi1 = 0, av_lo2 = 0, av_hi2 = 0, av_lo3 = 0, av_hi3 = 0, n_av = 0;
  {
    const unsigned long long avoid = op->avoid_ops;
    for (int oj = 0; avoid != 0ull && oj < oi && oj < 64; ++oj) {
      if (!((avoid >> oj) & 1ull)) continue;
      const int lo = P->ops[oj].slot0, hi = lo + P->ops[oj].count_max;
      if (n_av == 1 && av_hi0 == lo) av_hi0 = hi;
      else if (n_av == 2 && av_hi1 == lo) av_hi1 = hi;
      else if (n_av == 3 && av_hi2 == lo) av_hi2 = hi;
      else if (n_av == 4 && av_hi3 == lo) av_hi3 = hi;
      else {
        if (n_av == 0) { av_lo0 = lo; av_hi0 = hi; }
        else if (n_av == 1) { av_lo1 = lo; av_hi1 = hi; }
        else if (n_av == 2) { av_lo2 = lo; av_hi2 = hi; }
        else if (n_av == 3) { av_lo3 = lo; av_hi3 = hi; }
        ++n_av;   // (5: too many, see below)
      }
    }
  }
  const int cmax = op->count_max, slot0 = op->slot0, disjoint = op->disjoint & 1, max_tries = op->max_tries,
            graceful = op->fail_gracefully;
  // cell_arg > 0 (plain sprite ops): the call's k-th sprite lives in slot slot0 + cand[cell_arg - 1 + k] (the config spread
  // the call's sprites over its state in another order, red_green.py:157-183); earlier sprites of the call = the live ones
  const int perm = (op->cell_sel == MOOG_CELL_NONE && op->cell_arg > 0) ? op->cell_arg - 1 : -1;
  for (int k = 0; k < cmax; ++k) {
    int s = slot0 + (perm >= 0 ? (int)P->cand[perm + k] : k);
    if (k >= n) {
      wsync();
      if (e.lane == 0) { FLAGS(s) = 0; NV(s) = 0; }
      wsync();
      continue;
    }
    int count = 0;
    for (;;) {
      double fac[MOOG_NUM_FACTORS];
      int vel_f32, angvel_f32;
      if constexpr (FULL) e.cur_slot = s;
      { PROF_T0; sample_op_factors<FULL>(e, op, fac, vel_f32, angvel_f32); PROF_ADD(e, 15); }
      { PROF_T0; create_sprite<FULL>(e, s, fac, vel_f32, angvel_f32); PROF_ADD(e, 12); }
      bool ov = false;
      if (n_av <= 4) {
        if (n_av > 0) ov = overlaps_any(e, s, av_lo0, av_hi0);
        if (n_av > 1 && !ov) ov = overlaps_any(e, s, av_lo1, av_hi1);
        if (n_av > 2 && !ov) ov = overlaps_any(e, s, av_lo2, av_hi2);
        if (n_av > 3 && !ov) ov = overlaps_any(e, s, av_lo3, av_hi3);
      } else {
        for (int oj = 0; oj < oi && oj < 64 && !ov; ++oj) {
          if (!((op->avoid_ops >> oj) & 1)) continue;
          PGenop o2 = &P->ops[oj];
          ov = overlaps_any(e, s, o2->slot0, o2->slot0 + o2->count_max);
        }
      }
      if (disjoint && !ov) ov = overlaps_any(e, s, slot0, perm >= 0 ? slot0 + cmax : s);
      if (!ov) break;
      if (count > max_tries) {
        wsync();
        if (graceful) {   // `return sprites` (sprite_generators.py:93-95): this sprite and the rest of the call are dropped
          for (int t = k + e.lane; t < cmax; t += 64) {
            const int st = slot0 + (perm >= 0 ? (int)P->cand[perm + t] : t);
            FLAGS(st) = 0; NV(st) = 0;
          }
          wsync();
          if (graceful == 2) e.restart = 1;   // the config measures the result and starts over (red_green.py:152-155)
          return;
        }
        if (e.lane == 0) EQ(e)[EL(e).o_fault] |= MOOG_FAULT_SAMPLER_EXHAUSTED;
        wsync();
        break;
      }
      ++count;
    }
    wsync();
    if (e.lane == 0) FLAGS(s) |= MOOG_F_ALIVE;
    wsync();
  }
}

// A run of consecutive ONE-SPRITE ops without draws, rejection tests or computed factors (op->disjoint == 2, set by the
// lowering: the wall squares and pellets of a maze, the constant sprites of any config), lanes = ops: every lane builds
// its own sprite with the arithmetic of sample_factors / create_sprite / set_position / dop_scan.  Such ops do not see
// each other, so building up to 64 of them at once is the same as one after the other (pacman: 184 of 189 ops, each
// ~30 k cycles of dependent constant-memory reads on its own).  Needs the rank -> cell tables of this reset in LDS when
// cells are selected by rank (otherwise the caller takes the ordinary path).
__device__ inline void run_static_batch(Env& e, int oi0, int nb) {
  PProg P = EP(e);
  wsync();
  if (e.lane < nb) {
    PGenop op = &P->ops[oi0 + e.lane];
    const int s = op->slot0;
    bool have = true;
    int ci = 0, cj = 0;
    if (op->cell_sel != MOOG_CELL_NONE) {
      int p = -1;
      if (op->cell_sel == MOOG_CELL_SAMPLED) p = EQ(e)[EL(e).o_maze + MOOG_MAX_MAZE + op->cell_arg];
      else {
        const unsigned short* tab = reinterpret_cast<const unsigned short*>(EROWM(e));
        if (op->cell_sel == MOOG_CELL_WALL_RANK) p = op->cell_arg < e.cell_nw ? (int)tab[op->cell_arg] : -1;
        else p = op->cell_arg < e.cell_tab_n - e.cell_nw ? (int)tab[e.cell_nw + op->cell_arg] : -1;
      }
      have = p >= 0;
      ci = p >> 8; cj = p & 255;
    }
    if (!have) { FLAGS(s) = 0; NV(s) = 0; }
    else {
      double fac[MOOG_NUM_FACTORS];
#pragma unroll
      for (int q = 0; q < MOOG_NUM_FACTORS; ++q) {
        PFactor F = &op->factors[q];
        double v = F->a;                                                              // MOOG_DIST_CONST
        if (F->kind == MOOG_DIST_MAZE_COORD) v = P->cand[F->cand_off + (F->n_cand ? cj : ci)];
        else if (F->kind == MOOG_DIST_MAZE_SHAPE) v = F->a + (double)(cj * P->maze.size + ci);
        fac[q] = v;
      }
      const int sid = (int)fac[MOOG_FAC_SHAPE];
      PShape sh = &P->shapes[sid];
      const double x = fac[MOOG_FAC_X], y = fac[MOOG_FAC_Y];
      const double angle = fac[MOOG_FAC_ANGLE], scale = fac[MOOG_FAC_SCALE], aspect = fac[MOOG_FAC_ASPECT];
      const double sx = scale, sy = scale * aspect;
      const double c = cos(angle), sn = sin(angle);
      const double m00 = c * sx, m01 = (-sn) * sy, m10 = sn * sx, m11 = c * sy;
      int n = sh->nverts;
      if (n > P->slot_vcap[s]) n = P->slot_vcap[s];
      double* v = VERT(s);
      double r = -1.0;
      for (int k = 0; k < n; ++k) {
        const double ux = P->shape_verts[sh->voff + k][0], uy = P->shape_verts[sh->voff + k][1];
        const double vx = (m00 * ux + m01 * uy) + x, vy = (m10 * ux + m11 * uy) + y;
        v[2 * k] = vx; v[2 * k + 1] = vy;
        r = fmax(r, norm2(vx - x, vy - y));
      }
      NV(s) = n;
      SHAPEID_SET(s, sid);
      MAXR(s) = r;
      INER(s, 0) = sh->inertia[0] * (sx * sx);
      INER(s, 1) = sh->inertia[1] * (sy * sy);
      ANG(s) = angle;
      VELX(s) = fac[MOOG_FAC_XVEL]; VELY(s) = fac[MOOG_FAC_YVEL];
      ANGV(s) = fac[MOOG_FAC_ANGVEL];
      MASS(s) = fac[MOOG_FAC_MASS];
      COL_SET(s, 0, fac[MOOG_FAC_C0]); COL_SET(s, 1, fac[MOOG_FAC_C1]); COL_SET(s, 2, fac[MOOG_FAC_C2]);
      OPAC_SET(s, (int32_t)fac[MOOG_FAC_OPACITY]);
      TELE_SET(s, 0);
      if (P->vel_alias) VALIAS(s) = 0;
      int fl = MOOG_F_ALIVE;
      if (sh->is_circle && aspect == 1) fl |= MOOG_F_SYM_CIRCLE;
      FLAGS(s) = fl;
      if (P->sprite_factors) { SCALE(s) = scale; ASPECT(s) = aspect; FMASK(s) = 0; }
      // set_position(x + centroid): the path moves by the difference to the position just stored (sprite.py:616-633)
      const double nx = x + sh->centroid[0], ny = y + sh->centroid[1];
      const double dx = nx - x, dy = ny - y;
      for (int k = 0; k < n; ++k) { v[2 * k] = v[2 * k] + dx; v[2 * k + 1] = v[2 * k + 1] + dy; }
      PX(s) = nx; PY(s) = ny;
      dop_scan(v, n, &BB(s, 0));
    }
  }
  wave_global_fence();
  wsync();
}

// environment.py:82-96
template <bool DYN>
__device__ inline void env_reset(Env& e) {
  PProg P = EP(e);
  wsync();
  // sprites the config built outside its initializer are not rebuilt once the env has been reset before (program.born_rule)
  bool born = P->born_rule > 0 && EF(e)[EL(e).o_rule + P->born_rule - 1] != 0.0;
  // every episode draws from its own segment of the env's stream (counter = episode << 32 | draw): a reset's draws do not
  // depend on how many the previous episode took, so episode E + 1 can be built while E is running (the reset pool)
  if (e.lane == 0) { EQ(e)[EL(e).o_rng + 1] = (int32_t)((uint32_t)EQ(e)[EL(e).o_rng + 1] + 1u); EQ(e)[EL(e).o_rng] = 0; }
  wsync();
  for (int attempt = 0;; ++attempt) {   // (an initializer may start over: `return state_initializer()`, red_green.py:155,203)
    for (int s = e.lane; s < P->n_slots; s += 64) {
      if (born && P->slot_persist[s]) { TELE_SET(s, 0); continue; }
      FLAGS(s) = 0; NV(s) = 0; TELE_SET(s, 0); vel_unshare(e, s);
    }
    wave_global_fence();
    if (e.lane == 0) { EQ(e)[EL(e).o_step_count] = 0; }
    wsync();
    if (born) bbox_build_all(e);   // the kept sprites' boxes (scratch, normally made when a sprite is built): the sampler tests against them
    e.restart = 0;
    for (int oi = 0; oi < P->n_ops && !e.restart; ++oi) {
      if (born && P->ops[oi].cell_sel == MOOG_CELL_NONE && !P->ops[oi].runtime && P->slot_persist[P->ops[oi].slot0]) continue;
      if (P->ops[oi].disjoint == 2) {   // constant one-sprite ops: as many as follow each other, 64 lanes at a time
        bool tables = true;
        int nb = 0;
        while (nb < 64 && oi + nb < P->n_ops && P->ops[oi + nb].disjoint == 2 &&
               !(born && P->ops[oi + nb].cell_sel == MOOG_CELL_NONE && P->slot_persist[P->ops[oi + nb].slot0])) {
          const int cs = P->ops[oi + nb].cell_sel;
          if ((cs == MOOG_CELL_WALL_RANK || cs == MOOG_CELL_OPEN_RANK) && e.cell_tab_n <= 0) tables = false;
          ++nb;
        }
        if (tables && nb >= 2) { run_static_batch(e, oi, nb); oi += nb - 1; continue; }
      }
      run_genop<DYN>(e, oi);
    }
    if (!e.restart) break;
    if (attempt >= 10000) { if (e.lane == 0) EQ(e)[EL(e).o_fault] |= MOOG_FAULT_SAMPLER_EXHAUSTED; break; }
    // the sprites built outside the initializer exist from the first pass on: a second pass keeps them as a later episode does
    if (P->born_rule > 0) born = true;
  }
  if (P->born_rule > 0) { wsync(); if (e.lane == 0) EF(e)[EL(e).o_rule + P->born_rule - 1] = 1.0; wsync(); }
  wave_global_fence();   // create_sprite wrote colours / opacity / shape ids to HBM; rules read them
  wsync();
  if (e.lane == 0) {
    for (int t = 0; t < P->n_tasks; ++t) EF(e)[EL(e).o_task + t] = DINF;
    for (int k = 0; k < 2 * (P->n_actions > 1 ? P->n_actions : 1); ++k) EF(e)[EL(e).o_action + k] = 0;
  }
  wsync();
  { PROF_T0;
  for (int r = 0; r < P->n_rules; ++r)
    if (P->rules[r].parent < 0) { rule_reset_tree(e, r); rule_step<DYN>(e, r); }
  PROF_ADD(e, 10); }
}
