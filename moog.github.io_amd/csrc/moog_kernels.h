// moog_kernels.h -- the step / reset kernels of the engine (templates and launch arguments), shared by
// the translation units that instantiate them (moog_step_inst.hip, four times; moog_reset.hip) and by
// the host side (moog_engine.hip), which only sees the launch functions declared at the bottom.
#pragma once
#include "moog_device.h"
#include "moog_draw_record.h"   // the rasteriser's input, written where the record is on chip (step_env's epilogue)

// What the launch order of the next call sorts by (moog_sched_kernel): 0.6 x this call's cycles + 0.4 x the previous value.  An
// env that was expensive lately tends to be expensive again (a pile of sprites in contact) even when one call in between was
// cheap; with the bare cycle count such an env starts late in the next launch and the launch lasts as long as it does.
// Measured (profiles/r05_step_experiments.txt 4): headline step launch 610 -> 595 us, config 5 unchanged; fmaxf(now, 0.9 x
// before) gains as much on the headline and loses 2 % on config 5.  -DMOOG_COST_DECAY=d [-DMOOG_COST_EMA] are the experiment's knobs.
#if defined(MOOG_COST_DECAY) && defined(MOOG_COST_EMA)
#define MOOG_COST_OF(now, before) (MOOG_COST_DECAY * (before) + (1.0f - MOOG_COST_DECAY) * (now))
#elif defined(MOOG_COST_DECAY)
#define MOOG_COST_OF(now, before) fmaxf((now), MOOG_COST_DECAY * (before))
#else
#define MOOG_COST_OF(now, before) moog_cost_ema((now), (before))
#endif
// (a `before` that is not a finite number -- a caller's cost array that was never initialised -- would stay in the average for good)
static __device__ __forceinline__ float moog_cost_ema(float now, float before) {
  return (before >= 0.f && before < 3.0e38f) ? 0.4f * before + 0.6f * now : now;
}

// =====================================================================================
// record staging: HBM <-> LDS, 16 bytes per lane, coalesced
// =====================================================================================
// Hot layout: the records as staged in LDS.  The colour triples (f64) and the opacity / shape-id
// words (i32) are read by the rasteriser only, so the step / reset kernels leave them in HBM
// (Env::gcol / gopa / gshape): the staged records are the HBM records with those two ranges,
// shrunk inward to 16-byte boundaries, cut out.
struct HotLayout {
  moog_layout_t L;            // offsets inside the LDS records
  int32_t f_cut0, f_cut1;     // removed range of the f64 record (doubles, multiples of 2)
  int32_t i_cut0, i_cut1;     // removed range of the i32 record (ints, multiples of 4)
};

__host__ __device__ inline HotLayout hot_layout(const moog_layout_t& G) {
  HotLayout h;
  h.L = G;
  const int S = G.S;
  // A range is cut only when it is 16-byte aligned as a whole (S even): a partially staged
  // field would be written back over the values the kernels write to HBM directly.
  const bool f_ok = (G.o_color % 2 == 0) && ((3 * S) % 2 == 0);
  h.f_cut0 = G.o_color;
  h.f_cut1 = f_ok ? G.o_color + 3 * S : G.o_color;
  const int fc = h.f_cut1 - h.f_cut0;
  // fields behind the colours (moog_layout(): inertia, maxr, action, task, rule, scale, aspect, verts)
  h.L.o_inertia -= fc; h.L.o_maxr -= fc; h.L.o_action -= fc; h.L.o_task -= fc; h.L.o_rule -= fc;
  if (G.o_hdraw >= 0) h.L.o_hdraw -= fc;
  if (G.o_rule2 >= 0) h.L.o_rule2 -= fc;
  if (G.o_scale >= 0) { h.L.o_scale -= fc; h.L.o_aspect -= fc; }
  h.L.o_verts -= fc; h.L.f64_per_env -= fc;
  // opacity, shape ids and the Portal bits are adjacent ([S] each)
  const bool i_ok = (G.o_opacity % 4 == 0) && ((3 * S) % 4 == 0) && (G.o_shape == G.o_opacity + S) &&
                    (G.o_tele == G.o_shape + S);
  h.i_cut0 = G.o_opacity;
  h.i_cut1 = i_ok ? G.o_opacity + 3 * S : G.o_opacity;
  const int ic = h.i_cut1 - h.i_cut0;
  if (G.o_valias >= 0) h.L.o_valias -= ic;
  if (G.o_fmask >= 0) h.L.o_fmask -= ic;
  h.L.o_step_count -= ic; h.L.o_reset_next -= ic; h.L.o_fault -= ic; h.L.o_rng -= ic;
  if (G.o_maze >= 0) h.L.o_maze -= ic;
  h.L.i32_per_env -= ic;
  return h;   // o_color / o_opacity / o_shape keep their values: valid in LDS when nothing was cut
}

#ifdef MOOG_SPEC_PROGRAM_INC
__device__ __forceinline__ moog_layout_t moog_spec_hot_layout_fn() {
  moog_layout_t L;
  moog_layout(&MOOG_SPEC_PROGRAM, &L);
  return hot_layout(L).L;
}
#endif

// =====================================================================================
// record staging: HBM <-> LDS, 16 bytes per lane, coalesced
// =====================================================================================
__device__ inline void load_record(const Env& e, const HotLayout& h, const moog_layout_t& G,
                                   const double* gf, const int32_t* gq) {
  const double2* src = reinterpret_cast<const double2*>(gf);
  double2* dst = reinterpret_cast<double2*>(EF(e));
  const int fa = h.f_cut0 / 2, fb = h.f_cut1 / 2;
  for (int i = e.lane; i < G.f64_per_env / 2; i += 64) {
    if (i < fa) dst[i] = src[i];
    else if (i >= fb) dst[i - (fb - fa)] = src[i];
  }
  const int4* srci = reinterpret_cast<const int4*>(gq);
  int4* dsti = reinterpret_cast<int4*>(EQ(e));
  const int ia = h.i_cut0 / 4, ib = h.i_cut1 / 4;
  for (int i = e.lane; i < G.i32_per_env / 4; i += 64) {
    if (i < ia) dsti[i] = srci[i];
    else if (i >= ib) dsti[i - (ib - ia)] = srci[i];
  }
  for (int i = e.lane; i < EL(e).S; i += 64) EVOFF(e)[i] = EP(e)->slot_voff[i];
  wsync();
}

__device__ inline void store_record(const Env& e, const HotLayout& h, const moog_layout_t& G,
                                    double* gf, int32_t* gq, int32_t* fault_flag = nullptr) {
  wsync();
  if (fault_flag && e.lane == 0) {   // rare: tell the host without waiting for it to look at every record
    const int32_t fw = EQ(e)[EL(e).o_fault];
    if (fw) __hip_atomic_fetch_or(fault_flag, fw, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
  }
  double2* dst = reinterpret_cast<double2*>(gf);
  const double2* src = reinterpret_cast<const double2*>(EF(e));
  const int fa = h.f_cut0 / 2, fb = h.f_cut1 / 2;
  int4* dsti = reinterpret_cast<int4*>(gq);
  const int4* srci = reinterpret_cast<const int4*>(EQ(e));
  const int ia = h.i_cut0 / 4, ib = h.i_cut1 / 4;
  for (int i = e.lane; i < G.f64_per_env / 2; i += 64) {
    if (i < fa) dst[i] = src[i];
    else if (i >= fb) dst[i] = src[i - (fb - fa)];
  }
  for (int i = e.lane; i < G.i32_per_env / 4; i += 64) {
    if (i < ia) dsti[i] = srci[i];
    else if (i >= ib) dsti[i] = srci[i - (ib - ia)];
  }
}

// The (force, layer a, layer b) combinations of a program in the order physics.py:96-108 visits them.
#include <vector>
inline std::vector<FOp> moog_flatten_forces(const moog_program_t* p) {
  std::vector<FOp> out;
  for (int fi = 0; fi < p->n_forces; ++fi) {
    const moog_force_t& F = p->forces[fi];
    for (int a = 0; a < F.n_a; ++a) {
      FOp op = {};
      op.fi = fi; op.kind = F.kind; op.symmetric = F.symmetric; op.i0 = F.i0; op.i1 = F.i1; op.p0 = F.p0; op.p1 = F.p1;
      op.a0 = p->layer_slot0[F.layers_a[a]]; op.a1 = op.a0 + p->layer_nslots[F.layers_a[a]];
      op.n_b = F.n_b;
      if (F.n_b == 0) { out.push_back(op); continue; }
      for (int b = 0; b < F.n_b; ++b) {
        op.b0 = p->layer_slot0[F.layers_b[b]]; op.b1 = op.b0 + p->layer_nslots[F.layers_b[b]];
        out.push_back(op);
      }
    }
  }
  return out;
}

struct KArgs {
  const moog_program_t* P;
  moog_layout_t L;       // layout of the records in HBM (the ABI's)
  HotLayout H;           // layout of the records staged in LDS
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
  int32_t dbg;
  const int32_t* perm;   // launch order (or null)
  float* cost;           // per-env cycles of this step (or null)
  int32_t* fault_flag;   // host-visible word: OR of every fault bit raised by any env (deferred fault surfacing)
  int32_t* layer_hw;     // usage of the dynamic layers (Env::layer_hw) or null
  int32_t act_f32;       // 1: `actions` holds float32 values (moog_engine_set_action_dtype)
  int32_t xstack_off;    // byte offset of the per-lane expression stacks in a wave's LDS area (Env::xstack), 0: none
  const FOp* fops;       // flattened force list (moog_flatten_forces) and its length
  int32_t n_fops;
  int32_t* watch;        // section sampling (moog_engine_read_watch): [n_envs][MOOG_WATCH_SECTIONS] sample counts, or null
  int32_t watch_off;     // byte offset of the watcher's words in the workgroup's LDS
  // reset pool (moog_engine_set_reset_pool; the kernels that carry every component only): per env one record of the NEXT
  // episode, built by a fill launch (MODE_FILL) beside the step kernels and adopted by the step kernel when the episode ends
  // pool_depth records per env, record (d, env) at index d * n_envs + env of every pool array
  int32_t* pool_state;   // 0 empty, 3 claimed (no tag yet), 1 being filled, 2 ready; null: no pool
  int32_t* pool_tag;     // the episode (high word of the draw counter) the record opens
  int32_t* pool_lock;    // [n_envs] 4 while the step kernel is opening an episode of the env (fills keep off), else 0
  int32_t pool_depth;
  double* pool_f64[2];   // [0] the record as the fill read it (the reset's inputs are validated against it), [1] the record after
  int32_t* pool_i32[2];  //     the reset; same layout and strides as the live records
  uint8_t* late_mask;    // [n_envs] late reset (see step_env): the step kernel marks the envs it could not open an episode for,
                         // the full reset kernel behind it serves and clears them; null: the step kernel resets in place
  unsigned long long* pool_stats;   // [4] episodes opened from the pool / by a reset in place / pool records rejected / adoptions that waited for a fill
  const double* live_f64;    // MODE_FILL: the live records (a.f64 / a.i32 are pool_f64[1] / pool_i32[1] then)
  const int32_t* live_i32;
  int32_t rank0;         // launch rank of workgroup 0 (0; a launch split by rank was measured in round 5, profiles/r05_step_experiments.txt)
  const uint32_t* draw_vinfo;   // vertex slot -> sprite slot | index within the sprite << 8 (the emitter's)
  RmEmit draw;           // draw.out != null: a step (MODE_STEP) also writes the env's draw record (moog_draw_record.h) for the raster launch behind it
  int32_t prio_t[3];     // wave priorities by launch rank (with `perm`: descending cost of the previous step): workgroups
                         // [0, t0) issue at priority 3, [t0, t1) at 2, [t1, t2) at 1, the rest at 0; all zero: off
};

enum { MODE_STEP = 0, MODE_PHYSICS = 1, MODE_RESET_MASK = 2, MODE_FILL = 3 };

extern __shared__ __attribute__((aligned(16))) unsigned char moog_lds[];

// lds: this wave's record area (the whole dynamic LDS of a one-wave workgroup)
__device__ inline void bind_env(Env& e, const KArgs& a, int env, unsigned char* lds = moog_lds, int lane = (int)threadIdx.x) {
  e.P = as_const_prog(a.P);
  e.fops = (PFOp)(unsigned long long)a.fops;
  e.n_fops = a.n_fops;
  e.L = a.H.L;
  const moog_layout_t& H = a.H.L;
  e.f = reinterpret_cast<double*>(lds);
  e.q = reinterpret_cast<int32_t*>(lds + (size_t)H.f64_per_env * 8);
  e.bb = reinterpret_cast<float*>(lds + (size_t)H.f64_per_env * 8 + (size_t)H.i32_per_env * 4);
  e.voff = reinterpret_cast<int32_t*>((e).bb + 8 * H.S);
  e.cand = reinterpret_cast<uint16_t*>((e).voff + ((H.S + 3) & ~3));
  e.lst = reinterpret_cast<uint8_t*>((e).cand + CAND_CAP);
  e.rowm = reinterpret_cast<unsigned long long*>((e).lst + 128);
  e.xstack = a.xstack_off > 0 ? reinterpret_cast<double*>(lds + a.xstack_off) : nullptr;
#ifdef MOOG_WATCH
  e.secw = a.watch ? reinterpret_cast<int32_t*>(__builtin_assume_aligned(lds + a.watch_off, 16)) : nullptr;
#endif
  if (a.H.f_cut1 > a.H.f_cut0) e.gcol = a.f64 + (size_t)env * a.L.f64_per_env + a.L.o_color;
  else e.gcol = EF(e) + H.o_color;
  if (a.H.i_cut1 > a.H.i_cut0) {
    e.gopa = a.i32 + (size_t)env * a.L.i32_per_env + a.L.o_opacity;
    e.gshape = a.i32 + (size_t)env * a.L.i32_per_env + a.L.o_shape;
    e.gtele = a.i32 + (size_t)env * a.L.i32_per_env + a.L.o_tele;
  } else {
    e.gopa = EQ(e) + H.o_opacity;
    e.gshape = EQ(e) + H.o_shape;
    e.gtele = EQ(e) + H.o_tele;
  }
  e.vslot = a.vslot;
  e.dbg = a.dbg;
  e.n_path = 0; e.n_resp = 0; e.n_disj = 0;
  e.layer_hw = a.layer_hw;
  e.cell_tab_n = 0; e.cell_nw = 0;
#ifdef MOOG_PROFILE
  for (int k = 0; k < 16; ++k) e.prof[k] = 0;
#endif
  e.inj = a.inj ? a.inj + (size_t)env * a.inj_n : nullptr;
  e.inj_n = a.inj_n;
  e.seed = a.seed;
  e.env_index = a.env_index0 + env;
  e.lane = lane;
}

// The env's record as the step kernel holds it (LDS; colours and opacities where bind_env says) for the draw-record emitter
struct RmSrcEnv {
  static constexpr bool kGlobalRecord = false;   // (the record is in LDS)
  const Env* e; const uint32_t* vi;   // vi: vertex slot -> sprite slot | index within the sprite << 8 (KArgs::draw_vinfo)
  __device__ __forceinline__ int flags(int s) const { return EQ(*e)[EL(*e).o_flags + s]; }
  __device__ __forceinline__ int nv(int s) const { return EQ(*e)[EL(*e).o_nverts + s]; }
  __device__ __forceinline__ int opa(int s) const { return static_cast<const int32_t*>(e->gopa)[s]; }
  __device__ __forceinline__ int voff(int s) const { return EVOFF(*e)[s]; }
  __device__ __forceinline__ int vcap(int s) const { return EP(*e)->slot_vcap[s]; }
  __device__ __forceinline__ double col(int s, int c) const { return static_cast<const double*>(e->gcol)[3 * s + c]; }
  __device__ __forceinline__ uint32_t vinfo(int idx) const { return vi[idx]; }
  __device__ __forceinline__ const double* vbase() const { return &EF(*e)[EL(*e).o_verts]; }
  __device__ __forceinline__ const double* pos(int s) const { return &EF(*e)[EL(*e).o_pos + 2 * s]; }
};
// Beside store_record: the frame the rasteriser is about to draw, from the record in LDS (what the reset path wrote straight
// to HBM -- colours, opacities -- is read back from there: same wavefront, stores and loads in order behind wsync).
__device__ __forceinline__ void emit_draw_record(const Env& e, const KArgs& a, int env) {
  if (!a.draw.out) return;
  const long long t_emit = (a.dbg & 256) ? clock64() : 0;   // (profiling aid: the emitter's cycles instead of the step type, tools/emit_cycles.py)
  wsync();
  RmSrcEnv src;
  src.e = &e; src.vi = a.draw_vinfo;
  // scratch: one copy per sprite -- six words per slot -- in the bounding volumes (eight per slot; nothing reads them after the
  // last sub-step); the nine copies of a torus -- 46 words per slot -- in the candidate list / list / row mask area behind the
  // vertex offsets (CAND_CAP * 2 + 128 + 64 * 8 bytes: the engine checks that they fit, step_emits_draw)
  RmEmitScratch sc;
  rm_emit_scratch(a.draw.ncopy > 1 ? reinterpret_cast<int32_t*>(ECAND(e)) : reinterpret_cast<int32_t*>(EBB(e)), a.draw.slots, a.draw.ncopy, &sc);
  long long clk[5];
  rm_emit(a.draw, src, env, e.lane, sc, EL(e).TOTV, (a.dbg & 256) ? clk : nullptr);
  if ((a.dbg & 256) && e.lane == 0 && a.step_type) {   // (the emitter's cycles, and phase by phase: prefix | slots | vertex slots | items)
    a.step_type[env] = (int32_t)(clock64() - t_emit);
    if (a.discount) a.discount[env] = (double)(clk[1] - clk[0]) + 65536.0 * (double)(clk[2] - clk[1]);
    if (a.reward) a.reward[env] = (double)(clk[3] - clk[2]) + 65536.0 * (double)(clk[4] - clk[3]);
  }
}

// =====================================================================================
// reset pool (the kernels that carry every component): DESIGN 3.2
// =====================================================================================
// A reset's result is a function of (seed, env, episode) -- every episode draws from its own segment of the stream,
// env_reset -- and of what outlives resets: the sprites built outside the initializer (program.slot_persist, kept once
// born_rule's slot is set) and the MOOG_RULE_STATE_SLOT scalars.  So episode E + 1 can be built while E runs: a fill
// launch copies the live record, resets the copy and raises the env's flag; when E ends the step kernel checks that the
// inputs the fill read still equal the live ones, bit for bit, and takes the record over instead of running the reset
// (otherwise it resets in place, as without a pool).  Programs whose reset reads a state scalar (MOOG_CELL_PSTATE)
// are not eligible (moog_engine_set_reset_pool refuses them).
#if MOOG_WITH_MAZE
#define MOOG_WITH_POOL 1
#elif defined(MOOG_STEP_DYN)
#define MOOG_WITH_POOL (MOOG_STEP_DYN >= 1)   // (the step kernels with the expression evaluator step late-reset programs)
#else
#define MOOG_WITH_POOL 0
#endif
#if MOOG_WITH_POOL
// the per-slot words of two records (HBM layout) agree, for every slot the reset keeps
__device__ inline bool pool_inputs_equal(const Env& e, const moog_layout_t& G, const double* af, const int32_t* aq,
                                         const double* bf, const int32_t* bq) {
  PProg P = EP(e);
  bool same = true;
  const bool born = P->born_rule > 0;
  if (born) {
    same = __double_as_longlong(af[G.o_rule + P->born_rule - 1]) == __double_as_longlong(bf[G.o_rule + P->born_rule - 1]);
    for (int s = e.lane; s < G.S; s += 64) {
      if (!P->slot_persist[s]) continue;
      #define PEQ_F(o, k) (__double_as_longlong(af[(o) + (k)]) == __double_as_longlong(bf[(o) + (k)]))
      #define PEQ_I(o) (aq[(o) + s] == bq[(o) + s])
      bool q = PEQ_F(G.o_pos, 2 * s) && PEQ_F(G.o_pos, 2 * s + 1) && PEQ_F(G.o_vel, 2 * s) && PEQ_F(G.o_vel, 2 * s + 1) &&
               PEQ_F(G.o_angle, s) && PEQ_F(G.o_angvel, s) && PEQ_F(G.o_mass, s) && PEQ_F(G.o_color, 3 * s) &&
               PEQ_F(G.o_color, 3 * s + 1) && PEQ_F(G.o_color, 3 * s + 2) && PEQ_F(G.o_inertia, 2 * s) &&
               PEQ_F(G.o_inertia, 2 * s + 1) && PEQ_F(G.o_maxr, s);
      if (G.o_scale >= 0) q = q && PEQ_F(G.o_scale, s) && PEQ_F(G.o_aspect, s);
      q = q && PEQ_I(G.o_flags) && PEQ_I(G.o_nverts) && PEQ_I(G.o_opacity) && PEQ_I(G.o_shape);
      if (G.o_valias >= 0) q = q && PEQ_I(G.o_valias);
      if (G.o_fmask >= 0) q = q && PEQ_I(G.o_fmask);
      int nv = aq[G.o_nverts + s];
      if (nv > P->slot_vcap[s]) nv = P->slot_vcap[s];
      const int v0 = G.o_verts + 2 * P->slot_voff[s];
      for (int k = 0; k < 2 * nv; ++k) q = q && PEQ_F(v0, k);
      #undef PEQ_F
      #undef PEQ_I
      same = same && q;
    }
  }
  return __all(same);
}

__device__ inline void pool_copy_record(const moog_layout_t& G, int lane, const double* sf, const int32_t* sq, double* df, int32_t* dq) {
  const double2* a = reinterpret_cast<const double2*>(sf);
  double2* b = reinterpret_cast<double2*>(df);
  for (int i = lane; i < G.f64_per_env / 2; i += 64) b[i] = a[i];
  const int4* c = reinterpret_cast<const int4*>(sq);
  int4* d = reinterpret_cast<int4*>(dq);
  for (int i = lane; i < G.i32_per_env / 4; i += 64) d[i] = c[i];
}

// Step kernel, an env whose episode has ended, its live record staged in LDS: takes over the pool record that opens the
// episode the live record expects, when there is one (waiting for a fill of it that is under way: its wave is resident and
// shorter than a reset from scratch) and it was built from the inputs the live record holds now.  true: the record in LDS
// (and the fields that live in HBM only) are the new episode's; false: the caller resets in place.
// The env's lock is taken first (*held) and handed back by pool_release once the new episode's record is in memory: a fill
// that copied the record in between would read the episode that has just ended.
__device__ inline bool pool_adopt(Env& e, const KArgs& a, int env, double* gf, int32_t* gq, int* held) {
  *held = 0;
  if (!a.pool_state || e.inj) return false;
  *held = 1;
  const unsigned episode = (unsigned)EQ(e)[EL(e).o_rng + 1] + 1u;
  int pick = -1, waited = 0;
  if (e.lane == 0) {
    __hip_atomic_store(&a.pool_lock[env], 4, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    for (int d = 0; d < a.pool_depth; ++d) {
      int32_t* sp = &a.pool_state[(size_t)d * a.n_envs + env];
      int st = __hip_atomic_load(sp, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
      for (int spins = 0; st == 3 && spins < 100000; ++spins) {   // (claimed this instant: its tag is a few stores away)
        __builtin_amdgcn_s_sleep(2);
        st = __hip_atomic_load(sp, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
      }
      if (st != 1 && st != 2) continue;
      const unsigned tag = (unsigned)__hip_atomic_load(&a.pool_tag[(size_t)d * a.n_envs + env], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
      if (tag == episode) { if (pick < 0) pick = d; }
      else if (st == 2 && (int)(tag - episode) < 0)   // an episode that has passed (the host reset the env, or a late fill)
        __hip_atomic_store(sp, 0, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    }
    if (pick >= 0) {
      int32_t* sp = &a.pool_state[(size_t)pick * a.n_envs + env];
      int st = __hip_atomic_load(sp, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
      for (int spins = 0; st == 1 && spins < 400000; ++spins) {   // (a bound, not a protocol: a few tenths of a second)
        waited = 1;
        __builtin_amdgcn_s_sleep(32);
        st = __hip_atomic_load(sp, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
      }
      if (st != 2) pick = -1;   // (it will raise its flag later, for an episode that will have passed by then)
    }
  }
  pick = __shfl(pick, 0);
  waited = __shfl(waited, 0);
  if (pick < 0) {
    if (e.lane == 0) atomicAdd(&a.pool_stats[1], 1ull);
    return false;
  }
  __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "agent");
  const size_t rec = (size_t)pick * a.n_envs + env;
  const double* pre_f = a.pool_f64[0] + rec * a.L.f64_per_env; const int32_t* pre_q = a.pool_i32[0] + rec * a.L.i32_per_env;
  const double* post_f = a.pool_f64[1] + rec * a.L.f64_per_env; const int32_t* post_q = a.pool_i32[1] + rec * a.L.i32_per_env;
  const bool ok = pool_inputs_equal(e, a.L, pre_f, pre_q, gf, gq);
  wsync();
  if (!ok) {   // stale: something changed a kept sprite after the fill read it
    if (e.lane == 0) {
      __hip_atomic_store(&a.pool_state[rec], 0, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
      atomicAdd(&a.pool_stats[1], 1ull); atomicAdd(&a.pool_stats[2], 1ull);
    }
    return false;
  }
  // what outlives a reset: sticky fault bits, the state slots (never reset; a reset that reads one is not eligible)
  const int32_t fault = EQ(e)[EL(e).o_fault];
  const int r = e.lane < a.L.R ? e.lane : 0;
  const double keep = EF(e)[EL(e).o_rule + r];
  const double keep2 = EL(e).o_rule2 >= 0 ? EF(e)[EL(e).o_rule2 + r] : 0.0;
  wsync();
  load_record(e, a.H, a.L, post_f, post_q);
  if (a.H.f_cut1 > a.H.f_cut0)
    for (int i = a.H.f_cut0 + e.lane; i < a.H.f_cut1; i += 64) gf[i] = post_f[i];
  if (a.H.i_cut1 > a.H.i_cut0)
    for (int i = a.H.i_cut0 + e.lane; i < a.H.i_cut1; i += 64) gq[i] = post_q[i];
  if (e.lane < a.L.R && EP(e)->rules[e.lane].kind == MOOG_RULE_STATE_SLOT) {
    EF(e)[EL(e).o_rule + e.lane] = keep;
    if (EL(e).o_rule2 >= 0) EF(e)[EL(e).o_rule2 + e.lane] = keep2;
  }
  wave_global_fence();
  wsync();
  if (e.lane == 0) {
    EQ(e)[EL(e).o_fault] |= fault;
    __hip_atomic_store(&a.pool_state[rec], 0, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);   // (the lock keeps the fills off the env)
    atomicAdd(&a.pool_stats[0], 1ull);
    if (waited) atomicAdd(&a.pool_stats[3], 1ull);
  }
  wsync();
  return true;
}

// the new episode's record is stored: fills may read the env again (from memory: this XCD's L2 first)
__device__ inline void pool_release(const KArgs& a, int env, int lane) {
  __builtin_amdgcn_fence(__ATOMIC_RELEASE, "agent");
  if (lane == 0) __hip_atomic_store(&a.pool_lock[env], 0, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
}
#else
__device__ __forceinline__ bool pool_adopt(Env&, const KArgs&, int, double*, int32_t*, int* held) { *held = 0; return false; }
__device__ __forceinline__ void pool_release(const KArgs&, int, int) {}
#endif

// reset_next word: 0 = running, 1 = reset on the next call (environment.py:100-101): the step kernel
// resets such an env instead of stepping it (its action is ignored, the timestep is FIRST).
#ifdef MOOG_DEFINE_RESET_KERNELS
template <int VARIANT>   // VARIANT only names the instantiation (one per translation unit)
__global__ __launch_bounds__(64) void moog_reset_kernel(KArgs a) {
  int env = blockIdx.x;
#if MOOG_WITH_MAZE
  const int pool_d = a.mode == MODE_FILL ? env / a.n_envs : 0;   // reset pool: one workgroup per (record, env)
  if (a.mode == MODE_FILL) env -= pool_d * a.n_envs;
  const size_t rec = (size_t)pool_d * a.n_envs + env;
#else
  const size_t rec = (size_t)env;
#endif
  if (env >= a.n_envs) return;
  int32_t* gq = a.i32 + rec * a.L.i32_per_env;
  if (a.mask != nullptr && a.mask[env] == 0) return;
  if (a.late_mask != nullptr && a.mode != MODE_FILL && a.late_mask[env] == 0) return;   // late reset: the envs the step kernel marked
  double* gf = a.f64 + rec * a.L.f64_per_env;
#if MOOG_WITH_MAZE
  if (a.mode == MODE_FILL) {   // a.f64 / a.i32 are the pool's records: claim record pool_d of the env, copy the live record, reset the copy
    if (pool_d >= a.pool_depth) return;
    const int32_t* lq = a.live_i32 + (size_t)env * a.L.i32_per_env;
    int won = 0;
    if (threadIdx.x == 0 && __hip_atomic_load(&a.pool_lock[env], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) == 0) {
      bool lower_busy = true;   // the records of an env fill in order: a later one waits for the earlier ones to have been claimed
      for (int d = 0; d < pool_d; ++d)
        lower_busy = lower_busy && __hip_atomic_load(&a.pool_state[(size_t)d * a.n_envs + env], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) != 0;
      int expect = 0;
      won = lower_busy && __hip_atomic_compare_exchange_strong(&a.pool_state[rec], &expect, 3, __ATOMIC_RELAXED, __ATOMIC_RELAXED,
                                                               __HIP_MEMORY_SCOPE_AGENT) ? 1 : 0;
    }
    if (!__shfl(won, 0)) return;
    __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "agent");
    const double* lf = a.live_f64 + (size_t)env * a.L.f64_per_env;
    // (the step kernels may be storing this record right now: what the reset reads of it is validated when the record is
    //  adopted, pool_adopt; the rest it overwrites)
    pool_copy_record(a.L, (int)threadIdx.x, lf, lq, a.pool_f64[0] + rec * a.L.f64_per_env, a.pool_i32[0] + rec * a.L.i32_per_env);
    pool_copy_record(a.L, (int)threadIdx.x, a.pool_f64[0] + rec * a.L.f64_per_env, a.pool_i32[0] + rec * a.L.i32_per_env, gf, gq);
    __builtin_amdgcn_fence(__ATOMIC_SEQ_CST, "workgroup");
    __builtin_amdgcn_wave_barrier();
    int go = 0;
    if (threadIdx.x == 0) {
      // the episode this record opens: the next one the live record expects that no other record of the env holds or is being
      // filled for (a later record waits for an earlier one's tag; an earlier one does not wait for a later one)
      const unsigned ep_live = (unsigned)gq[a.L.o_rng + 1];
      unsigned target = ep_live + 1u;
      for (int round = 0; round < a.pool_depth; ++round)
        for (int d = 0; d < a.pool_depth; ++d) {
          if (d == pool_d) continue;
          const size_t o = (size_t)d * a.n_envs + env;
          int st = __hip_atomic_load(&a.pool_state[o], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
          for (int spins = 0; st == 3 && d < pool_d && spins < 100000; ++spins) {
            __builtin_amdgcn_s_sleep(2);
            st = __hip_atomic_load(&a.pool_state[o], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
          }
          if ((st == 1 || st == 2) &&
              (unsigned)__hip_atomic_load(&a.pool_tag[o], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) == target) ++target;
        }
      __hip_atomic_store(&a.pool_tag[rec], (int32_t)target, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
      __hip_atomic_store(&a.pool_state[rec], 1, __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_AGENT);
      // the copy is of a record that was not being replaced: the lock is free and the live record still says the same episode
      const bool stable = __hip_atomic_load(&a.pool_lock[env], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) == 0 &&
                          (unsigned)__hip_atomic_load(&lq[a.L.o_rng + 1], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) == ep_live;
      if (stable) { gq[a.L.o_rng + 1] = (int32_t)(target - 1u); go = 1; }   // (env_reset opens the next segment)
      else __hip_atomic_store(&a.pool_state[rec], 0, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    }
    if (!__shfl(go, 0)) return;
    __builtin_amdgcn_fence(__ATOMIC_SEQ_CST, "workgroup");
    __builtin_amdgcn_wave_barrier();
  }
#endif
  Env e;
  bind_env(e, a, env);
#if MOOG_WITH_MAZE
  if (a.mode == MODE_FILL) {   // (the fields the kernels keep in HBM: this record's, not record 0's)
    if (a.H.f_cut1 > a.H.f_cut0) e.gcol = gf + a.L.o_color;
    if (a.H.i_cut1 > a.H.i_cut0) { e.gopa = gq + a.L.o_opacity; e.gshape = gq + a.L.o_shape; e.gtele = gq + a.L.o_tele; }
  }
#endif
  load_record(e, a.H, a.L, gf, gq);
  if (e.inj && e.lane == 0) EQ(e)[EL(e).o_rng + 2] = 0;
  wsync();
  env_reset<true>(e);
  wsync();
#if MOOG_WITH_MAZE
  if (a.mode == MODE_FILL) {
    if (e.lane == 0) EQ(e)[EL(e).o_reset_next] = 0;
    store_record(e, a.H, a.L, gf, gq, nullptr);
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "agent");   // the record (and what the reset wrote straight to HBM) before the flag
    if (e.lane == 0) __hip_atomic_store(&a.pool_state[rec], 2, __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_AGENT);
    return;
  }
#endif
  if (e.lane == 0) {
    EQ(e)[EL(e).o_reset_next] = 0;
    if (a.reward) a.reward[env] = __builtin_nan("");
    if (a.discount) a.discount[env] = __builtin_nan("");
    if (a.step_type) a.step_type[env] = 0;
  }
  emit_draw_record(e, a, env);   // (a late reset behind a step launch that wrote draw records: this env's is the new episode's)
  store_record(e, a.H, a.L, gf, gq, a.fault_flag);
#if MOOG_WITH_MAZE
  if (a.late_mask != nullptr) {   // late reset: served; the step kernel took the pool's lock for this env when the episode ended
    if (a.pool_state) pool_release(a, env, e.lane);
    if (e.lane == 0) a.late_mask[env] = 0;
  }
#endif
}

#if !MOOG_RESET_FULL   // (the sort kernel lives in the first of the two reset translation units)
// Launch order for the next step: envs in (approximately) descending order of the cycles they
// took in this step (longest-processing-time first).  One workgroup of SCHED_THREADS threads: 1024-bin
// counting sort on cost / max(cost).  Runs on a side stream concurrently with the rasteriser.
// The order inside a bin is arbitrary -- the schedule never changes a result.
// Round 6: a thread keeps its envs' bins in registers (one pass over the costs instead of three), the prefix
// sum over the bins is a DPP scan per wavefront + one over the wave totals (three barriers instead of twenty-two), a cost that
// is not a finite number (a caller's uninitialised array) counts as zero, and the workgroup is four wavefronts instead of sixteen
// (it runs beside the rasteriser, whose workgroups leave a CU no room for sixteen until they drain).
#define SCHED_BINS 1024
#define SCHED_THREADS 256   // four wavefronts: a workgroup that finds room on a CU beside the rasteriser's (sixteen waited for a CU to drain)
#define SCHED_KEEP 16       // bins a thread keeps in registers: batches of up to 4096 envs make one pass over the costs
__device__ __forceinline__ int sched_wave_scan(int v) {   // inclusive, within a wavefront (the rasteriser's rm_wave_scan)
  v += __builtin_amdgcn_update_dpp(0, v, 0x111, 0xf, 0xf, false);   // row_shr:1
  v += __builtin_amdgcn_update_dpp(0, v, 0x112, 0xf, 0xf, false);   // row_shr:2
  v += __builtin_amdgcn_update_dpp(0, v, 0x114, 0xf, 0xf, false);   // row_shr:4
  v += __builtin_amdgcn_update_dpp(0, v, 0x118, 0xf, 0xf, false);   // row_shr:8
  v += __builtin_amdgcn_update_dpp(0, v, 0x142, 0xa, 0xf, false);   // row_bcast:15 -> rows 1, 3
  v += __builtin_amdgcn_update_dpp(0, v, 0x143, 0xc, 0xf, false);   // row_bcast:31 -> rows 2, 3
  return v;
}
__device__ __forceinline__ float sched_cost(const float* cost, int i) { const float c = cost[i]; return (c >= 0.f && c < 3.0e38f) ? c : 0.f; }
__global__ __launch_bounds__(SCHED_THREADS) void moog_sched_kernel(const float* cost, int32_t* perm, int n,
                                                                    const int32_t* reset_next, int stride) {
  __shared__ int hist[SCHED_BINS];
  __shared__ float red[SCHED_THREADS / 64];
  __shared__ int wsum[SCHED_THREADS / 64];
  const int t = threadIdx.x;
  float c[SCHED_KEEP];
  int rn[SCHED_KEEP];
  float m = 0.f;
#pragma unroll
  for (int q = 0; q < SCHED_KEEP; ++q) {
    const int i = t + SCHED_THREADS * q;
    c[q] = i < n ? sched_cost(cost, i) : 0.f;
    rn[q] = i < n ? reset_next[(size_t)i * stride] : 0;
    m = fmaxf(m, c[q]);
  }
  for (int i = t + SCHED_THREADS * SCHED_KEEP; i < n; i += SCHED_THREADS) m = fmaxf(m, sched_cost(cost, i));
  for (int o = 32; o > 0; o >>= 1) m = fmaxf(m, __shfl_xor(m, o));
  if ((t & 63) == 0) red[t >> 6] = m;
  for (int b = t; b < SCHED_BINS; b += SCHED_THREADS) hist[b] = 0;
  __syncthreads();
  m = red[0];
  for (int w = 1; w < SCHED_THREADS / 64; ++w) m = fmaxf(m, red[w]);
  const float scale = m > 0.f ? (float)(SCHED_BINS - 1) / m : 0.f;
  // bin 0 = most expensive
  // (an env whose episode just ended is reset inside the next step kernel: the sampler's rejection loop is
  //  as long as the heaviest step, so it starts first)
  auto bin_of = [&](float cv, int r) {
    int b = SCHED_BINS - 1 - (int)(cv * scale);
    b = b < 0 ? 0 : (b > SCHED_BINS - 1 ? SCHED_BINS - 1 : b);
    return r == 1 ? 0 : b;
  };
  int bq[SCHED_KEEP];
#pragma unroll
  for (int q = 0; q < SCHED_KEEP; ++q) {
    bq[q] = bin_of(c[q], rn[q]);
    if (t + SCHED_THREADS * q < n) atomicAdd(&hist[bq[q]], 1);
  }
  for (int i = t + SCHED_THREADS * SCHED_KEEP; i < n; i += SCHED_THREADS) atomicAdd(&hist[bin_of(sched_cost(cost, i), reset_next[(size_t)i * stride])], 1);
  __syncthreads();
  // exclusive prefix sum over the 1024 bins: a thread owns SCHED_BINS / SCHED_THREADS consecutive bins
  constexpr int PER = SCHED_BINS / SCHED_THREADS;
  int own[PER], v = 0;
#pragma unroll
  for (int k = 0; k < PER; ++k) { own[k] = hist[PER * t + k]; v += own[k]; }
  const int inc = sched_wave_scan(v);
  if ((t & 63) == 63) wsum[t >> 6] = inc;
  __syncthreads();
  int before = inc - v;
  for (int w = 0; w < (t >> 6); ++w) before += wsum[w];
#pragma unroll
  for (int k = 0; k < PER; ++k) { hist[PER * t + k] = before; before += own[k]; }
  __syncthreads();
#pragma unroll
  for (int q = 0; q < SCHED_KEEP; ++q) {
    const int i = t + SCHED_THREADS * q;
    if (i < n) perm[atomicAdd(&hist[bq[q]], 1)] = i;
  }
  for (int i = t + SCHED_THREADS * SCHED_KEEP; i < n; i += SCHED_THREADS)
    perm[atomicAdd(&hist[bin_of(sched_cost(cost, i), reset_next[(size_t)i * stride])], 1)] = i;
}
#endif

#endif  // MOOG_DEFINE_RESET_KERNELS

// DYN = the program has rules that create / move / filter sprites at run time (CreateSprites,
// ChangeLayer, VanishByFilter): that variant carries the reset path's sampler; the plain one
// is what the benchmark configs run.
// WPS = waves per SIMD the register allocation is sized for: 4 (128 VGPRs, some scratch) keeps sixteen
// envs per CU in flight, which is what programs with small state records want; 3 (168 VGPRs, no
// scratch in the hot loops) is faster once LDS holds fewer than fifteen records per CU anyway.
// One env's step (or auto-reset), one wavefront.
// Returns true when the call reset the env (the reset path writes colours / opacities / shapes straight to HBM).
template <bool DYN>
__device__ __forceinline__ void step_env(const KArgs& a, const int env, unsigned char* lds, const int lane) {
  const long long t_sched = a.cost ? clock64() : 0;
  int32_t* gq = a.i32 + (size_t)env * a.L.i32_per_env;
  Env e;
  bind_env(e, a, env, lds, lane);
  const long long t_begin = (a.dbg & 128) ? clock64() : 0;
  double* gf = a.f64 + (size_t)env * a.L.f64_per_env;
  { PROF_T0; load_record(e, a.H, a.L, gf, gq); PROF_ADD(e, 9); }
  if (e.inj && e.lane == 0) EQ(e)[EL(e).o_rng + 2] = 0;
  wsync();
#ifndef MOOG_NO_FUSED_RESET   // (A/B builds only: the step path without the sampler compiled in)
  if (a.mode == MODE_STEP && uni(EQ(e)[EL(e).o_reset_next]) == 1) {   // auto-reset (environment.py:100-101)
    int held = 0;
    if (!(DYN && pool_adopt(e, a, env, gf, gq, &held))) {   // (the next episode may be waiting in the reset pool)
      if (DYN && a.late_mask) {
        // Late reset: this kernel does not carry the program's initializer (it is one of the kernels that leave the rare
        // components out, chosen because the program needs them only to build an episode: the same program on the kernel
        // that carries everything steps 1.4 - 2.7 times slower, profiles/r04_variant_tax.txt).  The env is marked and left
        // as it is; the full reset kernel, launched behind this one, opens its episode and releases the pool's lock.
        if (e.lane == 0) a.late_mask[env] = 1;
        return;
      }
      env_reset<DYN>(e);
    }
    wsync();
    if (e.lane == 0) {
      EQ(e)[EL(e).o_reset_next] = 0;
      if (a.reward) a.reward[env] = __builtin_nan("");
      if (a.discount) a.discount[env] = __builtin_nan("");
      if (a.step_type) a.step_type[env] = 0;
#ifdef MOOG_PROFILE   // tools/reset_profile.py: cycles of the reset and of one of its sections instead of NaN
      if ((a.dbg & 128) && a.discount) {
        a.discount[env] = (double)(clock64() - t_begin);
        if (a.reward) a.reward[env] = (a.dbg >> 8) ? (double)e.prof[((a.dbg >> 8) & 31) - 1] : 0.0;
      }
#endif
    }
    emit_draw_record(e, a, env);   // (before the record's stores: a wave waits once for its stores to drain, at its end)
    store_record(e, a.H, a.L, gf, gq, a.fault_flag);
    if (DYN && held) pool_release(a, env, e.lane);
    if (a.cost && e.lane == 0) a.cost[env] = MOOG_COST_OF((float)(clock64() - t_sched), a.cost[env]);
    return;
  }
#endif
  { PROF_T0; bbox_build_all(e); PROF_ADD(e, 9); }
  PProg P = as_const_prog(a.P);
  const int K = uni(P->updates_per_env_step);
  if (a.mode == MODE_PHYSICS) {
    for (int k = 0; k < K; ++k) apply_physics<DYN>(e);
    store_record(e, a.H, a.L, gf, gq, a.fault_flag);
    return;
  }
  {
  PROF_T0;
  SEC(e, SEC_RULES);
  // environment.py:98-126
  const int n_rules = uni(P->n_rules);
  for (int r = 0; r < n_rules; ++r)
    if (P->rules[r].parent < 0) rule_step<DYN>(e, r);
  const bool af32 = a.act_f32 != 0;
  if (uni(P->n_actions) > 1) {   // composite.py:61-62: every sub-space, in keyword order
    const int na = uni(P->n_actions);
    for (int k = 0; k < na; ++k) {
      double x, y;
      if (af32) { const float* act = reinterpret_cast<const float*>(a.actions) + (size_t)2 * na * env; x = act[2 * k]; y = act[2 * k + 1]; }
      else { const double* act = reinterpret_cast<const double*>(a.actions) + (size_t)2 * na * env; x = act[2 * k]; y = act[2 * k + 1]; }
      action_step(e, k, x, y, (int)x, af32);
    }
  } else {
    double ax = 0, ay = 0;
    int ga = 4;
    if (P->action.kind == MOOG_ACTION_GRID) ga = reinterpret_cast<const int32_t*>(a.actions)[env];
    else if (af32) {
      ax = reinterpret_cast<const float*>(a.actions)[2 * env];
      ay = reinterpret_cast<const float*>(a.actions)[2 * env + 1];
    } else {
      ax = reinterpret_cast<const double*>(a.actions)[2 * env];
      ay = reinterpret_cast<const double*>(a.actions)[2 * env + 1];
    }
    action_step(e, 0, ax, ay, ga, af32);
  }
  PROF_ADD(e, 10);
  }
  { PROF_T0; for (int k = 0; k < K; ++k) apply_physics<DYN>(e); PROF_ADD(e, 6); }
  SEC(e, SEC_TASK);
  int sc = EQ(e)[EL(e).o_step_count] + 1;
  wsync();
  if (e.lane == 0) EQ(e)[EL(e).o_step_count] = sc;
  wsync();
  int sr = 0;
  double r;
  { PROF_T0; r = task_reward<DYN>(e, sc, &sr); PROF_ADD(e, 11); }
  wsync();
  if (e.lane == 0) {
    if (sr) EQ(e)[EL(e).o_reset_next] = 1;
    if (a.reward) a.reward[env] = r;
    if (a.discount) a.discount[env] = sr ? 0.0 : 1.0;
    if (a.step_type) a.step_type[env] = sr ? 2 : 1;
  }
  SEC(e, SEC_STORE);
  emit_draw_record(e, a, env);   // (before the record's stores: a wave waits once for its stores to drain, at its end)
  store_record(e, a.H, a.L, gf, gq, a.fault_flag);
  if (a.cost && e.lane == 0) a.cost[env] = MOOG_COST_OF((float)(clock64() - t_sched), a.cost[env]);
  if ((a.dbg & 128) && e.lane == 0 && a.discount) {   // profiling aid: cycles and work counters instead of outputs
    a.discount[env] = (double)(clock64() - t_begin);
    if (a.reward) a.reward[env] = (double)(e.n_path + 100000 * e.n_resp) + 1e10 * (double)e.n_disj;
#ifdef MOOG_PROFILE
    if (a.reward && (a.dbg >> 8)) a.reward[env] = (double)e.prof[((a.dbg >> 8) & 31) - 1];
#endif
  }
}

template <bool DYN, int WPS, int VARIANT>   // VARIANT only names the instantiation (one per translation unit)
// Every device function must end up inlined into this kernel: the env's descriptor (`Env`: pointers and the layout's offsets)
// has to live in registers.  One function left out of line takes it by reference through scratch memory, ~1000 cycles per
// access: round 4 measured this kernel at 1270 instead of 800 us the day the inliner's cost model left apply_physics out
// (hence the __forceinline__ there; tests/test_host.py::test_step_kernels_have_no_calls checks the built objects).
#ifdef MOOG_WATCH
#define MOOG_STEP_THREADS 128   // the env's wavefront + its watcher
#else
#define MOOG_STEP_THREADS 64
#endif
__global__ __launch_bounds__(MOOG_STEP_THREADS, WPS) void moog_step_kernel(KArgs a) {
  int env = (int)blockIdx.x + a.rank0;
  if (env >= a.n_envs) return;
#ifdef MOOG_WATCH
  if (a.watch) {
    int32_t* w = reinterpret_cast<int32_t*>(__builtin_assume_aligned(moog_lds + a.watch_off, 16));   // w[0]: the section announced, w[1]: 1 when the env is done
    if (threadIdx.x == 0) { w[0] = 0; w[1] = 0; }
    __syncthreads();
    if (threadIdx.x >= 64) {   // the watcher: one sample every ~256 cycles, histogram in registers of lane 64 + section
      const int lane = (int)threadIdx.x - 64;
      int count = 0;
      while (__hip_atomic_load(&w[1], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP) == 0) {
        const int sec = __hip_atomic_load(&w[0], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
        if (lane == (sec & (MOOG_WATCH_SECTIONS - 1))) ++count;
        __builtin_amdgcn_s_sleep(3);
      }
      const int real = a.perm ? a.perm[env] : env;
      if (lane < MOOG_WATCH_SECTIONS) atomicAdd(&a.watch[(size_t)real * MOOG_WATCH_SECTIONS + lane], count);
      return;
    }
  } else if (threadIdx.x >= 64) return;
#endif
  // The launch lasts as long as its slowest env, and a wavefront that shares its SIMD with two others issues an
  // instruction every ~9 cycles instead of every ~5: the envs that were expensive in the previous step (they come first in
  // the launch order) get the SIMD's issue slots ahead of their neighbours.  A scheduling hint: no result depends on it.
  if (a.perm && a.prio_t[2] > 0) {
    const int b = env;
    if (b < a.prio_t[0]) __builtin_amdgcn_s_setprio(3);
    else if (b < a.prio_t[1]) __builtin_amdgcn_s_setprio(2);
    else if (b < a.prio_t[2]) __builtin_amdgcn_s_setprio(1);
  }
  if (a.perm) env = a.perm[env];
#ifdef MOOG_SPEC_PROGRAM_INC
  {   // a program-specialised build (moog_step_spec.hip): the layout is a compile-time constant like the program itself
    moog_layout_t L;
    moog_layout(&MOOG_SPEC_PROGRAM, &L);
    a.L = L;
    a.H = hot_layout(L);
  }
#endif
  step_env<DYN>(a, env, moog_lds, (int)threadIdx.x);
#ifdef MOOG_WATCH
  if (a.watch && threadIdx.x == 0) { int32_t* w = reinterpret_cast<int32_t*>(__builtin_assume_aligned(moog_lds + a.watch_off, 16)); __hip_atomic_store(w + 1, 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP); }
#endif
}


// ---- launch functions (one translation unit each, so that they compile in parallel) ------------------
// variant = (dynamic rules ? 2 : 0) + (waves per SIMD == 4 ? 1 : 0)
typedef void (*moog_step_launch_fn)(int n_envs, size_t lds, hipStream_t s, const KArgs& a);
void moog_launch_step_f2(int n_envs, size_t lds, hipStream_t s, const KArgs& a);
void moog_launch_step_f3(int n_envs, size_t lds, hipStream_t s, const KArgs& a);
void moog_launch_step_f4(int n_envs, size_t lds, hipStream_t s, const KArgs& a);
void moog_launch_step_t3(int n_envs, size_t lds, hipStream_t s, const KArgs& a);
void moog_launch_step_t4(int n_envs, size_t lds, hipStream_t s, const KArgs& a);
void moog_launch_step_m3(int n_envs, size_t lds, hipStream_t s, const KArgs& a);
void moog_launch_step_m4(int n_envs, size_t lds, hipStream_t s, const KArgs& a);
int moog_configure_step_f2(size_t lds);
int moog_configure_step_f3(size_t lds);
int moog_configure_step_f4(size_t lds);
int moog_configure_step_t3(size_t lds);
int moog_configure_step_t4(size_t lds);
int moog_configure_step_m3(size_t lds);
int moog_configure_step_m4(size_t lds);
void moog_launch_reset_plain(int n_envs, size_t lds, hipStream_t s, const KArgs& a);
void moog_launch_reset_full(int n_envs, size_t lds, hipStream_t s, const KArgs& a);
int moog_configure_reset_plain(size_t lds);
int moog_configure_reset_full(size_t lds);
void moog_launch_sched(hipStream_t s, const float* cost, int32_t* perm, int n, const int32_t* reset_next, int stride);
