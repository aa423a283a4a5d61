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
  h.L.i32_per_env -= ic;
  return h;   // o_color / o_opacity / o_shape keep their values: valid in LDS when nothing was cut
}

// =====================================================================================
// record staging: HBM <-> LDS, 16 bytes per lane, coalesced
// =====================================================================================
__device__ inline void load_record(const Env& e, const HotLayout& h, const moog_layout_t& G,
                                   const double* gf, const int32_t* gq) {
  const double2* src = reinterpret_cast<const double2*>(gf);
  double2* dst = reinterpret_cast<double2*>(e.f);
  const int fa = h.f_cut0 / 2, fb = h.f_cut1 / 2;
  for (int i = e.lane; i < G.f64_per_env / 2; i += 64) {
    if (i < fa) dst[i] = src[i];
    else if (i >= fb) dst[i - (fb - fa)] = src[i];
  }
  const int4* srci = reinterpret_cast<const int4*>(gq);
  int4* dsti = reinterpret_cast<int4*>(e.q);
  const int ia = h.i_cut0 / 4, ib = h.i_cut1 / 4;
  for (int i = e.lane; i < G.i32_per_env / 4; i += 64) {
    if (i < ia) dsti[i] = srci[i];
    else if (i >= ib) dsti[i - (ib - ia)] = srci[i];
  }
  for (int i = e.lane; i < e.L.S; i += 64) e.voff[i] = e.P->slot_voff[i];
  wsync();
}

__device__ inline void store_record(const Env& e, const HotLayout& h, const moog_layout_t& G,
                                    double* gf, int32_t* gq) {
  wsync();
  double2* dst = reinterpret_cast<double2*>(gf);
  const double2* src = reinterpret_cast<const double2*>(e.f);
  const int fa = h.f_cut0 / 2, fb = h.f_cut1 / 2;
  for (int i = e.lane; i < G.f64_per_env / 2; i += 64) {
    if (i < fa) dst[i] = src[i];
    else if (i >= fb) dst[i] = src[i - (fb - fa)];
  }
  int4* dsti = reinterpret_cast<int4*>(gq);
  const int4* srci = reinterpret_cast<const int4*>(e.q);
  const int ia = h.i_cut0 / 4, ib = h.i_cut1 / 4;
  for (int i = e.lane; i < G.i32_per_env / 4; i += 64) {
    if (i < ia) dsti[i] = srci[i];
    else if (i >= ib) dsti[i] = srci[i - (ib - ia)];
  }
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
};

enum { MODE_STEP = 0, MODE_PHYSICS = 1, MODE_RESET_MASK = 2, MODE_RESET_AUTO = 3 };

extern __shared__ __attribute__((aligned(16))) unsigned char moog_lds[];

__device__ inline void bind_env(Env& e, const KArgs& a, int env) {
  e.P = as_const_prog(a.P);
  e.L = a.H.L;
  const moog_layout_t& H = a.H.L;
  e.f = reinterpret_cast<double*>(moog_lds);
  e.q = reinterpret_cast<int32_t*>(moog_lds + (size_t)H.f64_per_env * 8);
  e.bb = reinterpret_cast<float*>(moog_lds + (size_t)H.f64_per_env * 8 + (size_t)H.i32_per_env * 4);
  e.xf = reinterpret_cast<double*>(e.bb + 8 * H.S);       // [S][8] only when S > 64
  double* after_xf = (H.S > 64) ? e.xf + 8 * H.S : e.xf;
  e.voff = reinterpret_cast<int32_t*>(after_xf);
  e.cand = reinterpret_cast<uint16_t*>(e.voff + ((H.S + 3) & ~3));
  e.lst = reinterpret_cast<uint8_t*>(e.cand + CAND_CAP);
  if (a.H.f_cut1 > a.H.f_cut0) e.gcol = a.f64 + (size_t)env * a.L.f64_per_env + a.L.o_color;
  else e.gcol = e.f + H.o_color;
  if (a.H.i_cut1 > a.H.i_cut0) {
    e.gopa = a.i32 + (size_t)env * a.L.i32_per_env + a.L.o_opacity;
    e.gshape = a.i32 + (size_t)env * a.L.i32_per_env + a.L.o_shape;
    e.gtele = a.i32 + (size_t)env * a.L.i32_per_env + a.L.o_tele;
  } else {
    e.gopa = e.q + H.o_opacity;
    e.gshape = e.q + H.o_shape;
    e.gtele = e.q + H.o_tele;
  }
  e.vslot = a.vslot;
  e.dbg = a.dbg;
  e.n_path = 0; e.n_resp = 0;
#ifdef MOOG_PROFILE
  for (int k = 0; k < 8; ++k) e.prof[k] = 0;
#endif
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
  load_record(e, a.H, a.L, gf, gq);
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
  store_record(e, a.H, a.L, gf, gq);
}

// Launch order for the next step: envs in (approximately) descending order of the cycles they
// took in this step (longest-processing-time first).  One 1024-thread workgroup: 1024-bin
// counting sort on cost / max(cost).  Runs on a side stream concurrently with the rasteriser.
// The order inside a bin is arbitrary -- the schedule never changes a result.
#define SCHED_BINS 1024
__global__ __launch_bounds__(1024) void moog_sched_kernel(const float* cost, int32_t* perm, int n) {
  __shared__ int hist[SCHED_BINS];
  __shared__ float red[16];
  const int t = threadIdx.x;
  float m = 0.f;
  for (int i = t; i < n; i += 1024) m = fmaxf(m, cost[i]);
  for (int o = 32; o > 0; o >>= 1) m = fmaxf(m, __shfl_xor(m, o));
  if ((t & 63) == 0) red[t >> 6] = m;
  hist[t] = 0;
  __syncthreads();
  m = red[0];
  for (int w = 1; w < 16; ++w) m = fmaxf(m, red[w]);
  const float scale = m > 0.f ? (float)(SCHED_BINS - 1) / m : 0.f;
  // bin 0 = most expensive
  for (int i = t; i < n; i += 1024) {
    int b = SCHED_BINS - 1 - (int)(cost[i] * scale);
    b = b < 0 ? 0 : (b > SCHED_BINS - 1 ? SCHED_BINS - 1 : b);
    atomicAdd(&hist[b], 1);
  }
  __syncthreads();
  // exclusive prefix sum over the 1024 bins (one bin per thread, Hillis-Steele in LDS)
  int v = hist[t];
  __syncthreads();
  for (int o = 1; o < SCHED_BINS; o <<= 1) {
    int add = (t >= o) ? hist[t - o] : 0;
    __syncthreads();
    hist[t] += add;
    __syncthreads();
  }
  const int start = hist[t] - v;
  __syncthreads();
  hist[t] = start;
  __syncthreads();
  for (int i = t; i < n; i += 1024) {
    int b = SCHED_BINS - 1 - (int)(cost[i] * scale);
    b = b < 0 ? 0 : (b > SCHED_BINS - 1 ? SCHED_BINS - 1 : b);
    perm[atomicAdd(&hist[b], 1)] = i;
  }
}

// DYN = the program has rules that create / move / filter sprites at run time (CreateSprites,
// ChangeLayer, VanishByFilter): that variant carries the reset path's sampler; the plain one
// is what the benchmark configs run.
// WPS = waves per SIMD the register allocation is sized for: 4 (128 VGPRs, some scratch) keeps sixteen
// envs per CU in flight, which is what programs with small state records want; 3 (168 VGPRs, no
// scratch in the hot loops) is faster once LDS holds fewer than fifteen records per CU anyway.
template <bool DYN, int WPS>
__global__ __launch_bounds__(64, WPS) void moog_step_kernel(KArgs a) {
  int env = blockIdx.x;
  if (env >= a.n_envs) return;
  if (a.perm) env = a.perm[env];
  const long long t_sched = a.cost ? clock64() : 0;
  int32_t* gq = a.i32 + (size_t)env * a.L.i32_per_env;
  if (a.mode == MODE_STEP && gq[a.L.o_reset_next] == 2) {  // reset earlier in this call
    if (threadIdx.x == 0) gq[a.L.o_reset_next] = 0;
    return;
  }
  Env e;
  bind_env(e, a, env);
  const long long t_begin = (a.dbg & 128) ? clock64() : 0;
  double* gf = a.f64 + (size_t)env * a.L.f64_per_env;
  load_record(e, a.H, a.L, gf, gq);
  if (e.inj && e.lane == 0) e.q[e.L.o_rng + 2] = 0;
  wsync();
  bbox_build_all(e);
  PProg P = as_const_prog(a.P);
  const int K = uni(P->updates_per_env_step);
  if (a.mode == MODE_PHYSICS) {
    for (int k = 0; k < K; ++k) apply_physics(e);
    store_record(e, a.H, a.L, gf, gq);
    return;
  }
  // environment.py:98-126
  const int n_rules = uni(P->n_rules);
  for (int r = 0; r < n_rules; ++r)
    if (P->rules[r].parent < 0) rule_step<DYN>(e, r);
  if (uni(P->n_actions) > 1) {   // composite.py:61-62: every sub-space, in keyword order
    const int na = uni(P->n_actions);
    const double* act = reinterpret_cast<const double*>(a.actions) + (size_t)2 * na * env;
    for (int k = 0; k < na; ++k) action_step(e, k, act[2 * k], act[2 * k + 1], (int)act[2 * k]);
  } else {
    double ax = 0, ay = 0;
    int ga = 4;
    if (P->action.kind == MOOG_ACTION_GRID) ga = reinterpret_cast<const int32_t*>(a.actions)[env];
    else {
      ax = reinterpret_cast<const double*>(a.actions)[2 * env];
      ay = reinterpret_cast<const double*>(a.actions)[2 * env + 1];
    }
    action_step(e, 0, ax, ay, ga);
  }
  { PROF_T0; for (int k = 0; k < K; ++k) apply_physics(e); PROF_ADD(e, 6); }
  int sc = e.q[e.L.o_step_count] + 1;
  wsync();
  if (e.lane == 0) e.q[e.L.o_step_count] = sc;
  wsync();
  int sr = 0;
  double r;
  r = task_reward<DYN>(e, sc, &sr);
  wsync();
  if (e.lane == 0) {
    if (sr) e.q[e.L.o_reset_next] = 1;
    if (a.reward) a.reward[env] = r;
    if (a.discount) a.discount[env] = sr ? 0.0 : 1.0;
    if (a.step_type) a.step_type[env] = sr ? 2 : 1;
  }
  store_record(e, a.H, a.L, gf, gq);
  if (a.cost && e.lane == 0) a.cost[env] = (float)(clock64() - t_sched);
  if ((a.dbg & 128) && e.lane == 0 && a.discount) {   // profiling aid: cycles and work counters instead of outputs
    a.discount[env] = (double)(clock64() - t_begin);
    if (a.reward) a.reward[env] = (double)(e.n_path + 100000 * e.n_resp);
#ifdef MOOG_PROFILE
    if (a.reward && (a.dbg >> 8)) a.reward[env] = (double)e.prof[((a.dbg >> 8) & 15) - 1];
#endif
  }
}

#include "moog_raster.h"

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
  uint32_t* d_vinfo = nullptr;   // rasteriser: vertex slot -> sprite slot | index within the sprite << 8
  int32_t n_envs = 0;
  int device = 0;
  uint64_t seed = 0;
  int64_t env_index0 = 0;
  moog_state_view_t view{nullptr, nullptr};
  size_t step_lds = 0, raster_lds = 0;
  int step_wps = 4;   // register-allocation variant of the step kernel (waves per SIMD)
  bool dynamic_rules = false;
  RPlan raster_plan_{};
  int raster_chunk = 0, raster_words = 0, raster_iwords = 0, raster_hwords = 1, raster_xxcap = 4;
  int timing = 0;   // bit k: launches of kernel k are bracketed by HIP events
  int32_t* perm = nullptr;
  hipStream_t sched_stream = nullptr;   // the launch-order sort runs beside the rasteriser
  hipEvent_t ev_step_done = nullptr, ev_sched_done = nullptr;
  bool sched_pending = false;
  float* cost = nullptr;
  TimedKernel timed[MOOG_K_COUNT];
  int step_dbg = 0, raster_stop = 0;   // profiling aids (MOOG_STEP_DEBUG / MOOG_RASTER_STOP at create, moog_engine_set_debug)
  // static prefix of the rasteriser (moog_raster.h): a scratch env record that holds the reference
  // sprites (what a reset makes of the constant generation ops) and their picture
  int n_static = 0, nsv = 0;
  double* s_f64 = nullptr;
  int32_t* s_i32 = nullptr;
  uint8_t* s_bg = nullptr;
};

static void free_engine(moog_engine* e) {
  if (e->d_prog) hipFree(e->d_prog);
  if (e->d_vslot) hipFree(e->d_vslot);
  if (e->d_vinfo) hipFree(e->d_vinfo);
  if (e->s_f64) hipFree(e->s_f64);
  if (e->s_i32) hipFree(e->s_i32);
  if (e->s_bg) hipFree(e->s_bg);
  delete e;
}

// Leading sprite slots that every reset creates identically and at rest: slots filled by generation
// ops without a random factor, zero velocity.  Whether a frame's prefix really equals the reference
// is checked by the raster kernel per frame, so this only has to be a good guess.
static int static_prefix_slots(const moog_program_t* p, int* nsv) {
  *nsv = 0;
  if (p->render.polymod != MOOG_POLYMOD_NONE) return 0;
  std::vector<char> ok((size_t)(p->n_slots > 0 ? p->n_slots : 1), 0);
  for (int oi = 0; oi < p->n_ops; ++oi) {
    const moog_genop_t& op = p->ops[oi];
    if (op.runtime || op.n_sampled != 0 || op.code_off >= 0 || op.count_min != op.count_max) continue;
    bool c = true;
    for (int k = 0; k < MOOG_NUM_FACTORS; ++k) c = c && op.factors[k].kind == MOOG_DIST_CONST;
    c = c && op.factors[MOOG_FAC_XVEL].a == 0 && op.factors[MOOG_FAC_YVEL].a == 0 && op.factors[MOOG_FAC_ANGVEL].a == 0;
    if (!c) continue;
    for (int sl = op.slot0; sl < op.slot0 + op.count_max && sl < p->n_slots; ++sl)
      if (sl >= 0 && !p->layer_dynamic[p->slot_layer[sl]]) ok[sl] = 1;
  }
  int ns = 0, nv = 0;
  while (ns < p->n_slots && ns < 32 && ok[ns] && p->slot_voff[ns] == nv) { nv += p->slot_vcap[ns]; ++ns; }
  *nsv = nv;
  return ns;
}

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
    if (p->slot_vcap[s] > 128) return fail(MOOG_E_UNSUPPORTED, "sprites with more than 128 vertices");
  for (int l = 0; l < p->n_layers; ++l)
    if (p->layer_nslots[l] > 64 * 2) return fail(MOOG_E_UNSUPPORTED, "layer too large");
  if (p->render.width % 16 != 0 || p->render.width > 128 || p->render.height > 1024 ||
      p->render.height > 1024)
    return fail(MOOG_E_UNSUPPORTED, "render size unsupported (width % 16 == 0, width <= 128)");
  return MOOG_OK;
}

static KArgs make_args(moog_engine* e, const void* actions, const moog_inject_t* inj,
                       const moog_step_out_t* out, int mode, const uint8_t* mask);
static RArgs raster_args(moog_engine* e, uint8_t* image);

// Resets one scratch env (the constant generation ops do not depend on the random stream) and renders
// its static prefix on top of the background colour: the reference record and picture of moog_raster.h.
static int build_static_prefix(moog_engine* e) {
  int nsv = 0;
  const int ns = getenv("MOOG_RASTER_NO_STATIC") ? 0 : static_prefix_slots(&e->prog, &nsv);
  if (ns == 0) return MOOG_OK;
  const size_t fb = (size_t)e->L.f64_per_env * 8, ib = (size_t)e->L.i32_per_env * 4;
  const size_t pb = (size_t)e->prog.render.width * e->prog.render.height * 3;
  if (hipMalloc(&e->s_f64, fb) != hipSuccess || hipMalloc(&e->s_i32, ib) != hipSuccess ||
      hipMalloc(&e->s_bg, pb) != hipSuccess)
    return fail(MOOG_E_NOMEM, "hipMalloc(static prefix) failed");
  HIPCHK(hipMemset(e->s_f64, 0, fb));
  HIPCHK(hipMemset(e->s_i32, 0, ib));
  const moog_state_view_t keep = e->view;
  const int32_t keep_n = e->n_envs;
  e->view.f64 = e->s_f64; e->view.i32 = e->s_i32; e->n_envs = 1;
  KArgs a = make_args(e, nullptr, nullptr, nullptr, MODE_RESET_MASK, nullptr);
  a.dbg = 0;
  hipLaunchKernelGGL(moog_reset_kernel, dim3(1), dim3(64), e->step_lds, 0, a);
  RArgs r = raster_args(e, e->s_bg);
  r.n_static = ns; r.nsv = nsv; r.build = 1; r.debug_stop = 0;
  moog_raster_launch(r, e->raster_lds, 0);
  e->view = keep; e->n_envs = keep_n;
  HIPCHK(hipGetLastError());
  HIPCHK(hipStreamSynchronize(0));
  e->n_static = ns; e->nsv = nsv;
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
  if (err != hipSuccess) { free_engine(e); return fail(MOOG_E_NOMEM, "hipMalloc(program) failed"); }
  err = hipMemcpy(e->d_prog, prog, sizeof(moog_program_t), hipMemcpyHostToDevice);
  if (err != hipSuccess) { free_engine(e); return fail(MOOG_E_HIP, "hipMemcpy(program) failed"); }
  {
    std::vector<int16_t> vs((size_t)(prog->n_total_verts > 0 ? prog->n_total_verts : 1), 0);
    for (int sl = 0; sl < prog->n_slots; ++sl)
      for (int k = 0; k < prog->slot_vcap[sl]; ++k) vs[prog->slot_voff[sl] + k] = (int16_t)sl;
    err = hipMalloc(&e->d_vslot, vs.size() * sizeof(int16_t));
    if (err == hipSuccess)
      err = hipMemcpy(e->d_vslot, vs.data(), vs.size() * sizeof(int16_t), hipMemcpyHostToDevice);
    if (err != hipSuccess) { free_engine(e); return fail(MOOG_E_NOMEM, "vertex table"); }
    std::vector<uint32_t> vi(vs.size(), 0u);
    for (int sl = 0; sl < prog->n_slots; ++sl)
      for (int k = 0; k < prog->slot_vcap[sl]; ++k) vi[prog->slot_voff[sl] + k] = (uint32_t)sl | ((uint32_t)k << 8);
    err = hipMalloc(&e->d_vinfo, vi.size() * sizeof(uint32_t));
    if (err == hipSuccess)
      err = hipMemcpy(e->d_vinfo, vi.data(), vi.size() * sizeof(uint32_t), hipMemcpyHostToDevice);
    if (err != hipSuccess) { free_engine(e); return fail(MOOG_E_NOMEM, "vertex table"); }
  }
  const moog_layout_t HL = hot_layout(e->L).L;   // the records as staged in LDS
  e->step_lds = (size_t)HL.f64_per_env * 8 + (size_t)HL.i32_per_env * 4 +
                (size_t)e->L.S * 4 * 8 + (e->L.S > 64 ? (size_t)e->L.S * 8 * 8 : 0) +
                (size_t)((e->L.S + 3) & ~3) * 4 + CAND_CAP * 2 + 128 + 16;
  { const char* pad = getenv("MOOG_LDS_PAD"); if (pad) e->step_lds += (size_t)atoi(pad); }  // occupancy experiments
  if (e->step_lds > 160 * 1024) {
    free_engine(e);
    return fail(MOOG_E_UNSUPPORTED, "state record does not fit in 160 KB of LDS");
  }
  // raster LDS plan (moog_raster.h): row records for `chunk` rows per pass
  {
    int W = prog->render.width, H = prog->render.height;
    int ncopy = prog->render.polymod == MOOG_POLYMOD_TORUS ? 9 : 1;
    int items = prog->n_slots * ncopy;
    if (items < 1) items = 1;
    e->raster_words = (W + 63) / 64;
    e->raster_iwords = (items + 31) / 32;
    int maxv = 2;
    for (int sl = 0; sl < prog->n_slots; ++sl) if (prog->slot_vcap[sl] > maxv) maxv = prog->slot_vcap[sl];
    e->raster_xxcap = 2 * maxv;
    e->raster_hwords = (maxv + 31) / 32;
    RPlan pl;
    // Row records per pass: 376 (>= H so that any item fits; sized so that six workgroups of the
    // 4096 x 32-sprite workload share a CU), then as many more as fit without costing a resident
    // workgroup (frames with more rows than records take several passes).
    int cap = items * H;
    if (cap > 376) cap = 376;
    if (cap < H) cap = H;
    raster_plan(prog->n_slots, e->L.TOTV, ncopy, W, H, cap, e->raster_iwords, e->raster_hwords, e->raster_xxcap, &pl);
    {
      const unsigned lds_cu = 160 * 1024;
      unsigned wgs = pl.total ? lds_cu / pl.total : 0;
      if (wgs > 6) wgs = 6;   // registers hold six workgroups per CU at most
      const int want = items * H < 4096 ? items * H : 4096;
      while (wgs && cap < want) {
        RPlan p2;
        int c2 = cap + 32 < want ? cap + 32 : want;
        raster_plan(prog->n_slots, e->L.TOTV, ncopy, W, H, c2, e->raster_iwords, e->raster_hwords, e->raster_xxcap, &p2);
        if (lds_cu / p2.total < wgs) break;
        cap = c2; pl = p2;
      }
    }
    { const char* rc = getenv("MOOG_RASTER_ROWS"); if (rc && atoi(rc) >= H) cap = atoi(rc); }   // tuning / tests of the multi-pass path
    raster_plan(prog->n_slots, e->L.TOTV, ncopy, W, H, cap, e->raster_iwords, e->raster_hwords, e->raster_xxcap, &pl);
    if (pl.total > 160 * 1024 || (size_t)e->L.TOTV * ncopy >= (1u << 20)) {
      free_engine(e);
      return fail(MOOG_E_UNSUPPORTED, "raster working set does not fit in LDS");
    }
    e->raster_chunk = cap;
    e->raster_plan_ = pl;
    e->raster_lds = pl.total;
    { const char* pad = getenv("MOOG_RASTER_LDS_PAD"); if (pad) e->raster_lds += (size_t)atoi(pad); }  // occupancy experiments
  }
  {
    const void* variants[4] = {reinterpret_cast<const void*>(moog_step_kernel<false, 3>),
                               reinterpret_cast<const void*>(moog_step_kernel<false, 4>),
                               reinterpret_cast<const void*>(moog_step_kernel<true, 3>),
                               reinterpret_cast<const void*>(moog_step_kernel<true, 4>)};
    for (int v = 0; v < 4 && err == hipSuccess; ++v)
      err = hipFuncSetAttribute(variants[v], hipFuncAttributeMaxDynamicSharedMemorySize, (int)e->step_lds);
    e->step_wps = (160 * 1024 / (e->step_lds ? e->step_lds : 1)) <= 14 ? 3 : 4;
    { const char* w = getenv("MOOG_STEP_WPS"); if (w && (atoi(w) == 3 || atoi(w) == 4)) e->step_wps = atoi(w); }   // experiments
  }
  for (int r = 0; r < prog->n_rules; ++r) {
    int k = prog->rules[r].kind;
    if (k == MOOG_RULE_VANISH_BY_FILTER || k == MOOG_RULE_CHANGE_LAYER || k == MOOG_RULE_CREATE_SPRITES ||
        k == MOOG_RULE_MODIFY_SPRITES || k == MOOG_RULE_MODIFY_ON_CONTACT)
      e->dynamic_rules = true;
  }
  for (int t = 0; t < prog->n_tasks; ++t) {
    if (prog->tasks[t].kind == MOOG_TASK_CONTACT_REWARD && (prog->tasks[t].xcond >= 0 || prog->tasks[t].xreward >= 0))
      e->dynamic_rules = true;
    if (prog->tasks[t].kind == MOOG_TASK_RESET && prog->tasks[t].cond >= MOOG_COND_ALL_EXPR)
      e->dynamic_rules = true;
  }
  for (int r = 0; r < prog->n_rules; ++r)
    if ((prog->rules[r].kind == MOOG_RULE_CONDITIONAL || prog->rules[r].kind == MOOG_RULE_PHASE) &&
        prog->rules[r].cond >= MOOG_RCOND_CONTACT_COUNT)
      e->dynamic_rules = true;   // (rule_gate is compiled into both variants; keep them together anyway)
  for (int l = 0; l < prog->n_layers; ++l) if (prog->layer_dynamic[l]) e->dynamic_rules = true;
  if (err == hipSuccess)
    err = hipFuncSetAttribute(reinterpret_cast<const void*>(moog_reset_kernel),
                              hipFuncAttributeMaxDynamicSharedMemorySize, (int)e->step_lds);
  if (err == hipSuccess)
    err = (hipError_t)moog_raster_configure(e->raster_lds);
  if (err != hipSuccess) {
    free_engine(e);
    return fail(MOOG_E_HIP, std::string("hipFuncSetAttribute: ") + hipGetErrorString(err));
  }
  { const char* ds = getenv("MOOG_STEP_DEBUG"); e->step_dbg = ds ? atoi(ds) : 0; }
  { const char* ds = getenv("MOOG_RASTER_STOP"); e->raster_stop = ds ? atoi(ds) : 0; }
  int rc2 = build_static_prefix(e);
  if (rc2) { free_engine(e); return rc2; }
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
  if (e->sched_stream) {
    hipStreamSynchronize(e->sched_stream);
    hipEventDestroy(e->ev_step_done); hipEventDestroy(e->ev_sched_done);
    hipStreamDestroy(e->sched_stream);
  }
  free_engine(e);
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
    if ((e->timing >> id) & 1) { hipEventCreate(&a); hipEventCreate(&b); hipEventRecord(a, s); }
  }
  ~Bracket() {
    if (a) { hipEventRecord(b, s); e->timed[id].pending.emplace_back(a, b); }
  }
};

static KArgs make_args(moog_engine* e, const void* actions, const moog_inject_t* inj,
                       const moog_step_out_t* out, int mode, const uint8_t* mask) {
  KArgs a;
  a.P = e->d_prog; a.L = e->L; a.H = hot_layout(e->L); a.f64 = e->view.f64; a.i32 = e->view.i32;
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
  a.perm = (mode == MODE_STEP) ? e->perm : nullptr;
  a.cost = (mode == MODE_STEP) ? e->cost : nullptr;
  a.dbg = e->step_dbg;
  return a;
}

static void launch_step(moog_engine* e, hipStream_t s, const KArgs& a) {
  const dim3 g(e->n_envs), b(64);
  if (e->dynamic_rules) {
    if (e->step_wps == 3) hipLaunchKernelGGL((moog_step_kernel<true, 3>), g, b, e->step_lds, s, a);
    else hipLaunchKernelGGL((moog_step_kernel<true, 4>), g, b, e->step_lds, s, a);
  } else {
    if (e->step_wps == 3) hipLaunchKernelGGL((moog_step_kernel<false, 3>), g, b, e->step_lds, s, a);
    else hipLaunchKernelGGL((moog_step_kernel<false, 4>), g, b, e->step_lds, s, a);
  }
}

static RArgs raster_args(moog_engine* e, uint8_t* image) {
  RArgs r;
  r.P = e->d_prog; r.L = e->L; r.f64 = e->view.f64; r.i32 = e->view.i32; r.image = image;
  r.vinfo = e->d_vinfo; r.plan = e->raster_plan_;
  r.n_envs = e->n_envs; r.chunk = e->raster_chunk; r.words = e->raster_words;
  r.iwords = e->raster_iwords; r.hwords = e->raster_hwords; r.xxcap = e->raster_xxcap;
  r.debug_stop = e->raster_stop;
  r.n_static = e->n_static; r.nsv = e->nsv; r.build = 0;
  r.sref_v = e->s_f64 ? e->s_f64 + e->L.o_verts : nullptr;
  r.sref_col = e->s_f64 ? e->s_f64 + e->L.o_color : nullptr;
  r.sref_flags = e->s_i32 ? e->s_i32 + e->L.o_flags : nullptr;
  r.sref_nv = e->s_i32 ? e->s_i32 + e->L.o_nverts : nullptr;
  r.sref_opa = e->s_i32 ? e->s_i32 + e->L.o_opacity : nullptr;
  r.sbg = e->s_bg;
  return r;
}

static int launch_raster(moog_engine* e, uint8_t* image, hipStream_t s) {
  RArgs r = raster_args(e, image);
  {
    Bracket br(e, MOOG_K_RASTER, s);
    moog_raster_launch(r, e->raster_lds, s);
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
  a.perm = e->perm;
  a.cost = e->cost;
  if (e->sched_pending) {   // the order computed from the previous step's costs
    HIPCHK(hipStreamWaitEvent(s, e->ev_sched_done, 0));
    e->sched_pending = false;
  }
  {
    Bracket br(e, MOOG_K_STEP, s);
    launch_step(e, s, a);
  }
  HIPCHK(hipGetLastError());
  if (e->perm && e->cost) {
    HIPCHK(hipEventRecord(e->ev_step_done, s));
    HIPCHK(hipStreamWaitEvent(e->sched_stream, e->ev_step_done, 0));
    hipLaunchKernelGGL(moog_sched_kernel, dim3(1), dim3(1024), 0, e->sched_stream, e->cost, e->perm, e->n_envs);
    HIPCHK(hipEventRecord(e->ev_sched_done, e->sched_stream));
    e->sched_pending = true;
  }
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
    launch_step(e, s, a);
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

int moog_engine_set_schedule(moog_engine_t* e, int32_t* perm_dev, float* cost_dev) {
  if (!e) return fail(MOOG_E_INVALID, "null engine");
  if (e->sched_pending) { hipEventSynchronize(e->ev_sched_done); e->sched_pending = false; }
  e->perm = perm_dev;
  e->cost = cost_dev;
  if (perm_dev && cost_dev && !e->sched_stream) {
    HIPCHK(hipSetDevice(e->device));
    HIPCHK(hipStreamCreateWithFlags(&e->sched_stream, hipStreamNonBlocking));
    HIPCHK(hipEventCreateWithFlags(&e->ev_step_done, hipEventDisableTiming));
    HIPCHK(hipEventCreateWithFlags(&e->ev_sched_done, hipEventDisableTiming));
  }
  return MOOG_OK;
}

int moog_engine_static_prefix(moog_engine_t* e, int32_t* n_slots, uint8_t* image_dev, void* hip_stream) {
  if (!e) return fail(MOOG_E_INVALID, "null engine");
  if (n_slots) *n_slots = e->n_static;
  if (image_dev && e->n_static > 0) {
    HIPCHK(hipSetDevice(e->device));
    HIPCHK(hipMemcpyAsync(image_dev, e->s_bg, (size_t)e->prog.render.width * e->prog.render.height * 3,
                          hipMemcpyDeviceToDevice, (hipStream_t)hip_stream));
  }
  return MOOG_OK;
}

int moog_engine_set_debug(moog_engine_t* e, int32_t step_debug, int32_t raster_stop) {
  if (!e) return fail(MOOG_E_INVALID, "null engine");
  e->step_dbg = step_debug;
  e->raster_stop = raster_stop;
  return MOOG_OK;
}

int moog_engine_set_timing(moog_engine_t* e, int32_t enabled) {
  if (!e) return fail(MOOG_E_INVALID, "null engine");
  e->timing = enabled;
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
