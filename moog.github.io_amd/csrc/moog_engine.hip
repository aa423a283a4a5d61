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

#include "moog_kernels.h"

#include "moog_raster.h"
#include <dlfcn.h>
#include <unistd.h>

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
  int64_t seq = 0;   // launches seen while timing is on (every `period`-th one is bracketed)
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
  void* spec_handle = nullptr;   // a step kernel compiled for this very program (load_spec_kernel), or null
  void (*spec_launch)(int, size_t, hipStream_t, const KArgs*) = nullptr;
  int prio_pm[3] = {0, 0, 0};   // wave priorities by launch rank, per mille of the batch (KArgs::prio_t)
  int32_t xstack_off = 0;
  FOp* d_fops = nullptr;        // flattened force list (moog_flatten_forces)
  int32_t n_fops = 0;
  int32_t* watch = nullptr;     // section sampling (MOOG_WATCH=1): [n_envs][MOOG_WATCH_SECTIONS]
  int32_t watch_off = 0;
  bool dynamic_rules = false;
  bool maze_kernel = false;   // the program uses MazePhysics / a maze walk / a per-reset maze
  bool late_reset = false;    // ... but only to build an episode: it is stepped by the kernels without the rare components, and
                              // the full reset kernel behind every step launch opens the episodes those could not (step_env)
  uint8_t* late_mask = nullptr;   // [n_envs]
  RPlan raster_plan_{};
  RmSetup mask_setup{};   // the mask rasteriser (moog_raster_mask_core.h): ok = this program's ordinary frames are drawn by it
  uint8_t* draw = nullptr;        // its input: a draw record per env (moog_draw_record.h), written by the step kernel or derived before the launch
  RmDrawLayout draw_lay{};
  int raster_persist = 0;         // MOOG_RASTER_PERSIST=k: frames per CU the mask rasteriser's launch keeps resident (one round of workgroups that draw several frames each); 0: a workgroup per frame
  int n_cus = 256;
  bool draw_in_step = true;       // MOOG_DRAW_IN_STEP=0: never by the step kernel (A/B runs, tests: the derive kernel for every launch)
  int raster_tile_w = 0, raster_band_h = 0, raster_tiles_x = 1, raster_bands = 1;   // one workgroup per tile of the canvas
  // anti_aliasing > 1: frames are drawn on a canvas aa x the observation (a chunk of envs at a time) and down-sampled
  int aa = 1, canvas_w = 0, canvas_h = 0, aa_chunk = 0;
  int pad_w = 0;                  // canvas_w rounded up to a multiple of 16: the width the rasteriser draws (pil_renderer.py:64-66
                                  // accepts any size; the extra columns are what Pillow would draw on a wider image, and are dropped)
  uint8_t* pad_img = nullptr;     // [n_envs][canvas_h][pad_w][3] when pad_w != canvas_w without anti-aliasing (cropped into the caller's frames)
  uint8_t* aa_canvas = nullptr;   // [aa_chunk][canvas_h][pad_w][3]
  uint8_t* aa_tmp = nullptr;      // [aa_chunk][canvas_h][width rounded up to 4][3]
  int32_t* aa_tables = nullptr;   // bounds + coefficients of both axes
  RResize aa_resize{};
  int raster_chunk = 0, raster_words = 0, raster_iwords = 0, raster_hwords = 1, raster_xxcap = 4;
  int timing = 0;   // bit k: launches of kernel k are bracketed by HIP events
  int32_t* perm = nullptr;
  hipStream_t sched_stream = nullptr;   // the launch-order sort runs beside the rasteriser
  hipEvent_t ev_step_done = nullptr, ev_sched_done = nullptr;
  bool sched_pending = false;
  float* cost = nullptr;
  // moog_engine_set_reset_pool: the next episode of every env is built beside the step kernels (moog_kernels.h "reset pool")
  static constexpr int POOL_STREAMS = 8;
  int pool_streams = 2;   // the ones in use: hardware queues the runtime has (GPU_MAX_HW_QUEUES, default 4) minus the caller's and the sort's
  bool pool_on = false, pool_ready = false;
  int pool_depth = 2;              // records per env: the next episode and the one after (an episode shorter than a fill does not stall its call)
  int32_t* pool_state = nullptr;   // [pool_depth][n_envs] 0 empty / 3 claimed / 1 being filled / 2 ready
  int32_t* pool_tag = nullptr;     // [pool_depth][n_envs] episode of the pool record
  int32_t* pool_lock = nullptr;    // [n_envs] 4 while the step kernel opens an episode of the env
  unsigned long long* pool_stats = nullptr;   // [4] KArgs::pool_stats
  double* pool_f64[2] = {nullptr, nullptr};   // [pool_depth][n_envs][f64_per_env]: the fill's inputs, the record after its reset
  int32_t* pool_i32[2] = {nullptr, nullptr};
  hipStream_t pool_stream[POOL_STREAMS] = {};   // fills run here, call k's on stream k % pool_streams
  hipEvent_t ev_pool = nullptr;    // the call a fill follows
  int64_t pool_fills = 0;          // fill launches so far
  int act_f32 = 0;               // moog_engine_set_action_dtype: the action buffer holds float32 values
  int32_t* layer_hw = nullptr;   // [2 * MOOG_MAX_LAYERS]: high-water mark / dropped appends of the dynamic layers
  TimedKernel timed[MOOG_K_COUNT];
  int32_t* fault_flag = nullptr;   // pinned host word the kernels OR fault bits into
  int32_t* rows_seen = nullptr;    // two pinned host words: the most polygon rows a frame wanted when that was more than the mask rasteriser's row records; how many frames did
  int raster_rows_fixed = 0;       // MOOG_RASTER_ROWS pins the records (tests of the multi-pass path)
  int mask_free_cap = 0;           // the most row records that cost no resident frame per CU (mask_free_rows)
  int step_dbg = 0, raster_stop = 0;   // profiling aids (MOOG_STEP_DEBUG / MOOG_RASTER_STOP at create, moog_engine_set_debug)
  // static prefix of the rasteriser (moog_raster.h): a scratch env record that holds the reference
  // sprites (what a reset makes of the constant generation ops) and their picture
  int n_static = 0, nsv = 0;
  double* s_f64 = nullptr;
  int32_t* s_i32 = nullptr;
  uint8_t* s_bg = nullptr;
  const uint32_t* rgb_override = nullptr;   // moog_engine_set_color_override
  int kept_n_static = 0, kept_pe_ns = 0, kept_pe_nsv = 0;   // the prefixes' sizes while a colour override has them switched off
  // per-env prefix (RArgs::sbg_env_stride): leading sprites that stay put within an episode but differ between envs
  int pe_ns = 0, pe_nsv = 0;        // slots / vertex slots of the prefix (0: off); shrinks to the slots that really stay put
  double* pe_f64 = nullptr;         // [n_envs] snapshot of the record each env's picture was drawn from
  int32_t* pe_i32 = nullptr;
  uint8_t* pe_bg = nullptr;         // [n_envs][canvas_h][pad_w][3]
  int32_t* pe_valid = nullptr;      // [n_envs] the env has a picture
  int32_t* pe_build = nullptr;      // [n_envs] this call's build launch draws the env's picture
  int32_t* pe_min = nullptr;        // pinned host word: first slot of the prefix seen changing in the middle of an episode
};

static void free_engine(moog_engine* e) {
  if (e->d_prog) hipFree(e->d_prog);
  if (e->d_vslot) hipFree(e->d_vslot);
  if (e->d_vinfo) hipFree(e->d_vinfo);
  if (e->draw) hipFree(e->draw);
  if (e->s_f64) hipFree(e->s_f64);
  if (e->s_i32) hipFree(e->s_i32);
  if (e->s_bg) hipFree(e->s_bg);
  if (e->aa_canvas) hipFree(e->aa_canvas);
  if (e->pad_img) hipFree(e->pad_img);
  if (e->aa_tmp) hipFree(e->aa_tmp);
  if (e->aa_tables) hipFree(e->aa_tables);
  if (e->fault_flag) hipHostFree(e->fault_flag);
  if (e->rows_seen) hipHostFree(e->rows_seen);
  if (e->layer_hw) hipFree(e->layer_hw);
  for (int k = 0; k < moog_engine::POOL_STREAMS; ++k)
    if (e->pool_stream[k]) { hipStreamSynchronize(e->pool_stream[k]); hipStreamDestroy(e->pool_stream[k]); }
  if (e->ev_pool) hipEventDestroy(e->ev_pool);
  if (e->pe_f64) hipFree(e->pe_f64);
  if (e->pe_i32) hipFree(e->pe_i32);
  if (e->pe_bg) hipFree(e->pe_bg);
  if (e->pe_valid) hipFree(e->pe_valid);
  if (e->pe_build) hipFree(e->pe_build);
  if (e->pe_min) hipHostFree(e->pe_min);
  if (e->pool_state) hipFree(e->pool_state);
  if (e->pool_tag) hipFree(e->pool_tag);
  if (e->pool_lock) hipFree(e->pool_lock);
  if (e->late_mask) hipFree(e->late_mask);
  if (e->pool_stats) hipFree(e->pool_stats);
  for (int k = 0; k < 2; ++k) { if (e->pool_f64[k]) hipFree(e->pool_f64[k]); if (e->pool_i32[k]) hipFree(e->pool_i32[k]); }
  if (e->watch) hipFree(e->watch);
  if (e->d_fops) hipFree(e->d_fops);
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

// Per-env prefix: leading slots whose sprites are created at rest by the initializer (whatever their positions: a maze's
// walls, food on its cells) and that no op's code writes to afterwards.  A guess like the one above: the check launch
// compares every frame's prefix with the env's snapshot, and slots that turn out to change within episodes (food that gets
// eaten) shorten the prefix (launch_raster).
static int env_prefix_slots(const moog_program_t* p, int* nsv) {
  *nsv = 0;
  if (p->render.polymod != MOOG_POLYMOD_NONE) return 0;
  std::vector<char> ok((size_t)(p->n_slots > 0 ? p->n_slots : 1), 0);
  for (int oi = 0; oi < p->n_ops; ++oi) {
    const moog_genop_t& op = p->ops[oi];
    if (op.runtime || op.cell_sel == MOOG_CELL_STORE) continue;
    bool c = true;
    for (int k = MOOG_FAC_XVEL; k <= MOOG_FAC_ANGVEL; ++k)
      if (k == MOOG_FAC_XVEL || k == MOOG_FAC_YVEL || k == MOOG_FAC_ANGVEL) c = c && op.factors[k].kind == MOOG_DIST_CONST && op.factors[k].a == 0;
    if (!c) continue;
    for (int sl = op.slot0; sl < op.slot0 + op.count_max && sl < p->n_slots; ++sl)
      if (sl >= 0 && !p->layer_dynamic[p->slot_layer[sl]]) ok[sl] = 1;
  }
  for (int oi = 0; oi < p->n_ops; ++oi)
    if (p->ops[oi].cell_sel == MOOG_CELL_STORE && p->ops[oi].cell_arg >= 0 && p->ops[oi].cell_arg < p->n_slots) ok[p->ops[oi].cell_arg] = 0;
  int ns = 0, nv = 0;
  while (ns < p->n_slots && ok[ns] && p->slot_voff[ns] == nv) { nv += p->slot_vcap[ns]; ++ns; }
  *nsv = nv;
  return ns;
}

// Which step kernel a program takes, from the program alone (no device): the variant (plain / + expression evaluator,
// run-time sampler, dynamic layers / + the rare components), whether its episodes are opened by the reset kernel behind the
// step kernel (late reset), the LDS of one env and the register-allocation variant that follows from it.  Used by
// moog_engine_create and, through moog_program_step_kernel, by the builder of program-specialised kernels (moog/_spec.py).
struct StepVariant { bool dynamic_rules = false, maze_kernel = false, late_reset = false; int wps = 4; size_t step_lds = 0; int32_t xstack_off = 0; };

static StepVariant step_variant_of(const moog_program_t* prog) {
  StepVariant v;
  moog_layout_t GL;
  moog_layout(prog, &GL);
  const moog_layout_t HL = hot_layout(GL).L;   // the records as staged in LDS
  v.step_lds = (size_t)HL.f64_per_env * 8 + (size_t)HL.i32_per_env * 4 + (size_t)GL.S * 4 * 8 +
               (size_t)((GL.S + 3) & ~3) * 4 + CAND_CAP * 2 + 128 + 64 * 8 + 16;
  if (prog->xstack_depth > 0) {   // per-lane value stacks of the lane-parallel filter evaluator (moog_device.h eval_expr_t<true>)
    v.step_lds = (v.step_lds + 15) & ~(size_t)15;
    v.xstack_off = (int32_t)v.step_lds;
    v.step_lds += (size_t)prog->xstack_depth * 64 * 8;
  }
  { const char* pad = getenv("MOOG_LDS_PAD"); if (pad) v.step_lds += (size_t)atoi(pad); }  // occupancy experiments
  for (int r = 0; r < prog->n_rules; ++r) {
    int k = prog->rules[r].kind;
    if (k == MOOG_RULE_VANISH_BY_FILTER || k == MOOG_RULE_CHANGE_LAYER || k == MOOG_RULE_CREATE_SPRITES ||
        k == MOOG_RULE_MODIFY_SPRITES || k == MOOG_RULE_MODIFY_ON_CONTACT || k == MOOG_RULE_DRAWS)
      v.dynamic_rules = true;
  }
  for (int t = 0; t < prog->n_tasks; ++t) {
    if (prog->tasks[t].kind == MOOG_TASK_CONTACT_REWARD && (prog->tasks[t].xcond >= 0 || prog->tasks[t].xreward >= 0))
      v.dynamic_rules = true;
    if (prog->tasks[t].kind == MOOG_TASK_RESET && (prog->tasks[t].cond >= MOOG_COND_ALL_EXPR || prog->tasks[t].xreward >= 0))
      v.dynamic_rules = true;   // (a reward_fn that reads the state is an expression too: task_reward<DYN> evaluates it)
  }
  for (int r = 0; r < prog->n_rules; ++r)
    if ((prog->rules[r].kind == MOOG_RULE_CONDITIONAL || prog->rules[r].kind == MOOG_RULE_PHASE) &&
        prog->rules[r].cond >= MOOG_RCOND_CONTACT_COUNT)
      v.dynamic_rules = true;   // (rule_gate is compiled into both variants; keep them together anyway)
  for (int l = 0; l < prog->n_layers; ++l) if (prog->layer_dynamic[l]) v.dynamic_rules = true;
  for (int f = 0; f < prog->n_forces; ++f)   // a traced force_fn runs in the expression evaluator
    if (prog->forces[f].kind == MOOG_FORCE_DISTANCE_EXPR) v.dynamic_rules = true;
  for (int r = 0; r < prog->n_rules; ++r)
    if (prog->rules[r].kind == MOOG_RULE_FIXATION || (prog->rules[r].kind == MOOG_RULE_PHASE && prog->rules[r].op == 1))
      v.dynamic_rules = true;
  for (int k = 0; k < prog->n_dcode; ++k) {   // assigning sprite.angle turns the path: in the kernels that carry every component
    if (prog->dcode[k].op == MOOG_X_STORE && prog->dcode[k].a == MOOG_XA_ANGLE) v.maze_kernel = true;
    // expressions that name sprites by slot (state-level task functions: Reset(condition / reward_fn), and the reset-time
    // code) use opcodes only those kernels' evaluator carries (moog_device.h eval_expr_t: MOOG_WITH_MAZE)
    const int op = prog->dcode[k].op;
    if (op == MOOG_X_SLOT_ATTR || op == MOOG_X_HDRAW || op == MOOG_X_HDRAW_T || op == MOOG_X_STORE_VERT || op == MOOG_X_FACTOR)
      v.maze_kernel = true;
  }
  for (int o = 0; o < prog->n_ops; ++o) if (prog->ops[o].cell_sel != MOOG_CELL_NONE) v.maze_kernel = true;   // maze / draw / shuffle ops
  if (prog->n_hdraws > 0) v.maze_kernel = true;   // reset-time expressions: in the kernels that carry every component (m3 / m4)
  for (int o = 0; o < prog->n_ops; ++o)
    for (int k = 0; k < MOOG_NUM_FACTORS; ++k)
      if (prog->ops[o].factors[k].kind == MOOG_DIST_EXPR || prog->ops[o].factors[k].kind == MOOG_DIST_EXPR_SHAPE)
        v.maze_kernel = true;
  // the maze components live in a kernel variant of their own (m3 / m4): their code would only enlarge the others
  for (int f = 0; f < prog->n_forces; ++f)
    if (prog->forces[f].kind == MOOG_FORCE_MAZE_WALK || prog->forces[f].kind == MOOG_FORCE_MAZE_WALK_DET) v.maze_kernel = true;
  for (int c = 0; c < prog->n_corrective; ++c) if (prog->corrective[c].kind == MOOG_CORR_MAZE) v.maze_kernel = true;
  if (prog->maze.random) v.maze_kernel = true;
  if (v.maze_kernel) {
    // What the program needs WHILE STEPPING of the components only the every-component kernels carry: maze walks and
    // MazePhysics, modifiers that assign sprite.angle, run-time generators with maze cells / computed factors.  Everything else
    // that selects those kernels (reset-time draws and expressions, maze generation, shuffles, choices, look-aheads, values an
    // initializer keeps across episodes) happens when an episode is built.
    bool stepping = false;
    for (int f = 0; f < prog->n_forces; ++f)
      stepping = stepping || prog->forces[f].kind == MOOG_FORCE_MAZE_WALK || prog->forces[f].kind == MOOG_FORCE_MAZE_WALK_DET;
    for (int c = 0; c < prog->n_corrective; ++c) stepping = stepping || prog->corrective[c].kind == MOOG_CORR_MAZE;
    for (int k = 0; k < prog->n_dcode; ++k) stepping = stepping || (prog->dcode[k].op == MOOG_X_STORE && prog->dcode[k].a == MOOG_XA_ANGLE);
    for (int o = 0; o < prog->n_ops; ++o) {
      const moog_genop_t& op = prog->ops[o];
      if (!op.runtime) continue;
      bool full = op.cell_sel != MOOG_CELL_NONE || op.code_off >= 0;
      for (int k = 0; k < MOOG_NUM_FACTORS; ++k)
        full = full || op.factors[k].kind == MOOG_DIST_EXPR || op.factors[k].kind == MOOG_DIST_EXPR_SHAPE;
      stepping = stepping || full;
    }
    const char* off = getenv("MOOG_NO_LATE_RESET");
    v.late_reset = !stepping && !(off && atoi(off));
    if (v.late_reset) v.dynamic_rules = true;   // (the variant with the expression evaluator)
  }
  {   // MOOG_STEP_VARIANT=t|m (experiments): run a program on a kernel variant that carries more than it needs (what the variant
      // itself costs: profiles/r04_variant_tax.txt)
    const char* sv = getenv("MOOG_STEP_VARIANT");
    if (sv && (sv[0] == 't' || sv[0] == 'm')) v.dynamic_rules = true;
    if (sv && sv[0] == 'm') { v.maze_kernel = true; v.late_reset = false; }
  }
  // register allocation: three waves per SIMD (168 VGPRs) when the LDS of an env allows no more than 14 envs per CU anyway
  v.wps = (160 * 1024 / (v.step_lds ? v.step_lds : 1)) <= 14 ? 3 : 4;
  { const char* w = getenv("MOOG_STEP_WPS"); if (w && (atoi(w) == 3 || atoi(w) == 4)) v.wps = atoi(w); }   // experiments
  { const char* w = getenv("MOOG_STEP_WPS"); if (w && atoi(w) == 2 && !v.dynamic_rules && !(v.maze_kernel && !v.late_reset)) v.wps = 2; }   // (plain programs only)
  return v;
}

static uint64_t program_hash(const moog_program_t* prog) {   // FNV-1a 64 over the program's bytes (moog/_spec.py computes the same)
  const unsigned char* b = reinterpret_cast<const unsigned char*>(prog);
  uint64_t h = 1469598103934665603ull;
  for (size_t i = 0; i < sizeof(moog_program_t); ++i) { h ^= b[i]; h *= 1099511628211ull; }
  return h;
}

// A step kernel compiled for exactly this program (csrc/moog_step_spec.hip), if one has been built: looked for as
// <MOOG_SPEC_DIR or the directory of this library + "/spec">/step_<hash>_d<variant>w<wps>.so.  Everything about it is
// checked -- the digest of the kernel sources and flags it was built from, ABI, argument struct, variant, and the embedded
// program byte for byte -- before it replaces the generic kernel.
static void load_spec_kernel(moog_engine* e, const moog_program_t* prog);

// The digest of the kernel sources and flags this library was built from (moog/_digest.py; -DMOOG_SRC_DIGEST=0x...ull on the
// command line of this unit and of every program-specialised step kernel).  0: built without one -- no specialised kernel loads.
#ifndef MOOG_SRC_DIGEST
#define MOOG_SRC_DIGEST 0ull
#endif
#define MOOG_STR2(x) #x
#define MOOG_STR(x) MOOG_STR2(x)
extern "C" const char moog_src_digest_marker[] = "MOOG_SRC_DIGEST=" MOOG_STR(MOOG_SRC_DIGEST);   // (read as text by moog/_digest.py)

extern "C" {

int moog_abi_version(void) { return MOOG_ABI_VERSION; }
unsigned long long moog_source_digest(void) { return MOOG_SRC_DIGEST; }
const char* moog_last_error(void) { return g_err.c_str(); }
int64_t moog_program_sizeof(void) { return (int64_t)sizeof(moog_program_t); }

static int validate(const moog_program_t* p) {
  if (!p) return fail(MOOG_E_INVALID, "null program");
  if (p->abi_version != MOOG_ABI_VERSION) return fail(MOOG_E_INVALID, "program abi_version mismatch");
  if (p->n_slots < 0 || p->n_slots > MOOG_MAX_SLOTS) return fail(MOOG_E_INVALID, "n_slots out of range");
  if (p->n_layers < 0 || p->n_layers > MOOG_MAX_LAYERS) return fail(MOOG_E_INVALID, "n_layers out of range");
  if (p->n_hdraws < 0 || p->n_hdraws > MOOG_MAX_HDRAWS) return fail(MOOG_E_INVALID, "n_hdraws out of range");
  if (p->updates_per_env_step < 1) return fail(MOOG_E_INVALID, "updates_per_env_step < 1");
  for (int s = 0; s < p->n_slots; ++s)
    if (p->slot_vcap[s] > 128) return fail(MOOG_E_UNSUPPORTED, "sprites with more than 128 vertices");
  // the pairwise forces (collisions, gravity, springs) scan a layer pair's candidates 64 / 128 at a time
  for (int f = 0; f < p->n_forces; ++f) {
    const moog_force_t* F = &p->forces[f];
    if (F->n_b <= 0) continue;
    for (int k = 0; k < F->n_a; ++k)
      if (p->layer_nslots[F->layers_a[k]] > 64 * 2) return fail(MOOG_E_UNSUPPORTED, "layer of a pairwise force too large");
    for (int k = 0; k < F->n_b; ++k)
      if (p->layer_nslots[F->layers_b[k]] > 64 * 2) return fail(MOOG_E_UNSUPPORTED, "layer of a pairwise force too large");
  }
  if (p->maze.random && (p->maze.gen_size < 1 || p->maze.gen_size > MOOG_MAX_MAZE_GEN || p->maze.size < p->maze.gen_size ||
                         p->maze.size > MOOG_MAX_MAZE))
    return fail(MOOG_E_UNSUPPORTED, "random maze size unsupported");
  {
    const int aa = p->render.aa > 1 ? p->render.aa : 1;
    const long long cw = (long long)aa * p->render.width, ch = (long long)aa * p->render.height;
    if (cw < 1 || cw > 8192 || ch < 1 || ch > 8192 || aa > 16)
      return fail(MOOG_E_UNSUPPORTED, "render size unsupported (1 <= anti_aliasing x width, height <= 8192, anti_aliasing <= 16)");
  }
  return MOOG_OK;
}

static KArgs make_args(moog_engine* e, const void* actions, const moog_inject_t* inj,
                       const moog_step_out_t* out, int mode, const uint8_t* mask);
static RArgs raster_args(moog_engine* e, uint8_t* image);

// Pillow Resample.c precompute_coeffs + normalize_coeffs_8bpc for the LANCZOS filter (support 3): the window of
// output sample xx is centred on (xx + 0.5) * scale, weights are normalised in double and rounded to fixed
// point with 22 fractional bits.  Returns the taps per output sample.
static int resize_coeffs(int in_size, int out_size, std::vector<int32_t>& bounds, std::vector<int32_t>& kk) {
  const double scale = (double)in_size / out_size;
  const double filterscale = scale < 1.0 ? 1.0 : scale;
  const double support = 3.0 * filterscale;
  const int ksize = (int)std::ceil(support) * 2 + 1;
  auto sinc = [](double x) { return x == 0.0 ? 1.0 : std::sin(x * M_PI) / (x * M_PI); };
  auto lanczos = [&](double x) { return (-3.0 <= x && x < 3.0) ? sinc(x) * sinc(x / 3) : 0.0; };
  bounds.assign(2 * (size_t)out_size, 0);
  kk.assign((size_t)out_size * ksize, 0);
  std::vector<double> pre((size_t)ksize);
  for (int xx = 0; xx < out_size; ++xx) {
    const double center = 0 + (xx + 0.5) * scale, ss = 1.0 / filterscale;
    double ww = 0.0;
    int xmin = (int)(center - support + 0.5);
    if (xmin < 0) xmin = 0;
    int xmax = (int)(center + support + 0.5);
    if (xmax > in_size) xmax = in_size;
    xmax -= xmin;
    int x;
    for (x = 0; x < xmax; ++x) {
      const double w = lanczos((x + xmin - center + 0.5) * ss);
      pre[x] = w;
      ww += w;
    }
    for (x = 0; x < xmax; ++x)
      if (ww != 0.0) pre[x] /= ww;
    for (; x < ksize; ++x) pre[x] = 0;
    for (x = 0; x < ksize; ++x)
      kk[(size_t)xx * ksize + x] = pre[x] < 0 ? (int)(-0.5 + pre[x] * (1 << 22)) : (int)(0.5 + pre[x] * (1 << 22));
    bounds[2 * xx] = xmin;
    bounds[2 * xx + 1] = xmax;
  }
  return ksize;
}

static int setup_anti_aliasing(moog_engine* e) {
  if (e->aa <= 1) return MOOG_OK;
  const int ow = e->prog.render.width, oh = e->prog.render.height;
  std::vector<int32_t> bh, bv, ch, cv;
  const int kh = resize_coeffs(e->canvas_w, ow, bh, ch), kv = resize_coeffs(e->canvas_h, oh, bv, cv);
  const size_t words = bh.size() + bv.size() + ch.size() + cv.size();
  if (hipMalloc(&e->aa_tables, words * sizeof(int32_t)) != hipSuccess) return fail(MOOG_E_NOMEM, "hipMalloc(resize tables) failed");
  int32_t* d = e->aa_tables;
  HIPCHK(hipMemcpy(d, bh.data(), bh.size() * 4, hipMemcpyHostToDevice)); const int32_t* dbh = d; d += bh.size();
  HIPCHK(hipMemcpy(d, bv.data(), bv.size() * 4, hipMemcpyHostToDevice)); const int32_t* dbv = d; d += bv.size();
  HIPCHK(hipMemcpy(d, ch.data(), ch.size() * 4, hipMemcpyHostToDevice)); const int32_t* dch = d; d += ch.size();
  HIPCHK(hipMemcpy(d, cv.data(), cv.size() * 4, hipMemcpyHostToDevice)); const int32_t* dcv = d;
  int hspan = 0;   // the horizontal pass stages the bytes a block of 256 output columns reads in LDS
  for (int x0 = 0; x0 < ow; x0 += 256) {
    const int xl = (x0 + 255 < ow ? x0 + 255 : ow - 1);
    const int b0 = 3 * bh[2 * x0] & ~3, b1 = 3 * (bh[2 * xl] + bh[2 * xl + 1]);
    if (b1 - b0 > hspan) hspan = b1 - b0;
  }
  hspan = (hspan + 7) & ~3;
  const int tstride = (ow + 3) & ~3;
  e->aa_resize = RResize{e->canvas_w, e->canvas_h, ow, oh, kh, kv, dbh, dbv, dch, dcv, hspan, e->pad_w, tstride};
  // canvases of a chunk of envs at a time: at most 1 GiB of scratch
  const size_t canvas = (size_t)e->pad_w * e->canvas_h * 3;
  size_t chunk = ((size_t)1 << 30) / canvas;
  if (chunk < 1) chunk = 1;
  if (chunk > (size_t)e->n_envs) chunk = (size_t)e->n_envs;
  e->aa_chunk = (int)chunk;
  if (hipMalloc(&e->aa_canvas, chunk * canvas) != hipSuccess ||
      hipMalloc(&e->aa_tmp, chunk * (size_t)e->canvas_h * tstride * 3) != hipSuccess)
    return fail(MOOG_E_NOMEM, "hipMalloc(anti-aliasing canvas) failed");
  return MOOG_OK;
}

// Resets one scratch env (the constant generation ops do not depend on the random stream) and renders
// its static prefix on top of the background colour: the reference record and picture of moog_raster.h.
static int build_static_prefix(moog_engine* e) {
  int nsv = 0;
  const int ns = getenv("MOOG_RASTER_NO_STATIC") ? 0 : static_prefix_slots(&e->prog, &nsv);
  if (ns == 0) return MOOG_OK;
  const size_t fb = (size_t)e->L.f64_per_env * 8, ib = (size_t)e->L.i32_per_env * 4;
  const size_t pb = (size_t)e->pad_w * e->canvas_h * 3;
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
  a.fault_flag = nullptr;   // (faults of the scratch env are nobody's business)
  (e->maze_kernel ? moog_launch_reset_full : moog_launch_reset_plain)(1, e->step_lds, 0, a);
  RArgs r = raster_args(e, e->s_bg);
  r.n_static = ns; r.nsv = nsv; r.build = 1; r.debug_stop = 0;
  moog_raster_launch(r, e->raster_lds, 0);
  e->view = keep; e->n_envs = keep_n;
  HIPCHK(hipGetLastError());
  HIPCHK(hipStreamSynchronize(0));
  e->n_static = ns; e->nsv = nsv;
  return MOOG_OK;
}

// Per-env prefix (RArgs::sbg_env_stride): worth its memory (a frame and a record per env) and its two extra launches per
// call when it covers many sprites of a frame that takes several workgroups -- measured (profiles/r04_env_prefix.txt):
// pacman's 256 x 256 frames (136 walls, 8 tiles) 2.51 -> 1.73 ms per 4096; one-tile frames (maze_zoo, functional_maze)
// lose 0.09 ms to the check and the build launch and gain nothing, their workgroups being bound by fixed costs.
// MOOG_RASTER_ENV_BG=0 turns it off, =1 turns it on whatever the frame size (A/B runs, tests).
// The mask rasteriser's row records per pass (moog_raster_mask_core.h): `cap` of them, at least a canvas height so that any one
// polygon fits, at most one per (polygon, canvas row) and 4096 (the row sort's 12-bit places), and no more than 64 KB of LDS
// leave room for.  ok = 0 when even the smallest plan does not fit.
static void mask_plan_rows(moog_engine* e, int cap) {
  RmSetup& ms = e->mask_setup;
  if (cap > 4096) cap = 4096;
  if (cap > ms.S * e->canvas_h) cap = ms.S * e->canvas_h;
  if (cap < e->canvas_h) cap = e->canvas_h;
  for (;;) {
    rm_plan(ms.S, e->L.TOTV * ms.ncopy, e->pad_w, e->canvas_h, cap, ms.iwords, RM_THREADS / 64, ms.big, &ms.plan, ms.compact);
    if (ms.plan.total <= 64u * 1024u || cap <= e->canvas_h) break;
    cap = cap - 64 > e->canvas_h ? cap - 64 : e->canvas_h;
  }
  ms.cap_rows = cap;
  ms.lds = ms.plan.total;
  { const char* pad = getenv("MOOG_RASTER_LDS_PAD"); if (pad && ms.lds + (uint32_t)atoi(pad) <= 64u * 1024u) ms.lds += (uint32_t)atoi(pad); }  // occupancy experiments
  if (ms.plan.total > 64u * 1024u) ms.ok = 0;
}

// A frame with more polygon rows than row records is drawn in several passes, each a full sweep over the frame: correct, and
// cheaper than it sounds -- what costs is a resident frame less per CU (profiles/r05_raster.txt 5: cleanup's raster launch takes 97 us
// with 192 or 384 records and 1390 us with the 1500 its busiest frames would like).  So frames that want more records say so
// through a host-mapped word and before a later launch the records grow AS FAR AS THAT IS FREE: the same number of frames per
// CU (LDS; registers hold ten, and a plan within 1 KB of losing one counts as losing it).  falling_balls_64's launch 628 -> 471 us.
// They never shrink; the picture does not depend on their number.  Once they are as many as is free the frames stop reporting
// (an atomic on host memory per frame and launch: 8.6 ms per launch when every frame of a batch of 4096 keeps doing it).
static int mask_frames_per_cu(uint32_t lds) {   // by LDS, and by registers: RM_WAVES_PER_SIMD waves on each of four SIMDs, RM_THREADS / 64 waves a frame
  const int n = (int)(160u * 1024u / (lds + 1024u)), by_regs = RM_WAVES_PER_SIMD * 4 / (RM_THREADS / 64);
  return n > by_regs ? by_regs : n;
}
// the most row records that cost no resident frame per CU against the plan in hand (called once, at create)
static int mask_free_rows(moog_engine* e) {
  RmSetup& ms = e->mask_setup;
  const RmSetup first = ms;
  int best = first.cap_rows;
  for (int cap = first.cap_rows + 32; cap <= 4096; cap += 32) {
    ms = first;
    mask_plan_rows(e, cap);
    if (!ms.ok || ms.cap_rows < cap || mask_frames_per_cu(ms.lds) < mask_frames_per_cu(first.lds)) break;
    best = cap;
  }
  ms = first;
  return best;
}
static void mask_rows_grow(moog_engine* e) {
  if (!e->rows_seen || e->raster_rows_fixed || e->mask_setup.cap_rows >= e->mask_free_cap) return;
  const int want = __atomic_load_n(&e->rows_seen[0], __ATOMIC_RELAXED);
  RmSetup& ms = e->mask_setup;
  if (want <= ms.cap_rows) return;
  const RmSetup before = ms;
  int cap = (want + want / 8 + 31) & ~31;
  if (cap > e->mask_free_cap) cap = e->mask_free_cap;
  mask_plan_rows(e, cap);
  if (!ms.ok || ms.cap_rows <= before.cap_rows) ms = before;
}

static int setup_env_prefix(moog_engine* e) {
  const char* sw = getenv("MOOG_RASTER_ENV_BG");
  if ((sw && atoi(sw) == 0) || e->aa > 1) return MOOG_OK;
  const bool forced = sw && atoi(sw) == 1;
  int nsv = 0;
  const int ns = env_prefix_slots(&e->prog, &nsv);
  const size_t frame = (size_t)e->canvas_h * e->pad_w * 3;
  if (ns < 16 || ns < e->n_static + 8 || (size_t)e->n_envs * frame > ((size_t)4 << 30)) return MOOG_OK;
  if (!forced && (e->raster_tiles_x * e->raster_bands < 2 || ns < 32)) return MOOG_OK;
  const size_t n = (size_t)e->n_envs;
  if (hipMalloc(&e->pe_f64, n * e->L.f64_per_env * 8) != hipSuccess || hipMalloc(&e->pe_i32, n * e->L.i32_per_env * 4) != hipSuccess ||
      hipMalloc(&e->pe_bg, n * frame) != hipSuccess || hipMalloc(&e->pe_valid, n * 4) != hipSuccess ||
      hipMalloc(&e->pe_build, n * 4) != hipSuccess ||
      hipHostMalloc(reinterpret_cast<void**>(&e->pe_min), sizeof(int32_t), hipHostMallocMapped | hipHostMallocCoherent) != hipSuccess) {
    (void)hipGetLastError();   // no room: the frames are drawn without it
    return MOOG_OK;
  }
  HIPCHK(hipMemset(e->pe_valid, 0, n * 4));
  *e->pe_min = INT32_MAX;
  e->pe_ns = ns; e->pe_nsv = nsv;
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
  {
    const std::vector<FOp> fops = moog_flatten_forces(prog);
    e->n_fops = (int32_t)fops.size();
    err = hipMalloc(&e->d_fops, (fops.size() + 1) * sizeof(FOp));
    if (err == hipSuccess && !fops.empty()) err = hipMemcpy(e->d_fops, fops.data(), fops.size() * sizeof(FOp), hipMemcpyHostToDevice);
    if (err != hipSuccess) { free_engine(e); return fail(MOOG_E_NOMEM, "force list"); }
  }
  {
    const StepVariant sv = step_variant_of(prog);
    e->step_lds = sv.step_lds; e->xstack_off = sv.xstack_off;
  }
  { const char* w = getenv("MOOG_WATCH");   // section sampling (moog_engine_read_watch): two words of LDS for the watcher
    if (w && atoi(w) == 1) {
      e->step_lds = (e->step_lds + 15) & ~(size_t)15;
      e->watch_off = (int32_t)e->step_lds;
      e->step_lds += 16;
      if (hipMalloc(&e->watch, (size_t)n_envs * MOOG_WATCH_SECTIONS * sizeof(int32_t)) != hipSuccess ||
          hipMemset(e->watch, 0, (size_t)n_envs * MOOG_WATCH_SECTIONS * sizeof(int32_t)) != hipSuccess) {
        free_engine(e);
        return fail(MOOG_E_NOMEM, "hipMalloc(watch) failed");
      }
    } }
  if (e->step_lds > 160 * 1024) {
    free_engine(e);
    return fail(MOOG_E_UNSUPPORTED, "state record does not fit in 160 KB of LDS");
  }
  // raster LDS plan (moog_raster.h): row records for `chunk` rows per pass
  {
    // Tiles: the row masks of the kernel are 128 bits, so a wider canvas is cut into columns of the widest
    // multiple of 16 <= 128 that divides the width, and a taller one into bands of 64 rows.
    e->aa = prog->render.aa > 1 ? prog->render.aa : 1;
    e->canvas_w = e->aa * prog->render.width;
    e->canvas_h = e->aa * prog->render.height;
    e->pad_w = (e->canvas_w + 15) & ~15;
    int tw = e->pad_w <= 128 ? e->pad_w : 128;
    while (e->pad_w % tw != 0) tw -= 16;
    e->raster_tile_w = tw;
    e->raster_tiles_x = e->pad_w / tw;
    e->raster_band_h = e->canvas_h <= 128 ? e->canvas_h : 64;
    {   // frames that will use the per-env prefix (setup_env_prefix): few sprites are left to draw per tile and the per-tile fixed
        // cost dominates -- whole-height tiles measured 1.65 against 1.73 ms per 4096 pacman frames (profiles/r04_env_prefix.txt)
        // (decided here, from what the PROGRAM allows: the tile plan sizes every LDS table.  A handle that later runs without the prefix --
        //  allocation failure, a prefix that shrank to nothing, a colour override -- keeps the whole-height tiles: 1.65 ms against 1.73 ms
        //  with 64-row bands WITH the prefix, 2.45 against 2.51 ms WITHOUT it in the same file: not a loss either way)
      int nsv_ = 0;
      const char* sw = getenv("MOOG_RASTER_ENV_BG");
      if (e->canvas_h > 128 && e->canvas_h <= 256 && e->aa <= 1 && !(sw && atoi(sw) == 0) && env_prefix_slots(prog, &nsv_) >= 32)
        e->raster_band_h = e->canvas_h;
    }
    { const char* bh = getenv("MOOG_RASTER_BAND_H"); if (bh && atoi(bh) >= 16 && atoi(bh) <= e->canvas_h) e->raster_band_h = atoi(bh); }   // experiments
    e->raster_bands = (e->canvas_h + e->raster_band_h - 1) / e->raster_band_h;
    int W = e->raster_tile_w, H = e->raster_band_h;   // (the LDS plan is per tile)
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
  {   // mask rasteriser (moog_raster_mask_core.h): one-tile frames, polygons of <= 128 vertices, at most 256 polygons (a torus has nine per sprite)
    RmSetup& ms = e->mask_setup;
    memset(&ms, 0, sizeof ms);
    int maxv = 1;
    for (int sl = 0; sl < prog->n_slots; ++sl) if (prog->slot_vcap[sl] > maxv) maxv = prog->slot_vcap[sl];
    const int ncopy = prog->render.polymod == MOOG_POLYMOD_TORUS ? 9 : 1;
    const char* sw = getenv("MOOG_RASTER_MASK");   // 0: the push / sort / span kernel for every frame (A/B runs, tests)
    ms.ok = !(sw && atoi(sw) == 0) && e->raster_tiles_x * e->raster_bands == 1 && e->pad_w <= 128 && e->canvas_h <= 128 &&
            maxv <= RM_BIG_NV && prog->n_slots >= 1 && prog->n_slots * ncopy <= 256 && e->L.TOTV >= 1;
    if (ms.ok) {
      ms.slots = prog->n_slots; ms.ncopy = ncopy; ms.big = maxv > RM_MAX_NV ? 1 : 0;
      ms.S = prog->n_slots * ncopy;
      ms.iwords = (ms.S + 31) / 32;
      ms.cmap = prog->render.cmap;
      ms.first_person = prog->render.polymod == MOOG_POLYMOD_FIRST_PERSON ? 1 : 0;
      if (ms.first_person) { ms.fp_slot0 = prog->layer_slot0[prog->render.polymod_layer]; ms.fp_nslots = prog->layer_nslots[prog->render.polymod_layer]; }
      ms.bg = ((uint32_t)prog->render.bg[0] & 255u) | (((uint32_t)prog->render.bg[1] & 255u) << 8) | (((uint32_t)prog->render.bg[2] & 255u) << 16);
      // row records per pass: 192 (the headline workload's frames have ~170 rows behind the cached walls), at least a
      // canvas height so that any one polygon fits; frames with more rows take several passes
      // (frames that want more keep the records growing: mask_rows_grow, before a launch)
      int cap = 192;
      { const char* rc = getenv("MOOG_RASTER_ROWS"); if (rc && atoi(rc) >= 1) { cap = atoi(rc); e->raster_rows_fixed = 1; } }   // tuning / tests of the multi-pass path
      mask_plan_rows(e, cap);
      {   // 4-byte edge records when the 16-byte ones keep frames off a CU (or do not fit at all): MOOG_RASTER_COMPACT=0 / 1 forces
        const char* cs = getenv("MOOG_RASTER_COMPACT");
        const RmSetup full = ms;
        ms.compact = 1; ms.ok = 1;
        mask_plan_rows(e, cap);
        // (the 4-byte records cost the rows phase a tenth more instructions -- headline workload 49.5 -> 53 us when they were picked for
        //  it by mistake -- so they are for programs whose 16-byte records leave a CU clearly short of the frames its registers allow)
        const int by_regs = RM_WAVES_PER_SIMD * 4 / (RM_THREADS / 64);
        const bool better = ms.ok && (!full.ok || ((int)(160u * 1024u / full.lds) < by_regs - 1 && mask_frames_per_cu(ms.lds) > mask_frames_per_cu(full.lds)));
        if (cs ? atoi(cs) == 0 : !better) ms = full;
      }
      e->mask_free_cap = (ms.ok && !e->raster_rows_fixed) ? mask_free_rows(e) : ms.cap_rows;
      if (ms.ok) {   // a draw record per env (moog_draw_record.h): header + an item per slot and copy + every vertex slot's point and owner byte
        e->draw_lay = rm_draw_layout(ms.S, e->L.TOTV * ms.ncopy);
        if (hipMalloc(&e->draw, (size_t)e->n_envs * e->draw_lay.stride) != hipSuccess) {
          free_engine(e);
          return fail(MOOG_E_NOMEM, "out of device memory (draw records)");
        }
        const char* ds = getenv("MOOG_DRAW_IN_STEP");
        e->draw_in_step = !(ds && atoi(ds) == 0);
        { int cus = 0; if (hipDeviceGetAttribute(&cus, hipDeviceAttributeMultiprocessorCount, e->device) == hipSuccess && cus > 0) e->n_cus = cus; }
        { const char* ps = getenv("MOOG_RASTER_PERSIST"); if (ps) e->raster_persist = atoi(ps); }
      }
    }
  }
  {
    int (*const configure[6])(size_t) = {moog_configure_step_f3, moog_configure_step_f4, moog_configure_step_t3,
                                         moog_configure_step_t4, moog_configure_step_m3, moog_configure_step_m4};
    for (int v = 0; v < 6 && err == hipSuccess; ++v) err = (hipError_t)configure[v](e->step_lds);
    if (err == hipSuccess) err = (hipError_t)moog_configure_step_f2(e->step_lds);
    {   // MOOG_STEP_PRIO="a,b,c": per mille of the launch order that runs at wave priority 3 / >= 2 / >= 1 ("0": off)
      const char* pr = getenv("MOOG_STEP_PRIO");
      int a = 0, b = 0, c = 0;
      if (pr && sscanf(pr, "%d,%d,%d", &a, &b, &c) >= 1) {
        if (b < a) b = a;
        if (c < b) c = b;
        e->prio_pm[0] = a; e->prio_pm[1] = b; e->prio_pm[2] = c;
      }
    }
  }
  {
    const StepVariant sv = step_variant_of(prog);
    e->dynamic_rules = sv.dynamic_rules; e->maze_kernel = sv.maze_kernel; e->late_reset = sv.late_reset; e->step_wps = sv.wps;
  }
  if (!e->watch) load_spec_kernel(e, prog);
  if (e->late_reset) {
    if (hipMalloc(&e->late_mask, (size_t)n_envs) != hipSuccess || hipMemset(e->late_mask, 0, (size_t)n_envs) != hipSuccess) {
      free_engine(e);
      return fail(MOOG_E_NOMEM, "hipMalloc(late reset mask) failed");
    }
  }
  if (err == hipSuccess)
    err = (hipError_t)moog_configure_reset_plain(e->step_lds);
    if (err == hipSuccess) err = (hipError_t)moog_configure_reset_full(e->step_lds);
  if (err == hipSuccess)
    err = (hipError_t)moog_raster_configure(e->raster_lds);
  if (err == hipSuccess && e->mask_setup.ok) err = (hipError_t)moog_raster_configure_mask(64 * 1024);   // (the records may grow: mask_rows_grow)
  if (err != hipSuccess) {
    free_engine(e);
    return fail(MOOG_E_HIP, std::string("hipFuncSetAttribute: ") + hipGetErrorString(err));
  }
  { const char* ds = getenv("MOOG_STEP_DEBUG"); e->step_dbg = ds ? atoi(ds) : 0; }
  { const char* ds = getenv("MOOG_RASTER_STOP"); e->raster_stop = ds ? atoi(ds) : 0; }
  // (coherent = fine-grained: the kernels' system-scope atomics and the host's atomic read / clear meet in the same memory)
  if (hipHostMalloc(reinterpret_cast<void**>(&e->rows_seen), 2 * sizeof(int32_t), hipHostMallocMapped | hipHostMallocCoherent) == hipSuccess) { e->rows_seen[0] = 0; e->rows_seen[1] = 0; }   // ([1]: how many frames wanted more: for tools)
  else e->rows_seen = nullptr;   // (without it the records keep their first size)
  if (hipHostMalloc(reinterpret_cast<void**>(&e->fault_flag), sizeof(int32_t), hipHostMallocMapped | hipHostMallocCoherent) != hipSuccess) {
    free_engine(e);
    return fail(MOOG_E_NOMEM, "hipHostMalloc(fault flag) failed");
  }
  *e->fault_flag = 0;
  for (int l = 0; l < prog->n_layers; ++l)
    if (prog->layer_dynamic[l] && !e->layer_hw) {
      if (hipMalloc(&e->layer_hw, 2 * MOOG_MAX_LAYERS * sizeof(int32_t)) != hipSuccess ||
          hipMemset(e->layer_hw, 0, 2 * MOOG_MAX_LAYERS * sizeof(int32_t)) != hipSuccess) {
        free_engine(e);
        return fail(MOOG_E_NOMEM, "hipMalloc(layer usage) failed");
      }
    }
  int rc2 = build_static_prefix(e);
  if (rc2 == MOOG_OK) rc2 = setup_anti_aliasing(e);
  if (rc2 == MOOG_OK) rc2 = setup_env_prefix(e);
  if (rc2 == MOOG_OK && e->aa <= 1 && e->pad_w != e->canvas_w &&
      hipMalloc(&e->pad_img, (size_t)n_envs * e->canvas_h * e->pad_w * 3) != hipSuccess)
    rc2 = fail(MOOG_E_NOMEM, "hipMalloc(16-aligned frames) failed");
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
  void* spec = e->spec_handle;
  free_engine(e);
  (void)spec;   // (the object stays mapped: unloading a code object that a stream may still reference is not worth the risk)
  return MOOG_OK;
}

int moog_engine_layout(const moog_engine_t* e, moog_layout_t* out) {
  if (!e || !out) return fail(MOOG_E_INVALID, "null argument");
  *out = e->L;
  return MOOG_OK;
}

static int pool_drop(moog_engine* e, hipStream_t s);

int moog_engine_load_state(moog_engine_t* e, const moog_state_view_t* view) {
  if (!e || !view || !view->f64 || !view->i32) return fail(MOOG_E_INVALID, "null state view");
  if (((uintptr_t)view->f64 & 15) || ((uintptr_t)view->i32 & 15))
    return fail(MOOG_E_INVALID, "state buffers must be 16-byte aligned");
  e->view = *view;
  if (e->pool_on) {   // other records: whatever the pool holds was built from the old ones
    HIPCHK(hipSetDevice(e->device));
    const int rc = pool_drop(e, 0);
    if (rc) return rc;
    HIPCHK(hipStreamSynchronize(0));
  }
  return MOOG_OK;
}

// (an event pair costs ~5 us of stream time: `period` > 1 samples the launches instead of bracketing all of them)
static bool sampled(moog_engine* e, int id) {
  const int period = ((e->timing >> 8) & 255) + 1;
  return ((e->timing >> id) & 1) && (e->timed[id].seq++ % period) == 0;
}

struct Bracket {
  moog_engine* e; int id; hipStream_t s; hipEvent_t a = nullptr, b = nullptr;
  Bracket(moog_engine* e_, int id_, hipStream_t s_, int on = -1) : e(e_), id(id_), s(s_) {   // on: -1 = sample, else decided
    if (on < 0 ? sampled(e, id) : on != 0) { hipEventCreate(&a); hipEventCreate(&b); hipEventRecord(a, s); }
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
  a.perm = (mode == MODE_STEP && e->perm) ? e->perm : nullptr;
  a.cost = (mode == MODE_STEP) ? e->cost : nullptr;
  a.dbg = e->step_dbg;
  a.fault_flag = e->fault_flag;
  a.layer_hw = e->layer_hw;
  a.act_f32 = e->act_f32;
  a.xstack_off = e->xstack_off;
  a.watch = (mode == MODE_STEP) ? e->watch : nullptr; a.watch_off = e->watch_off;
  a.fops = e->d_fops; a.n_fops = e->n_fops;
  const bool pool = e->pool_on && mode == MODE_STEP && !(inj && inj->uniforms);
  a.pool_state = pool ? e->pool_state : nullptr; a.pool_tag = e->pool_tag; a.pool_stats = e->pool_stats;
  a.pool_lock = e->pool_lock; a.pool_depth = e->pool_depth;
  for (int k = 0; k < 2; ++k) { a.pool_f64[k] = e->pool_f64[k]; a.pool_i32[k] = e->pool_i32[k]; }
  a.live_f64 = nullptr; a.live_i32 = nullptr;
  a.late_mask = (mode == MODE_STEP && e->late_reset) ? e->late_mask : nullptr;
  for (int k = 0; k < 3; ++k) a.prio_t[k] = (int32_t)(((int64_t)e->prio_pm[k] * e->n_envs + 999) / 1000);
  a.rank0 = 0;
  memset(&a.draw, 0, sizeof a.draw);   // (moog_engine_step turns the draw records on)
  a.draw_vinfo = e->d_vinfo;
  return a;
}

static void launch_step(moog_engine* e, hipStream_t s, const KArgs& a) {
  static const moog_step_launch_fn launch[6] = {moog_launch_step_f3, moog_launch_step_f4, moog_launch_step_t3,
                                                moog_launch_step_t4, moog_launch_step_m3, moog_launch_step_m4};
  if (e->spec_launch) {
    e->spec_launch(e->n_envs, e->step_lds, s, &a);
    return;
  }
  const bool full = e->maze_kernel && !e->late_reset;
  if (e->step_wps == 2 && !full && !e->dynamic_rules) { moog_launch_step_f2(e->n_envs, e->step_lds, s, a); return; }
  launch[(full ? 4 : (e->dynamic_rules ? 2 : 0)) + (e->step_wps == 4 ? 1 : 0)](e->n_envs, e->step_lds, s, a);
}

// What the draw-record emitter needs (moog_draw_record.h): by value in the step kernel's and the derive kernel's arguments.
// out = null when this engine's frames are not the mask rasteriser's.
static RmEmit emit_args(moog_engine* e) {
  RmEmit m;
  memset(&m, 0, sizeof m);
  const RmSetup& ms = e->mask_setup;
  if (!ms.ok || !e->draw) return m;
  m.out = e->draw; m.lay = e->draw_lay;
  m.S = ms.S; m.slots = ms.slots; m.ncopy = ms.ncopy; m.W = e->pad_w; m.H = e->canvas_h; m.scale_w = e->canvas_w;
  m.cmap = ms.cmap; m.first_person = ms.first_person; m.fp_slot0 = ms.fp_slot0; m.fp_nslots = ms.fp_nslots;
  m.n_static = (e->s_f64 && ms.ncopy == 1) ? e->n_static : 0;
  m.sref_v = e->s_f64 ? e->s_f64 + e->L.o_verts : nullptr;
  m.sref_col = e->s_f64 ? e->s_f64 + e->L.o_color : nullptr;
  m.sref_flags = e->s_i32 ? e->s_i32 + e->L.o_flags : nullptr;
  m.sref_nv = e->s_i32 ? e->s_i32 + e->L.o_nverts : nullptr;
  m.sref_opa = e->s_i32 ? e->s_i32 + e->L.o_opacity : nullptr;
  m.rgb_override = e->rgb_override;
  return m;
}
// The step kernel writes the draw records of the frames the raster launch behind it draws (moog_engine_step with an image):
// when those frames are the mask rasteriser's, drawn in one launch over every env (a late-reset program's episodes are opened by
// the reset kernel behind the step kernel: that launch writes the draw records of the envs it resets).
static bool step_emits_draw(moog_engine* e) {
  // (the emitter's scratch in the step kernel's LDS, emit_draw_record: a torus's nine items per slot must fit behind the vertex offsets)
  const bool scratch_fits = e->mask_setup.ncopy == 1 ||
      4u * (size_t)RM_EMIT_SCRATCH_WORDS(e->mask_setup.slots, e->mask_setup.S, e->mask_setup.ncopy) <= (size_t)CAND_CAP * 2 + 128 + 64 * 8;
  return e->draw_in_step && e->mask_setup.ok && e->draw && e->pe_ns <= 0 && e->aa <= 1 && scratch_fits;
}

static RArgs raster_args(moog_engine* e, uint8_t* image) {
  RArgs r;
  if (e->mask_setup.ok) mask_rows_grow(e);
  r.ms = e->mask_setup;
  // (frames report only while the records can still grow: every report is an atomic on host memory)
  r.rows_seen = (e->raster_rows_fixed || e->mask_setup.cap_rows >= e->mask_free_cap) ? nullptr : e->rows_seen;
  r.P = e->d_prog; r.L = e->L; r.f64 = e->view.f64; r.i32 = e->view.i32; r.image = image;
  r.vinfo = e->d_vinfo; r.plan = e->raster_plan_;
  r.n_envs = e->n_envs; r.chunk = e->raster_chunk; r.words = e->raster_words;
  r.tile_w = e->raster_tile_w; r.band_h = e->raster_band_h; r.tiles_x = e->raster_tiles_x; r.bands = e->raster_bands;
  r.canvas_w = e->pad_w; r.scale_w = e->canvas_w; r.canvas_h = e->canvas_h; r.flip = e->aa > 1 ? 0 : 1;
  r.iwords = e->raster_iwords; r.hwords = e->raster_hwords; r.xxcap = e->raster_xxcap;
  r.debug_stop = e->raster_stop;
  r.n_static = e->n_static; r.nsv = e->nsv; r.build = 0;
  r.sref_v = e->s_f64 ? e->s_f64 + e->L.o_verts : nullptr;
  r.sref_col = e->s_f64 ? e->s_f64 + e->L.o_color : nullptr;
  r.sref_flags = e->s_i32 ? e->s_i32 + e->L.o_flags : nullptr;
  r.sref_nv = e->s_i32 ? e->s_i32 + e->L.o_nverts : nullptr;
  r.sref_opa = e->s_i32 ? e->s_i32 + e->L.o_opacity : nullptr;
  r.sbg = e->s_bg;
  r.sbg_env_stride = 0; r.env_build = nullptr; r.rgb_override = e->rgb_override;
  r.em = emit_args(e); r.draw_ready = 0; r.env0 = 0;
  {   // (one resident round: the frames a CU holds at once by LDS and registers, or fewer when asked)
    const int fit = mask_frames_per_cu(r.ms.lds);
    r.ms.persist_slots = e->raster_persist > 0 ? e->n_cus * (e->raster_persist < fit ? e->raster_persist : fit) : 0;
  }
  return r;
}

// Per-env prefix: validates every env's picture against its live record, draws the stale ones again, and points the frame
// launch's arguments at the pictures.  (The launches that do not pass through here --
// anti-aliased canvases -- draw every sprite; the next launch that does re-validates, so nothing goes stale unseen.)
static int use_env_prefix(moog_engine* e, RArgs& r, hipStream_t s) {
  if (e->pe_ns <= 0) return MOOG_OK;
  const int seen = __atomic_load_n(e->pe_min, __ATOMIC_RELAXED);
  if (seen < e->pe_ns) {   // slots from `seen` on do change within episodes: they leave the prefix, every picture is drawn again
    e->pe_ns = seen < 8 ? 0 : seen;
    e->pe_nsv = e->pe_ns > 0 ? e->prog.slot_voff[e->pe_ns] : 0;
    __atomic_store_n(e->pe_min, INT32_MAX, __ATOMIC_RELAXED);
    if (e->pe_ns <= 0) return MOOG_OK;
    HIPCHK(hipMemsetAsync(e->pe_valid, 0, sizeof(int32_t) * (size_t)e->n_envs, s));
  }
  PCArgs c;
  c.P = e->d_prog; c.L = e->L; c.f64 = e->view.f64; c.i32 = e->view.i32; c.s_f64 = e->pe_f64; c.s_i32 = e->pe_i32;
  c.valid = e->pe_valid; c.build = e->pe_build; c.min_changed = e->pe_min; c.n_envs = e->n_envs; c.n_static = e->pe_ns;
  moog_prefix_check_launch(c, s);
  RArgs b = r;
  b.image = e->pe_bg; b.n_static = e->pe_ns; b.nsv = e->pe_nsv; b.build = 1; b.env_build = e->pe_build; b.debug_stop = 0;
  b.sbg = nullptr; b.sbg_env_stride = 0;
  moog_raster_launch(b, e->raster_lds, s);
  r.n_static = e->pe_ns; r.nsv = e->pe_nsv; r.sbg = e->pe_bg;
  r.sbg_env_stride = (size_t)e->canvas_h * e->pad_w * 3;
  return MOOG_OK;
}

static int launch_raster(moog_engine* e, uint8_t* image, hipStream_t s, int timed = -1, bool draw_ready = false) {
  RArgs r = raster_args(e, image);
  r.draw_ready = draw_ready ? 1 : 0;
  Bracket br(e, MOOG_K_RASTER, s, timed);
  if (e->aa <= 1) { const int rc = use_env_prefix(e, r, s); if (rc) return rc; }
  if (e->aa <= 1 && e->pad_w != e->canvas_w) {   // drawn 16-aligned, cropped into the caller's frames
    r.image = e->pad_img;
    moog_raster_launch(r, e->raster_lds, s);
    moog_crop_launch(e->pad_img, image, (size_t)e->n_envs * e->canvas_h, e->pad_w * 3, e->canvas_w * 3, s);
  } else if (e->aa <= 1) {
    moog_raster_launch(r, e->raster_lds, s);
  } else {   // pil_renderer.py:111-112: draw on the large canvas, then Image.resize(LANCZOS); a chunk of envs at a time
    const size_t frame = (size_t)e->prog.render.width * e->prog.render.height * 3;
    for (int e0 = 0; e0 < e->n_envs; e0 += e->aa_chunk) {
      const int n = e->n_envs - e0 < e->aa_chunk ? e->n_envs - e0 : e->aa_chunk;
      RArgs c = r;
      c.f64 = r.f64 + (size_t)e0 * e->L.f64_per_env;
      c.i32 = r.i32 + (size_t)e0 * e->L.i32_per_env;
      c.image = e->aa_canvas;
      c.n_envs = n;
      c.env0 = e0; c.draw_ready = 0;
      moog_raster_launch(c, e->raster_lds, s);
      moog_resize_launch(e->aa_resize, e->aa_canvas, e->aa_tmp, image + (size_t)e0 * frame, n, s);
    }
  }
  HIPCHK(hipGetLastError());
  return MOOG_OK;
}

// Reset pool: one fill launch behind the call that has just been enqueued on `s`, on the side streams in turn.  A fill
// serves every env whose pool is empty when it STARTS (not just this call's), so of the launches that pile up on a
// stream behind a fill that is under way (~10 ms of look-ahead per env; calls come every fraction of a millisecond) the
// first does the work and the rest are grids of early exits: each stream always has the next fill ready to start, whether
// the caller synchronises every call or runs hundreds of calls ahead of the device.
// The fill reads the live records while later calls step them: see pool_adopt for why that is sound.
static int pool_kick(moog_engine* e, hipStream_t s) {
  hipStream_t ps = e->pool_stream[e->pool_fills % e->pool_streams];
  HIPCHK(hipEventRecord(e->ev_pool, s));
  HIPCHK(hipStreamWaitEvent(ps, e->ev_pool, 0));
  KArgs a = make_args(e, nullptr, nullptr, nullptr, MODE_FILL, nullptr);
  a.live_f64 = e->view.f64; a.live_i32 = e->view.i32;
  a.f64 = e->pool_f64[1]; a.i32 = e->pool_i32[1];
  a.pool_state = e->pool_state;
  a.fault_flag = nullptr;   // (a fault of the pool's record reaches the host when the record is adopted and stored)
  a.dbg = 0;
  moog_launch_reset_full(e->n_envs * e->pool_depth, e->step_lds, ps, a);   // one workgroup per (record, env)
  HIPCHK(hipGetLastError());
  ++e->pool_fills;
  return MOOG_OK;
}

// Late reset (step_env): behind the step kernel of a program stepped by the kernels without the rare components, the full
// reset kernel opens the episodes of the envs that kernel marked (a grid of early exits when there are none).
static int late_reset_launch(moog_engine* e, const moog_inject_t* inject, const moog_step_out_t* out, hipStream_t s, const RmEmit* draw) {
  KArgs b = make_args(e, nullptr, inject, out, MODE_RESET_MASK, nullptr);
  if (draw) b.draw = *draw;   // (the step launch wrote draw records: the envs this launch resets get theirs from it)
  b.late_mask = e->late_mask;
  b.pool_state = (e->pool_on && !(inject && inject->uniforms)) ? e->pool_state : nullptr;   // (the step kernel took the env's lock then)
  moog_launch_reset_full(e->n_envs, e->step_lds, s, b);
  HIPCHK(hipGetLastError());
  return MOOG_OK;
}

// every pool record is dropped (the host reset the envs, or handed other records over): fills under way finish first
static int pool_drop(moog_engine* e, hipStream_t s) {
  for (int k = 0; k < e->pool_streams; ++k) HIPCHK(hipStreamSynchronize(e->pool_stream[k]));
  HIPCHK(hipMemsetAsync(e->pool_state, 0, sizeof(int32_t) * (size_t)e->n_envs * e->pool_depth, s));
  HIPCHK(hipMemsetAsync(e->pool_lock, 0, sizeof(int32_t) * (size_t)e->n_envs, s));
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
  if (e->pool_on && (rc = pool_drop(e, s)) != MOOG_OK) return rc;
  {
    Bracket br(e, MOOG_K_RESET, s);
    (e->maze_kernel ? moog_launch_reset_full : moog_launch_reset_plain)(e->n_envs, e->step_lds, s, a);
  }
  HIPCHK(hipGetLastError());
  if (e->pool_on && !(inject && inject->uniforms) && (rc = pool_kick(e, s)) != MOOG_OK) return rc;
  if (out && out->image) return launch_raster(e, out->image, s);
  return MOOG_OK;
}

int moog_engine_step(moog_engine_t* e, const void* actions_dev, const moog_inject_t* inject,
                     const moog_step_out_t* out, void* hip_stream) {
  int rc = ready(e);
  if (rc) return rc;
  if (!actions_dev) return fail(MOOG_E_INVALID, "null actions");
  hipStream_t s = (hipStream_t)hip_stream;
  // (envs whose episode ended in the previous call are reset inside the step kernel, environment.py:100-101)
  KArgs a = make_args(e, actions_dev, inject, out, MODE_STEP, nullptr);
  const bool emit = out && out->image && step_emits_draw(e);
  if (emit) a.draw = emit_args(e);
  if (e->sched_pending) {   // the order computed from the previous step's costs
    HIPCHK(hipStreamWaitEvent(s, e->ev_sched_done, 0));
    e->sched_pending = false;
  }
  {
    Bracket br(e, MOOG_K_STEP, s);
    launch_step(e, s, a);
    if (e->late_reset && (rc = late_reset_launch(e, inject, out, s, emit ? &a.draw : nullptr)) != MOOG_OK) return rc;
  }
  HIPCHK(hipGetLastError());
  if (a.pool_state && (rc = pool_kick(e, s)) != MOOG_OK) return rc;
  if (e->perm && e->cost) {
    HIPCHK(hipEventRecord(e->ev_step_done, s));
    HIPCHK(hipStreamWaitEvent(e->sched_stream, e->ev_step_done, 0));
    moog_launch_sched(e->sched_stream, e->cost, e->perm, e->n_envs,
                      e->view.i32 + e->L.o_reset_next, e->L.i32_per_env);
    HIPCHK(hipEventRecord(e->ev_sched_done, e->sched_stream));
    e->sched_pending = true;
  }
  if (out && out->image) return launch_raster(e, out->image, s, -1, emit);
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
  if (perm_dev && cost_dev) {   // the step kernel reads the costs (a moving average): they start from zero, whatever the caller's memory held
    HIPCHK(hipSetDevice(e->device));
    HIPCHK(hipMemset(cost_dev, 0, (size_t)e->n_envs * sizeof(float)));
  }
  if (perm_dev && cost_dev && !e->sched_stream) {
    HIPCHK(hipSetDevice(e->device));
    HIPCHK(hipStreamCreateWithFlags(&e->sched_stream, hipStreamNonBlocking));
    HIPCHK(hipEventCreateWithFlags(&e->ev_step_done, hipEventDisableTiming));
    HIPCHK(hipEventCreateWithFlags(&e->ev_sched_done, hipEventDisableTiming));
  }
  return MOOG_OK;
}

int moog_engine_set_reset_pool(moog_engine_t* e, int32_t enabled) {
  if (!e) return fail(MOOG_E_INVALID, "null engine");
  HIPCHK(hipSetDevice(e->device));
  if (!enabled) {
    if (e->pool_on) { const int rc = pool_drop(e, 0); if (rc) return rc; HIPCHK(hipStreamSynchronize(0)); }
    e->pool_on = false;
    return MOOG_OK;
  }
  if (!e->maze_kernel)
    return fail(MOOG_E_UNSUPPORTED, "the reset pool lives in the kernels that carry every component; this program runs the plain ones (its resets are cheap)");
  for (int o = 0; o < e->prog.n_ops; ++o)
    if (e->prog.ops[o].cell_sel == MOOG_CELL_PSTATE)
      return fail(MOOG_E_UNSUPPORTED, "the reset pool cannot serve a program whose initializer keeps a number across episodes (MOOG_CELL_PSTATE): its resets depend on how the previous episode went");
  {
    const char* ser = getenv("AMD_SERIALIZE_KERNEL");
    const char* blk = getenv("HIP_LAUNCH_BLOCKING");
    const char* hwq = getenv("GPU_MAX_HW_QUEUES");
    const char* cc = getenv("ROCPROF_COUNTER_COLLECTION");
    if ((ser && atoi(ser)) || (blk && atoi(blk)) || (hwq && atoi(hwq) == 1) || (cc && atoi(cc)))
      return fail(MOOG_E_UNSUPPORTED, "the reset pool needs kernels to run beside each other (AMD_SERIALIZE_KERNEL / HIP_LAUNCH_BLOCKING / GPU_MAX_HW_QUEUES=1 / counter collection serialise them)");
  }
  if (!e->pool_ready) {
    { const char* pd = getenv("MOOG_POOL_DEPTH"); if (pd && atoi(pd) >= 1 && atoi(pd) <= 4) e->pool_depth = atoi(pd); }   // experiments
    const size_t n = (size_t)e->n_envs * e->pool_depth;
    if (!e->pool_state) HIPCHK(hipMalloc(&e->pool_state, sizeof(int32_t) * n));
    if (!e->pool_tag) HIPCHK(hipMalloc(&e->pool_tag, sizeof(int32_t) * n));
    if (!e->pool_lock) HIPCHK(hipMalloc(&e->pool_lock, sizeof(int32_t) * (size_t)e->n_envs));
    if (!e->pool_stats) { HIPCHK(hipMalloc(&e->pool_stats, 4 * sizeof(unsigned long long))); HIPCHK(hipMemset(e->pool_stats, 0, 4 * sizeof(unsigned long long))); }
    for (int k = 0; k < 2; ++k) {
      if (!e->pool_f64[k]) HIPCHK(hipMalloc(&e->pool_f64[k], sizeof(double) * n * (size_t)e->L.f64_per_env));
      if (!e->pool_i32[k]) HIPCHK(hipMalloc(&e->pool_i32[k], sizeof(int32_t) * n * (size_t)e->L.i32_per_env));
    }
    {   // streams that share a hardware queue with the caller's would put its step kernels behind a fill
      const char* hwq = getenv("GPU_MAX_HW_QUEUES");
      const int queues = (hwq && atoi(hwq) > 0) ? atoi(hwq) : 4;
      e->pool_streams = queues - 2 < 1 ? 1 : (queues - 2 > moog_engine::POOL_STREAMS ? moog_engine::POOL_STREAMS : queues - 2);
      const char* ps = getenv("MOOG_POOL_STREAMS");   // experiments
      if (ps && atoi(ps) > 0 && atoi(ps) <= moog_engine::POOL_STREAMS) e->pool_streams = atoi(ps);
    }
    for (int k = 0; k < e->pool_streams; ++k)
      if (!e->pool_stream[k]) HIPCHK(hipStreamCreateWithFlags(&e->pool_stream[k], hipStreamNonBlocking));
    if (!e->ev_pool) HIPCHK(hipEventCreateWithFlags(&e->ev_pool, hipEventDisableTiming));
    e->pool_ready = true;
  }
  HIPCHK(hipMemset(e->pool_state, 0, sizeof(int32_t) * (size_t)e->n_envs * e->pool_depth));
  HIPCHK(hipMemset(e->pool_tag, 0, sizeof(int32_t) * (size_t)e->n_envs * e->pool_depth));
  HIPCHK(hipMemset(e->pool_lock, 0, sizeof(int32_t) * (size_t)e->n_envs));
  e->pool_on = true;
  return MOOG_OK;
}

int moog_engine_get_reset_pool(moog_engine_t* e, int32_t* enabled, int64_t* stats) {
  if (!e) return fail(MOOG_E_INVALID, "null engine");
  if (enabled) *enabled = e->pool_on ? 1 : 0;
  if (stats) {
    stats[0] = e->pool_fills;
    for (int k = 1; k < 5; ++k) stats[k] = 0;
    if (e->pool_stats) {   // (synchronises the device)
      unsigned long long h[4];
      HIPCHK(hipSetDevice(e->device));
      HIPCHK(hipDeviceSynchronize());
      HIPCHK(hipMemcpy(h, e->pool_stats, sizeof(h), hipMemcpyDeviceToHost));
      for (int k = 0; k < 4; ++k) stats[1 + k] = (int64_t)h[k];
    }
  }
  return MOOG_OK;
}

int moog_engine_set_action_dtype(moog_engine_t* e, int32_t float32) {
  if (!e) return fail(MOOG_E_INVALID, "null engine");
  e->act_f32 = float32 ? 1 : 0;
  return MOOG_OK;
}

int moog_engine_layer_usage(moog_engine_t* e, int32_t* high_water, int32_t* dropped) {
  if (!e || !high_water || !dropped) return fail(MOOG_E_INVALID, "null argument");
  int32_t host[2 * MOOG_MAX_LAYERS] = {0};
  if (e->layer_hw) {
    HIPCHK(hipSetDevice(e->device));
    HIPCHK(hipMemcpy(host, e->layer_hw, sizeof(host), hipMemcpyDeviceToHost));   // (synchronises with the null stream's work)
  }
  for (int l = 0; l < MOOG_MAX_LAYERS; ++l) { high_water[l] = host[l]; dropped[l] = host[MOOG_MAX_LAYERS + l]; }
  return MOOG_OK;
}

int moog_program_step_kernel(const moog_program_t* prog, int32_t* variant, int32_t* wps, uint64_t* hash) {
  if (!prog) return fail(MOOG_E_INVALID, "null program");
  const StepVariant v = step_variant_of(prog);
  if (variant) *variant = (v.maze_kernel && !v.late_reset) ? 2 : (v.dynamic_rules ? 1 : 0);
  if (wps) *wps = v.wps;
  if (hash) *hash = program_hash(prog);
  return MOOG_OK;
}

int moog_engine_step_kernel(moog_engine_t* e, int32_t* specialised) {
  if (!e || !specialised) return fail(MOOG_E_INVALID, "null argument");
  *specialised = e->spec_launch ? 1 : 0;
  return MOOG_OK;
}

int moog_engine_read_draw_records(moog_engine_t* e, uint8_t* host_out, int64_t bytes, int64_t* stride, int32_t* in_step) {
  if (!e) return fail(MOOG_E_INVALID, "null engine");
  if (!e->mask_setup.ok || !e->draw) return fail(MOOG_E_UNSUPPORTED, "this program's frames are not the mask rasteriser's: no draw records");
  if (stride) *stride = (int64_t)e->draw_lay.stride;
  if (in_step) *in_step = step_emits_draw(e) ? 1 : 0;
  if (!host_out) return MOOG_OK;
  const int64_t need = (int64_t)e->n_envs * (int64_t)e->draw_lay.stride;
  if (bytes < need) return fail(MOOG_E_INVALID, "host buffer too small for the draw records");
  HIPCHK(hipSetDevice(e->device));
  HIPCHK(hipDeviceSynchronize());
  HIPCHK(hipMemcpy(host_out, e->draw, (size_t)need, hipMemcpyDeviceToHost));
  return MOOG_OK;
}

int moog_engine_raster_path(moog_engine_t* e, int32_t* path) {
  if (!e || !path) return fail(MOOG_E_INVALID, "null argument");
  *path = (e->mask_setup.ok && e->pe_ns <= 0) ? (e->mask_setup.compact ? MOOG_RASTER_MASK_COMPACT : MOOG_RASTER_MASK) : MOOG_RASTER_SPANS;
  return MOOG_OK;
}

int moog_engine_kernel_variant(moog_engine_t* e, int32_t* variant, int32_t* late_reset) {
  if (!e) return fail(MOOG_E_INVALID, "null engine");
  const bool full = e->maze_kernel && !e->late_reset;
  if (variant) *variant = full ? 2 : (e->dynamic_rules ? 1 : 0);
  if (late_reset) *late_reset = e->late_reset ? 1 : 0;
  return MOOG_OK;
}

int moog_engine_set_color_override(moog_engine_t* e, const uint32_t* rgb_dev) {
  if (!e) return fail(MOOG_E_INVALID, "null engine");
  if (rgb_dev && !e->rgb_override) {   // the cached pictures hold the colour map's colours: off while the host supplies them
    e->kept_n_static = e->n_static; e->kept_pe_ns = e->pe_ns; e->kept_pe_nsv = e->pe_nsv;
    e->n_static = 0; e->pe_ns = 0; e->pe_nsv = 0;
  } else if (!rgb_dev && e->rgb_override) {   // back to render.cmap: the prefixes as they were (the per-env pictures are drawn again)
    e->n_static = e->kept_n_static; e->pe_ns = e->kept_pe_ns; e->pe_nsv = e->kept_pe_nsv;
    if (e->pe_ns > 0 && e->pe_valid) HIPCHK(hipMemset(e->pe_valid, 0, sizeof(int32_t) * (size_t)e->n_envs));
  }
  e->rgb_override = rgb_dev;
  return MOOG_OK;
}

int moog_engine_env_prefix(moog_engine_t* e, int32_t* n_slots) {
  if (!e || !n_slots) return fail(MOOG_E_INVALID, "null argument");
  int ns = e->pe_ns;
  if (ns > 0) { const int seen = __atomic_load_n(e->pe_min, __ATOMIC_RELAXED); if (seen < ns) ns = seen < 8 ? 0 : seen; }
  *n_slots = ns;
  return MOOG_OK;
}

int moog_engine_static_prefix(moog_engine_t* e, int32_t* n_slots, uint8_t* image_dev, void* hip_stream) {
  if (!e) return fail(MOOG_E_INVALID, "null engine");
  if (n_slots) *n_slots = e->n_static;
  if (image_dev && e->n_static > 0) {
    HIPCHK(hipSetDevice(e->device));
    if (e->aa > 1) return fail(MOOG_E_UNSUPPORTED, "the cached picture of an anti-aliased renderer is canvas sized");
    if (e->pad_w != e->canvas_w)
      moog_crop_launch(e->s_bg, image_dev, (size_t)e->canvas_h, e->pad_w * 3, e->canvas_w * 3, (hipStream_t)hip_stream);
    else
      HIPCHK(hipMemcpyAsync(image_dev, e->s_bg, (size_t)e->prog.render.width * e->prog.render.height * 3,
                            hipMemcpyDeviceToDevice, (hipStream_t)hip_stream));
  }
  return MOOG_OK;
}

int moog_engine_poll_faults(moog_engine_t* e, int32_t clear, int32_t* bits) {
  if (!e || !bits) return fail(MOOG_E_INVALID, "null argument");
  *bits = __atomic_load_n(e->fault_flag, __ATOMIC_RELAXED);   // no stream synchronisation: what has arrived so far
  if (clear && *bits) __atomic_and_fetch(e->fault_flag, ~*bits, __ATOMIC_RELAXED);
  return MOOG_OK;
}

int moog_engine_set_debug(moog_engine_t* e, int32_t step_debug, int32_t raster_stop) {
  if (!e) return fail(MOOG_E_INVALID, "null engine");
  e->step_dbg = step_debug;
  e->raster_stop = raster_stop;
  return MOOG_OK;
}

int moog_engine_read_watch(moog_engine_t* e, int32_t* host_out, int32_t clear) {
  if (!e || !host_out) return fail(MOOG_E_INVALID, "null argument");
  if (!e->watch) return fail(MOOG_E_UNSUPPORTED, "section sampling needs an engine created with MOOG_WATCH=1");
  HIPCHK(hipSetDevice(e->device));
  HIPCHK(hipDeviceSynchronize());
  const size_t bytes = (size_t)e->n_envs * MOOG_WATCH_SECTIONS * sizeof(int32_t);
  HIPCHK(hipMemcpy(host_out, e->watch, bytes, hipMemcpyDeviceToHost));
  if (clear) HIPCHK(hipMemset(e->watch, 0, bytes));
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

static void load_spec_kernel(moog_engine* e, const moog_program_t* prog) {
  { const char* sw = getenv("MOOG_STEP_SPEC"); if (sw && atoi(sw) == 0) return; }   // 0: the generic kernels (A/B runs, tests)
  std::string dir;
  if (const char* d = getenv("MOOG_SPEC_DIR")) dir = d;
  else {
    Dl_info info;
    if (!dladdr(reinterpret_cast<const void*>(&moog_abi_version), &info) || !info.dli_fname) return;
    dir = info.dli_fname;
    const size_t cut = dir.find_last_of('/');
    dir = (cut == std::string::npos ? std::string(".") : dir.substr(0, cut)) + "/spec";
  }
  const bool full = e->maze_kernel && !e->late_reset;
  char name[96];
  snprintf(name, sizeof name, "/step_%016llx_d%dw%d.so", (unsigned long long)program_hash(prog), full ? 2 : (e->dynamic_rules ? 1 : 0), e->step_wps);
  const std::string path = dir + name;
  if (access(path.c_str(), R_OK) != 0) return;
  void* h = dlopen(path.c_str(), RTLD_NOW | RTLD_LOCAL);
  if (!h) { fprintf(stderr, "moog: %s does not load (%s); the generic step kernel is used\n", path.c_str(), dlerror()); return; }
  typedef int (*fn_i)(void);
  typedef unsigned long long (*fn_u)(void);
  typedef const void* (*fn_p)(void);
  typedef int (*fn_cfg)(size_t);
  fn_i abi = reinterpret_cast<fn_i>(dlsym(h, "moog_spec_abi")), var = reinterpret_cast<fn_i>(dlsym(h, "moog_spec_variant"));
  fn_u hash = reinterpret_cast<fn_u>(dlsym(h, "moog_spec_hash")), ksz = reinterpret_cast<fn_u>(dlsym(h, "moog_spec_kargs_size"));
  fn_u dig = reinterpret_cast<fn_u>(dlsym(h, "moog_spec_source_digest"));
  // (the sources first: an object of another build of the same ABI number may differ in anything)
  if (!dig || dig() != MOOG_SRC_DIGEST || MOOG_SRC_DIGEST == 0ull) {
    fprintf(stderr, "moog: %s was built from other kernel sources or flags (digest %016llx, this library %016llx); the generic step "
            "kernel is used -- rebuild it (python -m moog._spec ..., __graft_entry__.build())\n", path.c_str(),
            dig ? dig() : 0ull, (unsigned long long)MOOG_SRC_DIGEST);
    dlclose(h);
    return;
  }
  fn_p pr = reinterpret_cast<fn_p>(dlsym(h, "moog_spec_program"));
  fn_cfg cfg = reinterpret_cast<fn_cfg>(dlsym(h, "moog_spec_configure"));
  void* launch = dlsym(h, "moog_spec_launch");
  const bool ok = abi && var && hash && ksz && pr && cfg && launch && abi() == MOOG_ABI_VERSION && ksz() == sizeof(KArgs) &&
                  var() == ((full ? 2 : (e->dynamic_rules ? 1 : 0)) | (e->step_wps << 8)) && hash() == program_hash(prog) &&
                  memcmp(pr(), prog, sizeof(moog_program_t)) == 0 && cfg(e->step_lds) == (int)hipSuccess;
  if (!ok) {
    fprintf(stderr, "moog: %s was built for another program, variant or ABI; the generic step kernel is used\n", path.c_str());
    dlclose(h);
    return;
  }
  e->spec_handle = h;
  e->spec_launch = reinterpret_cast<void (*)(int, size_t, hipStream_t, const KArgs*)>(launch);
}
