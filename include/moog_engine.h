/*
 * moog_engine.h -- C ABI of the MI355X batched MOOG step engine.
 *
 * The reference (jazlab/moog.github.io, pure Python) has no FFI for this path;
 * its boundary is the config dict consumed by
 *   moog.environment.Environment.__init__   (moog/environment.py:28-80)
 * and the dm_env surface reset()/step()/observation() (moog/environment.py:82-131).
 * This header is the C boundary a binding for that surface talks to: a config
 * dict is lowered (host side, Python) to the plain-old-data `moog_program_t`
 * below, sprite state lives in two caller-owned device buffers per engine
 * (`moog_state_view_t`), and every entry point takes raw device pointers plus a
 * HIP stream.  No C++ / torch types cross the boundary.
 *
 * The same structs and the `moog_layout()` helper are shared with the CPU
 * oracle (oracle/moog_oracle.c), which operates on host buffers of the same
 * layout so that parity tests can compare records word for word.
 *
 * Conventions: every function returns 0 on success, a negative MOOG_E_* code on
 * failure; the message is available from moog_last_error() (thread local).
 */
#ifndef MOOG_ENGINE_H_
#define MOOG_ENGINE_H_

#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define MOOG_ABI_VERSION 30

/* ---- capacity limits of the program blob -------------------------------- */
#define MOOG_MAX_LAYERS 16
#define MOOG_MAX_FORCES 16
#define MOOG_MAX_CORRECTIVE 4
#define MOOG_MAX_RULES 16
#define MOOG_MAX_TASKS 8
#define MOOG_MAX_OPS 256
#define MOOG_MAX_SHAPES 256
#define MOOG_MAX_SHAPE_VERTS 2048
#define MOOG_MAX_CAND 256
#define MOOG_MAX_DCODE 2048
#define MOOG_X_STACK 16
#define MOOG_MAX_SLOTS 256
#define MOOG_MAX_ACTIONS 4
#define MOOG_MAX_MORE_ACTIONS 3 /* MOOG_MAX_ACTIONS - 1 */
#define MOOG_NUM_FACTORS 14
#define MOOG_MAX_HDRAWS 32   /* draws a state_initializer takes from np.random directly (per reset) */
#define MOOG_MAX_OP_DRAWS 24 /* draws of one generation op: its sampled factors + the direct draws made between them */

/* ---- error codes --------------------------------------------------------- */
#define MOOG_OK 0
#define MOOG_E_INVALID (-1)   /* bad argument / malformed program            */
#define MOOG_E_HIP (-2)       /* HIP runtime error (message has the detail)  */
#define MOOG_E_NOMEM (-3)
#define MOOG_E_UNSUPPORTED (-4)

/* ---- per-env fault bits (i32 record, word o_fault) ----------------------- */
/* sprite_generators.py:92-98  -> RecursionError */
#define MOOG_FAULT_SAMPLER_EXHAUSTED 1
/* portal.py:51-54            -> ValueError (odd number of portals) */
#define MOOG_FAULT_ODD_PORTALS 2
/* collisions.py:322-326      -> ValueError (collision normal not unit) */
#define MOOG_FAULT_BAD_NORMAL 4
/* injected-uniform buffer ran dry (test harness error) */
#define MOOG_FAULT_INJECT_UNDERRUN 8
/* TetherZippedLayers over layers of different lengths (tether_physics.py:192-198) */
#define MOOG_FAULT_TETHER_ZIP 16
/* distributions.py:242-249,344-351: Intersection / SetMinus / Selection exceeded
 * _MAX_TRIES = 1e5 rejections (ValueError)                                      */
#define MOOG_FAULT_DIST_EXHAUSTED 32
/* CreateSprites / ChangeLayer appended to a layer whose slot capacity is used up (the
 * reference's lists are unbounded; the engine's layers have a fixed capacity)      */
#define MOOG_FAULT_LAYER_FULL 64
/* PhaseSequence stepped past its last phase (task_phases.py:136-139 raises IndexError) */
#define MOOG_FAULT_PHASE_END 128
/* MazePhysics: an avatar is on no grid line of the maze (maze_physics.py:87-97 raises ValueError) */
#define MOOG_FAULT_OFF_GRID 256
/* (512, 1024: the fault bits of the "frames follow their env's step" launch structure, retired with ABI 30) */
#define MOOG_MAX_MAZE 32
#define MOOG_MAX_MAZE_POINTS 8 /* cells of one sample_distinct_open_points() call */
#define MOOG_MAX_MAZE_GEN 16   /* size of a maze drawn on the device (its frontier list holds size^2 one-byte cells) */

/* ---- sprite flag bits (i32 record, o_flags[slot]) ------------------------ */
#define MOOG_F_ALIVE 1
#define MOOG_F_SYM_CIRCLE 2 /* sprite.py:501-502 is_symmetric_circle          */
#define MOOG_F_VEL_F32 4    /* velocity is a float32 ndarray in the reference */
#define MOOG_F_ANGVEL_F32 8 /* angle_vel is a float32 0-d array               */

/* ---- factor indices: Sprite.FACTOR_NAMES order (sprite.py:237-253),
 *      metadata dropped ----------------------------------------------------- */
enum {
  MOOG_FAC_X = 0, MOOG_FAC_Y, MOOG_FAC_SHAPE, MOOG_FAC_ANGLE, MOOG_FAC_SCALE,
  MOOG_FAC_ASPECT, MOOG_FAC_C0, MOOG_FAC_C1, MOOG_FAC_C2, MOOG_FAC_OPACITY,
  MOOG_FAC_XVEL, MOOG_FAC_YVEL, MOOG_FAC_ANGVEL, MOOG_FAC_MASS
};

/* factor distribution kinds (state_initialization/distributions.py).  TREE: the
 * factor is written by the op's distribution program (moog_dinstr_t below). */
enum { MOOG_DIST_CONST = 0, MOOG_DIST_CONTINUOUS = 1, MOOG_DIST_DISCRETE = 2, MOOG_DIST_TREE = 3,
       /* factors of a sprite placed on a cell (i, j) of the per-episode random maze (pacman.py:40-65), see
        * moog_genop_t.cell_sel.  MAZE_COORD: cand[cand_off + (n_cand ? j : i)] -- the config's arithmetic on the
        * cell index (`grid_side * (0.5 + index)`), tabulated on the host for every index.  MAZE_SHAPE: shape id
        * a + j * N + i, the wall square of cell (row i, column j) (maze.py:98-111)                          */
       MOOG_DIST_MAZE_COORD = 4, MOOG_DIST_MAZE_SHAPE = 5,
       /* EXPR: the value of the postfix expression at dcode[cand_off] (leaves: constants, MOOG_X_HDRAW, MOOG_X_SLOT_ATTR).
        * EXPR_SHAPE (factor `shape` only): a raw polygon of n_cand vertices whose coordinates the code at
        * dcode[cand_off] stores with MOOG_X_STORE_VERT; centroid, inertia and the centred path are then derived on
        * the device as Sprite.__init__ derives them (sprite.py:360-401)                                          */
       MOOG_DIST_EXPR = 6, MOOG_DIST_EXPR_SHAPE = 7 };
/* moog_genop_t.cell_sel: which maze cell (row i, column j of Maze.maze) a one-sprite op is placed on.  The op's
 * slot stays dead when the maze has fewer such cells.  GENERATE / SAMPLE are ops without sprites.           */
enum {
  MOOG_CELL_NONE = 0,
  MOOG_CELL_GENERATE,   /* maze_generators.py:96-171 generate_random_maze_matrix -> the record's maze rows */
  MOOG_CELL_SAMPLE,     /* maze.py:200-214 sample_distinct_open_points(cell_arg) -> the record's point words */
  MOOG_CELL_SAMPLED,    /* the cell_arg-th point of that sample                                            */
  MOOG_CELL_OPEN_RANK,  /* the cell_arg-th open cell in np.argwhere order (rows outer; pacman.py:62-65)     */
  MOOG_CELL_WALL_RANK,  /* the cell_arg-th wall cell in Maze.to_sprites order (columns outer; maze.py:101-103) */
  MOOG_CELL_HDRAW,      /* not a cell: an op without sprites that takes direct draw cell_arg (MOOG_X_HDRAW).  code_off >= 0:
                         * the accept test of a rejection loop over this draw (`while not ok: a = np.random.uniform(..);
                         * ok = test(a)`, match_to_sample.py:33-43): the draw is retaken, one uniform per try, until the
                         * expression at dcode[code_off] is true (MOOG_DIST_MAX_TRIES tries: MOOG_FAULT_SAMPLER_EXHAUSTED) */
  MOOG_CELL_CHOICE,     /* not a cell: np.random.choice(count_max alternatives, p) of a sample_generator: the picked index
                         * goes to o_hdraw[cell_arg]; factors[0].cand_off >= 0: the normalised cumulative
                         * probabilities in program.cand (searchsorted(cdf, u, 'right')), else int(u * n) (no draw
                         * for n == 1)                                                                            */
  MOOG_CELL_SHUFFLE,    /* not a cell: sprite_generators.shuffle (sprite_generators.py:157-183): the live sprites in
                         * slots slot0 .. slot0 + cell_arg - 1 (a packed prefix) are permuted as np.random.shuffle
                         * permutes their list; slot slot0 + cell_arg is a spare used while swapping              */
  MOOG_CELL_HEXPR,      /* not a cell: o_hdraw[cell_arg] = the value of the expression at dcode[code_off] -- a value the
                         * initializer computed from its draws and uses several times (the compare-exchange outputs of
                         * np.sort over drawn values, match_to_sample.py:46); read back with MOOG_X_HDRAW            */
  MOOG_CELL_SIMULATE,   /* not a cell: the look-ahead an initializer runs on the state it has just built (bounce_box_contact_
                         * prediction.py:40-50: `while True: if <test>: return ..; physics.step(state)`).  Repeats:
                         * evaluate the expression at dcode[code_off] -- 0: go on, k > 0: the loop's k-th exit --; on an
                         * exit store k in o_hdraw[cell_arg] and stop, otherwise run one env step of the program's
                         * physics (updates_per_env_step sub-steps, physics.py:88-117; no rules, no task, no action).
                         * count_max = iteration limit (MOOG_FAULT_SAMPLER_EXHAUSTED beyond it)                     */
  MOOG_CELL_STORE,      /* not a cell: attribute writes to the already built sprite in slot cell_arg (`s.position = ..`,
                         * `s.velocity = ..` after the look-ahead, :117-119): the modifier code at dcode[code_off],
                         * applied with the setters' semantics (sprite.py:616-652)                                 */
  MOOG_CELL_PSTATE      /* not a cell: a number the initializer's object keeps ACROSS episodes (predators_arena.py:56,
                         * 88-94: the auto-curriculum's predator mass).  It lives in the state scalar of the
                         * MOOG_RULE_STATE_SLOT rule cell_arg, which resets never clear: the env's first reset stores
                         * factors[0].a there (and marks the slot's second scalar), every later reset stores the value
                         * of the update expression at dcode[code_off] (which reads the slot with MOOG_X_RULE_STATE) */
};

/* Distribution programs.  A factor distribution that is not a flat Product of
 * Continuous / Discrete / constants (distributions.py:159-420: Mixture, Intersection,
 * SetMinus, Selection, Discrete with probs) is lowered to a small branching program
 * that draws uniforms in exactly the order the reference's recursive `.sample()`
 * does.  `contains()` predicates are postfix boolean programs in the same table.
 *
 *   sampling ops
 *   CONT    fac[a] = x + (y - x) * u, cast to float32 when b           (:95-100)
 *   DISC    fac[a] = cand[c + int(u * b)]; no draw when b == 1          (:137-140)
 *   DISCP   as DISC with probabilities cand[d .. d + b): index =
 *           searchsorted(cumsum(p) / sum(p), u, side='right')           (numpy choice)
 *   CONST   fac[a] = x                                                   (Product constants)
 *   CHOICE  Mixture: index as DISCP over b components with probs cand[d..]; the next b
 *           instructions are JUMPs to the components                     (:176-180)
 *   JUMP    pc = a
 *   LOOP    tries[a] = 0 (a = nesting depth of the rejection loop, 0 or 1)
 *   TEST    evaluate the predicate code[c .. c + b); accepted when its value == d;
 *           otherwise ++tries[a], fault at 1e5, pc = loop body start (x as int)
 *   END
 *   predicate ops (value stack of booleans)
 *   P_RANGE push x <= fac[a] < y, compared in float32 when fac[a] is a float32 sample
 *   P_SET   push fac[a] in cand[c .. c + b)
 *   P_AND / P_OR  pop b values, push their conjunction / disjunction
 *   P_NOT
 */
enum {
  MOOG_D_CONT = 0, MOOG_D_DISC, MOOG_D_DISCP, MOOG_D_CONST, MOOG_D_CHOICE, MOOG_D_JUMP,
  MOOG_D_LOOP, MOOG_D_TEST, MOOG_D_END,
  MOOG_P_RANGE = 16, MOOG_P_SET, MOOG_P_AND, MOOG_P_OR, MOOG_P_NOT
};
#define MOOG_DIST_MAX_TRIES 100000

/* Sprite expressions.  Config callables (sprite filters, modifiers, pair conditions, reward
 * functions; e.g. modify_sprites.py:41-52, vanish.py:58-61, contact_reward.py:85-92) are run
 * once at build time on symbolic sprites (moog/_symbolic.py) and stored as postfix code in
 * the same table.  Every value carries a dtype tag so that scalar arithmetic follows numpy 2
 * promotion (NEP 50): 0 = weak Python scalar, 1 = float32, 2 = float64; an operation is
 * carried out in float32 when an operand is float32 and none is float64.
 *   X_CONST  push x (tag 2 when b, else 0)     X_ATTR  push attribute a of sprite b
 *   binary   ADD SUB MUL DIV REM(np.remainder) MIN MAX  LT LE GT GE EQ NE  AND OR
 *   unary    NEG ABS SQRT SIN COS FLOOR NOT SIGN
 *   X_SELECT pop b, a, c; push c ? a : b       X_STORE pop v; sprite 0 attribute a := v
 *   X_END
 */
enum {
  MOOG_X_CONST = 32, MOOG_X_ATTR, MOOG_X_ADD, MOOG_X_SUB, MOOG_X_MUL, MOOG_X_DIV, MOOG_X_REM,
  MOOG_X_MIN, MOOG_X_MAX, MOOG_X_LT, MOOG_X_LE, MOOG_X_GT, MOOG_X_GE, MOOG_X_EQ, MOOG_X_NE,
  MOOG_X_AND, MOOG_X_OR, MOOG_X_NEG, MOOG_X_ABS, MOOG_X_SQRT, MOOG_X_SIN, MOOG_X_COS,
  MOOG_X_FLOOR, MOOG_X_NOT, MOOG_X_SIGN, MOOG_X_SELECT, MOOG_X_STORE, MOOG_X_END,
  MOOG_X_RULE_STATE,  /* push the state scalar of rule a (e.g. a PhaseSequence's current phase index) */
  MOOG_X_OVERLAPS_FIRST, /* push sprite b .overlaps_sprite(first live sprite of layer a) (0 when the layer is empty) */
  /* reset-time expressions (factor values computed by the state_initializer's own arithmetic on np.random draws,
   * e.g. parallelogram_catch.py:34-68, multi_tracking_with_feature.py:41-46,136): */
  MOOG_X_HDRAW,      /* push uniform a of this reset (the a-th direct np.random call of the initializer)        */
  MOOG_X_SLOT_ATTR,  /* push attribute a (MOOG_XA_*) of sprite slot b (a factor copied from an earlier sprite)  */
  MOOG_X_STORE_VERT, /* pop -> component a of the raw shape being built (vertex a / 2, x or y)                   */
  MOOG_X_FACTOR,     /* push factor a (MOOG_FAC_*) of the sprite being created, as just sampled: the argument of a
                      * DependentDistribution's dependent_fn (distributions.py:420-475)                           */
  MOOG_X_SLOT_CONST, /* push cand[a + slot of sprite b]: `sprite.metadata[key]`, a per-slot constant of the config
                      * (match_to_sample.py:120-124,171; NaN where the config stored none)                        */
  MOOG_X_FMA,        /* pop c, b, a; push fma(a, b, c) -- float64: ONE rounding, what np.dot / 1-D np.linalg.norm of
                      * float64 2-vectors do (OpenBLAS ddot); float32 operands: a * b + c, two roundings */
  MOOG_X_RULE_STATE2,/* push the second state scalar of rule a (o_rule2: a Phase's drawn duration, the second
                      * uniform of a MOOG_RULE_DRAWS rule)                                                       */
  MOOG_X_ZIP_ATTR,   /* push attribute a of the sprite at sprite 0's list position in layer b: the partner of a
                      * `for t, c in zip(state[A], state[B])` loop in a config-local rule (match_to_sample.py:71) */
  MOOG_X_HDRAW_T,    /* push o_hdraw[a] with the dtype tag kept in o_hdraw[a + 1] (0 weak, 1 float32, 2 float64): a
                      * MOOG_CELL_HEXPR cell with count_min = 1, e.g. np.copy(sprite.velocity) of a float32 velocity
                      * that is assigned back later (bounce_box_contact_prediction.py:113-119)                     */
  MOOG_X_OVERLAPS_SLOTS,/* push (sprite in slot a).overlaps_sprite(sprite in slot b), 0 when either is gone: tests
                      * between fixed sprites in an initializer's look-ahead and in state-level task functions
                      * (bounce_box_contact_prediction.py:42,125-131)                                             */
  MOOG_X_ARG         /* push the scalar argument of the function being evaluated, a float64: the distance handed to a
                      * DistanceForce's force_fn (distance_fn_force.py:36, MOOG_FORCE_DISTANCE_EXPR)               */
};
/* sprite attributes of X_ATTR / X_STORE (sprite.py:505-664 properties) */
enum {
  MOOG_XA_X = 0, MOOG_XA_Y, MOOG_XA_XVEL, MOOG_XA_YVEL, MOOG_XA_ANGLE, MOOG_XA_ANGVEL, MOOG_XA_MASS,
  MOOG_XA_C0, MOOG_XA_C1, MOOG_XA_C2, MOOG_XA_OPACITY, MOOG_XA_SCALE, MOOG_XA_ASPECT
};

typedef struct {
  int32_t op;
  int32_t a, b, c, d;
  int32_t pad_;
  double x, y;
} moog_dinstr_t;

typedef struct {
  int32_t kind;     /* MOOG_DIST_*                                            */
  int32_t f32;      /* CONTINUOUS: sample is cast to float32 (:81,99)          */
  int32_t n_cand;   /* DISCRETE: number of candidates                         */
  int32_t cand_off; /* DISCRETE: first candidate in program.cand[]            */
  double a, b;      /* CONST: a ; CONTINUOUS: [a, b)                           */
  int32_t draw_pos; /* position of this factor's uniform among the draws of one sample of the op (sample_order
                     * with the direct draws counted in), -1: takes no draw.  Lets the device evaluate the factors
                     * of a sample in parallel lanes instead of walking sample_order                             */
  int32_t pad_;
} moog_factor_t;

/* One sprite-generation op = one `generate_sprites(...)._generate(...)` call
 * (sprite_generators.py:77-105), one direct `Sprite(**dist.sample())`, or one
 * static sprite built at config time.  Ops run in the order the reference's
 * state_initializer consumes randomness. */
typedef struct {
  int32_t slot0;        /* first sprite slot filled                           */
  int32_t count_min;    /* n ~ randint(count_min, count_max + 1)              */
  int32_t count_max;    /* slots reserved                                     */
  int32_t disjoint;     /* bit 0: _generate(disjoint=True).  bit 1 (value 2): a constant one-sprite op -- no draw,
                         * no rejection test, factors CONST / MAZE_COORD / MAZE_SHAPE only: an engine may build a run of
                         * such ops at once (they do not see each other); the result is the same either way     */
  int32_t max_tries;    /* max_recursion_depth (default 1e4)                  */
  int32_t n_sampled;    /* number of random factors                           */
  int32_t sample_order[MOOG_MAX_OP_DRAWS]; /* factor ids in draw order; MOOG_NUM_FACTORS + k: direct draw k */
  uint64_t avoid_ops;   /* without_overlapping: bitmask of earlier ops        */
  int32_t code_off;     /* distribution program in program.dcode, or -1       */
  int32_t runtime;      /* 1: run by a CREATE_SPRITES rule, not at reset; slot0
                         * is unused (sprites are appended to the rule's layer) */
  int32_t cell_sel;     /* MOOG_CELL_*                                        */
  int32_t cell_arg;
  int32_t fail_gracefully; /* sprite_generators.py:93-95: when max_tries is exceeded the call returns the
                            * sprites made so far (this and the op's remaining slots stay dead) instead of
                            * raising RecursionError (MOOG_FAULT_SAMPLER_EXHAUSTED)                         */
  int32_t cond_hdraw;      /* 1 + k: the op belongs to one alternative of a sample_generator (sprite_generators.py:
                            * 131-154) and runs only when the MOOG_CELL_CHOICE op that drew into o_hdraw[k] picked
                            * alternative cond_value; 0: unconditional.  The alternatives share their slots.      */
  int32_t cond_value;
  int32_t n_draws;         /* uniforms one sample of the op takes (its sampled factors + the direct draws between them) */
  moog_factor_t factors[MOOG_NUM_FACTORS];
} moog_genop_t;

/* unit shape table entry (sprite.py:329-409 precomputed per distinct shape) */
typedef struct {
  int32_t nverts;
  int32_t voff;        /* first vertex in program.shape_verts                 */
  int32_t is_circle;   /* shape name == 'circle' (sprite.py:259)              */
  int32_t pad_;
  double centroid[2];  /* centroid of the raw shape (added to position, :406) */
  double inertia[2];   /* (I_x, I_y) / area about the centroid (:400-401)     */
} moog_shape_t;

/* ---- forces (moog/physics) ----------------------------------------------- */
enum {
  MOOG_FORCE_DRAG = 1,          /* friction.py:54-56   p0 = coeff_friction    */
  MOOG_FORCE_KINETIC_FRICTION,  /* friction.py:25-33   p0 = coeff_friction    */
  MOOG_FORCE_DOWN_GRAVITY,      /* gravity.py:21-23    p0 = g                 */
  MOOG_FORCE_GRAVITY,           /* gravity.py:44-60    p0 = g, symmetric      */
  MOOG_FORCE_DISTANCE_LINEAR,   /* distance_fn_force.py:30-74 p0 = zero_intercept,
                                   p1 = slope, i0 = apply_distant, i1 = apply_nearby */
  MOOG_FORCE_DISTANCE_SPRING,   /* distance_fn_force.py:77-89 p0 = k, p1 = equilibrium */
  MOOG_FORCE_RANDOM,            /* random_force.py:22-26 p0 = max magnitude   */
  MOOG_FORCE_COLLISION,         /* collisions.py:457-584 p0 = elasticity,
                                   symmetric, i0 = update_angle_vel,
                                   i1 = max_recursion_depth                   */
  MOOG_FORCE_MAZE_WALK,         /* maze_walk.py:96-193 RandomMazeWalk: p0 = speed, i0 bit 0 prevent_backtracking,
                                   bit 1 allow_wall_backtracking, bit 2 only_turn_at_wall; the maze is
                                   program.maze (Maze.from_state of the wall layer, maze.py:39-84)      */
  MOOG_FORCE_MAZE_WALK_DET      /* maze_walk.py:203-243 DeterministicMazeWalk: p0 = speed; the prescribed velocities are
                                   cand[i0 .. i0 + 2 * i1) (x, y pairs), read front to back each time a sprite of the
                                   layer enters an intersection or stands still -- by whichever sprite comes next, over
                                   the whole life of the environment (the reference never rewinds the list: reset()
                                   only re-infers the maze): the number consumed so far is the scalar of the
                                   MOOG_RULE_STATE_SLOT entry `symmetric` of the rule table                     */
  ,
  MOOG_FORCE_DISTANCE_EXPR      /* distance_fn_force.py:16-47 DistanceForce with any scalar force_fn(distance), traced by
                                   moog/_symbolic.py trace_scalar_fn: the magnitude is the expression at dcode[i0] with the
                                   distance as MOOG_X_ARG; `symmetric` as for the other pair forces               */
};

typedef struct {
  int32_t kind;
  int32_t n_a, n_b;     /* layer-list lengths; n_b = 0 for one-sprite forces  */
  int32_t symmetric;
  int32_t i0, i1;
  int32_t layers_a[MOOG_MAX_LAYERS];
  int32_t layers_b[MOOG_MAX_LAYERS];
  double p0, p1;
} moog_force_t;

/* corrective physics (physics.py:110-111), applied after the forces of a substep:
 * ConstantSpeed (constant_speed.py:34-46), Tether and TetherZippedLayers
 * (tether_physics.py:43-201) */
enum { MOOG_CORR_CONSTANT_SPEED = 0, MOOG_CORR_TETHER = 1, MOOG_CORR_TETHER_ZIPPED = 2,
       MOOG_CORR_MAZE = 3 /* maze_physics.py:18-211 MazePhysics: layers = avatar layers, speed = constant_speed
                           * (NaN: None), anchor[0] = max_speed (NaN: None)                            */ };

typedef struct {
  int32_t kind;              /* MOOG_CORR_*                                      */
  int32_t n_layers;
  int32_t layers[MOOG_MAX_LAYERS];
  int32_t update_angle_vel;  /* TETHER*: tether_physics.py:70-91                 */
  int32_t has_anchor;        /* TETHER*: anchor given                            */
  double speed;              /* CONSTANT_SPEED                                   */
  double anchor[2];          /* TETHER*                                          */
} moog_corrective_t;

/* ---- game rules ----------------------------------------------------------- */
enum {
  MOOG_RULE_VANISH_ON_CONTACT = 1, /* vanish.py:63-86 l0 vanishing, l1 contacting */
  MOOG_RULE_TORUS_WRAP,            /* ModifySprites(pos = remainder(pos,1)),
                                      modify_sprites.py:35-52 + chase_avoid_torus.py:144-149 */
  MOOG_RULE_PORTAL,                /* portal.py:41-76 l0 teleporting, l1 portals */
  MOOG_RULE_BOOSTER,               /* functional_maze.py:23-78 l0 agent, l1 boosters,
                                      p0 mass mult, p1 c2 mult, p2 duration   */
  MOOG_RULE_VANISH_BY_FILTER,      /* vanish.py:42-61 l0 layer, filter             */
  MOOG_RULE_CHANGE_LAYER,          /* change_layer.py:36-46 l0 old, l1 new, filter */
  MOOG_RULE_CREATE_SPRITES,        /* create_sprites.py:31-37 l0 layer, op = runtime
                                      generation op, layers[] = without_overlapping */
  MOOG_RULE_TIMED,                 /* timing.py:46-59 TimedRule / DelayedRule /
                                      TemporaryRule: p0 start, p1 stop (inf allowed);
                                      steps its children while start <= 0 < stop.  op = 1: a callable
                                      interval -- the start is np.random.randint(p0, p2), drawn when the
                                      rule is reset (timing.py:47, before its children are), the stop is
                                      start + p1; op = 2: the start is p0, the stop np.random.randint(p1,
                                      p2).  The drawn width stop - start is the rule's second scalar
                                      (o_rule2)                                                          */
  MOOG_RULE_CONDITIONAL,           /* conditional.py:60-63: steps its children
                                      cond(state) times                           */
  MOOG_RULE_MODIFY_SPRITES,        /* modify_sprites.py:35-52: layers[], filter,
                                      i0 = sample_one, modifier = expression xmod  */
  MOOG_RULE_PHASE,                 /* task_phases.py:17-98: children = i0 one-time rules
                                      (stepped on the phase's first step) followed by the
                                      continual rules; ends after p0 steps or when cond
                                      (MOOG_RCOND_*, 0 = never) holds; state = step count,
                                      -1 once ended.  op = 1: the duration is drawn when the
                                      phase is reset, np.random.randint(p0, p2) (task_phases.py:72),
                                      and kept in o_rule2                           */
  MOOG_RULE_PHASE_SEQUENCE,        /* task_phases.py:101-141: children = PHASE rules, one
                                      current at a time; state = index of the current one */
  MOOG_RULE_KEEP_NEAR_CENTER,      /* re_center.py:48-58: l0 agent layer, layers[] to move,
                                      p0 / p1 grid cell                            */
  MOOG_RULE_MODIFY_ON_CONTACT,     /* contact_rules.py:112-141: layers[] x layers1[];
                                      xmod / filter for side 0, xmod1 / filter1 for
                                      side 1 (xmod < 0: no modifier on that side)  */
  MOOG_RULE_FIXATION,              /* fixation.py:17-58: state = consecutive steps during which the first sprite
                                      of l0 has been within p0 of the first sprite of l1 (the number the
                                      reference keeps in meta_state[key]); conditions read it through
                                      MOOG_X_RULE_STATE                                                   */
  MOOG_RULE_STATE_SLOT,            /* not a rule of the reference: a per-env scalar that belongs to another component
                                      (DeterministicMazeWalk's read position); never stepped, never reset -- it lives
                                      as long as the environment, like the Python object it stands for    */
  MOOG_RULE_DRAWS                  /* the np.random calls at the top of a config-local rule's step (match_to_sample.py:
                                      68-69): i0 = 1 or 2 uniforms taken when the rule is stepped, kept in the rule's
                                      two state scalars (o_rule, o_rule2) for the MODIFY_SPRITES rules that follow it
                                      (MOOG_X_RULE_STATE / MOOG_X_RULE_STATE2)                             */
};
/* sprite filters: ALWAYS, or the expression at rule.xfilter */
enum { MOOG_FILTER_ALWAYS = 0, MOOG_FILTER_EXPR = 1,
       MOOG_FILTER_EXPR_LANES = 2   /* as EXPR, and the expression only reads its own sprite (constants, attributes, metadata,
                                     * rule scalars; arithmetic, comparisons, select): the engine may evaluate it for the 64
                                     * sprites of a layer at once, one per lane (program.xstack_depth stack entries per lane) */
};
/* conditions of CONDITIONAL */
enum {
  MOOG_RCOND_BERNOULLI = 1,     /* np.random.binomial(1, p0): one uniform u, value u < p0 */
  MOOG_RCOND_CONTACT_COUNT,     /* contact_rules.get_contact_counter(l0, l1): overlapping pairs */
  MOOG_RCOND_ALL_EXPR,          /* as MOOG_COND_ALL_EXPR .. FIRST_EXPR: layer l0, expression   */
  MOOG_RCOND_ANY_EXPR,          /*   xfilter; FIRST_EXPR's value is the repeat count           */
  MOOG_RCOND_FIRST_EXPR,
  MOOG_RCOND_COUNT_EXPR,        /* sum of expression xfilter over the sprites of layer l0 (a loop that
                                   accumulates per-sprite terms, cleanup.py:181-190)                 */
  MOOG_RCOND_STATE_EXPR         /* = MOOG_COND_STATE_EXPR (7): the expression reads no layer's sprites (the current phase,
                                   rule scalars, the meta-state): evaluated unconditionally             */
};

/* Rules form a forest in pre-order: `parent` is the index of the enclosing TIMED /
 * CONDITIONAL rule or -1; the children of rule r are the later entries whose parent
 * is r, in order.  Nesting depth is at most 2 combinators.  Per-rule scalar state
 * (f64 record, o_rule[r]): BOOSTER countdown, TIMED _steps_until_start. */
typedef struct {
  int32_t kind;
  int32_t l0, l1;
  int32_t n_layers;
  int32_t layers[MOOG_MAX_LAYERS];
  int32_t parent;
  int32_t filter;      /* MOOG_FILTER_*                                           */
  int32_t cond;        /* MOOG_RCOND_*                                            */
  int32_t op;          /* CREATE_SPRITES: index of the runtime op in program.ops  */
  int32_t xfilter;     /* filter == EXPR: expression offset in program.dcode      */
  int32_t xmod;        /* MODIFY_*: modifier code (X_STORE ...), or -1            */
  int32_t filter1, xfilter1, xmod1;   /* MODIFY_ON_CONTACT, side 1                */
  int32_t i0;          /* MODIFY_SPRITES: sample_one; velocity assigned as a whole
                        * by xmod (bit 1) / xmod1 (bit 2); bit 3: only the layer's first
                        * sprite is modified (a rule written as `s = state[L][0]; s.attr = ...`); bit 4 (a hint):
                        * the stored values are constants and none of them moves vertices */
  int32_t n_layers1;
  int32_t layers1[MOOG_MAX_LAYERS];
  double p0, p1, p2;
} moog_rule_t;

/* ---- tasks ---------------------------------------------------------------- */
enum {
  MOOG_TASK_CONTACT_REWARD = 1, /* contact_reward.py:70-102 p0 reward, p1 reset_steps_after_contact */
  MOOG_TASK_RESET,              /* reset.py:48-61 cond, p0 reward, p1 steps_after_condition */
  MOOG_TASK_STAY_ALIVE          /* stay_alive.py:22-32 i0 period, p0 value    */
};
/* state conditions of RESET tasks (cond, cond_layer, cond_value / xcond); 3-5 are traced
 * symbolically (moog/_symbolic.py trace_state_condition): xcond is a one-sprite expression */
enum {
  MOOG_COND_LAYER_EMPTY = 1,    /* lambda state: len(state[L]) == 0           */
  MOOG_COND_ALL_Y_LT,           /* lambda state: all(s.y < c for s in state[L]) */
  MOOG_COND_ALL_EXPR,           /* all(pred(s) for s in state[L])             */
  MOOG_COND_ANY_EXPR,           /* any(pred(s) for s in state[L])             */
  MOOG_COND_FIRST_EXPR,         /* expr(state[L][0])                          */
  MOOG_COND_STATE_EXPR = 7      /* expr(state, meta_state): sprites named by slot (state[L][k]) or no sprite at all --
                                   evaluated whatever the layers hold (no "first live sprite" to anchor it to) */
};

typedef struct {
  int32_t kind;
  int32_t n0, n1;
  int32_t layers0[MOOG_MAX_LAYERS];
  int32_t layers1[MOOG_MAX_LAYERS];
  int32_t cond, cond_layer;
  int32_t i0;
  int32_t xcond;       /* CONTACT_REWARD: pair condition(s0, s1) expression, or -1
                        * (contact_reward.py:52-58)                               */
  int32_t xreward;     /* CONTACT_REWARD: reward_fn(s0, s1) expression, or -1 (p0) */
  int32_t pad_;
  double cond_value;
  double p0, p1;
} moog_task_t;

/* ---- action space ---------------------------------------------------------- */
enum { MOOG_ACTION_JOYSTICK = 1,      /* joystick.py:45-70 */
       MOOG_ACTION_GRID = 2,          /* grid.py:15-21,52-75 */
       MOOG_ACTION_SET_POSITION = 3 };/* set_position.py:41-58: position = inertia * position
                                         + (1 - inertia) * action; `momentum` holds inertia */

typedef struct {
  int32_t kind;
  int32_t n_layers;
  int32_t layers[MOOG_MAX_LAYERS];
  int32_t constrained_lr;
  int32_t control_velocity;
  double scaling_factor;
  double momentum;
} moog_action_t;

/* ---- renderer (observers/pil_renderer.py:37-120) --------------------------- */
enum { MOOG_CMAP_IDENTITY = 0, MOOG_CMAP_HSV = 1 };     /* color_maps.py:21-23 */
enum { MOOG_POLYMOD_NONE = 0, MOOG_POLYMOD_TORUS = 1,   /* polygon_modifiers.py:32-38,67-98 */
       MOOG_POLYMOD_FIRST_PERSON = 2 };                  /* polygon_modifiers.py:41-64 */

typedef struct {
  int32_t width, height; /* the observation: image_size[0], image_size[1]      */
  int32_t cmap;
  int32_t polymod;
  int32_t bg[3];
  int32_t polymod_layer; /* FIRST_PERSON: agent layer (its first sprite is drawn at (0.5, 0.5)) */
  int32_t aa;            /* anti_aliasing: the canvas is aa * width x aa * height and is
                          * down-sampled with Pillow's LANCZOS filter (pil_renderer.py:64-66,112); 0 / 1: none */
  int32_t pad_;
} moog_render_t;

/* ---- maze (maze_lib/maze.py:20-36): the binary matrix MazePhysics / the maze walks infer from the wall
 *      sprites when they are reset (Maze.from_state, maze.py:39-84) ------------------------------------- */
typedef struct {
  int32_t size;                    /* N: the maze is N x N cells (0: the program has no maze)      */
  int32_t random;                  /* 1: the matrix is drawn at every reset (a MOOG_CELL_GENERATE op) and lives in
                                    * the env's record (o_maze); `rows` is unused                  */
  int32_t gen_size;                /* random: `size` argument of generate_random_maze_matrix, centred in the
                                    * N x N ambient matrix of walls (maze_generators.py:160-167)   */
  int32_t flip;                    /* random: np.flip(matrix, axis=0) before use (pacman.py:42)     */
  uint32_t rows[MOOG_MAX_MAZE];    /* rows[j] bit i = Maze.maze[j, i] (1 = wall)                    */
} moog_maze_t;

/* ---- the lowered config ----------------------------------------------------- */
typedef struct {
  int32_t abi_version;
  int32_t n_layers;
  int32_t n_slots;                 /* S                                        */
  int32_t n_total_verts;           /* TOTV = sum of slot vertex capacities     */
  int32_t layer_slot0[MOOG_MAX_LAYERS];
  int32_t layer_nslots[MOOG_MAX_LAYERS];
  int32_t layer_dynamic[MOOG_MAX_LAYERS]; /* 1: the layer is a Python list that rules
                                    * append to / pop from; its live sprites are kept
                                    * packed at the front of its slots, in list order */
  int32_t slot_layer[MOOG_MAX_SLOTS];
  int32_t slot_voff[MOOG_MAX_SLOTS];
  int32_t slot_vcap[MOOG_MAX_SLOTS];

  int32_t updates_per_env_step;    /* K, physics.py:15                         */
  int32_t sprite_factors;          /* some expression reads scale / aspect_ratio or the
                                    * float32-ness of mass / colours: the records carry
                                    * o_scale, o_aspect, o_fmask                       */
  int32_t vel_alias;               /* some Tether has update_angle_vel=False: its
                                    * sprites share ONE velocity ndarray afterwards
                                    * (tether_physics.py:90), tracked in o_valias  */
  int32_t n_forces;
  int32_t n_corrective;
  int32_t n_rules;
  int32_t n_tasks;
  int32_t n_ops;
  int32_t n_shapes;
  int32_t n_cand;
  int32_t n_hdraws;                /* direct np.random draws per reset (o_hdraw) */
  int32_t rule_state2;             /* some rule keeps a second scalar (o_rule2) */
  double timeout_steps;            /* CompositeTask timeout (inf allowed)      */

  moog_force_t forces[MOOG_MAX_FORCES];
  moog_corrective_t corrective[MOOG_MAX_CORRECTIVE];
  moog_rule_t rules[MOOG_MAX_RULES];
  moog_task_t tasks[MOOG_MAX_TASKS];
  moog_action_t action;            /* the action space, or the first of a Composite      */
  int32_t n_actions;               /* sub-spaces of a Composite (composite.py:38-49), in
                                    * keyword order; 0 / 1: `action` alone.  With more than
                                    * one the action buffer is f64 [n_envs][n_actions][2]
                                    * (a Grid move in component 0) and o_action holds
                                    * 2 words per sub-space                              */
  int32_t pad_actions_;
  moog_action_t more_actions[MOOG_MAX_MORE_ACTIONS];
  moog_render_t render;
  moog_maze_t maze;
  moog_genop_t ops[MOOG_MAX_OPS];
  moog_shape_t shapes[MOOG_MAX_SHAPES];
  double shape_verts[MOOG_MAX_SHAPE_VERTS][2]; /* centred, CCW, unit shapes    */
  double cand[MOOG_MAX_CAND];                  /* DISCRETE candidates / probs  */
  int32_t n_dcode;
  int32_t xstack_depth; /* deepest value stack among the MOOG_FILTER_EXPR_LANES expressions (0: none) */
  int32_t pad_x_;
  int32_t pad_y_;
  int32_t born_rule;   /* Sprites the config builds OUTSIDE its state_initializer are the same Python objects in every
                        * episode of the reference: whatever a rule did to them (match_to_sample.py:89-90,205: the
                        * screen, made transparent once) they keep across resets.  1 + index of the MOOG_RULE_STATE_SLOT
                        * (op 2) that says "this env has been reset before": from then on the slots marked in
                        * slot_persist are not rebuilt by a reset.  0: no such sprites.                            */
  moog_dinstr_t dcode[MOOG_MAX_DCODE];         /* distribution programs        */
  uint8_t slot_persist[MOOG_MAX_SLOTS];        /* 1: built outside the initializer (see born_rule) */
} moog_program_t;

/* ---- state record layout ------------------------------------------------------
 * Per env one f64 record and one i32 record, env-major:
 *   f64[n_envs][f64_per_env], i32[n_envs][i32_per_env].
 * One wavefront owns one env, so a record is one contiguous, coalesced block.
 * Offsets below are in elements from the start of the env's record. */
typedef struct {
  int32_t S, TOTV, T, R;
  int32_t f64_per_env, i32_per_env;
  /* f64 record */
  int32_t o_pos;      /* [S][2]   sprite.position                              */
  int32_t o_vel;      /* [S][2]   sprite.velocity                              */
  int32_t o_angle;    /* [S]                                                   */
  int32_t o_angvel;   /* [S]                                                   */
  int32_t o_mass;     /* [S]                                                   */
  int32_t o_color;    /* [S][3]   c0,c1,c2                                     */
  int32_t o_inertia;  /* [S][2]   _x_y_rotational_inertia                      */
  int32_t o_maxr;     /* [S]      _max_radius                                  */
  int32_t o_action;   /* [2 * max(1, n_actions)] action-space memory (_action)    */
  int32_t o_task;     /* [T]      per-task _steps_until_reset (inf sentinel)   */
  int32_t o_rule;     /* [R]      per-rule scalar (Booster countdown)          */
  int32_t o_hdraw;    /* [n_hdraws] the uniforms of this episode's direct draws; -1 when there are none */
  int32_t o_rule2;    /* [R] second per-rule scalar (the drawn duration of a PHASE); -1 unless program.rule_state2 */
  int32_t o_scale;    /* [S] sprite.scale, [S] sprite.aspect_ratio at o_aspect; -1 when
                       *     program.sprite_factors == 0                              */
  int32_t o_aspect;
  int32_t o_verts;    /* [TOTV][2] world vertices (sprite.vertices)            */
  /* i32 record */
  int32_t o_flags;    /* [S] MOOG_F_*                                          */
  int32_t o_nverts;   /* [S]                                                   */
  int32_t o_opacity;  /* [S]                                                   */
  int32_t o_shape;    /* [S] shape-table id                                    */
  int32_t o_tele;     /* [S] bit r set: slot is in rule r's _currently_teleporting */
  int32_t o_fmask;    /* [S] bit MOOG_FAC_* set: that factor currently holds a float32
                       *     sample (numpy scalar promotion in expressions); -1 when
                       *     program.sprite_factors == 0                              */
  int32_t o_valias;   /* [S] 0, or 1 + id of the group of slots whose velocity is one
                       *     shared ndarray in the reference; -1 when the program has
                       *     no such tether (vel_alias == 0)                          */
  int32_t o_step_count;
  int32_t o_reset_next;
  int32_t o_fault;
  int32_t o_rng;      /* [4] rng draw counter lo/hi, injected cursor, spare    */
  int32_t o_maze;     /* [MOOG_MAX_MAZE] rows of the episode's maze as in moog_maze_t.rows, then
                       * [MOOG_MAX_MAZE_POINTS] sampled cells (i << 8 | j); -1 unless maze.random */
} moog_layout_t;

#if defined(__HIPCC__)
#define MOOG_HD __host__ __device__   /* (a program-specialised kernel computes its layout at compile time) */
#else
#define MOOG_HD
#endif
MOOG_HD static inline int32_t moog_align_(int32_t x, int32_t a) { return (x + a - 1) / a * a; }

MOOG_HD static inline void moog_layout(const moog_program_t* p, moog_layout_t* L) {
  int32_t S = p->n_slots, o = 0;
  L->S = S; L->TOTV = p->n_total_verts; L->T = p->n_tasks; L->R = p->n_rules;
  L->o_pos = o; o += 2 * S;
  L->o_vel = o; o += 2 * S;
  L->o_angle = o; o += S;
  L->o_angvel = o; o += S;
  L->o_mass = o; o += S;
  L->o_color = o; o += 3 * S;
  L->o_inertia = o; o += 2 * S;
  L->o_maxr = o; o += S;
  L->o_action = o; o += 2 * (p->n_actions > 1 ? p->n_actions : 1);
  L->o_task = o; o += p->n_tasks;
  L->o_rule = o; o += p->n_rules;
  if (p->n_hdraws > 0) { L->o_hdraw = o; o += p->n_hdraws; } else L->o_hdraw = -1;
  if (p->rule_state2) { L->o_rule2 = o; o += p->n_rules; } else L->o_rule2 = -1;
  if (p->sprite_factors) { L->o_scale = o; o += S; L->o_aspect = o; o += S; }
  else { L->o_scale = -1; L->o_aspect = -1; }
  o = moog_align_(o, 2);
  L->o_verts = o; o += 2 * p->n_total_verts;
  L->f64_per_env = moog_align_(o, 2);          /* 16-byte multiple */
  o = 0;
  L->o_flags = o; o += S;
  L->o_nverts = o; o += S;
  L->o_opacity = o; o += S;
  L->o_shape = o; o += S;
  L->o_tele = o; o += S;
  if (p->vel_alias) { L->o_valias = o; o += S; } else L->o_valias = -1;
  if (p->sprite_factors) { L->o_fmask = o; o += S; } else L->o_fmask = -1;
  L->o_step_count = o; o += 1;
  L->o_reset_next = o; o += 1;
  L->o_fault = o; o += 1;
  L->o_rng = o; o += 4;
  if (p->maze.random) { L->o_maze = o; o += MOOG_MAX_MAZE + MOOG_MAX_MAZE_POINTS; } else L->o_maze = -1;
  L->i32_per_env = moog_align_(o, 4);          /* 16-byte multiple */
}

/* Borrowed device pointers (owned by the caller, e.g. torch tensors). */
typedef struct {
  double* f64;   /* [n_envs][f64_per_env] */
  int32_t* i32;  /* [n_envs][i32_per_env] */
} moog_state_view_t;

/* Per-call outputs, device pointers, any may be NULL. */
typedef struct {
  double* reward;     /* [n_envs]  NaN encodes dm_env's None (FIRST steps)      */
  double* discount;   /* [n_envs]  NaN = None, 1.0 MID, 0.0 LAST                */
  int32_t* step_type; /* [n_envs]  0 FIRST, 1 MID, 2 LAST                       */
  uint8_t* image;     /* [n_envs][height][width][3]                             */
} moog_step_out_t;

/* Optional per-call randomness injection for parity runs:
 * uniforms in [0,1) consumed in the reference's draw order.  NULL = the
 * engine's own counter RNG (Philox4x32-10, key = (seed, env_index0 + env)). */
typedef struct {
  const double* uniforms; /* [n_envs][per_env] device pointer or NULL          */
  int32_t per_env;
} moog_inject_t;

typedef struct moog_engine moog_engine_t;

/* kernel ids for moog_engine_kernel_time() */
enum { MOOG_K_STEP = 0, MOOG_K_RASTER = 1, MOOG_K_RESET = 2, MOOG_K_COUNT = 3 };

int moog_abi_version(void);
/* The digest of the kernel sources and hipcc flags the library was built from (moog/_digest.py: 64 bits of SHA-256 over
 * csrc/ and this header, passed to the build as -DMOOG_SRC_DIGEST).  A program-specialised step kernel is used only when
 * it carries the same digest (the symbol `moog_spec_source_digest` of the object); 0 = built without one. */
unsigned long long moog_source_digest(void);
const char* moog_last_error(void);
/* sizeof(moog_program_t) as compiled, for binding self-checks */
int64_t moog_program_sizeof(void);

/* Replaces Environment.__init__ (environment.py:28-80). `seed`/`env_index0`
 * key the per-env RNG streams (global env index = env_index0 + local index so
 * results do not depend on how envs are sharded over GPUs). */
int moog_engine_create(const moog_program_t* prog, int32_t n_envs, int32_t device_id,
                       uint64_t seed, int64_t env_index0, moog_engine_t** out);
int moog_engine_destroy(moog_engine_t* e);
int moog_engine_layout(const moog_engine_t* e, moog_layout_t* out);
int moog_engine_load_state(moog_engine_t* e, const moog_state_view_t* view);

/* Replaces Environment.reset (environment.py:82-96) for the envs whose mask
 * byte is non-zero (NULL = all).  Writes FIRST timesteps + images to `out`. */
int moog_engine_reset(moog_engine_t* e, const uint8_t* env_mask_dev,
                      const moog_inject_t* inject, const moog_step_out_t* out,
                      void* hip_stream);
/* Replaces Environment.step (environment.py:98-126).  actions: f64 [n_envs][n_actions][2] for
 * a Composite action space, else f64 [n_envs][2]
 * (Joystick) or i32 [n_envs] (Grid).  Envs with reset_next set are reset
 * instead (auto-reset, :100-101). */
int moog_engine_step(moog_engine_t* e, const void* actions_dev,
                     const moog_inject_t* inject, const moog_step_out_t* out,
                     void* hip_stream);
/* Element type of the action buffer of a Joystick (and of the Joystick components of a Composite): 0 = float64 (default),
 * 1 = float32, the dtype of the reference's action spec (joystick.py:42-43).  With float32 actions the reference's
 * `self._scaling_factor * action` is a float32 product (numpy: the Python float is the weak operand); the engine then
 * computes exactly that.  Grid actions stay int32; SetPosition components read the same buffer and use the values as they are. */
int moog_engine_set_action_dtype(moog_engine_t* e, int32_t float32);
/* env.physics.step(env.state) only (tests/runtime_benchmark.py:101-107). */
int moog_engine_physics_only(moog_engine_t* e, const moog_inject_t* inject,
                             void* hip_stream);
/* env.observation() only (environment.py:128-131, runtime_benchmark.py:113-130). */
int moog_engine_render(moog_engine_t* e, uint8_t* image_dev, void* hip_stream);

/* Optional launch-order schedule for the step kernel (pure performance hint, results do
 * not depend on it).  `cost_dev` (float[n_envs], borrowed; zeroed by this call) holds a moving average of every env's
 * shader-clock cycles per step (0.6 x the last step + 0.4 x the value before: the step kernel reads and writes it); `perm_dev` (int32[n_envs], borrowed, initialised by the caller to a
 * permutation, e.g. the identity) is the order in which workgroups pick envs.  After every
 * step the engine re-sorts it by descending cost (counting sort on a side stream, overlapped
 * with the rasteriser) so that the expensive envs (clustered contacts) start first instead of
 * landing in the under-filled tail of the launch.  NULLs disable. */
int moog_engine_set_schedule(moog_engine_t* e, int32_t* perm_dev, float* cost_dev);

/* (ABI <= 29 had moog_engine_set_fused / moog_engine_get_fused here: a raster grid beside the step kernel that drew each
 * frame as soon as its env's step was stored.  It paid 3 % when the step kernel took 750 us and the rasteriser 95; with the
 * step kernel at 650 us and the rasteriser at 57 it measured slower than the two plain launches and was removed --
 * profiles/HISTORY.md 3.4.) */

/* Reset pool (programs whose initializer is expensive: bounce_box_contact_prediction.py:88-119 and red_green.py:120-203
 * play the episode forward inside state_initializer -- MOOG_CELL_SIMULATE -- which on one wavefront takes as long as a
 * hundred env steps of the whole batch, inside the launch every env waits for).  With the pool on, the NEXT episode of
 * every env is built by a background launch on a stream of the engine's own while the current one is being stepped, and
 * the step kernel takes the finished record over when the episode ends (environment.py:100-101) instead of running the
 * initializer.  Results are the same with and without the pool, bit for bit: a reset's draws come from the episode's
 * own segment of the env's stream (counter = episode << 32 | draw), and what else a reset reads -- the sprites built
 * outside the initializer (slot_persist) -- is compared with the live record when the pool's record is taken over; a
 * record that does not match (the host reset the env or loaded a state meanwhile) is dropped and the env is reset in
 * place.  The pool is two episodes deep (records for the next episode and the one after it: an episode shorter than a fill
 * does not stall the call it ends in).  An env whose record is not ready when its episode ends waits for a fill that is
 * under way, or is reset in place.  Not available (MOOG_E_UNSUPPORTED): programs of the plain kernels (their resets are cheap), programs with a
 * MOOG_CELL_PSTATE op (the reset depends on the episode that has just ended), a runtime that serialises kernels.
 * Calls with injected uniforms never use the pool.  Costs 4 records per env of device memory.
 * moog_engine_get_reset_pool: whether it is on, and (synchronising the device) stats[5] = fill launches so far, episodes
 * opened from the pool, episodes opened by a reset in place, pool records rejected, take-overs that had to wait for a fill. */
int moog_engine_set_reset_pool(moog_engine_t* e, int32_t enabled);
int moog_engine_get_reset_pool(moog_engine_t* e, int32_t* enabled, int64_t* stats);

/* Usage of the dynamic layers (layers that rules append to: the reference's unbounded Python lists, create_sprites.py,
 * change_layer.py; here `layer_capacity` slots, overflow = MOOG_FAULT_LAYER_FULL).  Per layer, over all envs and calls
 * since create: high_water[l] = the most sprites an append ever needed room for (capacity + 1 once the layer overflowed),
 * dropped[l] = appends that found the layer full.  Both arrays have MOOG_MAX_LAYERS entries.  Synchronises the device. */
int moog_engine_layer_usage(moog_engine_t* e, int32_t* high_water, int32_t* dropped);

/* Per-kernel device timing: bits 0-7 of `enabled` are a mask over MOOG_K_* (bit k set: launches of kernel k are
 * bracketed by HIP events on the launch stream; 0 disables), bits 8-15 hold period - 1: every period-th launch of
 * a kernel is bracketed (an event pair costs about 5 us of stream time; period 1 = every launch).  Totals and the
 * number of bracketed launches are read back (synchronising) by moog_engine_kernel_time. */
int moog_engine_set_timing(moog_engine_t* e, int32_t enabled);
int moog_engine_kernel_time(moog_engine_t* e, int32_t kernel_id, double* total_ms,
                            int64_t* launches);

/* Deferred fault surfacing.  Device-side per-env faults (MOOG_FAULT_* bits in the i32 record, the
 * engine's counterpart of the reference's synchronous exceptions: sprite_generators.py:92-98 RecursionError,
 * portal.py:51-54 / collisions.py:322-326 ValueError, ...) are also OR-ed by the kernels into one
 * host-visible word.  This call reads that word WITHOUT synchronising the stream (it reports what has
 * arrived so far) and optionally clears the bits it returns; a binding polls it at the start of every
 * call and, when non-zero, synchronises and raises from the per-env fault words. */
int moog_engine_poll_faults(moog_engine_t* e, int32_t clear, int32_t* bits);

/* The rasteriser's static prefix: the leading sprite slots that every reset creates identically and
 * at rest (border walls) are rendered once at create; frames whose prefix equals that reference
 * bit for bit are composed on top of the cached picture (pil_renderer.py:104-111 draws them first,
 * so the result is the same).  Returns the number of slots in the prefix (0: feature unused) and,
 * when `image_dev` is not NULL, copies the cached picture [height][width][3] there. */
int moog_engine_static_prefix(moog_engine_t* e, int32_t* n_slots, uint8_t* image_dev, void* hip_stream);

/* The per-env prefix: leading sprites that stay put within an episode but differ between envs and episodes (a random
 * maze's walls, pacman.py:62-65).  Every env keeps a picture of its own prefix and a snapshot of the record it was drawn
 * from; before the frames of a call are drawn, a check launch compares each env's live prefix with its snapshot bit for
 * bit, the pictures of the envs that differ (a reset, an edited record) are drawn again, and the frame launch composes
 * the remaining sprites on top of the pictures -- the same pixels as drawing everything (pil_renderer.py:104-111 draws in
 * slot order).  The prefix starts as the leading slots the initializer creates at rest and SHRINKS to the first slot seen
 * changing in the middle of an episode (food that gets eaten), so what it covers is what really stays put.  Costs a frame
 * and a record per env of device memory and two small launches per call; used for frames of several tiles (wider or
 * taller than 128 pixels) when it covers at least 32 slots and 8 more than the static prefix; MOOG_RASTER_ENV_BG=0 in the
 * environment turns it off, =1 on for any frame size.  Returns the number of slots it covers now (0: unused). */
int moog_engine_env_prefix(moog_engine_t* e, int32_t* n_slots);

/* Which step kernel a program runs on: variant 0 = plain, 1 = + expression evaluator, run-time sampler, dynamic layers,
 * 2 = + the rare components (maze physics and walks, reset-time expressions, computed shapes, look-aheads).  The same program
 * steps 1.4 - 2.7 times slower on variant 2 (profiles/r04_variant_tax.txt), so a program that needs the rare components only
 * to BUILD an episode (everything but maze physics / maze walks / modifiers that assign sprite.angle / run-time generators with
 * computed factors) is stepped by variant 1 with a LATE RESET: a step kernel that cannot open an env's next episode -- from
 * the reset pool, if that is on -- marks the env, and the full reset kernel, launched behind every step launch, opens it
 * before the frames are drawn: the same time steps and records as a reset inside the step kernel (late_reset = 1).
 * MOOG_NO_LATE_RESET=1 in the environment: such programs are stepped by variant 2 as before. */
int moog_engine_kernel_variant(moog_engine_t* e, int32_t* variant, int32_t* late_reset);

/* Program-specialised step kernels.  The generic step kernels read the lowered config through the scalar cache and carry
 * every component of their variant; the same source compiled with the program as a compile-time constant
 * (csrc/moog_step_spec.hip, built ahead of time by moog/_spec.py -- `python -m moog._spec <config>` or
 * BatchedEnvironment(specialize=True)) drops what the program does not use and folds its parameters: same arithmetic, bit-identical
 * results, 7 % faster on the headline workload.  moog_engine_create looks for
 *     <MOOG_SPEC_DIR, default: the directory of the engine library + "/spec">/step_<hash>_d<variant>w<wps>.so
 * and uses it after checking the digest of the sources it was built from (moog_source_digest), ABI, variant and the embedded
 * program byte for byte (MOOG_STEP_SPEC=0: never).  An object that fails a check is reported on stderr and left alone.
 * moog_program_step_kernel (no device needed): the variant (0 plain, 1 + evaluator / sampler / dynamic layers, 2 + the rare
 * components), the register-allocation variant (waves per SIMD) and the hash the file name carries (FNV-1a 64 of the
 * program's bytes).  moog_engine_step_kernel: whether this engine steps with a specialised kernel. */
int moog_program_step_kernel(const moog_program_t* program, int32_t* variant, int32_t* wps, uint64_t* hash);
int moog_engine_step_kernel(moog_engine_t* e, int32_t* specialised);

/* Which rasteriser draws this engine's ordinary frames (what a profile of the run names): MOOG_RASTER_MASK = the mask
 * rasteriser (csrc/moog_raster_mask_core.h: one-tile frames -- padded width and height <= 128 --, polygons of <= 128
 * vertices, slots x copies <= 256, tables within 64 KB of LDS; plain, first-person and torus frames),
 * MOOG_RASTER_SPANS = the push / sort / span kernel (csrc/moog_raster_kernel.h: everything else, every prefix picture, and
 * every frame while a per-env prefix is active). */
enum { MOOG_RASTER_SPANS = 0, MOOG_RASTER_MASK = 1, MOOG_RASTER_MASK_COMPACT = 3 };   /* 3: the mask rasteriser with 4-byte edge records (programs whose 16-byte records keep frames off a CU) */
int moog_engine_raster_path(moog_engine_t* e, int32_t* path);

/* The frames' draw records (csrc/moog_draw_record.h: what the mask rasteriser reads -- per env a header, an item per sprite slot
 * and copy, the live vertices as integer canvas points) as the last launch left them, copied to the host (synchronises the
 * device).  *stride: bytes per env; *in_step: 1 when moog_engine_step's step kernel writes them (else the derive kernel in
 * front of every raster launch does).  host_out may be NULL (sizes only); bytes = the room at host_out, at least
 * n_envs x *stride.  MOOG_E_UNSUPPORTED when the program's frames are not the mask rasteriser's.  For tests: the record
 * the step kernel writes from LDS and the one derived from the stored state must be equal byte for byte. */
int moog_engine_read_draw_records(moog_engine_t* e, uint8_t* host_out, int64_t bytes, int64_t* stride, int32_t* in_step);

/* PILRenderer(color_to_rgb=<any callable>) (pil_renderer.py:72-76,108: the renderer calls it on every sprite's colour
 * triple when it draws): the callable is Python and stays on the host.  The binding evaluates it once per DISTINCT colour
 * triple (colours rarely change after a reset), keeps r | g << 8 | b << 16 per (env, sprite slot) in a device array
 * [n_envs][n_slots] and hands the array over here; the rasteriser then takes a live sprite's colour from it instead of
 * applying render.cmap (opacity still comes from the record).  NULL: back to render.cmap.  While set, the engine draws
 * every sprite every frame (no cached prefix pictures) and frames do not follow their env's step. */
int moog_engine_set_color_override(moog_engine_t* e, const uint32_t* rgb_dev);

/* Profiling aids, both 0 in production (they make results wrong: timing only).  `step_debug`: bit
 * mask that switches parts of the step kernel off / writes cycle counters instead of outputs;
 * `raster_stop` = k truncates the raster kernel after phase k.  The initial values come from the
 * environment variables MOOG_STEP_DEBUG / MOOG_RASTER_STOP, read once by moog_engine_create. */
int moog_engine_set_debug(moog_engine_t* e, int32_t step_debug, int32_t raster_stop);

/* Section sampling of the step kernel (a profiling aid; an engine created with the environment variable MOOG_WATCH=1 and a
 * library whose step kernel was built with -DMOOG_WATCH, tools/build_variant.sh): a second wavefront beside every env's
 * samples, every few hundred cycles, the section id the stepping wavefront last announced (one LDS store per section
 * entry: the stepped wavefront is not slowed by clock reads).  Copies the per-env sample counts of the calls since the
 * last clear, [n_envs][MOOG_WATCH_SECTIONS], to `host_out` (synchronises) and optionally clears them.  Without
 * MOOG_WATCH=1 the call fails with MOOG_E_UNSUPPORTED. */
#define MOOG_WATCH_SECTIONS 32
int moog_engine_read_watch(moog_engine_t* e, int32_t* host_out, int32_t clear);

#ifdef __cplusplus
}
#endif
#endif /* MOOG_ENGINE_H_ */
