/*
 * paintrl.h -- C ABI of the MI355X batched paint-coverage simulator.
 *
 * The reference (translearn/PaintRL) is pure Python and has no FFI layer; its
 * boundary for this path is the module-level simulator API of
 * PaintRLEnv/bullet_paint_wrapper.py:1327-1400 (load_part, fast_paint, paint,
 * get_guided_point, get_observation, get_normalized_pose, reset_part,
 * get_job_status ...) as driven by Robot.apply_action (PaintRLEnv/robot.py:383-433)
 * and PaintGymEnv.step/reset (PaintRLEnv/robot_gym_env.py:349-387).  Each entry
 * point below names the reference interface it replaces.  A maintainer binds
 * these with ctypes; paintrl_amd/_lib.py is that binding.
 *
 * Conventions: every function returns 0 on success and a negative PRL_E* code
 * on failure (never throws); prl_last_error() gives the text.  The caller owns
 * all buffers; device buffers are plain device pointers (e.g. torch
 * tensor.data_ptr()).  The library owns only the opaque handles.  All launches
 * go on the caller's HIP stream (`stream` is a hipStream_t passed as void*,
 * NULL = the default stream).  A handle is not thread-safe; distinct handles are
 * independent.  All floating point is float64, like the reference.
 */
#ifndef PAINTRL_H
#define PAINTRL_H

#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define PRL_ABI_VERSION 3

enum { PRL_OK = 0, PRL_E_INVALID = -1, PRL_E_HIP = -2, PRL_E_UNSUPPORTED = -3, PRL_E_NOMEM = -4 };
enum { PRL_OBS_SECTION = 0, PRL_OBS_GRID = 1, PRL_OBS_SIMPLE = 2, PRL_OBS_DISCRETE = 3 };   /* rge:166-173 */
enum { PRL_ACT_DISCRETE = 0, PRL_ACT_CONTINUOUS = 1 };                                     /* rge:159-165 */
enum { PRL_TERM_LATE = 0, PRL_TERM_EARLY = 1, PRL_TERM_HYBRID = 2 };                       /* rge:142-147 */
enum { PRL_PAINT_FAST = 0, PRL_PAINT_NORMAL = 1 };                                         /* rob:171-172 */
enum { PRL_COLOR_RGB = 0, PRL_COLOR_HSI = 1 };                                            /* rge:156 COLOR_MODE */

#define PRL_STATE_DOUBLES 16     /* per-env scalar state record, see prl_batch_get_state */
#define PRL_MAX_DISCRETE 64

typedef struct PrlPart PrlPart;     /* static tables of one part, resident in HBM */
typedef struct PrlBatch PrlBatch;   /* N environments: coverage masks + scalar state in HBM */

/*
 * Host-side static tables of one part, produced by paintrl_amd.device_tables
 * from paintrl_amd.part_tables (the restatement of bpw.load_part, bpw:1327-1335).
 * All arrays are host pointers, copied by prl_part_create.
 */
typedef struct {
    /* coverage samples in device order, padded to a multiple of 64: sorted by sample-grid cell row, then
       ascending on axis a1 (which also orders a row by cell).  prl_part_create rejects a table in which the
       64 samples of a word do not ascend on axis a1: the 4-sector observation finds the samples left and
       right of the tool by binary search in the word. */
    int32_t n_samples;            /* real samples (bpw Part.get_job_limit) */
    int32_t n_samples_pad;
    const double *sample_xyz[3];  /* world x, y, z, [n_samples_pad]; pads are far away */
    const double *word_bbox;      /* [n_samples_pad/64][4]: min/max on axis a1, min/max on axis a2 */
    const uint64_t *word_valid;   /* [n_samples_pad/64] bit set = real sample */
    const int32_t *sample_rank;   /* [n_samples_pad] index in the reference order (nearest-sample tie break), pads INT32_MAX */
    /* uniform grid over the principal plane that orders the samples */
    double sgrid_origin[2], sgrid_inv_cell;
    int32_t sgrid_nx, sgrid_ny;
    const int32_t *sgrid_start;   /* [nx*ny+1] first sample of each cell; every cell row starts on a 64-sample
                                     word, the pads in between are far away and never valid */
    /* grid observation (bpw GridObservation): one sample bitmask per observation cell */
    int32_t n_obs_cells;
    const uint64_t *obs_cell_mask;   /* [n_obs_cells][n_samples_pad/64] */
    const int32_t *obs_cell_count;   /* [n_obs_cells] */
    /* same-side vertices (bpw vertices_kd_tree[side]) sorted by vertex-grid cell */
    int32_t n_vertices;
    const double *vertex_xyz[3];
    const int32_t *vertex_rank;      /* original order, the nearest-vertex tie break */
    int32_t adj_width;               /* incident same-side triangles per vertex, padded (<= 64) */
    const int32_t *vertex_adj;       /* [n_vertices][adj_width] triangle ids in file order (bpw uv_map), -1 = pad */
    double vgrid_origin[2], vgrid_inv_cell;
    double vgrid_accept;             /* 0.99 * cell edge: a ring-k search is exact within k * this distance */
    int32_t vgrid_nx, vgrid_ny;
    const int32_t *vgrid_start;      /* [nx*ny+1] */
    /* Only for parts whose vertex rows the reference moved AFTER building its cKDTree (bpw:943-946, sparse grid rows):
       that tree, flattened (paintrl_amd/part_tables.py), so that the nearest-vertex query of bpw:526 is walked the
       way scipy walks it -- split planes of the old rows, leaf distances to the moved ones.  n_kd_nodes = 0: the
       exact nearest vertex (every part the reference does not touch, all synthetic parts). */
    int32_t n_kd_nodes;
    const int32_t *kd_node;          /* [n_kd_nodes][4]: split dim (-1 = leaf), lesser | first point, greater | end, 0 */
    const double *kd_split;          /* [n_kd_nodes] */
    int32_t n_kd_points;
    const int32_t *kd_points;        /* [n_kd_points] positions in vertex_xyz, tree order; -1 = a row of another side */
    double kd_box[6];                /* mins[3], maxes[3] of the rows when the tree was built */
    /* same-side triangle records (bpw BarycentricInterpolator), 16 doubles each:
       a[3] v0[3] v1[3] d00 d01 d11 inv_denom normal[3] */
    int32_t n_triangles;
    const double *tri_records;
    /* collision triangle set (what pybullet.rayTestBatch hits), padded to a multiple of 64 */
    int32_t n_collision;
    int32_t n_collision_pad;
    const double *col_v0e1e2[9];     /* v0.xyz e1.xyz e2.xyz, [n_collision_pad] each */
    const float *col_bbox;           /* [n_collision_pad][8] outward-rounded box: lo,hi on axis1, axis2, axis0, 2 pad */
    const int32_t *col_rank;         /* [n_collision_pad] index in the reference (tie-break) order */
    int32_t col_convex;              /* 1: the set is the boundary of a convex polytope (hull mode): enables the
                                        neighbourhood-first ray path */
    int32_t nbr_width;               /* <= 64 */
    const int32_t *col_nbr;          /* [n_collision_pad][nbr_width] facets sharing a vertex, itself first, -1 = pad;
                                        a row of -1 disables the fast path for that facet */
    const int32_t *col_orient;       /* [n_collision_pad] +1/-1: orient * det > 0 <=> the segment enters through it */
    int32_t n_col_chunks;            /* n_collision_pad / 64 */
    const float *col_chunk_bbox;     /* [n_col_chunks padded to 64][8] union box of each 64-triangle chunk */
    /* grid rows (bpw grid_dict), extents, axes */
    const double *grid_lo, *grid_hi; /* [100] */
    double range1[2], range2[2], length_width_ratio;
    int32_t axis0, axis1, axis2;
    /* start points (bpw _start_points) with the quaternion of rob:93-100 */
    int32_t n_start;
    const double *start_pos;         /* [n_start][3] */
    const double *start_quat;        /* [n_start][4] */
    /* cone beams for PAINT_METHOD='normal': the uniform lattice of rob:23-35, or for COLOR_MODE 'HSI' the beta-profile rings of
       rob:38-69 (paintrl_amd.part_tables.beta_plain); any table of beam end points in the tool frame will do */
    int32_t n_beams;
    const double *beams;             /* [n_beams][3] */
} PrlPartTables;

/* PaintGymEnv configuration (rge:127-157) as a POD. */
typedef struct {
    int32_t obs_mode, obs_grad;
    int32_t action_mode, action_dim, n_discrete;
    int32_t termination_mode;
    int32_t turning_penalty, overlap_penalty;
    int32_t paint_method;
    int32_t max_episode_len, expected_episode_len;
    int32_t auto_reset;              /* 1: a finished env is reset inside the step kernel */
    int32_t color_mode;              /* PRL_COLOR_RGB: a texel is painted or not.  PRL_COLOR_HSI (bpw:384-434): every
                                        front texel carries a uint8 that shots lower by 1..26 per hit (wrapping, as
                                        the reference's numpy uint8 does); "painted" for observations stays byte == 255 */
    int32_t reserved_;
    double switch_threshold;
    double paint_radius, step_size;  /* PaintToolProfile.PAINT_RADIUS / STEP_SIZE (bpw:40-43), default 0.051 both;
                                        the part must have been built and packed for the same radius */
    double max_possible_point[8];    /* Part_Dict[...][1] per part id (rge:106-117) */
    uint64_t seed;                   /* start-point RNG for auto_reset / reset without indices */
    /* discrete action -> (delta_axis1, delta_axis2, turning angle), evaluated on the host with the
       reference's own numpy calls (rge:342-347, rob:151-153, 396-397, 352-356) */
    double act_delta1[PRL_MAX_DISCRETE], act_delta2[PRL_MAX_DISCRETE], act_angle[PRL_MAX_DISCRETE];
} PrlConfig;

int prl_abi_version(void);
const char *prl_last_error(void);
int prl_obs_dim(const PrlConfig *cfg);                       /* rge:166-173 */
/* sizeof(PrlConfig) / sizeof(PrlPartTables) as compiled, so a binding can check its struct layout */
int prl_struct_sizes(int *config_bytes, int *part_tables_bytes);

/* Replaces bullet_paint_wrapper.load_part (bpw:1327-1335): upload one part's tables to `device`. */
int prl_part_create(const PrlPartTables *host_tables, int device, PrlPart **out);
void prl_part_destroy(PrlPart *part);
int prl_part_mask_words(const PrlPart *part);                /* 64-bit words of one env's coverage mask */

/* Replaces PaintGymEnv.__init__ (rge:207-229) for N envs; env i paints parts[env_part_id[i]]
 * (env_part_id host pointer, NULL = all part 0).  Allocates coverage masks and scalar state. */
int prl_batch_create(PrlPart *const *parts, int n_parts, const int32_t *env_part_id, int n_envs,
                     const PrlConfig *cfg, PrlBatch **out);
void prl_batch_destroy(PrlBatch *batch);
int prl_batch_mask_stride(const PrlBatch *batch);            /* 64-bit words per env in get_mask */

/* Replaces PaintGymEnv.reset + reset_part + Robot.reset (rge:370-387, bpw:706-712, rob:366-372).
 * reset_mask: device u8[N] or NULL (= all).  start_idx: device i32[N] or NULL (= library RNG).
 * obs: device f64[N][obs_dim] or NULL; rows of envs that are not reset are left untouched. */
int prl_batch_reset(PrlBatch *batch, const uint8_t *reset_mask, const int32_t *start_idx, double *obs, void *stream);

/* Replaces PaintGymEnv.step (rge:349-368) and everything under it for all N envs in ONE kernel.
 * actions: device i32[N] (discrete) or f64[N][action_dim] (continuous).
 * obs f64[N][obs_dim], reward f64[N] (= reward - penalty), done u8[N], info f64[N][2] (reward, penalty).
 * With cfg.auto_reset the obs row of a finished env is its post-reset observation and, if
 * final_obs != NULL, the terminal observation goes to final_obs f64[N][obs_dim].
 * start_idx: device i32[N] start points for auto-reset, or NULL (= library RNG). */
int prl_batch_step(PrlBatch *batch, const void *actions, double *obs, double *reward, uint8_t *done, double *info,
                   double *final_obs, const int32_t *start_idx, void *stream);

/* How a prl_batch_step launch of this batch (PAINT_METHOD 'fast': its one kernel) occupies the chip -- for bench.py's
 * line, no reference counterpart: out[0] = wavefronts (= envs) per workgroup, out[1] = workgroups resident per CU as the
 * runtime's occupancy calculator reports them (registers, LDS), out[2] = dynamic LDS bytes per workgroup, out[3] = CUs. */
int prl_batch_step_occupancy(PrlBatch *batch, int32_t *out /* host, [4] */);

/* Replaces Robot.reset(pose) (rob:366-372, used by spiral.py:38): place env `env_index` at `pos` with tool
 * quaternion `quat` (host pointers, xyzw) and clear its off-part bookkeeping (rob:208-212).  Coverage,
 * step counter and reward accumulators are left as they are, like the reference.  Synchronous. */
int prl_batch_set_pose(PrlBatch *batch, int env_index, const double *pos, const double *quat);
/* rge:306-319 _augmented_observation of the current state (no state change): obs f64[N][obs_dim], device.
 * What PaintGymEnv.reset() returns after env.robot.reset(pose) in the reference's scripts (spiral.py:38). */
int prl_batch_observe(PrlBatch *batch, double *obs, void *stream);

/* Replaces get_job_status / get_texture_image style read-back (bpw:727-738): coverage bits in
 * device sample order, u64[N][mask_stride]. */
int prl_batch_get_mask(PrlBatch *batch, uint64_t *painted, void *stream);
/* The previous shot's affected set (Part._last_painted_pixels, bpw:483, 575-576: what OVERLAP_PENALTY's pixel counter is
 * measured against) as bits in device sample order, u64[N][mask_stride], device.  nonzero_words (device, or NULL):
 * u64[N][*nonzero_stride], the library's own index of that row -- bit w & 63 of word w >> 6 is set for every word w of the row
 * that may be non-zero (a superset; kernels read and clear only those) -- for tests; nonzero_stride: host int, or NULL. */
int prl_batch_get_last_mask(PrlBatch *batch, uint64_t *last, uint64_t *nonzero_words, int32_t *nonzero_stride, void *stream);
/* COLOR_MODE 'HSI' only: the thickness byte of every sample, u8[N][64 * mask_stride] in device sample order
 * (what bpw texels[get_texel(i, j)] holds for the front samples). */
int prl_batch_get_thickness(PrlBatch *batch, uint8_t *thick, void *stream);
/* Per-env scalar state f64[N][PRL_STATE_DOUBLES]: pose[3] quat[4] last_turning_angle total_reward
 * total_return {i32 terminate, terminate_counter} {i32 last_on_part, step_counter}
 * {u32 episode_count, i32 facet_hint} last_episode_return last_episode_reward
 * {i32 last_episode_len, last_episode_painted}.  facet_hint is a cache (the collision triangle the
 * previous ray hit, -1 = none), not reference state: any value gives the same results. */
int prl_batch_get_state(PrlBatch *batch, double *state, void *stream);
/* Episode returns (rge:359-360 _total_return of the last finished episode) -- the RCCL gather payload. */
int prl_batch_get_returns(PrlBatch *batch, double *episode_return, void *stream);

/* Replaces pybullet.rayTestBatch (bpw:873,918; rob:282) against a part's collision triangles:
 * from/to f64[n][3] -> tri i32[n] (-1 = miss), frac f64[n], pos f64[n][3]; all device pointers. */
int prl_ray_batch(PrlPart *part, int n, const double *from, const double *to, int32_t *tri, double *frac,
                  double *pos, void *stream);

/* Kernel timing for bench.py: HIP events recorded on the launch stream around every `every`-th step launch
 * (0 = off, 1 = every launch).  prl_batch_timing_read synchronises, returns the summed duration and the
 * number of launches that were timed, and resets the counters. */
int prl_batch_timing_enable(PrlBatch *batch, int every);
int prl_batch_timing_read(PrlBatch *batch, double *total_ms, int64_t *launches);

/* ---- rollout policy: the caller of the step in BASELINE.json configs 3-4 ----------
 * The policy of paint_ppo.py:170-195 (model fcnet_hiddens [256, 128], RLlib's default tanh, a linear
 * logits head and a linear value head) as ONE launch on the env's stream, so that a rollout worker keeps
 * the host out of the per-step loop: obs f64[N][in_dim] (what prl_batch_step wrote) -> action i32[N]
 * (what the next prl_batch_step reads), log-probability, value estimate, optionally the logits.
 * Weights are f32, row-major [in][out] (the transpose of torch.nn.Linear.weight); w3/b3 hold the
 * n_actions logit columns followed by the value column.  Arithmetic is f32 (a k-ordered fmaf chain per
 * output, a fast tanh, expf/logf); the action is the inverse-CDF draw for uniform[env] in [0, 1), or,
 * with uniform == NULL, for a counter-based random number keyed by (rng_seed, env, rng_count[env]++):
 * rng_count is a zero-initialised u32[N] device array the caller keeps alive (race-free, and safe to
 * replay from a captured HIP graph).  Limits: hidden sizes multiples of 16, n_actions <= 15, 64 KB of
 * LDS per 16 envs. */
typedef struct {
    int32_t in_dim, h1, h2, n_actions;
    const float *w1, *b1;        /* [in_dim][h1], [h1] */
    const float *w2, *b2;        /* [h1][h2], [h2] */
    const float *w3, *b3;        /* [h2][n_actions + 1], [n_actions + 1] */
} PrlPolicyWeights;              /* host struct holding device pointers */
int prl_policy_act(const PrlPolicyWeights *weights, int n, const double *obs, const float *uniform /* or NULL */,
                   uint32_t *rng_count /* or NULL if uniform is given */, uint64_t rng_seed, int32_t *action,
                   float *logp /* or NULL */, float *value /* or NULL */, float *logits /* or NULL, [N][n_actions] */,
                   void *stream);

/* Policy + env step in ONE launch: what a rollout worker does per step (paint_ppo.py:170-195: policy forward on the
 * current observations, sample, env.step).  Sixteen envs per workgroup run the policy together, each wave then steps
 * its env with the sampled action.  obs_in f64[N][obs_dim] is what the policy sees (the obs a previous step or
 * reset wrote; it may be the same buffer as obs only if the caller no longer needs it: rows are read before they are
 * written, by the same workgroup).  action i32[N], logp f32[N], value f32[N] out; the remaining arguments are
 * prl_batch_step's.  Results are bit for bit those of prl_policy_act (in-kernel sampling stream: rng_count u32[N],
 * rng_seed) followed by prl_batch_step.  The batch must have auto_reset and discrete actions.  ONE launch for the ball
 * painter (PAINT_METHOD 'fast', COLOR_MODE 'RGB', 4-sector / grid / simple observation) on parts of at most 16 384
 * samples; any other configuration is served by the two launches this call is defined to equal. */
int prl_batch_act_step(PrlBatch *batch, const PrlPolicyWeights *weights, const double *obs_in, uint32_t *rng_count,
                       uint64_t rng_seed, int32_t *action, float *logp, float *value, double *obs, double *reward,
                       uint8_t *done, double *info, double *final_obs /* or NULL */, void *stream);

/* A whole rollout fragment -- what one RLlib rollout worker does between two learner updates
 * (paint_ppo.py:170-195, sample_batch_size steps of policy forward + env.step) -- enqueued by ONE call, no host code of
 * the caller between the steps.  The batch must have been created with auto_reset and discrete actions.  All buffers
 * are device pointers, row-major [t][env]:
 *   obs        f64[n_steps + 1][N][obs_dim]   row 0: observations before the first step (input), row t + 1: after step t
 *   final_obs  f64[n_steps][N][obs_dim] or NULL: terminal observation of envs that finished in step t
 *   reward f64[n_steps][N], done u8[n_steps][N], info f64[n_steps][N][2]
 *   action     i32[n_steps][N]: written when `weights` is given, otherwise READ (replay / scripted / random actions)
 *   logp, value f32[n_steps][N], last_value f32[N] (value estimate of row n_steps), rng_count u32[N]: policy only
 * ONE persistent launch either way, sixteen envs per workgroup (the configurations of prl_batch_act_step's one launch).
 * Given actions (weights == NULL): every wave walks its
 * env through all n_steps on its own (no barrier, no launch boundary: a slow step delays nobody), coverage masks in LDS
 * throughout.  With `weights`: the sixteen waves of a workgroup alternate the policy (together) and the step (each its
 * env) and meet only at the policy's barriers; after the last step the policy runs once more for last_value (its draw
 * is discarded).  The rows are bit for bit what n_steps rounds of prl_policy_act + prl_batch_step produce -- and for the
 * configurations the persistent kernels are not built for (cone beams, COLOR_MODE 'HSI', atan2 sectors, parts beyond 16 384
 * samples) that is how the call is carried out: launch by launch, still with no host code of the caller in between. */
int prl_rollout_fragment(PrlBatch *batch, const PrlPolicyWeights *weights /* or NULL */, int n_steps, double *obs,
                         double *final_obs, double *reward, uint8_t *done, double *info, int32_t *action, float *logp,
                         float *value, float *last_value, uint32_t *rng_count, uint64_t rng_seed, void *stream);

#ifdef __cplusplus
}
#endif
#endif /* PAINTRL_H */
