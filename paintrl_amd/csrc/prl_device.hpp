// prl_device.hpp -- constants, table descriptor, wave-level helpers, reference arithmetic.
// Part of the single translation unit paintrl_hip.hip (device code, anonymous namespace); see that
// file for the overall design.  Compile with -ffp-contract=off.
#pragma once

namespace {


// Envs (= waves) per workgroup of step_kernel: 8 while every workgroup of the launch is resident at once (N <= 16 waves x
// CUs: fewer workgroups to dispatch, 41.8 -> 41.05 us at 4 096 envs), 4 beyond that (a finishing 4-wave workgroup is
// backfilled sooner: at 8 192 envs 80.2 vs 90.2 us, at 32 768 298 vs 319); cone beams: 4.
constexpr int STEP_WAVES_WIDE = 8, STEP_WAVES_NARROW = 4;
constexpr int MAX_WAVES_PER_WG = STEP_WAVES_WIDE;   // per-wave LDS scratch rows
constexpr int KW_MAX = 4;                       // mask slots per lane: up to 64*64*4 = 16384 samples in registers
constexpr int BIG_MAX_WORDS = 1600;             // larger parts: masks in LDS, 3-5 copies per wave, as many waves (<= 4) as fit next to the static LDS (k_big.hip)
constexpr double PAINT_RADIUS = 0.051;          // bpw:42
constexpr double STEP_SIZE = 0.051;             // bpw:43
constexpr double HOOK_DISTANCE = 0.1;           // bpw:443
constexpr int GRID_GRANULARITY = 100;           // bpw:447
constexpr int PAINT_PER_ACTION = 5;             // rob:165
constexpr int NOT_ON_PART_TERMINATE = 1000;     // rob:167
constexpr double RAY_EPS_DET = 1e-12;
constexpr double RAY_EPS_BARY = 1e-9;
constexpr double PI = 3.141592653589793;
constexpr double SHOT_CENTRE_OFFSET = 0.1;      // rob:277-278: the shot centre lies 0.1 ahead of the tool
// Device triangle record: a[3] v0[3] v1[3] d00 d01 d11 inv normal[3] | quat[4] centre_off[3] pad.  The tail is what
// a hit on this triangle makes of its normal, computed once on upload with the device's own arithmetic:
// quat = get_pose_orn(-normal) (rob:93-100), centre_off = R(quat) (0, 0, 0.1) (rob:277-278).
constexpr int TRI_REC = 24;
#define PRL_CONE_WORK_LISTS 256                 // sub-lists of the far list and of the ray list (k_cone_beams.hip)
// entries of a ray sub-list: eight leftover rays per beam trip on average and a few full trips (a step has ~0.03 per trip;
// a full sub-list sends the trip through the general code): host and device size the list alike
__host__ __device__ constexpr int prl_cone_ray_sub_cap(int items) { return 256 + 8 * ((items + PRL_CONE_WORK_LISTS - 1) / PRL_CONE_WORK_LISTS); }
constexpr int PY_MAX_LEVELS = 13;               // box pyramid over the fine sample grid (PartDev::py_*): grids up to 4096 cells wide

// Table pointers are read from a descriptor in memory, so the compiler cannot infer their address
// space and would emit flat_load (out-of-order, waits on vmcnt AND lgkmcnt).  Typing them as global
// (address space 1) gives global_load with counted vmcnt waits.
#define GAS __attribute__((address_space(1)))
#define CAS __attribute__((address_space(4)))
typedef const double GAS *gdouble_p;
typedef const float GAS *gfloat_p;
typedef const int GAS *gint_p;
typedef const uint64_t GAS *gu64_p;
typedef const uint8_t GAS *gu8_p;
typedef uint64_t GAS *gu64_rw_p;                          // a mask row in HBM, read and written (HbmMasks)
typedef float f32x4 __attribute__((ext_vector_type(4)));     // native vectors: loadable through GAS pointers
typedef float f32x2 __attribute__((ext_vector_type(2)));
typedef double f64x2 __attribute__((ext_vector_type(2)));
typedef double f64x4 __attribute__((ext_vector_type(4)));
typedef int i32x4 __attribute__((ext_vector_type(4)));

struct PartDev {
    int n_samples, n_samples_pad, n_words;
    gdouble_p samp[3];
    gdouble_p samp_a1, samp_a2;   // = samp[a1], samp[a2]: a dynamic index into samp[] would be a memory load of the pointer
    gdouble_p word_bbox;
    gu64_p word_valid;
    gint_p samp_rank;             // tie rank of each device sample: its place in the reference cKDTree's own index array
                                  // (part_tables._sample_tie_rank; equal distances resolve to the lowest), pads = INT_MAX
    gu8_p samp_ub;                // in-word index one past the last sample with the same a1 coordinate (derived in part_fill)
    gdouble_p word_pivot;         // [n_words][8]: a1 coordinate of in-word samples 7, 15, .. 63 (derived in part_fill)
    // large parts (more than 16 384 samples; derived in part_fill, else null): per word its valid samples' a2 coordinates in
    // ascending order (64, padded with +inf) and, for r = 0 .. 64, the mask of the samples from sorted position r on -- with
    // r = the number of samples below (or not above) the tool's a2 line the samples above it (or not below it) are one mask:
    // the observation's pass over the tool's own cell row counts whole words instead of samples (section4_big, round 5)
    gdouble_p word_ysort;         // [n_words][64]
    gdouble_p word_ypivot;        // [n_words][8]: entries 7, 15, .. 63 of word_ysort (the group of eight that holds a given a2 in ONE probe)
    gu64_p word_ymask;            // [n_words][65]
    gfloat_p samp_a2_f32;         // [n_samples_pad]: the a2 coordinate rounded to the nearest float (derived in part_fill)
    // large parts' observation (prl_observe.hpp section4_big; derived in part_fill):
    gint_p word_cells;            // [n_words]: the (at most four) grid-observation cells the word's valid samples lie in, one byte each,
                                  // 0xff = none; 0xfffffffe: more than four (or a cell index beyond 254): every cell's mask is tried
    gint_p word_info;             // [n_words]: the word's cell row (low 16 bits) | its number of valid samples << 16
    gfloat_p word_x32;            // [n_words][2]: float interval holding the a1 coordinates of its valid samples (rounded outward)
    // fine sample grid for the cone-beam painter's nearest-sample queries, one query per lane (prl_cone.hpp; derived
    // in part_fill): cells of ~2.6 samples, fg_rec = the samples sorted by cell as (x, y, z, {i32 rank, i32 device pos})
    double fg_o1, fg_o2, fg_inv, fg_accept;       // fg_accept = 0.99 * cell edge
    int fg_nx, fg_ny;
    gint_p fg_start;              // [fg_nx * fg_ny + 1]
    gdouble_p fg_rec;             // [n_samples][4]
    gfloat_p fg_rec32;            // [n_samples][4]: the same records as x y z rounded to float | device position (int bits)
    // box pyramid over the fine grid (derived in part_fill): level 0 = the cells, level l = 2^l x 2^l of them; a node is the
    // bounding box of its samples as 8 floats (lo x y z, -, hi x y z, -; rounded outward; empty: lo = +inf, hi = -inf; the
    // spare floats of a CELL hold its record range fg_start[c], fg_start[c + 1] as int bits).
    // nearest_sample_bfs walks it branch and bound: exact nearest samples of points centimetres to decimetres from the
    // sampled surface (the collision hull spans windows and recesses of the part), where a ring of cells has no grip.
    int py_levels;                // 0: no pyramid
    int py_off[PY_MAX_LEVELS], py_nx[PY_MAX_LEVELS], py_ny[PY_MAX_LEVELS];      // first node / dimensions of each level
    gfloat_p py_box;
    gint_p fg_seed;               // [fg_nx * fg_ny]: a record near the cell (its own centre-nearest sample, or that of the nearest
                                  // cell that has one; -1: the part has no sample): the first bound of nearest_sample_bfs
    // outline of the collision set in the principal plane (convex polygon, derived in part_fill) and its extent along
    // the third axis: a beam whose stretch inside that slab projects outside the outline misses the part (prl_cone.hpp)
    int n_outline;                // edges, 0 = no test; the table is padded to a multiple of 64 rows
    gdouble_p outline;            // [n_outline_pad][4]: a point of the edge (axis a1, a2), its outward unit normal
    double slab_lo, slab_hi;      // collision vertices' range on axis a0
    gfloat_p samp_f32;            // [n_samples_pad][4]: x y z 0 rounded to float (derived): the paint pre-filter
    double samp_absmax;           // largest |coordinate| of a real sample: bounds the pre-filter's rounding error
    double sg_o1, sg_o2, sg_inv;
    int sg_nx, sg_ny;
    gint_p sg_start;
    int n_obs_cells;
    gu64_p cell_mask;
    gint_p cell_count;
    int n_vertices;
    gdouble_p vert4;              // [n_vertices][4]: x y z | {i32 reference rank, 0}  (derived in part_fill: one record
                                  // per candidate = two 16-byte loads instead of four loads from four tables)
    int adj_width;
    gint_p vadj;
    double vg_o1, vg_o2, vg_inv, vg_accept;
    int vg_nx, vg_ny;
    gint_p vg_start;
    int n_kd_nodes;               // > 0: the reference's stale vertex kd-tree (include/paintrl.h); nearest_vertex_kd walks it
    gint_p kd_node;               // [n][4] split dim | lesser or first point | greater or end | 0
    gdouble_p kd_split;
    gint_p kd_points;
    gdouble_p kd_rec;             // [n_kd_points][4]: x y z | {i32 vertex or -1, i32 the leaf's node} of the leaves' points in tree order (derived in part_fill)
    int n_kd_points;
    // the tree for the lane-parallel query (nearest_vertex_kd_lanes; derived in part_fill when it has <= 64 nodes, <= 32 leaves of
    // <= 16 points): node n's static word for lane n {split dim or 3 | parent's dim << 2 | lesser child << 4 | depth << 5 |
    // parent << 11 | lesser << 17 | greater << 23, the leaf's ordinal or -1, the parent's split value}; the leaves' points as
    // x y z | vertex in rows of 16 (leaf ordinal * 16, padded with vertex -1)
    gint_p kd_lane;
    gu64_p kd_anc;                // [n] the nodes on the way from the root to node n's parent (root excluded) that were reached across a plane of
                                  // the same dimension as node n was: the deepest of them that the query reached as the FAR child set the old side distance
    gdouble_p kd_rec16;
    int n_kd_leaves;              // 0: no such tables
    int kd_depth;
    double kd_box[6];
    int n_triangles;
    gdouble_p tri_rec;            // [n_triangles][TRI_REC]: the 16 doubles of the host table + derived tail (part_fill)
    int n_col, n_col_pad;
    gdouble_p col[9];
    gfloat_p col_bbox;
    gint_p col_rank;
    int col_convex, nbr_width;
    gint_p col_nbr, col_orient;
    gint_p col_enbr;              // convex sets: [n_col_pad][4] facet across the edge u = 0 / v = 0 / u + v = 1, or -1 | col_rank (derived)
    // convex sets: the facet met by a line along axis a0 through the centre of each cell of a grid over the set's outline
    // (the one on the tool's side; cells beside the set hold their nearest neighbour's): where a cone beam's walk starts
    double hg_o1, hg_o2, hg_inv;
    int hg_nx, hg_ny;             // 0: no grid
    gint_p hg_facet;              // [hg_ny][hg_nx] (derived in part_fill)
    gdouble_p col_rec;            // convex sets: [n_col_pad][12] v0 e1 e2 | edge margin | |e1 x e2|^2 | orient (derived in part_fill)
    int n_col_chunks;
    gfloat_p col_chunk_bbox;
    gdouble_p grid_lo, grid_hi;
    double r1min, r1max, r2min, r2max, lwr;
    int a0, a1, a2;
    int n_start;
    gdouble_p start_pos, start_quat;
    int n_beams;
    gdouble_p beams;
    // Per BATCH (filled in the batch's own copy of the descriptor, prl_batch_create): the observation a reset to each
    // start point returns, [n_start][obs_dim].  Right after a reset nothing is painted, so that observation depends
    // on the start point and the configuration only; computing it once (reset_obs_kernel, the same device code)
    // takes a whole observation pass out of every step in which an episode ends.
    gdouble_p reset_obs;
};

// The part descriptor and the batch configuration are read-only for every kernel: typed as constant
// address space so that their fields are fetched with scalar loads (s_load through the K$) instead
// of wave-uniform vector loads the compiler has to assume the kernel's own stores may clobber.
typedef const PartDev CAS &PartRef;
// The batch configuration as the kernels see it: the caller's PrlConfig plus what prl_batch_create derives from it once,
// with the same IEEE divisions the step would otherwise repeat per env and step (a float64 division is ~14 vector
// instructions, its reciprocal seed 16 cycles: the dozen the step used to make were ~5 % of its vector time).
struct CfgDev : PrlConfig {
    double act_d1[PRL_MAX_DISCRETE], act_d2[PRL_MAX_DISCRETE];     // act_delta1 / 2 over PAINT_PER_ACTION (rob:403-409)
    double expected_reward[8];                                      // max_possible_point / (Expected_Episode_Length * 100), rge:296
    double switch_points[8];                                        // SWITCH_THRESHOLD * max_possible_point / 100, rge:301
};
typedef const CfgDev CAS &CfgRef;

struct StepArgs {
    const PartDev *parts;
    const CfgDev *cfg;
    const int *env_part;          // device, or nullptr
    const int *slot_env;          // device, or nullptr: step_kernel's wave slot -> env (mixed batches: every XCD's workgroups get the
                                  // envs of as few parts as possible, so that its L2 holds one part's tables; paintrl_hip.hip)
    int n_envs, mask_stride;
    uint64_t *painted, *last;
    uint64_t *last_nz;            // [n_envs][nz_stride]: bit w & 63 of word w >> 6 = word w of the env's last-shot row is not zero (every
                                  // writer of `last` keeps it, or sets it to all ones)
    int nz_stride;                // KW_MAX for parts of up to 16 384 samples, (mask_stride + 63) / 64 for larger ones (HbmMasks)
    uint8_t *thick;               // COLOR_MODE 'HSI': one byte per sample, [n_envs][64 * mask_stride]; else nullptr
    double *state;
    const void *actions;
    double *obs, *reward, *info, *final_obs;
    uint8_t *done;
    const int *start_idx;
    const uint8_t *reset_mask;
    // PAINT_METHOD 'normal' only: what the five cone-beam launches of a step hand to each other (k_cone_beams.hip, k_cone.hip)
    double *cone_shots;           // [n_envs][5][8]: tool pose after each sub-shot (pos, quat) | {i32 facet hint, 0}
    double *cone_aux;             // [n_envs][2]: new turning angle | {i32 off-part counter before the step, i32 facet hint}
    int *cone_hits;               // [n_envs][5][cone_nb]: device position of the sample each beam paints, or -1
    int *cone_work;               // counters and work lists of a cone-beam step (k_cone_beams.hip)
    double *cone_far;             // far list, WORK_LISTS sub-lists of far_cap entries [4]: hit point x y z | {hi: f32 bits of the distance
                                  // bound the beams kernel saw, lo: i32 index into cone_hits}
    int cone_nb;                  // beams per shot, padded to 64 (the largest beam count of the batch's parts)
};

// ---------------------------------------------------------------- table loads
// Table indices are non-negative ints.  Indexing a global pointer with a SIGNED 32-bit value makes the
// compiler sign-extend it and build a 64-bit address per lane (v_ashrrev + v_lshl_add_u64, then a load
// with a VGPR-pair address); a 32-bit unsigned BYTE offset lets it keep the table base in scalar registers
// and use the "saddr + 32-bit voffset" form of global_load (one v_lshlrev per load).  Every table is far
// smaller than 4 GB, so the byte offset cannot wrap.
template <typename T>
__device__ __forceinline__ T ldg(const T GAS *p, int i) {
    const uint32_t off = (uint32_t)i * (uint32_t)sizeof(T);
    return *reinterpret_cast<const T GAS *>(reinterpret_cast<const char GAS *>(p) + off);
}

// ---------------------------------------------------------------- wave helpers
// HIP's __ballot(int) compares its argument with zero again (a v_cndmask + a v_cmp per call on top of the compare that
// produced the predicate); the builtin takes the predicate's own lane mask.  Same result: bit l = predicate of active lane l.
#ifdef PRL_OLD_BALLOT                    // (A/B switch)
__device__ __forceinline__ uint64_t ballot64(bool p) { return __ballot(p); }
#else
__device__ __forceinline__ uint64_t ballot64(bool p) { return __builtin_amdgcn_ballot_w64(p); }
#endif

__device__ __forceinline__ int rfl(int v) { return __builtin_amdgcn_readfirstlane(v); }

// Per-wave LDS scratch of a workgroup (MAX_WAVES_PER_WG waves): the ray's candidate list, the five shot centres of
// a step, the sector counters of the atan2 observation.  The rows are picked ONCE, at kernel entry, from the wave
// index held in a scalar register: indexed with `threadIdx.x >> 6` at the point of use the compiler keeps
// threadIdx.x and 64-bit generic row addresses alive in vector registers through the whole kernel -- and, at
// this kernel's register budget, spills them (a scratch access is 64 separate cache lines on gfx950).
constexpr int KD_HEAP = 64;         // queued cells of the stale kd-tree walk (5 doubles each)
constexpr int KD_LDS_NODES = 64;    // trees of at most this many nodes are walked from a copy in LDS (kd_stage, prl_search.hpp)
constexpr int KD_ROW = KD_HEAP * 5 + 3 * KD_LDS_NODES;      // doubles per wave: the queue (lane-parallel query: kd_anc) | nodes as int4 (kd_lane) | split values

// Convex collision sets: the records of ONE facet's vertex neighbourhood (the facet itself in lane 0, PartDev::col_nbr) staged
// in this wave's LDS -- the "LDS-staged triangle tile" of the ray (prl_ray.hpp).  A sub-shot's ray ends on the facet the
// previous one hit or on one that shares a vertex with it, so the five rays of a step test against LDS (one ds_read of 96
// bytes per lane) instead of walking global memory twice per ray (neighbour list, then records); the tile of a newly entered
// facet is fetched by LDS-DMA (global_load_lds_dwordx4: no vector registers) under the hook-point search of the same
// sub-shot.  Chunk c of lane l's record lives at rec[c][l]: what one DMA instruction writes (wave-uniform base + lane * 16).
constexpr int TILE_LANES = 32;      // = NBR_WIDTH of paintrl_amd/device_tables.py; sets with wider lists take the global path
struct FacetTile {
    f64x2 rec[6][TILE_LANES];       // v0 e1 e2 | edge margin | |e1 x e2|^2 | orient   (PartDev::col_rec)
    int id[TILE_LANES];             // facet of lane l, -1: none
    int rank[TILE_LANES];           // its reference index (PartDev::col_rank)
    int facet;                      // whose neighbourhood is resident, -1: none
    int pad_[3];
};

// Records gathered by a few lanes -- the <= 12 triangles around a vertex (192 bytes each), the <= 32 facets around a facet (96
// bytes) -- cost one vector-memory instruction per 16-byte chunk of the record when every candidate lane fetches its own
// (thirteen for a hook point), and the texture addresser takes ~19 cycles per instruction whatever its width: at sixteen
// waves a CU it was busy 61 % of the step (TA_TA_BUSY, profiles/r03_sq_counters.txt).  record_gather spreads the chunks of
// all candidates over the 64 lanes instead (chunk f = record f / C, piece f % C: two or three instructions), parks them in
// this wave's LDS in that order, and the candidate lanes read their records back from there -- as does the winner's tail,
// which no longer is a dependent round trip of its own.
constexpr int GATHER_CHUNKS = 192;  // 16 triangle records of 12 chunks, or 32 facet records of 6

struct WaveLds {
    int *cand;          // [64]
    double *cen;        // [PAINT_PER_ACTION * 3] (+ 1 pad)
    int *cnt;           // [128], only with the atan2-sector observation
    double *kd_heap;    // [KD_ROW]: the queue [KD_HEAP][5], then the staged tree; only in the kernels for parts with the stale kd-tree
    uint64_t *lastrow;  // [2][64 * KW]: the last-shot mask and its successor while the ball painter runs (step_kernel; nullptr:
                        // both stay in registers)
    FacetTile *tile;    // the ray's facet tile, or nullptr (kernels that do not stage one: the ray then reads global memory)
    f64x2 *gather;      // [GATHER_CHUNKS]: where a few lanes' records are fetched by ALL lanes (record_gather below), or nullptr
    int kd_staged;      // kd_heap has KD_ROW doubles: small trees are walked from their copy behind the queue (kd_stage); 0: KD_HEAP * 5
    // the part's two cell-start tables (PartDev::vg_start, sg_start) copied into the workgroup's LDS where the kernel stages
    // them (step_kernel: batches of one part whose grids have at most GRID_LDS entries), or nullptr: read from global memory
    const int *vg_lds = nullptr, *sg_lds = nullptr;
};
constexpr int GRID_LDS = 512;
template <bool GENSEC, bool KD = false, int LASTROW_KW = 0, int WAVES = MAX_WAVES_PER_WG, bool TILE = false, bool GATHER = false>
__device__ __forceinline__ WaveLds wave_lds() {
    __shared__ int s_cand[WAVES][64];
    __shared__ double s_centres[WAVES][PAINT_PER_ACTION * 3 + 1];
    __shared__ int s_cnt[GENSEC ? WAVES : 1][128];
    __shared__ double s_kd[KD ? WAVES : 1][KD ? KD_ROW : 1];
    __shared__ uint64_t s_last[LASTROW_KW ? WAVES : 1][LASTROW_KW ? 2 * 64 * LASTROW_KW : 1];
    __shared__ FacetTile s_tile[TILE ? WAVES : 1];
    __shared__ f64x2 s_gather[GATHER ? WAVES : 1][GATHER ? GATHER_CHUNKS : 1];
    const int w = rfl((int)(threadIdx.x >> 6));
    return WaveLds{s_cand[w], s_centres[w], s_cnt[GENSEC ? w : 0], s_kd[KD ? w : 0], LASTROW_KW ? s_last[w] : nullptr,
                   TILE ? &s_tile[w] : nullptr, GATHER ? s_gather[w] : nullptr, KD ? 1 : 0};
}

__device__ __forceinline__ double bcast_d(double v, int src) {
    src = rfl(src);
    int lo = __double2loint(v), hi = __double2hiint(v);
    lo = __builtin_amdgcn_readlane(lo, src);
    hi = __builtin_amdgcn_readlane(hi, src);
    return __hiloint2double(hi, lo);
}

// reg[lane `slot`] = val for wave-uniform val and slot (v_writelane_b32: the lane select goes through M0 -- with the value in a
// scalar register too the instruction would read two of them, one more than the constant bus of this family allows)
__device__ __forceinline__ int writelane_i(int val, int slot, int reg) {
    asm volatile("s_mov_b32 m0, %2\n\tv_writelane_b32 %0, %1, m0" : "+v"(reg) : "s"(val), "s"(slot) : "m0");
    return reg;
}

// a wave-uniform 64-bit value that arrived in vector registers (a load every lane made from the same address)
__device__ __forceinline__ uint64_t uni_u64(uint64_t v) {
    return ((uint64_t)(uint32_t)rfl((int)(uint32_t)(v >> 32)) << 32) | (uint32_t)rfl((int)(uint32_t)v);
}

__device__ __forceinline__ uint64_t bcast_u64(uint64_t v, int src) {
    src = rfl(src);
    int lo = (int)(uint32_t)v, hi = (int)(uint32_t)(v >> 32);
    lo = __builtin_amdgcn_readlane(lo, src);
    hi = __builtin_amdgcn_readlane(hi, src);
    return ((uint64_t)(uint32_t)hi << 32) | (uint32_t)lo;
}

// Wave-wide min/max by DPP row shifts + row broadcasts (VALU speed) instead of ds_bpermute chains.
// After the six steps lane 63 holds the reduction of all 64 lanes; it is broadcast with readlane.
// dpp_ctrl: row_shr:n = 0x110+n, row_bcast:15 = 0x142, row_bcast:31 = 0x143.  Lanes with no source
// (bound_ctrl off) keep `old`, which is the lane's own value -- harmless for idempotent min/max.
template <int CTRL, int ROW_MASK>
__device__ __forceinline__ int dpp_i(int v) {
    return __builtin_amdgcn_update_dpp(v, v, CTRL, ROW_MASK, 0xf, false);
}

template <int CTRL, int ROW_MASK>
__device__ __forceinline__ double dpp_d(double v) {
    const int lo = dpp_i<CTRL, ROW_MASK>(__double2loint(v)), hi = dpp_i<CTRL, ROW_MASK>(__double2hiint(v));
    return __hiloint2double(hi, lo);
}

#define WAVE_REDUCE_DPP(T, v, OP, DPPF)               \
    do {                                              \
        T x_;                                         \
        x_ = DPPF<0x111, 0xf>(v); v = OP(x_, v);      \
        x_ = DPPF<0x112, 0xf>(v); v = OP(x_, v);      \
        x_ = DPPF<0x114, 0xf>(v); v = OP(x_, v);      \
        x_ = DPPF<0x118, 0xf>(v); v = OP(x_, v);      \
        x_ = DPPF<0x142, 0xa>(v); v = OP(x_, v);      \
        x_ = DPPF<0x143, 0xc>(v); v = OP(x_, v);      \
    } while (0)

#define OP_MIN(x, y) ((x) < (y) ? (x) : (y))
#define OP_MAX(x, y) ((x) > (y) ? (x) : (y))

__device__ __forceinline__ double wave_min_d(double v) {
    WAVE_REDUCE_DPP(double, v, OP_MIN, dpp_d);
    return bcast_d(v, 63);
}

// The same for values known to be >= +0.0 or NaN (squared distances, ray parameters): such doubles order like their
// bit patterns, and a 32-bit unsigned minimum takes its DPP operand directly (one instruction per step instead of
// two moves, a compare and two selects per half).  High words first, then the low words of the lanes that hold the
// smallest high word.  A NaN orders above +inf here instead of poisoning nothing: callers compare lanes with `==`.
template <int CTRL, int ROW_MASK>
__device__ __forceinline__ uint32_t dpp_umin(uint32_t v) {
    return min(v, (uint32_t)__builtin_amdgcn_update_dpp(-1, (int)v, CTRL, ROW_MASK, 0xf, false));
}

__device__ __forceinline__ uint32_t wave_min_u32(uint32_t v) {
    v = dpp_umin<0x111, 0xf>(v);
    v = dpp_umin<0x112, 0xf>(v);
    v = dpp_umin<0x114, 0xf>(v);
    v = dpp_umin<0x118, 0xf>(v);
    v = dpp_umin<0x142, 0xa>(v);
    v = dpp_umin<0x143, 0xc>(v);
    return (uint32_t)__builtin_amdgcn_readlane((int)v, 63);
}

__device__ __forceinline__ double wave_min_nonneg_d(double v) {
    const uint32_t hi = (uint32_t)__double2hiint(v), lo = (uint32_t)__double2loint(v);
    const uint32_t mh = wave_min_u32(hi);
    const uint32_t ml = wave_min_u32(hi == mh ? lo : 0xffffffffu);
    return __hiloint2double((int)mh, (int)ml);
}

__device__ __forceinline__ double wave_max_d(double v) {
    WAVE_REDUCE_DPP(double, v, OP_MAX, dpp_d);
    return bcast_d(v, 63);
}

__device__ __forceinline__ int wave_min_i(int v) {
    WAVE_REDUCE_DPP(int, v, OP_MIN, dpp_i);
    return __builtin_amdgcn_readlane(v, 63);
}

// Wave-wide sum the same way; a lane without a source contributes 0 (old = 0).  Used on packed
// 16-bit counters too: partial sums never carry across fields as long as the totals fit.
template <int CTRL, int ROW_MASK>
__device__ __forceinline__ uint32_t dpp_add(uint32_t v) {
    return v + (uint32_t)__builtin_amdgcn_update_dpp(0, (int)v, CTRL, ROW_MASK, 0xf, false);
}

__device__ __forceinline__ uint32_t wave_sum_u32(uint32_t v) {       // (the adds take their DPP operand directly)
    v = dpp_add<0x111, 0xf>(v);
    v = dpp_add<0x112, 0xf>(v);
    v = dpp_add<0x114, 0xf>(v);
    v = dpp_add<0x118, 0xf>(v);
    v = dpp_add<0x142, 0xa>(v);
    v = dpp_add<0x143, 0xc>(v);
    return (uint32_t)__builtin_amdgcn_readlane((int)v, 63);
}

// Every caller packs counters that never carry across bit 32 (4 x 16-bit or 2 x 32-bit fields whose totals fit):
// the two halves are summed on their own, without the carry chain of a 64-bit add.
__device__ __forceinline__ uint64_t wave_sum_u64(uint64_t v) {
    return ((uint64_t)wave_sum_u32((uint32_t)(v >> 32)) << 32) | wave_sum_u32((uint32_t)v);
}

// Wave-wide sum of doubles (same shifts; a lane without a source adds +0.0).  The order of the additions is this
// tree's, not any reference order: only used where the result is compared with a tolerance (HSI deposits).
template <int CTRL, int ROW_MASK>
__device__ __forceinline__ double dpp0_d(double v) {
    const int lo = __builtin_amdgcn_update_dpp(0, __double2loint(v), CTRL, ROW_MASK, 0xf, false);
    const int hi = __builtin_amdgcn_update_dpp(0, __double2hiint(v), CTRL, ROW_MASK, 0xf, false);
    return __hiloint2double(hi, lo);
}

__device__ __forceinline__ double wave_sum_d(double v) {
    v += dpp0_d<0x111, 0xf>(v);
    v += dpp0_d<0x112, 0xf>(v);
    v += dpp0_d<0x114, 0xf>(v);
    v += dpp0_d<0x118, 0xf>(v);
    v += dpp0_d<0x142, 0xa>(v);
    v += dpp0_d<0x143, 0xc>(v);
    return bcast_d(v, 63);
}

// A wave-uniform double computed by the vector ALU sits in a VGPR pair; moving it to scalar
// registers frees vector registers for the per-lane work (masks, Moller-Trumbore temporaries).
__device__ __forceinline__ double uni_d(double v) {
    return __hiloint2double(rfl(__double2hiint(v)), rfl(__double2loint(v)));
}

__device__ __forceinline__ double sel3(double x, double y, double z, int axis) {
    return axis == 0 ? x : (axis == 1 ? y : z);
}

// ---------------------------------------------------------------- reference arithmetic
// numpy.dot on 3-vectors = OpenBLAS ddot = fused chain (oracle/paint_oracle.c dot3_np)
__host__ __device__ __forceinline__ double dot3_np(double a0, double a1, double a2, double b0, double b1, double b2) {
    return __builtin_fma(a2, b2, __builtin_fma(a1, b1, a0 * b0));
}

// this project's multiplyTransforms rotation (paintrl_amd/geometry.py quat_rotate)
__host__ __device__ __forceinline__ void quat_rotate(const double q[4], double v0, double v1, double v2, double o[3]) {
    double t0 = 2.0 * (q[1] * v2 - q[2] * v1);
    double t1 = 2.0 * (q[2] * v0 - q[0] * v2);
    double t2 = 2.0 * (q[0] * v1 - q[1] * v0);
    o[0] = (v0 + q[3] * t0) + (q[1] * t2 - q[2] * t1);
    o[1] = (v1 + q[3] * t1) + (q[2] * t0 - q[0] * t2);
    o[2] = (v2 + q[3] * t2) + (q[0] * t1 - q[1] * t0);
}

__host__ __device__ __forceinline__ void transform_point(const double pos[3], const double q[4], double v0, double v1,
                                                double v2, double o[3]) {
    double r[3];
    quat_rotate(q, v0, v1, v2, r);
    o[0] = pos[0] + r[0];
    o[1] = pos[1] + r[1];
    o[2] = pos[2] + r[2];
}

// rob:93-100 get_pose_orn + bpw:32-37 normalize
__host__ __device__ __forceinline__ void pose_orn_quat(const double orn[3], double q[4]) {
    double x = 0.0 * orn[2] - 1.0 * orn[1];
    double y = 1.0 * orn[0] - 0.0 * orn[2];
    double z = 0.0 * orn[1] - 0.0 * orn[0];
    double w = 1.0 + __builtin_fma(1.0, orn[2], __builtin_fma(0.0, orn[1], 0.0 * orn[0]));
    double mag2 = (((0.0 + x * x) + y * y) + z * z) + w * w;
    if (fabs(mag2 - 1.0) > 0.00001) {
        double mag = sqrt(mag2);
        x /= mag;
        y /= mag;
        z /= mag;
        w /= mag;
    }
    q[0] = x;
    q[1] = y;
    q[2] = z;
    q[3] = w;
}

// rob:266-271 _get_tcp_orn_norm
__device__ __forceinline__ void tcp_orn_norm(const double pose[3], const double quat[4], double n[3]) {
    double along[3];
    transform_point(pose, quat, 0.0, 0.0, 1.0, along);
    double v0 = along[0] - pose[0], v1 = along[1] - pose[1], v2 = along[2] - pose[2];
    double norm = sqrt(dot3_np(v0, v1, v2, v0, v1, v2));
    // the three quotients in three lanes at once (wave-uniform inputs): one division sequence instead of three
    const int lane = threadIdx.x & 63;
    const double q = (lane == 0 ? v0 : (lane == 1 ? v1 : v2)) / norm;
    n[0] = bcast_d(q, 0);
    n[1] = bcast_d(q, 1);
    n[2] = bcast_d(q, 2);
}

__host__ __device__ __forceinline__ int cell_coord(double x, double origin, double inv, int n) {
    double f = floor((x - origin) * inv);
    f = f < -2.0 ? -2.0 : f;                       // NaN stays NaN -> comparison below sends it out of range
    f = f > (double)(n + 1) ? (double)(n + 1) : f;
    return (f == f) ? (int)f : -2;
}

}  // namespace
