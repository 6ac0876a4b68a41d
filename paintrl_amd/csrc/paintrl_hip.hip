// paintrl_hip.hip -- MI355X (gfx950) batched paint-coverage simulator: kernels + C ABI.
//
// One wavefront (64 lanes) advances one environment by one PaintGymEnv.step()
// (PaintRLEnv/robot_gym_env.py:349-368): five dependent sub-shots (tool move ->
// ray onto the collision triangles -> nearest vertex -> closest incident triangle
// -> hook pose -> ball paint) and then the observation, all in ONE kernel.
//
//  * the env's coverage state (painted mask, last-shot mask, this-shot mask,
//    union-of-valid mask) lives in registers for the whole step: 64-bit word w
//    of a mask is owned by lane (w & 63), slot (w >> 6); HBM traffic per env-step
//    is one coalesced read and one coalesced write of the two persistent masks
//    plus a 128-byte scalar record;
//  * static part tables are shared by all envs and stay L2-resident; samples and
//    vertices are sorted by uniform-grid cell so a sub-shot touches 3 short
//    contiguous ranges (coalesced 512-B loads, one sample per lane, hit mask by
//    ballot);
//  * collision triangles are culled with a 16-byte box per lane-triangle before
//    the float64 Moller-Trumbore test; closest hit by wave min-reduction;
//  * section / grid observations are popcounts over mask words; only words whose
//    bounding box straddles the tool position are classified per sample.
//
// All arithmetic is float64 in the reference's operation order (see
// oracle/paint_oracle.c for the scalar statement; numpy.dot -> explicit fma chain,
// everything else unfused: this file must be compiled with -ffp-contract=off).
// No MFMA: this is gather / scan / bit work.
#include <hip/hip_runtime.h>

#include <cmath>
#include <cstdarg>
#include <cstdio>
#include <cstring>
#include <new>
#include <vector>

#include "paintrl.h"

namespace {

constexpr int KW_MAX = 4;                       // mask slots per lane: up to 64*64*4 = 16384 samples
constexpr double PAINT_RADIUS = 0.051;          // bpw:42
constexpr double STEP_SIZE = 0.051;             // bpw:43
constexpr double HOOK_DISTANCE = 0.1;           // bpw:443
constexpr int GRID_GRANULARITY = 100;           // bpw:447
constexpr int PAINT_PER_ACTION = 5;             // rob:165
constexpr int NOT_ON_PART_TERMINATE = 1000;     // rob:167
constexpr double RAY_EPS_DET = 1e-12;
constexpr double RAY_EPS_BARY = 1e-9;
constexpr double PI = 3.141592653589793;

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
typedef float f32x4 __attribute__((ext_vector_type(4)));     // native vectors: loadable through GAS pointers
typedef double f64x2 __attribute__((ext_vector_type(2)));
typedef double f64x4 __attribute__((ext_vector_type(4)));

struct PartDev {
    int n_samples, n_samples_pad, n_words;
    gdouble_p samp[3];
    gdouble_p samp_a1, samp_a2;   // = samp[a1], samp[a2]: a dynamic index into samp[] would be a memory load of the pointer
    gdouble_p word_bbox;
    gu64_p word_valid;
    gint_p samp_rank;             // canonical (reference-order) index of each device sample, pads = INT_MAX
    gu8_p samp_ub;                // in-word index one past the last sample with the same a1 coordinate (derived in part_fill)
    double sg_o1, sg_o2, sg_inv;
    int sg_nx, sg_ny;
    gint_p sg_start;
    int n_obs_cells;
    gu64_p cell_mask;
    gint_p cell_count;
    int n_vertices;
    gdouble_p vert[3];
    gint_p vert_rank;
    int adj_width;
    gint_p vadj;
    double vg_o1, vg_o2, vg_inv, vg_accept;
    int vg_nx, vg_ny;
    gint_p vg_start;
    int n_triangles;
    gdouble_p tri_rec;
    int n_col, n_col_pad;
    gdouble_p col[9];
    gfloat_p col_bbox;
    gint_p col_rank;
    int col_convex, nbr_width;
    gint_p col_nbr, col_orient;
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
};

// The part descriptor and the batch configuration are read-only for every kernel: typed as constant
// address space so that their fields are fetched with scalar loads (s_load through the K$) instead
// of wave-uniform vector loads the compiler has to assume the kernel's own stores may clobber.
typedef const PartDev CAS &PartRef;
typedef const PrlConfig CAS &CfgRef;

struct StepArgs {
    const PartDev *parts;
    const PrlConfig *cfg;
    const int *env_part;          // device, or nullptr
    int n_envs, mask_stride;
    uint64_t *painted, *last;
    double *state;
    const void *actions;
    double *obs, *reward, *info, *final_obs;
    uint8_t *done;
    const int *start_idx;
    const uint8_t *reset_mask;
};

#ifdef PRL_WAVE_TIMES      // per-wave trip counters of the data-dependent loops (diagnostic build only)
__device__ uint32_t g_wcnt[1 << 16][8];
#define WCNT(slot, v)                                                                        \
    do {                                                                                     \
        if ((threadIdx.x & 63) == 0) g_wcnt[(blockIdx.x * 4 + (threadIdx.x >> 6)) & 0xffff][slot] += (v); \
    } while (0)
#else
#define WCNT(slot, v)
#endif

// ---------------------------------------------------------------- diagnostic build only (-DPRL_PHASE_TIMING)
// Per-phase s_memtime deltas summed over all waves into a buffer nothing else reads
// (cdna_hip_programming.md "In-kernel stamps").  The product build contains no stamp.
#ifdef PRL_PHASE_TIMING
enum { PH_LOAD = 0, PH_RAY, PH_VERTEX, PH_BARY, PH_MATH, PH_BALL, PH_APPLY, PH_OBS, PH_STORE, PH_COUNT };
__device__ unsigned long long g_phase_cycles[16];
struct Prof {
    unsigned long long acc[PH_COUNT];
    unsigned long long prev;
};
#define PROF_ARG , Prof &prof
#define PROF_PASS , prof
#define STAMP(ph)                                                         \
    do {                                                                  \
        __builtin_amdgcn_sched_barrier(0);                                \
        __builtin_amdgcn_s_waitcnt(0);                                    \
        const unsigned long long now_ = __builtin_amdgcn_s_memtime();     \
        __builtin_amdgcn_s_waitcnt(0xC07F);                               \
        prof.acc[ph] += now_ - prof.prev;                                 \
        prof.prev = now_;                                                 \
        __builtin_amdgcn_sched_barrier(0);                                \
    } while (0)
#else
#define PROF_ARG
#define PROF_PASS
#define STAMP(ph) \
    do {          \
    } while (0)
#endif

// ---------------------------------------------------------------- wave helpers
__device__ __forceinline__ int rfl(int v) { return __builtin_amdgcn_readfirstlane(v); }

__device__ __forceinline__ double bcast_d(double v, int src) {
    src = rfl(src);
    int lo = __double2loint(v), hi = __double2hiint(v);
    lo = __builtin_amdgcn_readlane(lo, src);
    hi = __builtin_amdgcn_readlane(hi, src);
    return __hiloint2double(hi, lo);
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
__device__ __forceinline__ uint64_t dpp0_u64(uint64_t v) {
    const uint32_t lo = (uint32_t)__builtin_amdgcn_update_dpp(0, (int)(uint32_t)v, CTRL, ROW_MASK, 0xf, false);
    const uint32_t hi = (uint32_t)__builtin_amdgcn_update_dpp(0, (int)(uint32_t)(v >> 32), CTRL, ROW_MASK, 0xf, false);
    return ((uint64_t)hi << 32) | lo;
}

__device__ __forceinline__ uint64_t wave_sum_u64(uint64_t v) {
    v += dpp0_u64<0x111, 0xf>(v);
    v += dpp0_u64<0x112, 0xf>(v);
    v += dpp0_u64<0x114, 0xf>(v);
    v += dpp0_u64<0x118, 0xf>(v);
    v += dpp0_u64<0x142, 0xa>(v);
    v += dpp0_u64<0x143, 0xc>(v);
    return bcast_u64(v, 63);
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
__device__ __forceinline__ double dot3_np(double a0, double a1, double a2, double b0, double b1, double b2) {
    return __builtin_fma(a2, b2, __builtin_fma(a1, b1, a0 * b0));
}

// this project's multiplyTransforms rotation (paintrl_amd/geometry.py quat_rotate)
__device__ __forceinline__ void quat_rotate(const double q[4], double v0, double v1, double v2, double o[3]) {
    double t0 = 2.0 * (q[1] * v2 - q[2] * v1);
    double t1 = 2.0 * (q[2] * v0 - q[0] * v2);
    double t2 = 2.0 * (q[0] * v1 - q[1] * v0);
    o[0] = (v0 + q[3] * t0) + (q[1] * t2 - q[2] * t1);
    o[1] = (v1 + q[3] * t1) + (q[2] * t0 - q[0] * t2);
    o[2] = (v2 + q[3] * t2) + (q[0] * t1 - q[1] * t0);
}

__device__ __forceinline__ void transform_point(const double pos[3], const double q[4], double v0, double v1,
                                                double v2, double o[3]) {
    double r[3];
    quat_rotate(q, v0, v1, v2, r);
    o[0] = pos[0] + r[0];
    o[1] = pos[1] + r[1];
    o[2] = pos[2] + r[2];
}

// rob:93-100 get_pose_orn + bpw:32-37 normalize
__device__ __forceinline__ void pose_orn_quat(const double orn[3], double q[4]) {
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
    n[0] = v0 / norm;
    n[1] = v1 / norm;
    n[2] = v2 / norm;
}

__device__ __forceinline__ int cell_coord(double x, double origin, double inv, int n) {
    double f = floor((x - origin) * inv);
    f = f < -2.0 ? -2.0 : f;                       // NaN stays NaN -> comparison below sends it out of range
    f = f > (double)(n + 1) ? (double)(n + 1) : f;
    return (f == f) ? (int)f : -2;
}

// ---------------------------------------------------------------- ray: closest two-sided hit (rayTestBatch)
#define FACET_EDGE_MARGIN 1.0e-6     // metres from every edge of the entered facet (single-facet fast path)
#define FACET_MIN_COS2 0.01          // squared cosine between segment and facet normal: no grazing entries
// Cull before the float64 Moller-Trumbore test:
//   * 3-D float boxes (rounded outward) per triangle and per 64-triangle chunk; lane c tests chunk c,
//     only surviving chunks are visited (one triangle per lane, boxes + 9 doubles in one round trip);
//   * two stages: first only the near part of the segment, t <= 0.125 (the tool hovers 0.1 above
//     the part, so this is where the hit almost always is, and the short box excludes back and side
//     facets); the whole segment only if nothing was hit.  The closest hit of the whole segment is
//     the closest hit of the near part whenever the latter exists, so the result is unchanged.
// Equal t resolves to the lowest reference-order index (col_rank), as in paintrl_amd/geometry.py.
struct SegBox {
    float lo[3], hi[3];      // axis1, axis2, axis0
};

__device__ __forceinline__ SegBox seg_box(const double o3[3], const double d3[3], double tmax) {
    SegBox b;
#pragma unroll
    for (int k = 0; k < 3; ++k) {
        const double e = o3[k] + tmax * d3[k];
        b.lo[k] = nextafterf((float)fmin(o3[k], e), -INFINITY);
        b.hi[k] = nextafterf((float)fmax(o3[k], e), INFINITY);
    }
    return b;
}

__device__ __forceinline__ bool box_overlap(const SegBox &s, const f32x4 a, const f32x4 b) {
    return (s.lo[0] <= a.y) && (s.hi[0] >= a.x) && (s.lo[1] <= a.w) && (s.hi[1] >= a.z) && (s.lo[2] <= b.y) &&
           (s.hi[2] >= b.x);
}

// The triangles whose own box passes are first compacted (their ids go to a per-wave LDS list, slot =
// running count + number of passing lanes below), then the float64 test runs ONCE over the list
// with one candidate per lane, instead of once per visited chunk with a handful of active lanes.
// One float64 Moller-Trumbore test per lane (triangle i, or none if i < 0); keeps the lane's best
// (t, reference rank) and remembers which triangle and which determinant produced it.
__device__ __forceinline__ void mt_one(PartRef P, int i, const double o[3], double d0, double d1, double d2,
                                       double tmax, double &best_t, int &best_r, int &best_i, double &best_det) {
    if (i >= 0) {
        const double v00 = P.col[0][i], v01 = P.col[1][i], v02 = P.col[2][i];
        const double e10 = P.col[3][i], e11 = P.col[4][i], e12 = P.col[5][i];
        const double e20 = P.col[6][i], e21 = P.col[7][i], e22 = P.col[8][i];
        const int rk = P.col_rank[i];
        const double p0 = d1 * e22 - d2 * e21;
        const double p1 = d2 * e20 - d0 * e22;
        const double p2 = d0 * e21 - d1 * e20;
        const double det = (e10 * p0 + e11 * p1) + e12 * p2;
        if (fabs(det) >= RAY_EPS_DET) {
            const double inv = 1.0 / det;
            const double s0 = o[0] - v00, s1 = o[1] - v01, s2 = o[2] - v02;
            const double u = ((s0 * p0 + s1 * p1) + s2 * p2) * inv;
            const double q0 = s1 * e12 - s2 * e11;
            const double q1 = s2 * e10 - s0 * e12;
            const double q2 = s0 * e11 - s1 * e10;
            const double v = ((d0 * q0 + d1 * q1) + d2 * q2) * inv;
            const double t = ((e20 * q0 + e21 * q1) + e22 * q2) * inv;
            if (u >= -RAY_EPS_BARY && v >= -RAY_EPS_BARY && (u + v) <= 1.0 + RAY_EPS_BARY && t >= 0.0 && t <= tmax &&
                (t < best_t || (t == best_t && rk < best_r))) {
                best_t = t;
                best_r = rk;
                best_i = i;
                best_det = det;
            }
        }
    }
}

// The same test on the facet record of a convex set (one 96-byte gather per lane instead of ten
// strided loads); `interior` reports a hit that meets the single-facet criterion of ray_closest_wave.
__device__ __forceinline__ void mt_rec(PartRef P, int i, const double o[3], double d0, double d1, double d2, double dd,
                                       double &best_t, int &best_r, int &best_i, double &best_det, bool &interior) {
    interior = false;
    if (i >= 0) {
        const f64x2 GAS *r = reinterpret_cast<const f64x2 GAS *>(P.col_rec + (size_t)i * 12);
        const f64x2 r0 = r[0], r1 = r[1], r2 = r[2], r3 = r[3], r4 = r[4], r5 = r[5];
        const int rk = P.col_rank[i];
        const double v00 = r0.x, v01 = r0.y, v02 = r1.x, e10 = r1.y, e11 = r2.x, e12 = r2.y;
        const double e20 = r3.x, e21 = r3.y, e22 = r4.x, m = r4.y, nn = r5.x, orient = r5.y;
        const double p0 = d1 * e22 - d2 * e21;
        const double p1 = d2 * e20 - d0 * e22;
        const double p2 = d0 * e21 - d1 * e20;
        const double det = (e10 * p0 + e11 * p1) + e12 * p2;
        if (fabs(det) >= RAY_EPS_DET) {
            const double inv = 1.0 / det;
            const double s0 = o[0] - v00, s1 = o[1] - v01, s2 = o[2] - v02;
            const double u = ((s0 * p0 + s1 * p1) + s2 * p2) * inv;
            const double q0 = s1 * e12 - s2 * e11;
            const double q1 = s2 * e10 - s0 * e12;
            const double q2 = s0 * e11 - s1 * e10;
            const double v = ((d0 * q0 + d1 * q1) + d2 * q2) * inv;
            const double t = ((e20 * q0 + e21 * q1) + e22 * q2) * inv;
            if (u >= -RAY_EPS_BARY && v >= -RAY_EPS_BARY && (u + v) <= 1.0 + RAY_EPS_BARY && t >= 0.0 && t <= 1.0 &&
                (t < best_t || (t == best_t && rk < best_r))) {
                best_t = t;
                best_r = rk;
                best_i = i;
                best_det = det;
                interior = u >= m && v >= m && (u + v) <= 1.0 - m && orient * det > 0 &&
                           det * det >= FACET_MIN_COS2 * dd * nn;
            }
        }
    }
}

// Lane holding the wave's best (t, rank); -1 if no lane has a hit.
__device__ __forceinline__ int ray_winner_lane(double best_t, int best_r, double &tmin) {
    if (__ballot(best_t < INFINITY) == 0) return -1;
    tmin = wave_min_d(best_t);
    const uint64_t tie = __ballot(best_t == tmin);
    if ((tie & (tie - 1)) == 0) return __builtin_ctzll(tie);
    const int rmin = wave_min_i(best_t == tmin ? best_r : 0x7fffffff);        // equal t: lowest reference index
    return __builtin_ctzll(__ballot(best_t == tmin && best_r == rmin));
}

// `hint` (in/out): collision-set position of the facet hit by the previous ray of this env, or -1.
//
// Convex fast path (collision set = boundary of a convex polytope, i.e. hull mode): a segment that
// starts outside enters the polytope at one point, so every facet with a valid hit at or before the
// entry parameter contains that point and therefore shares a vertex with any one of them.  If the
// vertex-neighbourhood of `hint` holds a valid hit whose facet is ENTERED (orient * det > 0), the
// closest hit of the whole set is the best over that facet's own neighbourhood.  Anything else (no
// hit there, an exit hit, a facet without a neighbour list) takes the general search below.
__device__ int ray_closest_wave(PartRef P, const double o[3], const double e[3], int lane, double &t_out,
                                double hit[3], int &hint) {
    __shared__ int s_cand[4][64];
    int *cand = s_cand[threadIdx.x >> 6];
    const double d0 = e[0] - o[0], d1 = e[1] - o[1], d2 = e[2] - o[2];
    double best_t = INFINITY, best_det = 0, tmin = INFINITY;
    int best_r = 0x7fffffff, best_i = -1, win = -1;
#ifdef PRL_FORCE_GENERAL_RAY                         // diagnostic build: never take the convex fast path
    hint = -1;
#endif
    if (P.col_convex && hint >= 0) {
        // (1) The previous facet alone, wave-uniform on scalar-loaded data.  If the segment ENTERS the hull
        // through it at a point at least FACET_EDGE_MARGIN away from its edges, and not at a grazing
        // angle, no other facet can report a hit at or before that point: a second hit there would lie
        // in the other facet's 1e-9 tolerance fringe, i.e. within nanometres of an edge of the entered
        // facet.  The result is then this facet's own Moller-Trumbore value, arithmetic as in mt_one.
        {
            const int h = rfl(hint);
            const double CAS *r = reinterpret_cast<const double CAS *>((uint64_t)P.col_rec) + (size_t)h * 12;
            const double e10 = r[3], e11 = r[4], e12 = r[5], e20 = r[6], e21 = r[7], e22 = r[8];
            const double p0 = d1 * e22 - d2 * e21;
            const double p1 = d2 * e20 - d0 * e22;
            const double p2 = d0 * e21 - d1 * e20;
            const double det = (e10 * p0 + e11 * p1) + e12 * p2;
            bool inside = false;
            double t = 0;
            if (fabs(det) >= RAY_EPS_DET) {
                const double inv = 1.0 / det;
                const double s0 = o[0] - r[0], s1 = o[1] - r[1], s2 = o[2] - r[2];
                const double u = ((s0 * p0 + s1 * p1) + s2 * p2) * inv;
                const double q0 = s1 * e12 - s2 * e11;
                const double q1 = s2 * e10 - s0 * e12;
                const double q2 = s0 * e11 - s1 * e10;
                const double v = ((d0 * q0 + d1 * q1) + d2 * q2) * inv;
                t = ((e20 * q0 + e21 * q1) + e22 * q2) * inv;
                const double m = r[9], dd = (d0 * d0 + d1 * d1) + d2 * d2;
                inside = u >= m && v >= m && (u + v) <= 1.0 - m && t >= 0.0 && t <= 1.0 && r[11] * det > 0 &&
                         det * det >= FACET_MIN_COS2 * dd * r[10];
            }
            if (rfl(inside)) {
                WCNT(7, 1);
                t_out = t;
                hit[0] = o[0] + t * d0;
                hit[1] = o[1] + t * d1;
                hit[2] = o[2] + t * d2;
                return reinterpret_cast<const int CAS *>((uint64_t)P.col_rank)[h];
            }
        }
        // (2) The facets that share a vertex with it, one per lane.  A lane whose facet is entered at an
        // interior point holds the closest hit of the whole set by the same argument (there is at most
        // one such lane): no reduction, no second round.
        WCNT(4, 1);
        const double dd = (d0 * d0 + d1 * d1) + d2 * d2;
        const int i1 = lane < P.nbr_width ? P.col_nbr[hint * P.nbr_width + lane] : -1;
        bool interior;
        mt_rec(P, i1, o, d0, d1, d2, dd, best_t, best_r, best_i, best_det, interior);
        const uint64_t im = __ballot(interior);
        if (im) {
            win = __builtin_ctzll(im);
            tmin = bcast_d(best_t, win);
        } else {
            // (3) otherwise the closest hit there, if it enters the hull, decides after a look at its own
            // neighbourhood
            win = ray_winner_lane(best_t, best_r, tmin);
            if (win >= 0) {
                const int f = __builtin_amdgcn_readlane(best_i, rfl(win));
                const double fdet = bcast_d(best_det, win);
                const int i2 = lane < P.nbr_width ? P.col_nbr[f * P.nbr_width + lane] : -1;
                const bool entering = (double)P.col_orient[f] * fdet > 0;
                if (entering && __ballot(i2 >= 0) != 0) {
                    if (f != hint) {
                        WCNT(4, 16);
                        mt_rec(P, i2, o, d0, d1, d2, dd, best_t, best_r, best_i, best_det, interior);
                        win = ray_winner_lane(best_t, best_r, tmin);
                    }
                } else {
                    win = -1;
                }
            }
        }
        if (win < 0) {
            best_t = INFINITY;
            best_r = 0x7fffffff;
            best_i = -1;
        }
    }
    if (win < 0) {
        WCNT(0, 1);
        const double o3[3] = {sel3(o[0], o[1], o[2], P.a1), sel3(o[0], o[1], o[2], P.a2), sel3(o[0], o[1], o[2], P.a0)};
        const double d3[3] = {sel3(d0, d1, d2, P.a1), sel3(d0, d1, d2, P.a2), sel3(d0, d1, d2, P.a0)};
        const f32x4 GAS *boxes = reinterpret_cast<const f32x4 GAS *>(P.col_bbox);
        const f32x4 GAS *chunk_boxes = reinterpret_cast<const f32x4 GAS *>(P.col_chunk_bbox);
        for (int stage = 0; stage < 2; ++stage) {
#ifdef PRL_PHASE_COUNTERS
            if (lane == 0) atomicAdd(&g_phase_cycles[10 + stage], 1ull);
#endif
            const double tmax = stage == 0 ? 0.125 : 1.0;
            if (stage == 1) WCNT(1, 1);
            const SegBox sb = seg_box(o3, d3, tmax);
            int n_cand = 0;
            for (int cbase = 0; cbase < P.n_col_chunks; cbase += 64) {
                const f32x4 ca = chunk_boxes[2 * (cbase + lane)], cb = chunk_boxes[2 * (cbase + lane) + 1];
                uint64_t cm = __ballot(box_overlap(sb, ca, cb));   // table is padded to 64 with empty boxes
                while (cm) {
                    WCNT(2, 1);
                    const int i = ((cbase + __builtin_ctzll(cm)) << 6) + lane;
                    cm &= cm - 1;
                    const f32x4 ba = boxes[2 * i], bb = boxes[2 * i + 1];
                    const bool pass = box_overlap(sb, ba, bb);
                    const uint64_t pm = __ballot(pass);
                    if (pm == 0) continue;
                    const int np = __popcll(pm);
                    if (n_cand + np > 64) {                        // list full: test what is queued first
                        __builtin_amdgcn_wave_barrier();
                        mt_one(P, lane < n_cand ? cand[lane] : -1, o, d0, d1, d2, tmax, best_t, best_r, best_i, best_det);
                        __builtin_amdgcn_wave_barrier();
                        n_cand = 0;
                    }
                    if (pass)
                        cand[n_cand + __builtin_amdgcn_mbcnt_hi((uint32_t)(pm >> 32),
                                                                __builtin_amdgcn_mbcnt_lo((uint32_t)pm, 0))] = i;
                    n_cand += np;
                }
            }
            if (n_cand) {
                __builtin_amdgcn_wave_barrier();
                mt_one(P, lane < n_cand ? cand[lane] : -1, o, d0, d1, d2, tmax, best_t, best_r, best_i, best_det);
                __builtin_amdgcn_wave_barrier();
            }
            if (__ballot(best_t < INFINITY)) break;
        }
        win = ray_winner_lane(best_t, best_r, tmin);
    }
    if (win < 0) {
        t_out = INFINITY;
        hint = -1;
        return -1;
    }
    hint = __builtin_amdgcn_readlane(best_i, rfl(win));
    t_out = tmin;
    hit[0] = o[0] + tmin * d0;
    hit[1] = o[1] + tmin * d1;
    hit[2] = o[2] + tmin * d2;
    return __builtin_amdgcn_readlane(best_r, rfl(win));
}

// ---------------------------------------------------------------- bpw:526 nearest same-side vertex
// Grid rows cy-1..cy+1 of a uniform grid: each row's three cells are one contiguous index range.
// Lanes 0..5 fetch the six range bounds in one load; `rows` returns them wave-uniform.
struct Rows3 {
    int begin[3], count[3];
};

__device__ __forceinline__ Rows3 grid_rows3(gint_p start, int nx, int ny, int icx, int icy, int lane) {
    const int r = lane >> 1, cy = icy - 1 + r;
    const int cx0 = icx - 1 < 0 ? 0 : icx - 1, cx1 = icx + 1 > nx - 1 ? nx - 1 : icx + 1;
    const bool ok = lane < 6 && cy >= 0 && cy < ny && cx0 <= cx1;
    const int v = ok ? start[cy * nx + ((lane & 1) ? cx1 + 1 : cx0)] : 0;
    Rows3 out;
#pragma unroll
    for (int k = 0; k < 3; ++k) {
        const int b = __builtin_amdgcn_readlane(v, 2 * k), e = __builtin_amdgcn_readlane(v, 2 * k + 1);
        out.begin[k] = b;
        out.count[k] = e - b;
    }
    return out;
}

__device__ __forceinline__ void nv_scan(PartRef P, int begin, int end, const double pt[3], int lane,
                                        double &best_d, int &best_rank, int &best_idx) {
    for (int b = begin; b < end; b += 128) {                   // two batches per trip: eight loads in flight
        const int v0 = b + lane, v1 = v0 + 64;
        const bool k0 = v0 < end, k1 = v1 < end;
        double x0 = 0, y0 = 0, z0 = 0, x1 = 0, y1 = 0, z1 = 0;
        int r0 = 0, r1 = 0;
        if (k0) {
            x0 = P.vert[0][v0];
            y0 = P.vert[1][v0];
            z0 = P.vert[2][v0];
            r0 = P.vert_rank[v0];
        }
        if (k1) {
            x1 = P.vert[0][v1];
            y1 = P.vert[1][v1];
            z1 = P.vert[2][v1];
            r1 = P.vert_rank[v1];
        }
        if (k0) {
            const double dx = x0 - pt[0], dy = y0 - pt[1], dz = z0 - pt[2];
            const double dd = (dx * dx + dy * dy) + dz * dz;
            if (dd < best_d || (dd == best_d && r0 < best_rank)) {
                best_d = dd;
                best_rank = r0;
                best_idx = v0;
            }
        }
        if (k1) {
            const double dx = x1 - pt[0], dy = y1 - pt[1], dz = z1 - pt[2];
            const double dd = (dx * dx + dy * dy) + dz * dz;
            if (dd < best_d || (dd == best_d && r1 < best_rank)) {
                best_d = dd;
                best_rank = r1;
                best_idx = v1;
            }
        }
    }
}

// Exact nearest neighbour by expanding rings: the (2k+1)^2 cell block around the query's cell is
// scanned (its rows are contiguous index ranges, flattened into one candidate list); every vertex
// outside the block is at least k cells away in the principal plane, so the result is exact once the
// best distance is within k * 0.99 * cell.  After ring 3 the whole table is scanned.
__device__ int nearest_vertex_wave(PartRef P, const double pt[3], int lane) {
    const double h1 = sel3(pt[0], pt[1], pt[2], P.a1), h2 = sel3(pt[0], pt[1], pt[2], P.a2);
    const int icx = cell_coord(h1, P.vg_o1, P.vg_inv, P.vg_nx), icy = cell_coord(h2, P.vg_o2, P.vg_inv, P.vg_ny);
    double best_d = INFINITY, dmin = INFINITY;
    int best_rank = 0x7fffffff, best_idx = -1;
    bool exact = false;
    for (int ring = 1; ring <= 3 && !exact; ++ring) {
        const int nrows = 2 * ring + 1;
        const int cx0 = icx - ring < 0 ? 0 : icx - ring, cx1 = icx + ring > P.vg_nx - 1 ? P.vg_nx - 1 : icx + ring;
        const int rcy = icy - ring + (lane >> 1);
        const bool okr = lane < 2 * nrows && rcy >= 0 && rcy < P.vg_ny && cx0 <= cx1;
        const int bound = okr ? P.vg_start[rcy * P.vg_nx + ((lane & 1) ? cx1 + 1 : cx0)] : 0;
        // per-row begin and exclusive prefix of counts, wave-uniform (<= 7 rows)
        int rbeg[7], rpre[8];
        rpre[0] = 0;
#pragma unroll
        for (int r = 0; r < 7; ++r) {
            const int b0 = __builtin_amdgcn_readlane(bound, 2 * r), e0 = __builtin_amdgcn_readlane(bound, 2 * r + 1);
            rbeg[r] = b0;
            rpre[r + 1] = rpre[r] + ((r < nrows) ? e0 - b0 : 0);
        }
        const int total = rpre[7];
        best_d = INFINITY;
        best_rank = 0x7fffffff;
        best_idx = -1;
        for (int c0 = 0; c0 < total; c0 += 64) {
            WCNT(3, 1);
            const int c = c0 + lane;
            if (c < total) {
                int v = rbeg[0] + c;
#pragma unroll
                for (int r = 1; r < 7; ++r)
                    if (c >= rpre[r]) v = rbeg[r] + (c - rpre[r]);
                const double dx = P.vert[0][v] - pt[0], dy = P.vert[1][v] - pt[1], dz = P.vert[2][v] - pt[2];
                const double dd = (dx * dx + dy * dy) + dz * dz;
                const int rk = P.vert_rank[v];
                if (dd < best_d || (dd == best_d && rk < best_rank)) {
                    best_d = dd;
                    best_rank = rk;
                    best_idx = v;
                }
            }
        }
        dmin = wave_min_d(best_d);
        const double lim = ring * P.vg_accept;          // ring * 0.99 * cell
        exact = dmin <= lim * lim;
    }
#ifdef PRL_FORCE_FULL_SCANS                          // diagnostic build: exercise the whole-table scans
    exact = false;
#endif
    if (!exact) {
        best_d = INFINITY;
        best_rank = 0x7fffffff;
        best_idx = -1;
        nv_scan(P, 0, P.n_vertices, pt, lane, best_d, best_rank, best_idx);
        dmin = wave_min_d(best_d);
    }
    const uint64_t tie = __ballot(best_d == dmin);
    if (tie == 0) return -1;                                        // NaN query point
    if ((tie & (tie - 1)) == 0) return __builtin_amdgcn_readlane(best_idx, rfl(__builtin_ctzll(tie)));
    const int rmin = wave_min_i(best_d == dmin ? best_rank : 0x7fffffff);
    const uint64_t win = __ballot(best_d == dmin && best_rank == rmin);
    return __builtin_amdgcn_readlane(best_idx, rfl(__builtin_ctzll(win)));
}

// ---------------------------------------------------------------- bpw:565 pixel_kd_tree.query(k=1): nearest sample
// Same exact expanding-ring search as for vertices, over the sample grid; equal distances resolve
// to the lowest reference-order index.  Returns the device position of the sample, or -1.
__device__ int nearest_sample_wave(PartRef P, const double pt[3], int lane) {
    const double h1 = sel3(pt[0], pt[1], pt[2], P.a1), h2 = sel3(pt[0], pt[1], pt[2], P.a2);
    const int icx = cell_coord(h1, P.sg_o1, P.sg_inv, P.sg_nx), icy = cell_coord(h2, P.sg_o2, P.sg_inv, P.sg_ny);
    double best_d = INFINITY, dmin = INFINITY;
    int best_rank = 0x7fffffff, best_idx = -1;
    bool exact = false;
    for (int ring = 1; ring <= 3 && !exact; ++ring) {
        const int nrows = 2 * ring + 1;
        const int cx0 = icx - ring < 0 ? 0 : icx - ring, cx1 = icx + ring > P.sg_nx - 1 ? P.sg_nx - 1 : icx + ring;
        const int rcy = icy - ring + (lane >> 1);
        const bool okr = lane < 2 * nrows && rcy >= 0 && rcy < P.sg_ny && cx0 <= cx1;
        const int bound = okr ? P.sg_start[rcy * P.sg_nx + ((lane & 1) ? cx1 + 1 : cx0)] : 0;
        best_d = INFINITY;
        best_rank = 0x7fffffff;
        best_idx = -1;
#pragma unroll
        for (int r = 0; r < 7; ++r) {
            const int b0 = __builtin_amdgcn_readlane(bound, 2 * r), e0 = __builtin_amdgcn_readlane(bound, 2 * r + 1);
            if (r >= nrows) continue;
            for (int s0 = b0; s0 < e0; s0 += 64) {
                const int sidx = s0 + lane;
                if (sidx < e0) {
                    const double dx = P.samp[0][sidx] - pt[0], dy = P.samp[1][sidx] - pt[1], dz = P.samp[2][sidx] - pt[2];
                    const double dd = (dx * dx + dy * dy) + dz * dz;
                    const int rk = P.samp_rank[sidx];
                    if (dd < best_d || (dd == best_d && rk < best_rank)) {
                        best_d = dd;
                        best_rank = rk;
                        best_idx = sidx;
                    }
                }
            }
        }
        dmin = wave_min_d(best_d);
        const double lim = ring * (0.99 / P.sg_inv);        // ring * 0.99 * sample cell
        exact = dmin <= lim * lim;
    }
#ifdef PRL_FORCE_FULL_SCANS
    exact = false;
#endif
    if (!exact) {                                   // far from every sample: scan the whole table
        best_d = INFINITY;
        best_rank = 0x7fffffff;
        best_idx = -1;
        for (int s0 = 0; s0 < P.n_samples_pad; s0 += 64) {
            const int sidx = s0 + lane;
            const double dx = P.samp[0][sidx] - pt[0], dy = P.samp[1][sidx] - pt[1], dz = P.samp[2][sidx] - pt[2];
            const double dd = (dx * dx + dy * dy) + dz * dz;
            const int rk = P.samp_rank[sidx];
            if (rk != 0x7fffffff && (dd < best_d || (dd == best_d && rk < best_rank))) {
                best_d = dd;
                best_rank = rk;
                best_idx = sidx;
            }
        }
        dmin = wave_min_d(best_d);
    }
    const int rmin = wave_min_i(best_d == dmin ? best_rank : 0x7fffffff);
    const uint64_t win = __ballot(best_d == dmin && best_rank == rmin && best_idx >= 0);
    if (win == 0) return -1;
    return __builtin_amdgcn_readlane(best_idx, rfl(__builtin_ctzll(win)));
}

// ---------------------------------------------------------------- bpw:525-534 _get_hook_point (+508-523)
__device__ bool hook_point_wave(PartRef P, const double pt[3], int lane, double pose[3], double orn[3] PROF_ARG) {
#ifdef PRL_ABLATE_VERTEX                    // diagnostic stand-in: some vertex near the right cell, no scan
    const int vidx = P.vg_start[0] + ((int)(fabs(pt[1] * 977.0 + pt[2] * 1543.0)) % P.n_vertices);
#else
#ifdef PRL_DOUBLE_VERTEX
    {
        const int v2 = nearest_vertex_wave(P, pt, lane);
        asm volatile("" ::"s"(v2));
    }
#endif
    const int vidx = nearest_vertex_wave(P, pt, lane);
#endif
    STAMP(PH_VERTEX);
    if (vidx < 0) return false;
    const int ti = lane < P.adj_width ? P.vadj[vidx * P.adj_width + lane] : -1;   // file order, -1 = pad
    if (__ballot(ti >= 0) == 0) return false;
    bool inside = false, ok = false;
    double m = -INFINITY, n0 = 0, n1 = 0, n2 = 0;
    if (ti >= 0) {
        gdouble_p r = P.tri_rec + (size_t)ti * 16;
        const f64x2 GAS *r2 = reinterpret_cast<const f64x2 GAS *>(r);
        const f64x2 q0 = r2[0], q1 = r2[1], q2 = r2[2], q3 = r2[3], q4 = r2[4], q5 = r2[5], q6 = r2[6], q7 = r2[7];
        // a = q0.x q0.y q1.x | v0 = q1.y q2.x q2.y | v1 = q3.x q3.y q4.x | d00 q4.y d01 q5.x d11 q5.y inv q6.x | n q6.y q7.x q7.y
        const double x0 = pt[0] - q0.x, x1 = pt[1] - q0.y, x2 = pt[2] - q1.x;
        const double d20 = dot3_np(x0, x1, x2, q1.y, q2.x, q2.y);
        const double d21 = dot3_np(x0, x1, x2, q3.x, q3.y, q4.x);
        const double inv = q6.x;
        double v = (q5.y * d20 - q5.x * d21) * inv;
        double w = (q4.y * d21 - q5.x * d20) * inv;
        double u = 1.0 - v - w;
        if (inv == 0) {
            u = -1;
            v = -1;
            w = -1;
        }
        inside = 0 <= u && u <= 1 && 0 <= v && v <= 1 && 0 <= w && w <= 1;
        m = v < u ? v : u;
        m = w < m ? w : m;
        ok = m >= -1.0;
        n0 = q6.y;
        n1 = q7.x;
        n2 = q7.y;
    }
    int j;
    const uint64_t in_mask = __ballot(inside);
    if (in_mask) {
        j = __builtin_ctzll(in_mask);                               // first triangle containing the point
    } else {
        const uint64_t ok_mask = __ballot(ok);
        if (ok_mask == 0) {
            j = 0;                                                   // nothing beat -1: the first candidate stays
        } else {
            const double mx = wave_max_d(ok ? m : -INFINITY);
            j = 63 - __builtin_clzll(__ballot(ok && m == mx));       // last one reaching the maximum
        }
    }
    n0 = bcast_d(n0, j);
    n1 = bcast_d(n1, j);
    n2 = bcast_d(n2, j);
    pose[0] = pt[0] + n0 * HOOK_DISTANCE;
    pose[1] = pt[1] + n1 * HOOK_DISTANCE;
    pose[2] = pt[2] + n2 * HOOK_DISTANCE;
    orn[0] = -n0;
    orn[1] = -n1;
    orn[2] = -n2;
    STAMP(PH_BARY);
    return true;
}

// ---------------------------------------------------------------- bpw:568-570 fast_paint (ball query)
template <int KW>
__device__ __forceinline__ void set_word(uint64_t cur[KW_MAX], int w, uint64_t b, int lane) {
    const int owner = w & 63, slot = w >> 6;
#pragma unroll
    for (int k = 0; k < KW; ++k)
        if (k == slot && lane == owner) cur[k] |= b;
}

template <int KW>
__device__ void ball_query_wave(PartRef P, double radius, const double c[3], int lane,
                                uint64_t cur[KW_MAX]) {
    const double r2 = radius * radius;
    const double c1 = sel3(c[0], c[1], c[2], P.a1), c2 = sel3(c[0], c[1], c[2], P.a2);
    const int icx = cell_coord(c1, P.sg_o1, P.sg_inv, P.sg_nx), icy = cell_coord(c2, P.sg_o2, P.sg_inv, P.sg_ny);
    const Rows3 R = grid_rows3(P.sg_start, P.sg_nx, P.sg_ny, icx, icy, lane);
#pragma unroll
    for (int r = 0; r < 3; ++r) {
        const int begin = R.begin[r], end = begin + R.count[r];
        if (R.count[r] <= 0) continue;
        const int wlast = (end - 1) >> 6;
        for (int w = begin >> 6; w <= wlast; w += 2) {            // two words per trip: six loads in flight
            const int s0 = (w << 6) + lane, s1 = s0 + 64;
            const bool two = w + 1 <= wlast;
            const double x0 = P.samp[0][s0], y0 = P.samp[1][s0], z0 = P.samp[2][s0];
            double x1 = 0, y1 = 0, z1 = 0;
            if (two) {
                x1 = P.samp[0][s1];
                y1 = P.samp[1][s1];
                z1 = P.samp[2][s1];
            }
            {
                const double dx = x0 - c[0], dy = y0 - c[1], dz = z0 - c[2];
                const double dd = (dx * dx + dy * dy) + dz * dz;
                const uint64_t b = __ballot(s0 >= begin && s0 < end && dd <= r2);
                if (b) set_word<KW>(cur, w, b, lane);
            }
            if (two) {
                const double dx = x1 - c[0], dy = y1 - c[1], dz = z1 - c[2];
                const double dd = (dx * dx + dy * dy) + dz * dz;
                const uint64_t b = __ballot(s1 >= begin && s1 < end && dd <= r2);
                if (b) set_word<KW>(cur, w + 1, b, lane);
            }
        }
    }
}

// ---------------------------------------------------------------- the five shots of one step, painted together
// The five shot centres of a step are 0.0102 apart, so their 3x3 neighbourhoods overlap almost
// entirely.  Each candidate sample is loaded once and tested against all five centres; the per-word
// hit ballots b_0..b_4 are wave-uniform, so the reference's shot-by-shot bookkeeping (bpw:572-577:
// count newly painted, paint, valid = affected minus last shot, last = affected) runs on the scalar
// unit for that word and is written back to the lane that owns the word.  Words outside the
// neighbourhood have no hits in any shot: painted is unchanged and their last-shot bits become 0.
// The centres are written to LDS by the shot loop (a register array indexed by the runtime shot
// number would live in scratch) and read back wave-uniformly here.
struct ShotCentres {
    double c[PAINT_PER_ACTION][3];
};

template <int KW>
__device__ bool paint_shots_union(PartRef P, double radius, const double *cen_lds, int lane,
                                  uint64_t painted[KW_MAX],
                                  const uint64_t last[KW_MAX], uint64_t new_last[KW_MAX], int &succeeded,
                                  int &pixel_counter) {
    const double r2 = radius * radius;
    ShotCentres sc;
#pragma unroll
    for (int k = 0; k < PAINT_PER_ACTION; ++k) {
        sc.c[k][0] = cen_lds[3 * k];
        sc.c[k][1] = cen_lds[3 * k + 1];
        sc.c[k][2] = cen_lds[3 * k + 2];
    }
    int cx_lo = 0x7fffffff, cx_hi = -0x7fffffff, cy_lo = 0x7fffffff, cy_hi = -0x7fffffff;
#pragma unroll
    for (int k = 0; k < PAINT_PER_ACTION; ++k) {
        const int icx = cell_coord(sel3(sc.c[k][0], sc.c[k][1], sc.c[k][2], P.a1), P.sg_o1, P.sg_inv, P.sg_nx);
        const int icy = cell_coord(sel3(sc.c[k][0], sc.c[k][1], sc.c[k][2], P.a2), P.sg_o2, P.sg_inv, P.sg_ny);
        cx_lo = icx < cx_lo ? icx : cx_lo;
        cx_hi = icx > cx_hi ? icx : cx_hi;
        cy_lo = icy < cy_lo ? icy : cy_lo;
        cy_hi = icy > cy_hi ? icy : cy_hi;
    }
#ifdef PRL_FORCE_PER_SHOT_PAINT                     // diagnostic build: exercise the general path in the parity tests
    return false;
#endif
    if (cy_hi - cy_lo > 1) return false;            // centres spread over > 2 cell rows: caller paints shot by shot
    // rows cy_lo-1 .. cy_hi+1 (<= 4), columns cx_lo-1 .. cx_hi+1: lanes 0..7 fetch the range bounds
    const int cx0 = cx_lo - 1 < 0 ? 0 : cx_lo - 1, cx1 = cx_hi + 1 > P.sg_nx - 1 ? P.sg_nx - 1 : cx_hi + 1;
    const int rcy = cy_lo - 1 + (lane >> 1);
    const bool ok = lane < 8 && rcy <= cy_hi + 1 && rcy >= 0 && rcy < P.sg_ny && cx0 <= cx1;
    const int bound = ok ? P.sg_start[rcy * P.sg_nx + ((lane & 1) ? cx1 + 1 : cx0)] : 0;
    int rb[4], re[4];
#pragma unroll
    for (int r = 0; r < 4; ++r) {
        rb[r] = __builtin_amdgcn_readlane(bound, 2 * r);
        re[r] = __builtin_amdgcn_readlane(bound, 2 * r + 1);
    }
    int done_w = -1;                                 // a word shared by two rows' ranges is handled once
#pragma unroll
    for (int r = 0; r < 4; ++r) {
        if (re[r] <= rb[r]) continue;
        const int wlast = (re[r] - 1) >> 6;
        for (int w = (rb[r] >> 6) > done_w ? (rb[r] >> 6) : done_w + 1; w <= wlast; ++w) {
            WCNT(5, 1);
            const int s = (w << 6) + lane;
            const double x = P.samp[0][s], y = P.samp[1][s], z = P.samp[2][s];
            const bool in = (s >= rb[0] && s < re[0]) || (s >= rb[1] && s < re[1]) || (s >= rb[2] && s < re[2]) ||
                            (s >= rb[3] && s < re[3]);
            uint64_t b[PAINT_PER_ACTION];
            uint64_t any = 0;
#pragma unroll
            for (int k = 0; k < PAINT_PER_ACTION; ++k) {
                const double dx = x - sc.c[k][0], dy = y - sc.c[k][1], dz = z - sc.c[k][2];
                const double dd = (dx * dx + dy * dy) + dz * dz;
                b[k] = __ballot(in && dd <= r2);
                any |= b[k];
            }
            done_w = w;
            const int owner = w & 63, slot = w >> 6;
            uint64_t pw = 0, lw = 0;
#pragma unroll
            for (int k = 0; k < KW; ++k)
                if (k == slot) {
                    pw = bcast_u64(painted[k], owner);
                    lw = bcast_u64(last[k], owner);
                }
            if (any == 0 && lw == 0) continue;       // nothing to record for this word
            // bpw:572-577 shot by shot (count newly painted, paint, valid = affected minus last shot, last =
            // affected), folded: the newly painted samples of the five shots are the union minus what was
            // painted before, and each shot's valid set only looks one shot back
            succeeded += __popcll(any & ~pw);
            pw |= any;
            uint64_t uw = b[0] & ~lw;
#pragma unroll
            for (int k = 1; k < PAINT_PER_ACTION; ++k) uw |= b[k] & ~b[k - 1];
            lw = b[PAINT_PER_ACTION - 1];
            pixel_counter += __popcll(uw);
#pragma unroll
            for (int k = 0; k < KW; ++k)
                if (k == slot && lane == owner) {
                    painted[k] = pw;
                    new_last[k] = lw;
                }
        }
    }
    return true;
}

// ---------------------------------------------------------------- observation (rge:306-319)
__device__ __forceinline__ int grid_index_2(PartRef P, double val) {
    const double rel = (val - P.r2min) / (P.r2max - P.r2min);
    const double g = rel * GRID_GRANULARITY;
    int gi;
    if (!(g > -2147483648.0 && g < 2147483648.0)) gi = g > 0 ? GRID_GRANULARITY - 1 : 0;
    else gi = (int)g;
    return gi < 0 ? 0 : (gi > GRID_GRANULARITY - 1 ? GRID_GRANULARITY - 1 : gi);
}

__device__ __forceinline__ double clip01(double v) { return v < 0 ? 0.0 : (v > 1 ? 1.0 : v); }

__device__ __forceinline__ int handle_pos(double v) {        // rge:92-98
    if (v == 0) return 0;
    if (v == 1) return 21;
    return (int)(v * 20) + 1;
}

// CPython float_floor_div, the `//` of bpw:1030 (exact floor of the true quotient)
__device__ __forceinline__ double py_floor_div(double vx, double wx) {
    const double mod = fmod(vx, wx);
    double div = (vx - mod) / wx;
    if (mod != 0.0 && ((wx < 0) != (mod < 0))) div -= 1.0;
    if (div != 0.0) {
        double fl = floor(div);
        if (div - fl > 0.5) fl += 1.0;
        return fl;
    }
    return copysign(0.0, vx / wx);
}

// bpw:1026-1031, 1045-1061 with section != 4: every sample is classified by atan2 (not tuned: this
// is the hand-selected OBS_GRAD variant; the default 4-sector rule takes the fast path below).
template <int KW>
__device__ void section_general_wave(PartRef P, int g, double x1, double x2, const uint64_t painted[KW_MAX],
                                     int lane, int *cnt /* LDS: [2][64] for this wave */, double *out) {
    gdouble_p sx = P.samp_a1, sy = P.samp_a2;
    cnt[lane] = 0;
    cnt[64 + lane] = 0;
    const double two_pi = 2 * PI, basis = two_pi / g;
    for (int w = 0; w < P.n_words; ++w) {
        const uint64_t vw = P.word_valid[w];
        uint64_t pw = 0;
#pragma unroll
        for (int k = 0; k < KW; ++k)
            if (k == (w >> 6)) pw = bcast_u64(painted[k], w & 63);
        if (!((vw >> lane) & 1)) continue;
        const int s = (w << 6) + lane;
        const double rx = sx[s] - x1, ry = sy[s] - x2;
        if (rx == 0 && ry == 0) continue;
        double ang = atan2(ry, rx);
        if (ang < 0) ang = two_pi + ang;
        int idx = (int)py_floor_div(ang, basis);
        idx = idx > g - 1 ? g - 1 : (idx < 0 ? 0 : idx);
        atomicAdd(&cnt[idx], 1);
        if (!((pw >> lane) & 1)) atomicAdd(&cnt[64 + idx], 1);
    }
    if (lane < g) {
        const int t = cnt[lane], u = cnt[64 + lane];
        out[lane] = t == 0 ? 0.0 : (double)u / (double)t;
    }
}

// GENSEC selects the atan2-sector variant at compile time so that the default kernel carries none of
// its registers or code.
template <int KW, bool GENSEC>
__device__ void observation_wave(PartRef P, CfgRef C, const double pose[3],
                                 const uint64_t painted[KW_MAX], int lane, double *out) {
    // bpw:965-978 get_normalized_pose
    const double r = C.paint_radius;
    const double x1 = sel3(pose[0], pose[1], pose[2], P.a1), x2 = sel3(pose[0], pose[1], pose[2], P.a2);
    const double in2 = (x2 - P.r2min + r) / (P.r2max - P.r2min + 2 * r);
    const int gi = grid_index_2(P, x2);
    const double lo = P.grid_lo[gi], hi = P.grid_hi[gi];
    double in1;
    if (hi - lo == 0) in1 = 0;
    else in1 = (x1 - lo + r) / (hi - lo + 2 * r);
    const double np0 = clip01(in1), np1 = clip01(in2);
    const int mode = C.obs_mode;
    if (mode == PRL_OBS_SIMPLE) {
        if (lane == 0) {
            out[0] = np0;
            out[1] = np1;
        }
        return;
    }
    if (mode == PRL_OBS_GRID) {                    // bpw:1126-1139: 1 - painted/num per cell
        // 16 cells per pass: four packed accumulators (4 x 16-bit per u64), four DPP sums, then lane j
        // finishes cell j (one division per lane, one coalesced store)
        const int cells = P.n_obs_cells;
        for (int c0 = 0; c0 < cells; c0 += 16) {
            uint64_t acc[4] = {0, 0, 0, 0};
#pragma unroll
            for (int k = 0; k < KW; ++k) {
                const int w = lane + 64 * k;
                if (w < P.n_words) {
#pragma unroll
                    for (int j = 0; j < 16; ++j)
                        if (c0 + j < cells)
                            acc[j >> 2] += (uint64_t)__popcll(painted[k] & P.cell_mask[(size_t)(c0 + j) * P.n_words + w])
                                           << (16 * (j & 3));
                }
            }
#pragma unroll
            for (int g = 0; g < 4; ++g) acc[g] = wave_sum_u64(acc[g]);
            const int cell = c0 + lane;
            if (lane < 16 && cell < cells) {
                uint64_t a4 = acc[0];
#pragma unroll
                for (int g = 1; g < 4; ++g) a4 = (lane >> 2) == g ? acc[g] : a4;
                const int dn = (int)((a4 >> (16 * (lane & 3))) & 0xffff);
                const int num = P.cell_count[cell];
                out[cell] = num == 0 ? 0.0 : 1.0 - (double)dn / (double)num;
            }
        }
        return;
    }
    if constexpr (GENSEC) {                        // section / discrete with atan2 sectors (OBS_GRAD != 4)
        __shared__ int s_cnt[4][128];
        section_general_wave<KW>(P, C.obs_grad, x1, x2, painted, lane, s_cnt[threadIdx.x >> 6], out);
        if (lane == 0) {
            if (mode == PRL_OBS_SECTION) {
                out[C.obs_grad] = np0;
                out[C.obs_grad + 1] = np1;
            } else {
                const int position = (handle_pos(np0) + 1) * 22 + handle_pos(np1);
                out[C.obs_grad] = 1.0 / (double)position;
            }
        }
        return;
    } else {
    // section / discrete, 4-sector rule bpw:1034-1043
    gdouble_p sx = P.samp_a1, sy = P.samp_a2;
    uint64_t tot_l = 0, und_l = 0;                 // 4 x 16-bit counters per lane (total / unpainted per sector)
    uint32_t tot_s = 0, und_s = 0;                 // 4 x 8-bit counters per lane for the straddling words
    // Pass 1, one word per lane and slot: a word whose box lies in one sector is counted whole.  A word
    // that straddles only the vertical line x1 (its row is clear of x2) is resolved by its own lane
    // below; only the rest -- the words of the row that x2 crosses -- is classified sample by sample.
    bool vline[KW_MAX] = {false, false, false, false}, above[KW_MAX] = {false, false, false, false};
    uint64_t valid[KW_MAX] = {0, 0, 0, 0}, smask[KW_MAX] = {0, 0, 0, 0};
#pragma unroll
    for (int k = 0; k < KW; ++k) {
        const int w = lane + 64 * k;
        bool straddle = false;
        if (w < P.n_words) {
            const f64x4 bb = reinterpret_cast<const f64x4 GAS *>(P.word_bbox)[w];
            valid[k] = P.word_valid[w];
            const bool xg = bb.x > x1, xl = bb.y < x1, yg = bb.z > x2, yl = bb.w < x2;
            if ((xg || xl) && (yg || yl)) {
                const int idx = (xg && yg) ? 0 : ((xl && yg) ? 1 : ((xl && yl) ? 2 : 3));
                tot_l += (uint64_t)__popcll(valid[k]) << (16 * idx);
                und_l += (uint64_t)__popcll(valid[k] & ~painted[k]) << (16 * idx);
            } else if (valid[k] != 0) {
                vline[k] = yg || yl;
                above[k] = yg;
                straddle = !vline[k];
            }
        }
        smask[k] = __ballot(straddle);
    }
#ifndef PRL_ABLATE_STRADDLE
    // Pass 2: the samples of a word ascend on axis a1 (device_tables), so { xs < x1 } is a prefix and
    // { xs > x1 } a suffix of the word: a 7-probe lower bound by the owning lane, all rows at once.
    // Above the line the rule reads  > -> 0, < -> 1, == -> 3;  below it  < -> 2, else 3  (bpw:1034-1043).
    {
        bool any = false;
#pragma unroll
        for (int k = 0; k < KW; ++k) any = any || vline[k];
        if (__ballot(any)) {
            int base[KW_MAX], pos[KW_MAX];
#pragma unroll
            for (int k = 0; k < KW; ++k) {
                base[k] = vline[k] ? (lane + 64 * k) << 6 : 0;      // other lanes probe word 0: harmless, in bounds
                pos[k] = 0;
            }
#pragma unroll
            for (int step = 32; step >= 0; step = step > 1 ? step >> 1 : step - 1) {   // 32 .. 1, then the closing probe
                double probe[KW_MAX];
#pragma unroll
                for (int k = 0; k < KW; ++k) probe[k] = sx[base[k] + pos[k] + (step ? step - 1 : 0)];
                __builtin_amdgcn_sched_barrier(0);      // the slots' probes travel together: one round trip per step
#pragma unroll
                for (int k = 0; k < KW; ++k) pos[k] += probe[k] < x1 ? (step ? step : 1) : 0;
            }
#pragma unroll
            for (int k = 0; k < KW; ++k) {
                const int at = base[k] + (pos[k] < 64 ? pos[k] : 63);
                int ub = (pos[k] < 64 && sx[at] == x1) ? (int)P.samp_ub[at] : pos[k];
                if (x1 != x1) ub = 64;                              // NaN: nothing is greater either
                if (vline[k]) {
                    const uint64_t lt = pos[k] >= 64 ? ~0ull : ((1ull << pos[k]) - 1);
                    const uint64_t ng = ub >= 64 ? ~0ull : ((1ull << ub) - 1);
                    const uint64_t v = valid[k], u = valid[k] & ~painted[k];
                    const uint64_t vg = __popcll(v & ~ng), vl = __popcll(v & lt), ve = __popcll(v) - vg - vl;
                    const uint64_t ug = __popcll(u & ~ng), ul = __popcll(u & lt), ue = __popcll(u) - ug - ul;
                    tot_l += above[k] ? (vg | (vl << 16) | (ve << 48)) : ((vl << 32) | ((vg + ve) << 48));
                    und_l += above[k] ? (ug | (ul << 16) | (ue << 48)) : ((ul << 32) | ((ug + ue) << 48));
                }
            }
        }
    }
#endif
#pragma unroll
    for (int k = 0; k < KW; ++k) {
        uint64_t sm = smask[k];
#ifdef PRL_ABLATE_STRADDLE
        sm = 0;
#endif
        while (sm) {                                // wave-uniform loop over the words that straddle the tool
            WCNT(6, 1);
            const int L = __builtin_ctzll(sm);
            sm &= sm - 1;
            const int w2 = L + 64 * k;
            const double xs = sx[(w2 << 6) + lane], ys = sy[(w2 << 6) + lane];
            const uint64_t vs = P.word_valid[w2];
            // one sample per lane, 32-bit work only: the uniform valid / painted words become lane
            // predicates (inverse ballot) and the counters are four 8-bit fields (a lane sees at most
            // 64 straddling words)
            const uint64_t pw = bcast_u64(painted[k], L);
            const bool cnt = __builtin_amdgcn_inverse_ballot_w64(vs) && !(xs == x1 && ys == x2);
            const bool gy = ys > x2, lx = xs < x1;
            const uint32_t sh = (xs > x1 && gy) ? 0u : ((lx && gy) ? 8u : ((lx && ys < x2) ? 16u : 24u));
            const uint32_t one = cnt ? (1u << sh) : 0u;
            tot_s += one;
            und_s += __builtin_amdgcn_inverse_ballot_w64(pw) ? 0u : one;
        }
    }
    // widen the 8-bit straddle counters into the 16-bit fields
#pragma unroll
    for (int q = 0; q < 4; ++q) {
        tot_l += (uint64_t)((tot_s >> (8 * q)) & 0xffu) << (16 * q);
        und_l += (uint64_t)((und_s >> (8 * q)) & 0xffu) << (16 * q);
    }
    tot_l = wave_sum_u64(tot_l);
    und_l = wave_sum_u64(und_l);
    if (lane == 0) {
        for (int q = 0; q < 4; ++q) {
            const uint32_t t = (uint32_t)((tot_l >> (16 * q)) & 0xffff);
            const uint32_t u = (uint32_t)((und_l >> (16 * q)) & 0xffff);
            out[q] = t == 0 ? 0.0 : (double)u / (double)t;
        }
        if (mode == PRL_OBS_SECTION) {
            out[4] = np0;
            out[5] = np1;
        } else {
            const int position = (handle_pos(np0) + 1) * 22 + handle_pos(np1);     // rge:101-103
            out[4] = 1.0 / (double)position;
        }
    }
    }
}

// ---------------------------------------------------------------- start-point RNG
__device__ __forceinline__ uint64_t splitmix64(uint64_t x) {
    x += 0x9E3779B97F4A7C15ull;
    x = (x ^ (x >> 30)) * 0xBF58476D1CE4E5B9ull;
    x = (x ^ (x >> 27)) * 0x94D049BB133111EBull;
    return x ^ (x >> 31);
}

__device__ __forceinline__ int draw_start(uint64_t seed, int env, uint64_t episode, int n_start) {
    const uint64_t h = splitmix64(seed ^ splitmix64(((uint64_t)env << 1) | 1) ^ (episode * 0xD1342543DE82EF95ull));
    return (int)(((h >> 32) * (uint64_t)n_start) >> 32);
}

struct EnvState {                 // PRL_STATE_DOUBLES record
    double pose[3], quat[4];
    double last_angle, total_reward, total_return;
    int terminate, terminate_counter, last_on_part, step_counter;
    uint32_t episode;
    int facet_hint;               // collision triangle the last ray of the previous step hit (-1: none); a cache
    double last_ep_return, last_ep_reward;
    int last_ep_len, last_ep_painted;
};
static_assert(sizeof(EnvState) == PRL_STATE_DOUBLES * 8, "state record layout");

// Store the 128-byte record as one coalesced write: lane l < 16 writes double l.
__device__ __forceinline__ void store_state(double *dst, const EnvState &S, int lane) {
    const double *src = reinterpret_cast<const double *>(&S);
    double v = 0;
#pragma unroll
    for (int k = 0; k < PRL_STATE_DOUBLES; ++k) v = lane == k ? src[k] : v;
    if (lane < PRL_STATE_DOUBLES) dst[lane] = v;
}

__device__ __forceinline__ void reset_state(PartRef P, EnvState &S, int start) {   // rge:370-387, rob:366-372
    S.pose[0] = P.start_pos[3 * start];
    S.pose[1] = P.start_pos[3 * start + 1];
    S.pose[2] = P.start_pos[3 * start + 2];
    S.quat[0] = P.start_quat[4 * start];
    S.quat[1] = P.start_quat[4 * start + 1];
    S.quat[2] = P.start_quat[4 * start + 2];
    S.quat[3] = P.start_quat[4 * start + 3];
    S.last_angle = 0;
    S.total_reward = 0;
    S.total_return = 0;
    S.terminate = 0;
    S.terminate_counter = 0;
    S.last_on_part = 1;
    S.step_counter = 0;
    S.episode += 1;
    S.facet_hint = -1;
}

template <int KW>
__device__ __forceinline__ void load_masks(const StepArgs &a, int env, int n_words, int lane, uint64_t painted[KW_MAX],
                                           uint64_t last[KW_MAX]) {
#pragma unroll
    for (int k = 0; k < KW; ++k) {
        const int w = lane + 64 * k;
        const bool in = w < n_words;
        painted[k] = in ? a.painted[(size_t)env * a.mask_stride + w] : 0;
        last[k] = in ? a.last[(size_t)env * a.mask_stride + w] : 0;
    }
}

template <int KW>
__device__ __forceinline__ void store_masks(const StepArgs &a, int env, int n_words, int lane,
                                            const uint64_t painted[KW_MAX], const uint64_t last[KW_MAX]) {
#pragma unroll
    for (int k = 0; k < KW; ++k) {
        const int w = lane + 64 * k;
        if (w < n_words) {
            a.painted[(size_t)env * a.mask_stride + w] = painted[k];
            a.last[(size_t)env * a.mask_stride + w] = last[k];
        }
    }
}

__host__ __device__ inline int obs_dim_of(int obs_mode, int obs_grad) {          // rge:166-173
    switch (obs_mode) {
    case PRL_OBS_SECTION: return obs_grad + 2;
    case PRL_OBS_GRID: return obs_grad * obs_grad;
    case PRL_OBS_SIMPLE: return 2;
    default: return obs_grad + 1;
    }
}

// ---------------------------------------------------------------- reset kernel (rge:370-387)
template <int KW, bool GENSEC>
__global__ __launch_bounds__(256) void reset_kernel(StepArgs a) {
    const int lane = threadIdx.x & 63;
    const int env = rfl(blockIdx.x * 4 + (threadIdx.x >> 6));
    if (env >= a.n_envs) return;
    if (a.reset_mask && !a.reset_mask[env]) return;
    PartRef P = *(const PartDev CAS *)(a.parts + (a.env_part ? a.env_part[env] : 0));
    CfgRef C = *(const PrlConfig CAS *)a.cfg;
    EnvState S = *reinterpret_cast<const EnvState *>(a.state + (size_t)env * PRL_STATE_DOUBLES);
    int start = a.start_idx ? a.start_idx[env] : draw_start(C.seed, env, S.episode, P.n_start);
    start = start < 0 ? 0 : (start >= P.n_start ? P.n_start - 1 : start);
    reset_state(P, S, start);
    uint64_t painted[KW_MAX] = {0, 0, 0, 0}, last[KW_MAX] = {0, 0, 0, 0};
    store_masks<KW>(a, env, P.n_words, lane, painted, last);
    store_state(a.state + (size_t)env * PRL_STATE_DOUBLES, S, lane);
    if (a.obs) observation_wave<KW, GENSEC>(P, C, S.pose, painted, lane, a.obs + (size_t)env * obs_dim_of(C.obs_mode, C.obs_grad));
}

// ---------------------------------------------------------------- observation of the current state (rge:306-319)
template <int KW, bool GENSEC>
__global__ __launch_bounds__(256) void observe_kernel(StepArgs a) {
    const int lane = threadIdx.x & 63;
    const int env = rfl(blockIdx.x * 4 + (threadIdx.x >> 6));
    if (env >= a.n_envs) return;
    PartRef P = *(const PartDev CAS *)(a.parts + (a.env_part ? a.env_part[env] : 0));
    CfgRef C = *(const PrlConfig CAS *)a.cfg;
    const EnvState S = *reinterpret_cast<const EnvState *>(a.state + (size_t)env * PRL_STATE_DOUBLES);
    uint64_t painted[KW_MAX] = {0, 0, 0, 0}, last[KW_MAX] = {0, 0, 0, 0};
    load_masks<KW>(a, env, P.n_words, lane, painted, last);
    observation_wave<KW, GENSEC>(P, C, S.pose, painted, lane, a.obs + (size_t)env * obs_dim_of(C.obs_mode, C.obs_grad));
}

// ---------------------------------------------------------------- step kernel (rge:349-368)
// NORMAL = PAINT_METHOD 'normal' (cone beams, rob:280-285 + bpw:562-566); false = 'fast' (ball query).
template <int KW, bool NORMAL, bool GENSEC>
__global__ __launch_bounds__(256, 4) void step_kernel(StepArgs a) {
    const int lane = threadIdx.x & 63;
    const int env = rfl(blockIdx.x * 4 + (threadIdx.x >> 6));
    if (env >= a.n_envs) return;
#ifdef PRL_WAVE_TIMES
    const unsigned long long wave_t0 = __builtin_amdgcn_s_memtime();
    const unsigned long long wave_r0 = __builtin_amdgcn_s_memrealtime();
#endif
    const int part_id = a.env_part ? a.env_part[env] : 0;
    PartRef P = *(const PartDev CAS *)(a.parts + part_id);
    CfgRef C = *(const PrlConfig CAS *)a.cfg;
    const int od = obs_dim_of(C.obs_mode, C.obs_grad);
#ifdef PRL_REPEAT          // diagnostic build: PRL_REPEAT whole steps per launch (cold-start vs steady-state cost)
  for (int prl_rep = 0; prl_rep < PRL_REPEAT; ++prl_rep) {
    __threadfence();
#endif
    EnvState S = *reinterpret_cast<const EnvState *>(a.state + (size_t)env * PRL_STATE_DOUBLES);
    uint64_t painted[KW_MAX] = {0, 0, 0, 0}, last[KW_MAX] = {0, 0, 0, 0};
#ifdef PRL_PHASE_TIMING
    Prof prof;
    for (int k = 0; k < PH_COUNT; ++k) prof.acc[k] = 0;
    prof.prev = __builtin_amdgcn_s_memtime();
#endif
    load_masks<KW>(a, env, P.n_words, lane, painted, last);
    STAMP(PH_LOAD);

    // ---- action -> (delta1, delta2, turning angle)   rge:342-347, rob:390-398, 352-358
    double delta1, delta2, new_angle;
    if (C.action_mode == PRL_ACT_DISCRETE) {
        int act = reinterpret_cast<const int *>(a.actions)[env];
        act = act < 0 ? 0 : (act >= C.n_discrete ? C.n_discrete - 1 : act);
        delta1 = C.act_delta1[act];
        delta2 = C.act_delta2[act];
        new_angle = C.act_angle[act];
    } else {
        const double *av = reinterpret_cast<const double *>(a.actions) + (size_t)env * C.action_dim;
        double a0 = av[0], a1 = C.action_dim > 1 ? av[1] : 0.0;
        if (!(-1 <= a0 && a0 <= 1)) a0 = a0 < -1 ? -1 : (a0 > 1 ? 1 : a0);
        if (!(-1 <= a1 && a1 <= 1)) a1 = a1 < -1 ? -1 : (a1 > 1 ? 1 : a1);
        double dx, dy;
        if (C.action_dim == 1) {                       // rob:152-153
            const double phi = (a0 + 1) * PI;
            dx = 1 * cos(phi);
            dy = 1 * sin(phi);
        } else {                                       // rob:154-160
            const double phi = atan2(a1, a0);
            const double ax = fabs(a0), ay = fabs(a1);
            if (ax == 0 && ay == 0) {
                dx = ax;
                dy = ay;
            } else {
                const double mx = ax > ay ? ax : ay;
                dx = mx * cos(phi);
                dy = mx * sin(phi);
            }
        }
        delta1 = dx * C.step_size;
        delta2 = dy * C.step_size;
        new_angle = delta1 != 0 ? atan(fabs(delta2 / delta1)) : PI / 2;
    }
    const double angle_diff = fabs(new_angle - S.last_angle);
    S.last_angle = new_angle;
    const int counter_before = S.terminate_counter;

    // ---- five chained sub-shots   rob:302-329 + 403-424
    double cur_pose[3] = {S.pose[0], S.pose[1], S.pose[2]}, cur_norm[3];
    tcp_orn_norm(S.pose, S.quat, cur_norm);
#pragma unroll
    for (int k = 0; k < 3; ++k) cur_norm[k] = uni_d(cur_norm[k]);
    const double d1 = uni_d(delta1 / PAINT_PER_ACTION), d2 = uni_d(delta2 / PAINT_PER_ACTION);
    // facet hit by the previous ray, also across steps (convex fast path); only a cache, but it indexes a table
    int facet_hint = (S.facet_hint >= 0 && S.facet_hint < P.n_col_pad) ? S.facet_hint : -1;
    uint64_t n_uni[KW_MAX] = {0, 0, 0, 0};      // NORMAL only: union of valid samples over the five shots
    uint32_t n_succeeded_l = 0;
    __shared__ double s_centres[4][PAINT_PER_ACTION * 3];
    double *cen = s_centres[threadIdx.x >> 6];
    for (int shot = 0; shot < PAINT_PER_ACTION; ++shot) {
        // bpw:865-880 get_guided_point
        double pt[3] = {cur_pose[0], cur_pose[1], cur_pose[2]};
        const double delta_2 = d2 * P.lwr;
        if (P.a1 == 0) pt[0] += d1; else if (P.a1 == 1) pt[1] += d1; else pt[2] += d1;
        if (P.a2 == 0) pt[0] += delta_2; else if (P.a2 == 1) pt[1] += delta_2; else pt[2] += delta_2;
        const double end[3] = {pt[0] + cur_norm[0], pt[1] + cur_norm[1], pt[2] + cur_norm[2]};
        double t, hit[3], pos[3], orn[3], quat[4];
        STAMP(PH_MATH);
#ifdef PRL_ABLATE_RAY                       // diagnostic stand-in: hit 0.1 along the normal, no table reads
        bool on = true;
        t = 0.1;
        hit[0] = pt[0] + 0.1 * cur_norm[0];
        hit[1] = pt[1] + 0.1 * cur_norm[1];
        hit[2] = pt[2] + 0.1 * cur_norm[2];
#else
#ifdef PRL_DOUBLE_RAY                       // diagnostic build: the phase runs twice, the first result is discarded
        {
            double t2, h2[3] = {0, 0, 0};
            int hint2 = facet_hint;
            const int r2 = ray_closest_wave(P, pt, end, lane, t2, h2, hint2);
            asm volatile("" ::"v"(t2), "v"(h2[0]), "v"(h2[1]), "v"(h2[2]), "s"(r2), "s"(hint2));
        }
#endif
        bool on = ray_closest_wave(P, pt, end, lane, t, hit, facet_hint) >= 0;
#endif
        STAMP(PH_RAY);
#ifdef PRL_DOUBLE_HOOK
        if (on) {
            double p2[3] = {0, 0, 0}, o2[3] = {0, 0, 0};
            const bool b2 = hook_point_wave(P, hit, lane, p2, o2 PROF_PASS);
            asm volatile("" ::"v"(p2[0]), "v"(p2[1]), "v"(p2[2]), "v"(o2[0]), "v"(o2[1]), "v"(o2[2]), "s"((int)b2));
        }
#endif
        if (on) on = hook_point_wave(P, hit, lane, pos, orn PROF_PASS);
        if (!on) {
            orn[0] = cur_norm[0];
            orn[1] = cur_norm[1];
            orn[2] = cur_norm[2];
        }
        pose_orn_quat(orn, quat);
        if (!on) {
            transform_point(cur_pose, quat, d2, d1, 0.0, pos);      // rob:317, tool frame [delta2, delta1, 0]
            if (S.last_on_part) {                                    // rob:292-300
                S.last_on_part = 0;
            } else {
                S.terminate_counter += 1;
                if (S.terminate_counter > NOT_ON_PART_TERMINATE) S.terminate = 1;
            }
        } else {
            S.last_on_part = 1;
        }
#pragma unroll
        for (int k = 0; k < 3; ++k) {
            pos[k] = uni_d(pos[k]);
            orn[k] = uni_d(orn[k]);
            cur_pose[k] = pos[k];
            cur_norm[k] = orn[k];
            S.pose[k] = pos[k];
        }
#pragma unroll
        for (int k = 0; k < 4; ++k) S.quat[k] = quat[k];       // stays in vector registers: scalar registers are the scarce kind
        // rob:277-278 shot centre; painting is deferred until all five centres are known
        double center[3];
        transform_point(pos, quat, 0.0, 0.0, 0.1, center);
#pragma unroll
        for (int k = 0; k < 3; ++k)
            if (lane == 0) cen[3 * shot + k] = center[k];
        STAMP(PH_MATH);
        if constexpr (NORMAL) {
            // rob:251-258, 280-285: one ray per cone beam from the tool to the beam's end point on the
            // plane 0.2 ahead; bpw:562-566: every hit paints the sample nearest to it.  No hit at all:
            // the reference returns early and leaves the last-shot set untouched.
            uint64_t cur[KW_MAX] = {0, 0, 0, 0};
            int beam_hits = 0;
            for (int bm = 0; bm < P.n_beams; ++bm) {
                double dst[3], bt, bh[3];
                transform_point(pos, quat, P.beams[3 * bm], P.beams[3 * bm + 1], P.beams[3 * bm + 2], dst);
                int beam_hint = -1;
                if (ray_closest_wave(P, pos, dst, lane, bt, bh, beam_hint) < 0) continue;
                ++beam_hits;
                const int sidx = nearest_sample_wave(P, bh, lane);
                if (sidx >= 0) set_word<KW>(cur, sidx >> 6, (uint64_t)1 << (sidx & 63), lane);
            }
            if (beam_hits > 0) {
#pragma unroll
                for (int k = 0; k < KW; ++k) {
                    n_succeeded_l += __popcll(cur[k] & ~painted[k]);
                    painted[k] |= cur[k];
                    n_uni[k] |= cur[k] & ~last[k];
                    last[k] = cur[k];
                }
            }
        }
    }
    // bpw:568-577 fast_paint + _paint for the five shots
    int succeeded = 0, pixel_counter = 0;
    if constexpr (NORMAL) {
        uint32_t pix_l = 0;
#pragma unroll
        for (int k = 0; k < KW; ++k) pix_l += __popcll(n_uni[k]);
        const uint64_t sums = wave_sum_u64(((uint64_t)n_succeeded_l << 32) | pix_l);
        succeeded = (int)(sums >> 32);
        pixel_counter = (int)(sums & 0xffffffffu);
    } else {
        uint64_t new_last[KW_MAX] = {0, 0, 0, 0};
#ifdef PRL_ABLATE_BALL
        if (true) {
#else
#ifdef PRL_DOUBLE_BALL
        {
            uint64_t p2[KW_MAX], l2[KW_MAX] = {0, 0, 0, 0};
            int s2 = 0, c2 = 0;
            for (int k = 0; k < KW_MAX; ++k) p2[k] = painted[k];
            const bool b2 = paint_shots_union<KW>(P, C.paint_radius, cen, lane, p2, last, l2, s2, c2);
            asm volatile("" ::"v"(p2[0]), "v"(l2[0]), "v"(p2[1]), "v"(l2[1]), "v"(p2[2]), "v"(l2[2]), "s"(s2), "s"(c2), "s"((int)b2));
        }
#endif
        if (paint_shots_union<KW>(P, C.paint_radius, cen, lane, painted, last, new_last, succeeded, pixel_counter)) {
#endif
#pragma unroll
            for (int k = 0; k < KW; ++k) last[k] = new_last[k];
        } else {                                   // general path: one ball query per shot
            uint64_t uni[KW_MAX] = {0, 0, 0, 0};
            uint32_t succeeded_l = 0, pix_l = 0;
            for (int shot = 0; shot < PAINT_PER_ACTION; ++shot) {
                uint64_t cur[KW_MAX] = {0, 0, 0, 0};
                const double c3[3] = {cen[3 * shot], cen[3 * shot + 1], cen[3 * shot + 2]};
                ball_query_wave<KW>(P, C.paint_radius, c3, lane, cur);
#pragma unroll
                for (int k = 0; k < KW; ++k) {
                    succeeded_l += __popcll(cur[k] & ~painted[k]);
                    painted[k] |= cur[k];
                    uni[k] |= cur[k] & ~last[k];
                    last[k] = cur[k];
                }
            }
#pragma unroll
            for (int k = 0; k < KW; ++k) pix_l += __popcll(uni[k]);
            const uint64_t sums = wave_sum_u64(((uint64_t)succeeded_l << 32) | pix_l);
            succeeded = (int)(sums >> 32);
            pixel_counter = (int)(sums & 0xffffffffu);
        }
    }
    STAMP(PH_BALL);
    S.facet_hint = facet_hint;
    const double rate = pixel_counter ? (double)succeeded / (double)pixel_counter : 0.0;      // rob:425-426
    if (S.terminate_counter - counter_before >= PAINT_PER_ACTION && pixel_counter == 0) S.terminate = 1;

    // ---- reward, penalty, termination   rge:321-340, 289-304
    const double rew = (double)succeeded / 100;
    S.total_reward += rew;
    double pen = 0.2;
    if (C.overlap_penalty) pen += 0.1 * (1 - rate);
    if (C.turning_penalty) pen += 0.1 * (angle_diff / PI);
    const double actual = rew - pen;
    S.step_counter += 1;
    const double max_pts = C.max_possible_point[part_id & 7];
    const int finished = max_pts > S.total_reward * 100 ? 0 : 1;
    const double avg = S.total_reward / S.step_counter;
    const double expected = max_pts / (C.expected_episode_len * 100);
    int dn;
    if (avg < expected && C.termination_mode != PRL_TERM_LATE &&
        (C.termination_mode == PRL_TERM_EARLY || S.total_reward < C.switch_threshold * max_pts / 100))
        dn = 1;
    else
        dn = finished || S.terminate || S.step_counter > C.max_episode_len - 1;
    if (!dn) S.total_return += actual;
    STAMP(PH_APPLY);

    const bool do_reset = dn && C.auto_reset;
    double *obs_row = a.obs + (size_t)env * od;
    double *term_row = do_reset ? (a.final_obs ? a.final_obs + (size_t)env * od : nullptr) : obs_row;
#ifndef PRL_ABLATE_OBS
#ifdef PRL_DOUBLE_OBS
    if (term_row) observation_wave<KW, GENSEC>(P, C, S.pose, painted, lane, term_row);
    __builtin_amdgcn_s_waitcnt(0);
#endif
    if (term_row) observation_wave<KW, GENSEC>(P, C, S.pose, painted, lane, term_row);
#endif
    if (lane == 0) {
        a.reward[env] = actual;
        a.done[env] = (uint8_t)dn;
        a.info[2 * (size_t)env] = rew;
        a.info[2 * (size_t)env + 1] = pen;
    }
    if (dn) {                                   // episode statistics (the RCCL gather payload)
        uint32_t cnt_l = 0;
#pragma unroll
        for (int k = 0; k < KW; ++k) cnt_l += __popcll(painted[k]);
        S.last_ep_painted = (int)wave_sum_u64(cnt_l);
        S.last_ep_return = S.total_return;
        S.last_ep_reward = S.total_reward;
        S.last_ep_len = S.step_counter;
    }
    if (do_reset) {
        int start = a.start_idx ? a.start_idx[env] : draw_start(C.seed, env, S.episode, P.n_start);
        start = start < 0 ? 0 : (start >= P.n_start ? P.n_start - 1 : start);
        reset_state(P, S, start);
#pragma unroll
        for (int k = 0; k < KW; ++k) {
            painted[k] = 0;
            last[k] = 0;
        }
        observation_wave<KW, GENSEC>(P, C, S.pose, painted, lane, obs_row);
    }
    STAMP(PH_OBS);
    store_masks<KW>(a, env, P.n_words, lane, painted, last);
    store_state(a.state + (size_t)env * PRL_STATE_DOUBLES, S, lane);
    STAMP(PH_STORE);
#ifdef PRL_WAVE_TIMES      // diagnostic build: wave lifetime and on-part shot count into final_obs[env][0..1]
    if (lane == 0 && a.final_obs) {
        a.final_obs[(size_t)env * od] = (double)(__builtin_amdgcn_s_memtime() - wave_t0);
        uint32_t *wc = g_wcnt[env & 0xffff];       // misses:3 dn:1 | general rays:4 | stage 1:4 | chunk tests:6 |
        const uint64_t packed =                    // vertex batches:8 | extra rings:4 | paint words:8 | straddle words:8
            (uint64_t)(S.terminate_counter - counter_before) | ((uint64_t)dn << 3) | ((uint64_t)(wc[0] & 15) << 4) |
            ((uint64_t)(wc[1] & 15) << 8) | ((uint64_t)(wc[2] & 63) << 12) | ((uint64_t)(wc[3] & 255) << 18) |
            ((uint64_t)(wc[4] & 15) << 26) | ((uint64_t)(wc[5] & 255) << 30) | ((uint64_t)(wc[6] & 255) << 38) |
            ((uint64_t)(wc[7] & 15) << 46) | ((uint64_t)((wc[4] >> 4) & 15) << 50);
        for (int k = 0; k < 8; ++k) wc[k] = 0;
        a.final_obs[(size_t)env * od + 1] = (double)packed;
        a.final_obs[(size_t)env * od + 2] = (double)dn;
        a.final_obs[(size_t)env * od + 3] = (double)wave_r0;                              // 100 MHz wall clock
        a.final_obs[(size_t)env * od + 4] = (double)__builtin_amdgcn_s_memrealtime();
        a.final_obs[(size_t)env * od + 5] =                                               // HW_ID + 2^32 * XCC_ID
            (double)(((uint64_t)__builtin_amdgcn_s_getreg((20 << 0) | (0 << 6) | (3 << 11)) << 32) |
                     __builtin_amdgcn_s_getreg((4 << 0) | (0 << 6) | (31 << 11)));
    }
#endif
#ifdef PRL_PHASE_TIMING
    if (lane == 0)
        for (int k = 0; k < PH_COUNT; ++k) atomicAdd(&g_phase_cycles[k], prof.acc[k]);
#endif
#ifdef PRL_REPEAT
  }
#endif
}

// ---------------------------------------------------------------- rayTestBatch drop-in: one wave per ray
__global__ __launch_bounds__(256) void ray_batch_kernel(const PartDev *part, int n, const double *from,
                                                        const double *to, int *tri, double *frac, double *pos) {
    const int lane = threadIdx.x & 63;
    const int r = rfl(blockIdx.x * 4 + (threadIdx.x >> 6));
    if (r >= n) return;
    const double o[3] = {from[3 * r], from[3 * r + 1], from[3 * r + 2]};
    const double e[3] = {to[3 * r], to[3 * r + 1], to[3 * r + 2]};
    double t, hit[3] = {0, 0, 0};
    int hint = -1;
    const int idx = ray_closest_wave(*(const PartDev CAS *)part, o, e, lane, t, hit, hint);
    if (lane == 0) {
        tri[r] = idx;
        frac[r] = t;
        pos[3 * r] = hit[0];
        pos[3 * r + 1] = hit[1];
        pos[3 * r + 2] = hit[2];
    }
}

// Copies the coverage words out (prl_batch_get_mask).  Same access shape as the step kernel's mask
// traffic (8 bytes per lane, coalesced), so tools/hbm_calibration.py uses it to calibrate FETCH_SIZE.
__global__ void copy_mask_kernel(const uint64_t *src, uint64_t *dst, size_t n) {
    const size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i < n) dst[i] = src[i];
}

__global__ void gather_state_kernel(const double *state, int n, int field, double *out) {
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i < n) out[i] = state[(size_t)i * PRL_STATE_DOUBLES + field];
}

// ================================================================= host side
thread_local char g_error[512] = "";

int fail(int code, const char *fmt, ...) {
    va_list ap;
    va_start(ap, fmt);
    vsnprintf(g_error, sizeof g_error, fmt, ap);
    va_end(ap);
    return code;
}

#define HIP_TRY(expr)                                                                              \
    do {                                                                                           \
        hipError_t e_ = (expr);                                                                    \
        if (e_ != hipSuccess) return fail(PRL_E_HIP, "%s: %s", #expr, hipGetErrorString(e_));     \
    } while (0)

}  // namespace

struct PrlPart {
    int device = 0;
    PartDev dev{};                 // device pointers inside
    std::vector<void *> allocs;
    int max_adj = 0;
    PartDev *dev_copy = nullptr;   // single-part device copy (prl_ray_batch)
};

struct PrlBatch {
    int device = 0, n_envs = 0, n_parts = 0, mask_stride = 0, kw = 0;
    PrlConfig cfg{};
    PartDev *parts_dev = nullptr;
    PrlConfig *cfg_dev = nullptr;
    int *env_part_dev = nullptr;
    uint64_t *painted = nullptr, *last = nullptr;
    double *state = nullptr;
    int timing_every = 0;          // 0 = off; k = HIP events around every k-th step launch
    long long launch_no = 0;
    std::vector<hipEvent_t> ev_start, ev_stop;
    size_t ev_used = 0;
};

namespace {

template <typename T>
int upload(PrlPart *p, const T *host, size_t count, const T GAS **out) {
    *out = nullptr;
    if (count == 0) return PRL_OK;
    if (!host) return fail(PRL_E_INVALID, "null table pointer");
    void *d = nullptr;
    HIP_TRY(hipMalloc(&d, count * sizeof(T)));
    p->allocs.push_back(d);
    HIP_TRY(hipMemcpy(d, host, count * sizeof(T), hipMemcpyHostToDevice));
    *out = (const T GAS *)(d);
    return PRL_OK;
}

#define UP(field, host, count)                                        \
    do {                                                              \
        int rc_ = upload(p, host, (size_t)(count), &p->dev.field);    \
        if (rc_) return rc_;                                          \
    } while (0)

// grid-cell start tables index the sorted arrays: a malformed one would send device loads out of bounds
bool monotone_starts(const int32_t *start, size_t n, int limit) {
    if (!start || n == 0 || start[0] < 0) return false;
    for (size_t k = 1; k < n; ++k)
        if (start[k] < start[k - 1]) return false;
    return start[n - 1] <= limit;
}

int part_fill(PrlPart *p, const PrlPartTables *t) {
    PartDev &d = p->dev;
    if (t->n_samples <= 0 || t->n_samples_pad % 64 || t->n_samples_pad < t->n_samples)
        return fail(PRL_E_INVALID, "bad sample counts %d/%d", t->n_samples, t->n_samples_pad);
    d.n_samples = t->n_samples;
    d.n_samples_pad = t->n_samples_pad;
    d.n_words = t->n_samples_pad / 64;
    if (d.n_words > 64 * KW_MAX)
        return fail(PRL_E_UNSUPPORTED, "part has %d samples; this build keeps at most %d per env in registers",
                    t->n_samples, 64 * 64 * KW_MAX);
    for (int k = 0; k < 3; ++k) UP(samp[k], t->sample_xyz[k], t->n_samples_pad);
    UP(word_bbox, t->word_bbox, (size_t)d.n_words * 4);
    UP(word_valid, t->word_valid, d.n_words);
    UP(samp_rank, t->sample_rank, t->n_samples_pad);
    d.sg_o1 = t->sgrid_origin[0];
    d.sg_o2 = t->sgrid_origin[1];
    d.sg_inv = t->sgrid_inv_cell;
    d.sg_nx = t->sgrid_nx;
    d.sg_ny = t->sgrid_ny;
    if (d.sg_nx <= 0 || d.sg_ny <= 0) return fail(PRL_E_INVALID, "empty sample grid");
    UP(sg_start, t->sgrid_start, (size_t)d.sg_nx * d.sg_ny + 1);
    if (!monotone_starts(t->sgrid_start, (size_t)d.sg_nx * d.sg_ny + 1, t->n_samples_pad))
        return fail(PRL_E_INVALID, "sample grid starts must be non-decreasing and within the padded sample count");
    d.n_obs_cells = t->n_obs_cells;
    if (d.n_obs_cells > 0) {
        UP(cell_mask, t->obs_cell_mask, (size_t)d.n_obs_cells * d.n_words);
        UP(cell_count, t->obs_cell_count, d.n_obs_cells);
    }
    d.n_vertices = t->n_vertices;
    if (d.n_vertices <= 0) return fail(PRL_E_INVALID, "part has no same-side vertices");
    for (int k = 0; k < 3; ++k) UP(vert[k], t->vertex_xyz[k], d.n_vertices);
    UP(vert_rank, t->vertex_rank, d.n_vertices);
    d.n_triangles = t->n_triangles;
    d.adj_width = t->adj_width;
    if (d.adj_width < 1 || d.adj_width > 64)
        return fail(PRL_E_UNSUPPORTED, "adjacency width %d (a vertex may have at most 64 incident triangles)", d.adj_width);
    if (!t->vertex_adj) return fail(PRL_E_INVALID, "null adjacency table");
    for (size_t k = 0; k < (size_t)d.n_vertices * d.adj_width; ++k)
        if (t->vertex_adj[k] < -1 || t->vertex_adj[k] >= d.n_triangles)
            return fail(PRL_E_INVALID, "adjacency entry %zu out of range", k);
    UP(vadj, t->vertex_adj, (size_t)d.n_vertices * d.adj_width);
    d.vg_o1 = t->vgrid_origin[0];
    d.vg_o2 = t->vgrid_origin[1];
    d.vg_inv = t->vgrid_inv_cell;
    d.vg_accept = t->vgrid_accept;
    d.vg_nx = t->vgrid_nx;
    d.vg_ny = t->vgrid_ny;
    if (d.vg_nx <= 0 || d.vg_ny <= 0) return fail(PRL_E_INVALID, "empty vertex grid");
    UP(vg_start, t->vgrid_start, (size_t)d.vg_nx * d.vg_ny + 1);
    if (!monotone_starts(t->vgrid_start, (size_t)d.vg_nx * d.vg_ny + 1, d.n_vertices) ||
        t->vgrid_start[(size_t)d.vg_nx * d.vg_ny] != d.n_vertices)
        return fail(PRL_E_INVALID, "vertex grid starts must be non-decreasing and end at n_vertices");
    UP(tri_rec, t->tri_records, (size_t)d.n_triangles * 16);
    d.n_col = t->n_collision;
    d.n_col_pad = t->n_collision_pad;
    if (d.n_col <= 0 || d.n_col_pad % 64 || d.n_col_pad < d.n_col) return fail(PRL_E_INVALID, "bad collision counts");
    for (int k = 0; k < 9; ++k) UP(col[k], t->col_v0e1e2[k], d.n_col_pad);
    UP(col_bbox, t->col_bbox, (size_t)d.n_col_pad * 8);
    UP(col_rank, t->col_rank, d.n_col_pad);
    d.col_convex = t->col_convex ? 1 : 0;
    d.nbr_width = t->nbr_width;
    if (d.col_convex) {
        if (d.nbr_width < 1 || d.nbr_width > 64) return fail(PRL_E_INVALID, "nbr_width must be 1..64");
        for (size_t k = 0; k < (size_t)d.n_col_pad * d.nbr_width; ++k)
            if (t->col_nbr[k] < -1 || t->col_nbr[k] >= d.n_col_pad) return fail(PRL_E_INVALID, "col_nbr entry out of range");
        UP(col_nbr, t->col_nbr, (size_t)d.n_col_pad * d.nbr_width);
        UP(col_orient, t->col_orient, d.n_col_pad);
        // per-facet record of the single-facet fast path (ray_closest_wave): geometry, the barycentric
        // margin that keeps a hit FACET_EDGE_MARGIN away from every edge, |e1 x e2|^2, orientation
        std::vector<double> rec((size_t)d.n_col_pad * 12, 0.0);
        for (int i = 0; i < d.n_col_pad; ++i) {
            double *r = rec.data() + (size_t)i * 12;
            for (int k = 0; k < 9; ++k) r[k] = t->col_v0e1e2[k][i];
            const double *e1 = r + 3, *e2 = r + 6;
            const double n0 = e1[1] * e2[2] - e1[2] * e2[1], n1 = e1[2] * e2[0] - e1[0] * e2[2],
                         n2 = e1[0] * e2[1] - e1[1] * e2[0];
            const double nn = n0 * n0 + n1 * n1 + n2 * n2;
            const double l1 = e1[0] * e1[0] + e1[1] * e1[1] + e1[2] * e1[2];
            const double l2 = e2[0] * e2[0] + e2[1] * e2[1] + e2[2] * e2[2];
            const double f0 = e2[0] - e1[0], f1 = e2[1] - e1[1], f2 = e2[2] - e1[2];
            const double l3 = f0 * f0 + f1 * f1 + f2 * f2;
            const double lmax = std::max(l1, std::max(l2, l3));
            const double hmin = lmax > 0 ? std::sqrt(nn / lmax) : 0.0;          // smallest height of the facet
            r[9] = hmin > 0 ? FACET_EDGE_MARGIN / hmin : INFINITY;              // never met by a sliver
            r[10] = nn;
            r[11] = (double)t->col_orient[i];                                   // 0 for pads: never entered
        }
        UP(col_rec, rec.data(), rec.size());
    }
    d.n_col_chunks = t->n_col_chunks;
    if (d.n_col_chunks != d.n_col_pad / 64) return fail(PRL_E_INVALID, "n_col_chunks must be n_collision_pad / 64");
    UP(col_chunk_bbox, t->col_chunk_bbox, (size_t)((d.n_col_chunks + 63) / 64) * 64 * 8);
    UP(grid_lo, t->grid_lo, GRID_GRANULARITY);
    UP(grid_hi, t->grid_hi, GRID_GRANULARITY);
    d.r1min = t->range1[0];
    d.r1max = t->range1[1];
    d.r2min = t->range2[0];
    d.r2max = t->range2[1];
    d.lwr = t->length_width_ratio;
    d.a0 = t->axis0;
    d.a1 = t->axis1;
    d.a2 = t->axis2;
    if (d.a0 < 0 || d.a0 > 2 || d.a1 < 0 || d.a1 > 2 || d.a2 < 0 || d.a2 > 2 || d.a0 == d.a1 || d.a1 == d.a2 ||
        d.a0 == d.a2)
        return fail(PRL_E_INVALID, "axes must be a permutation of 0,1,2");
    d.samp_a1 = d.samp[d.a1];
    d.samp_a2 = d.samp[d.a2];
    {   // the samples of a word ascend on axis a1 (paintrl.h); derive the equal-run ends the observation uses
        const double *x = t->sample_xyz[d.a1];
        std::vector<uint8_t> ub((size_t)t->n_samples_pad);
        for (int w = 0; w < d.n_words; ++w) {
            const double *xw = x + (size_t)w * 64;
            for (int j = 1; j < 64; ++j)
                if (!(xw[j - 1] <= xw[j]))
                    return fail(PRL_E_INVALID, "samples of word %d do not ascend on axis %d", w, d.a1);
            int end = 64;
            for (int j = 63; j >= 0; --j) {
                if (j < 63 && xw[j] != xw[j + 1]) end = j + 1;
                ub[(size_t)w * 64 + j] = (uint8_t)end;
            }
        }
        UP(samp_ub, ub.data(), ub.size());
    }
    d.n_start = t->n_start;
    if (d.n_start <= 0) return fail(PRL_E_INVALID, "part has no start points");
    UP(start_pos, t->start_pos, (size_t)d.n_start * 3);
    UP(start_quat, t->start_quat, (size_t)d.n_start * 4);
    d.n_beams = t->n_beams;
    if (d.n_beams > 0) UP(beams, t->beams, (size_t)d.n_beams * 3);
    return PRL_OK;
}

int check_config(const PrlConfig *c) {
    if (c->obs_mode < 0 || c->obs_mode > 3) return fail(PRL_E_INVALID, "obs_mode %d", c->obs_mode);
    if ((c->obs_mode == PRL_OBS_SECTION || c->obs_mode == PRL_OBS_DISCRETE) && (c->obs_grad < 1 || c->obs_grad > 62))
        return fail(PRL_E_UNSUPPORTED, "section/discrete observation: OBS_GRAD must be 1..62");
    if (c->obs_mode == PRL_OBS_GRID && (c->obs_grad < 1 || c->obs_grad > 16))
        return fail(PRL_E_UNSUPPORTED, "grid observation: OBS_GRAD must be 1..16");
    if (c->action_mode == PRL_ACT_DISCRETE) {
        if (c->n_discrete < 1 || c->n_discrete > PRL_MAX_DISCRETE) return fail(PRL_E_INVALID, "n_discrete %d", c->n_discrete);
    } else if (c->action_mode == PRL_ACT_CONTINUOUS) {
        if (c->action_dim < 1 || c->action_dim > 2) return fail(PRL_E_INVALID, "action_dim %d", c->action_dim);
    } else {
        return fail(PRL_E_INVALID, "action_mode %d", c->action_mode);
    }
    if (c->termination_mode < 0 || c->termination_mode > 2) return fail(PRL_E_INVALID, "termination_mode");
    if (c->paint_method != PRL_PAINT_FAST && c->paint_method != PRL_PAINT_NORMAL) return fail(PRL_E_INVALID, "paint_method");
    if (c->max_episode_len < 1 || c->expected_episode_len < 1) return fail(PRL_E_INVALID, "episode lengths");
    if (!(c->paint_radius > 0) || !(c->step_size > 0)) return fail(PRL_E_INVALID, "paint_radius and step_size must be positive");
    return PRL_OK;
}

template <int KW>
void launch_step(const StepArgs &a, bool normal, bool gensec, hipStream_t s) {
    const dim3 grid((a.n_envs + 3) / 4), block(256);
    if (normal && gensec) hipLaunchKernelGGL((step_kernel<KW, true, true>), grid, block, 0, s, a);
    else if (normal) hipLaunchKernelGGL((step_kernel<KW, true, false>), grid, block, 0, s, a);
    else if (gensec) hipLaunchKernelGGL((step_kernel<KW, false, true>), grid, block, 0, s, a);
    else hipLaunchKernelGGL((step_kernel<KW, false, false>), grid, block, 0, s, a);
}

template <int KW>
void launch_reset(const StepArgs &a, bool gensec, hipStream_t s) {
    const dim3 grid((a.n_envs + 3) / 4), block(256);
    if (gensec) hipLaunchKernelGGL((reset_kernel<KW, true>), grid, block, 0, s, a);
    else hipLaunchKernelGGL((reset_kernel<KW, false>), grid, block, 0, s, a);
}

template <int KW>
void launch_observe(const StepArgs &a, bool gensec, hipStream_t s) {
    const dim3 grid((a.n_envs + 3) / 4), block(256);
    if (gensec) hipLaunchKernelGGL((observe_kernel<KW, true>), grid, block, 0, s, a);
    else hipLaunchKernelGGL((observe_kernel<KW, false>), grid, block, 0, s, a);
}

bool general_section(const PrlConfig &c) {
    return (c.obs_mode == PRL_OBS_SECTION || c.obs_mode == PRL_OBS_DISCRETE) && c.obs_grad != 4;
}

StepArgs base_args(PrlBatch *b) {
    StepArgs a{};
    a.parts = b->parts_dev;
    a.cfg = b->cfg_dev;
    a.env_part = b->env_part_dev;
    a.n_envs = b->n_envs;
    a.mask_stride = b->mask_stride;
    a.painted = b->painted;
    a.last = b->last;
    a.state = b->state;
    return a;
}

}  // namespace

// ================================================================= C ABI
extern "C" {

int prl_abi_version(void) { return PRL_ABI_VERSION; }
const char *prl_last_error(void) { return g_error; }

int prl_obs_dim(const PrlConfig *cfg) {
    if (!cfg) return fail(PRL_E_INVALID, "null config");
    return obs_dim_of(cfg->obs_mode, cfg->obs_grad);
}

int prl_struct_sizes(int *config_bytes, int *part_tables_bytes) {
    if (config_bytes) *config_bytes = (int)sizeof(PrlConfig);
    if (part_tables_bytes) *part_tables_bytes = (int)sizeof(PrlPartTables);
    return PRL_OK;
}

int prl_part_create(const PrlPartTables *t, int device, PrlPart **out) {
    if (!t || !out) return fail(PRL_E_INVALID, "null argument");
    *out = nullptr;
    HIP_TRY(hipSetDevice(device));
    PrlPart *p = new (std::nothrow) PrlPart();
    if (!p) return fail(PRL_E_NOMEM, "out of host memory");
    p->device = device;
    int rc = part_fill(p, t);
    if (rc == PRL_OK) {
        hipError_t e = hipMalloc(reinterpret_cast<void **>(&p->dev_copy), sizeof(PartDev));
        if (e == hipSuccess) e = hipMemcpy(p->dev_copy, &p->dev, sizeof(PartDev), hipMemcpyHostToDevice);
        if (e != hipSuccess) rc = fail(PRL_E_HIP, "part descriptor upload: %s", hipGetErrorString(e));
    }
    if (rc != PRL_OK) {
        prl_part_destroy(p);
        return rc;
    }
    *out = p;
    return PRL_OK;
}

void prl_part_destroy(PrlPart *p) {
    if (!p) return;
    (void)hipSetDevice(p->device);
    for (void *d : p->allocs) (void)hipFree(d);
    if (p->dev_copy) (void)hipFree(p->dev_copy);
    delete p;
}

int prl_part_mask_words(const PrlPart *p) { return p ? p->dev.n_words : fail(PRL_E_INVALID, "null part"); }

int prl_batch_create(PrlPart *const *parts, int n_parts, const int32_t *env_part_id, int n_envs, const PrlConfig *cfg,
                     PrlBatch **out) {
    if (!parts || !cfg || !out || n_parts < 1 || n_parts > 8 || n_envs < 1)
        return fail(PRL_E_INVALID, "bad arguments (n_parts 1..8, n_envs >= 1)");
    *out = nullptr;
    int rc = check_config(cfg);
    if (rc) return rc;
    for (int i = 0; i < n_parts; ++i) {
        if (!parts[i]) return fail(PRL_E_INVALID, "null part %d", i);
        if (parts[i]->device != parts[0]->device) return fail(PRL_E_INVALID, "parts live on different devices");
        if (cfg->paint_radius * parts[i]->dev.sg_inv >= 1.0)
            return fail(PRL_E_INVALID, "part %d: sample grid cell %.4f does not exceed the paint radius %.4f", i,
                        1.0 / parts[i]->dev.sg_inv, cfg->paint_radius);
        if (cfg->paint_method == PRL_PAINT_NORMAL && parts[i]->dev.n_beams <= 0)
            return fail(PRL_E_INVALID, "part %d has no cone beams but PAINT_METHOD='normal' was requested", i);
        if (cfg->obs_mode == PRL_OBS_GRID && parts[i]->dev.n_obs_cells != cfg->obs_grad * cfg->obs_grad)
            return fail(PRL_E_INVALID, "part %d was packed for %d observation cells, config wants %d", i,
                        parts[i]->dev.n_obs_cells, cfg->obs_grad * cfg->obs_grad);
    }
    if (env_part_id)
        for (int i = 0; i < n_envs; ++i)
            if (env_part_id[i] < 0 || env_part_id[i] >= n_parts) return fail(PRL_E_INVALID, "env_part_id[%d] out of range", i);
    PrlBatch *b = new (std::nothrow) PrlBatch();
    if (!b) return fail(PRL_E_NOMEM, "out of host memory");
    b->device = parts[0]->device;
    b->n_envs = n_envs;
    b->n_parts = n_parts;
    b->cfg = *cfg;
    for (int i = 0; i < n_parts; ++i)
        if (parts[i]->dev.n_words > b->mask_stride) b->mask_stride = parts[i]->dev.n_words;
    b->kw = (b->mask_stride + 63) / 64;
    hipError_t e = hipSetDevice(b->device);
    std::vector<PartDev> pd(n_parts);
    for (int i = 0; i < n_parts; ++i) pd[i] = parts[i]->dev;
    const size_t mask_bytes = (size_t)n_envs * b->mask_stride * sizeof(uint64_t);
    const size_t state_bytes = (size_t)n_envs * PRL_STATE_DOUBLES * sizeof(double);
    if (e == hipSuccess) e = hipMalloc(reinterpret_cast<void **>(&b->parts_dev), sizeof(PartDev) * n_parts);
    if (e == hipSuccess) e = hipMemcpy(b->parts_dev, pd.data(), sizeof(PartDev) * n_parts, hipMemcpyHostToDevice);
    if (e == hipSuccess) e = hipMalloc(reinterpret_cast<void **>(&b->cfg_dev), sizeof(PrlConfig));
    if (e == hipSuccess) e = hipMemcpy(b->cfg_dev, cfg, sizeof(PrlConfig), hipMemcpyHostToDevice);
    if (e == hipSuccess && env_part_id) {
        e = hipMalloc(reinterpret_cast<void **>(&b->env_part_dev), sizeof(int) * n_envs);
        if (e == hipSuccess) e = hipMemcpy(b->env_part_dev, env_part_id, sizeof(int) * n_envs, hipMemcpyHostToDevice);
    }
    if (e == hipSuccess) e = hipMalloc(reinterpret_cast<void **>(&b->painted), mask_bytes);
    if (e == hipSuccess) e = hipMalloc(reinterpret_cast<void **>(&b->last), mask_bytes);
    if (e == hipSuccess) e = hipMalloc(reinterpret_cast<void **>(&b->state), state_bytes);
    if (e == hipSuccess) e = hipMemset(b->painted, 0, mask_bytes);
    if (e == hipSuccess) e = hipMemset(b->last, 0, mask_bytes);
    if (e == hipSuccess) e = hipMemset(b->state, 0, state_bytes);
    if (e != hipSuccess) {
        rc = fail(PRL_E_HIP, "batch allocation: %s", hipGetErrorString(e));
        prl_batch_destroy(b);
        return rc;
    }
    *out = b;
    return PRL_OK;
}

void prl_batch_destroy(PrlBatch *b) {
    if (!b) return;
    (void)hipSetDevice(b->device);
    for (hipEvent_t ev : b->ev_start) (void)hipEventDestroy(ev);
    for (hipEvent_t ev : b->ev_stop) (void)hipEventDestroy(ev);
    (void)hipFree(b->parts_dev);
    (void)hipFree(b->cfg_dev);
    (void)hipFree(b->env_part_dev);
    (void)hipFree(b->painted);
    (void)hipFree(b->last);
    (void)hipFree(b->state);
    delete b;
}

int prl_batch_mask_stride(const PrlBatch *b) { return b ? b->mask_stride : fail(PRL_E_INVALID, "null batch"); }

int prl_batch_reset(PrlBatch *b, const uint8_t *reset_mask, const int32_t *start_idx, double *obs, void *stream) {
    if (!b) return fail(PRL_E_INVALID, "null batch");
    StepArgs a = base_args(b);
    a.reset_mask = reset_mask;
    a.start_idx = start_idx;
    a.obs = obs;
    hipStream_t s = static_cast<hipStream_t>(stream);
    switch (b->kw) {
    case 1: launch_reset<1>(a, general_section(b->cfg), s); break;
    case 2: launch_reset<2>(a, general_section(b->cfg), s); break;
    case 3: launch_reset<3>(a, general_section(b->cfg), s); break;
    default: launch_reset<4>(a, general_section(b->cfg), s); break;
    }
    HIP_TRY(hipGetLastError());
    return PRL_OK;
}

int prl_batch_observe(PrlBatch *b, double *obs, void *stream) {
    if (!b || !obs) return fail(PRL_E_INVALID, "null argument");
    StepArgs a = base_args(b);
    a.obs = obs;
    hipStream_t s = static_cast<hipStream_t>(stream);
    switch (b->kw) {
    case 1: launch_observe<1>(a, general_section(b->cfg), s); break;
    case 2: launch_observe<2>(a, general_section(b->cfg), s); break;
    case 3: launch_observe<3>(a, general_section(b->cfg), s); break;
    default: launch_observe<4>(a, general_section(b->cfg), s); break;
    }
    HIP_TRY(hipGetLastError());
    return PRL_OK;
}

int prl_batch_step(PrlBatch *b, const void *actions, double *obs, double *reward, uint8_t *done, double *info,
                   double *final_obs, const int32_t *start_idx, void *stream) {
    if (!b || !actions || !obs || !reward || !done || !info) return fail(PRL_E_INVALID, "null argument");
    StepArgs a = base_args(b);
    a.actions = actions;
    a.obs = obs;
    a.reward = reward;
    a.done = done;
    a.info = info;
    a.final_obs = final_obs;
    a.start_idx = start_idx;
    hipStream_t s = static_cast<hipStream_t>(stream);
    const bool normal = b->cfg.paint_method == PRL_PAINT_NORMAL;
    const bool timed = b->timing_every > 0 && (b->launch_no++ % b->timing_every) == 0;
    if (timed) {
        if (b->ev_used == b->ev_start.size()) {
            hipEvent_t e0, e1;
            HIP_TRY(hipEventCreate(&e0));
            HIP_TRY(hipEventCreate(&e1));
            b->ev_start.push_back(e0);
            b->ev_stop.push_back(e1);
        }
        HIP_TRY(hipEventRecord(b->ev_start[b->ev_used], s));
    }
    switch (b->kw) {
    case 1: launch_step<1>(a, normal, general_section(b->cfg), s); break;
    case 2: launch_step<2>(a, normal, general_section(b->cfg), s); break;
    case 3: launch_step<3>(a, normal, general_section(b->cfg), s); break;
    default: launch_step<4>(a, normal, general_section(b->cfg), s); break;
    }
    HIP_TRY(hipGetLastError());
    if (timed) {
        HIP_TRY(hipEventRecord(b->ev_stop[b->ev_used], s));
        b->ev_used += 1;
    }
    return PRL_OK;
}

int prl_batch_set_pose(PrlBatch *b, int env_index, const double *pos, const double *quat) {
    if (!b || !pos || !quat || env_index < 0 || env_index >= b->n_envs) return fail(PRL_E_INVALID, "bad argument");
    HIP_TRY(hipSetDevice(b->device));
    HIP_TRY(hipDeviceSynchronize());
    EnvState S;
    double *rec = b->state + (size_t)env_index * PRL_STATE_DOUBLES;
    HIP_TRY(hipMemcpy(&S, rec, sizeof S, hipMemcpyDeviceToHost));
    for (int k = 0; k < 3; ++k) S.pose[k] = pos[k];
    for (int k = 0; k < 4; ++k) S.quat[k] = quat[k];
    S.terminate = 0;
    S.terminate_counter = 0;
    S.last_on_part = 1;
    S.last_angle = 0;
    S.facet_hint = -1;
    HIP_TRY(hipMemcpy(rec, &S, sizeof S, hipMemcpyHostToDevice));
    return PRL_OK;
}

int prl_batch_get_mask(PrlBatch *b, uint64_t *painted, void *stream) {
    if (!b || !painted) return fail(PRL_E_INVALID, "null argument");
    const size_t n = (size_t)b->n_envs * b->mask_stride;
    hipLaunchKernelGGL(copy_mask_kernel, dim3((unsigned)((n + 255) / 256)), dim3(256), 0,
                       static_cast<hipStream_t>(stream), b->painted, painted, n);
    HIP_TRY(hipGetLastError());
    return PRL_OK;
}

int prl_batch_get_state(PrlBatch *b, double *state, void *stream) {
    if (!b || !state) return fail(PRL_E_INVALID, "null argument");
    HIP_TRY(hipMemcpyAsync(state, b->state, (size_t)b->n_envs * PRL_STATE_DOUBLES * sizeof(double),
                           hipMemcpyDeviceToDevice, static_cast<hipStream_t>(stream)));
    return PRL_OK;
}

int prl_batch_get_returns(PrlBatch *b, double *episode_return, void *stream) {
    if (!b || !episode_return) return fail(PRL_E_INVALID, "null argument");
    hipLaunchKernelGGL(gather_state_kernel, dim3((b->n_envs + 255) / 256), dim3(256), 0,
                       static_cast<hipStream_t>(stream), b->state, b->n_envs, 13, episode_return);
    HIP_TRY(hipGetLastError());
    return PRL_OK;
}

int prl_ray_batch(PrlPart *p, int n, const double *from, const double *to, int32_t *tri, double *frac, double *pos,
                  void *stream) {
    if (!p || n < 0 || !from || !to || !tri || !frac || !pos) return fail(PRL_E_INVALID, "bad argument");
    if (n == 0) return PRL_OK;
    hipLaunchKernelGGL(ray_batch_kernel, dim3((n + 3) / 4), dim3(256), 0, static_cast<hipStream_t>(stream),
                       p->dev_copy, n, from, to, tri, frac, pos);
    HIP_TRY(hipGetLastError());
    return PRL_OK;
}

#ifdef PRL_PHASE_TIMING
// diagnostic build only: read and clear the per-phase cycle sums
int prl_debug_phase_cycles(unsigned long long *out, int n) {
    unsigned long long host[16] = {0};
    if (hipMemcpyFromSymbol(host, HIP_SYMBOL(g_phase_cycles), sizeof host) != hipSuccess) return PRL_E_HIP;
    for (int k = 0; k < n && k < 16; ++k) out[k] = host[k];
    unsigned long long zero[16] = {0};
    if (hipMemcpyToSymbol(HIP_SYMBOL(g_phase_cycles), zero, sizeof zero) != hipSuccess) return PRL_E_HIP;
    return PRL_OK;
}
#endif

int prl_batch_timing_enable(PrlBatch *b, int enable) {
    if (!b) return fail(PRL_E_INVALID, "null batch");
    b->timing_every = enable < 0 ? 0 : enable;
    b->launch_no = 0;
    b->ev_used = 0;
    return PRL_OK;
}

int prl_batch_timing_read(PrlBatch *b, double *total_ms, int64_t *launches) {
    if (!b || !total_ms || !launches) return fail(PRL_E_INVALID, "null argument");
    double sum = 0;
    for (size_t i = 0; i < b->ev_used; ++i) {
        HIP_TRY(hipEventSynchronize(b->ev_stop[i]));
        float ms = 0;
        HIP_TRY(hipEventElapsedTime(&ms, b->ev_start[i], b->ev_stop[i]));
        sum += ms;
    }
    *total_ms = sum;
    *launches = (int64_t)b->ev_used;
    b->ev_used = 0;
    return PRL_OK;
}

}  // extern "C"
