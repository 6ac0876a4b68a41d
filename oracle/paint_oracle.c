/*
 * paint_oracle.c -- CPU restatement (float64, scalar, brute force) of PaintRL's
 * per-step paint-coverage simulator.
 *
 * TEST INFRASTRUCTURE.  Only tests/, __graft_entry__.smoke() and bench.py's
 * cpu_baseline leg may load this library, and only as the checker / the timed
 * CPU baseline.  The product (paintrl_amd/) never calls into it.
 *
 * Parity is PINNED: tests/test_oracle_golden.py replays the golden vectors in
 * the tests/golden fixtures (.npz), which were produced by importing the reference itself
 * (tests/golden/make_golden.py) -- observations, rewards, done flags and painted
 * texel sets must agree exactly.
 *
 * Every function cites the reference code it follows.  Abbreviations:
 *   bpw = PaintRLEnv/bullet_paint_wrapper.py,  rob = PaintRLEnv/robot.py,
 *   rge = PaintRLEnv/robot_gym_env.py.
 * Arithmetic notes that matter for bit-exactness with the reference:
 *   - numpy.dot on short float64 vectors (bpw:154-163, rob:269) is OpenBLAS ddot,
 *     which accumulates with fused multiply-adds: fma(a2,b2, fma(a1,b1, a0*b0)).
 *   - everything else is plain IEEE double, left to right; compile with
 *     -ffp-contract=off so the compiler adds no fusing of its own.
 *   - the ray test and the rigid transform are this project's definitions of
 *     pybullet.rayTestBatch / multiplyTransforms (Bullet is not vendored):
 *     paintrl_amd/geometry.py states them in numpy.
 */
#include <math.h>
#include <stdint.h>
#include <stdlib.h>
#include <string.h>
#ifdef _OPENMP
#include <omp.h>
#endif

#define PAINT_RADIUS 0.051          /* bpw:42 */
#define STEP_SIZE 0.051             /* bpw:43 */
#define HOOK_DISTANCE 0.1           /* bpw:443 */
#define GRID_GRANULARITY 100        /* bpw:447 */
#define PAINT_PER_ACTION 5          /* rob:165 */
#define NOT_ON_PART_TERMINATE 1000  /* rob:167 */
#define RAY_EPS_DET 1e-12
#define RAY_EPS_BARY 1e-9

enum { OBS_SECTION = 0, OBS_GRID = 1, OBS_SIMPLE = 2, OBS_DISCRETE = 3 };
enum { ACT_DISCRETE = 0, ACT_CONTINUOUS = 1 };
enum { TERM_LATE = 0, TERM_EARLY = 1, TERM_HYBRID = 2 };
enum { PAINT_FAST = 0, PAINT_NORMAL = 1 };
enum { COLOR_RGB = 0, COLOR_HSI = 1 };

typedef struct {
    /* samples, canonical order (ascending j*W+i) */
    int32_t n_samples;
    const double *sample_pos;      /* [P][3] */
    const int32_t *sample_cell;    /* [P] grid-observation cell */
    /* side vertices (compact) + CSR to compact front-triangle ids */
    int32_t n_vertices;
    const double *vertex_pos;      /* [V][3] (after the reference's in-place row mutation) */
    const int32_t *adj_off;        /* [V+1] */
    const int32_t *adj_tri;
    /* front triangle records */
    int32_t n_triangles;
    const double *tri_a, *tri_v0, *tri_v1;   /* [T][3] */
    const double *tri_d00, *tri_d01, *tri_d11, *tri_inv;
    const double *tri_normal;      /* [T][3] corrected normals */
    /* collision triangles */
    int32_t n_collision;
    const double *col_v0, *col_e1, *col_e2;  /* [C][3] */
    /* grid rows and extents */
    const double *grid_lo, *grid_hi;         /* [100] */
    double range1_min, range1_max, range2_min, range2_max, lwr;
    int32_t a0, a1, a2;
    /* start points */
    int32_t n_start;
    const double *start_pos;       /* [S][3] */
    const double *start_quat;      /* [S][4] xyzw, from rob:93-100 on the host */
    /* cone beams (PAINT_METHOD='normal') */
    int32_t n_beams;
    const double *beams;           /* [B][3] */
    /* the reference's STALE vertex kd-tree, only for parts whose vertex rows _set_grid_dict moved after the tree was
     * built (bpw:943-946; paintrl_amd/part_tables.py stale_kd_query); n_kd_nodes = 0: exact nearest vertex */
    int32_t n_kd_nodes;
    const int32_t *kd_split_dim;   /* [n] -1 = leaf */
    const double *kd_split;        /* [n] */
    const int32_t *kd_less, *kd_greater, *kd_start, *kd_end;   /* [n] children / leaf range into kd_points */
    const int32_t *kd_points;      /* compact side-vertex ids in scipy's tree order, -1 = a row of another side */
    const double *kd_box;          /* [2][3] bounding box of the rows when the tree was built */
    /* pixel_kd_tree.query(k = 1) (bpw:565) between samples at EQUAL distance -- samples that share one 3-D position, texels
     * of different triangles on one mesh vertex: query.cxx keeps the first of a leaf in the tree's index order (strict <);
     * sample_rank[s] = place of sample s in that order (paintrl_amd/part_tables.py _sample_tie_rank), lower wins */
    const int32_t *sample_rank;    /* [P] */
    /* the same for vertices_kd_tree[side].query(k = 1) (bpw:526): vertices written twice in the OBJ (UV seams) share a position
     * and differ in their incident triangles; vertex_rank[v] = place of compact side vertex v in the reference tree's index
     * order (part_tables.py _vertex_tie_rank), lower wins */
    const int32_t *vertex_rank;    /* [V] */
} OrPart;

typedef struct {
    int32_t obs_mode, obs_grad;
    int32_t action_mode, action_dim, n_discrete;
    int32_t termination_mode;
    int32_t turning_penalty, overlap_penalty;
    int32_t paint_method;
    int32_t max_episode_len, expected_episode_len;
    double switch_threshold, max_possible_point;
    double paint_radius, step_size;                      /* PaintToolProfile (bpw:40-43) */
    const double *act_delta1, *act_delta2, *act_angle;   /* [n_discrete] host table */
    int32_t color_mode;                                  /* 0 = 'RGB', 1 = 'HSI' (rge:156, bpw:384-434) */
    int32_t pad_;
} OrConfig;

typedef struct {
    double pose[3], quat[4];
    double last_turning_angle, angle_diff;
    double total_reward, total_return;
    int32_t terminate, terminate_counter, last_on_part, step_counter;
} OrEnv;

static int or_threads = 1;
void or_set_threads(int n) { or_threads = n > 0 ? n : 1; }

int or_obs_dim(const OrConfig *c) {
    switch (c->obs_mode) {
    case OBS_SECTION: return c->obs_grad + 2;
    case OBS_GRID: return c->obs_grad * c->obs_grad;
    case OBS_SIMPLE: return 2;
    default: return c->obs_grad + 1;
    }
}

int or_mask_words(const OrPart *p) { return (p->n_samples + 63) / 64; }
int or_env_size(void) { return (int)sizeof(OrEnv); }

/* numpy.dot of two 3-vectors (OpenBLAS ddot rounding) */
static inline double dot3_np(const double *a, const double *b) {
    return fma(a[2], b[2], fma(a[1], b[1], a[0] * b[0]));
}

/* this project's multiplyTransforms rotation: v + w*t + qv x t, t = 2*(qv x v) */
static void quat_rotate(const double *q, const double *v, double *o) {
    double t0 = 2.0 * (q[1] * v[2] - q[2] * v[1]);
    double t1 = 2.0 * (q[2] * v[0] - q[0] * v[2]);
    double t2 = 2.0 * (q[0] * v[1] - q[1] * v[0]);
    o[0] = (v[0] + q[3] * t0) + (q[1] * t2 - q[2] * t1);
    o[1] = (v[1] + q[3] * t1) + (q[2] * t0 - q[0] * t2);
    o[2] = (v[2] + q[3] * t2) + (q[0] * t1 - q[1] * t0);
}

static void transform_point(const double *pos, const double *q, const double *pt, double *o) {
    double r[3];
    quat_rotate(q, pt, r);
    o[0] = pos[0] + r[0]; o[1] = pos[1] + r[1]; o[2] = pos[2] + r[2];
}

/* rob:93-100 get_pose_orn + bpw:32-37 normalize */
static void pose_orn_quat(const double *orn, double *q) {
    double x = 0.0 * orn[2] - 1.0 * orn[1];
    double y = 1.0 * orn[0] - 0.0 * orn[2];
    double z = 0.0 * orn[1] - 0.0 * orn[0];
    double w = 1.0 + fma(1.0, orn[2], fma(0.0, orn[1], 0.0 * orn[0]));
    double mag2 = (((0.0 + x * x) + y * y) + z * z) + w * w;
    if (fabs(mag2 - 1.0) > 0.00001) {
        double mag = sqrt(mag2);
        x /= mag; y /= mag; z /= mag; w /= mag;
    }
    q[0] = x; q[1] = y; q[2] = z; q[3] = w;
}

/* exported so the Python wrapper builds start-point quaternions with the oracle's own arithmetic */
void or_pose_orn_quat(const double *orn, double *q) { pose_orn_quat(orn, q); }

/* rob:266-271 _get_tcp_orn_norm */
static void tcp_orn_norm(const double *pose, const double *quat, double *n) {
    static const double zaxis[3] = {0.0, 0.0, 1.0};
    double along[3], v[3];
    transform_point(pose, quat, zaxis, along);
    v[0] = along[0] - pose[0]; v[1] = along[1] - pose[1]; v[2] = along[2] - pose[2];
    double norm = sqrt(dot3_np(v, v));
    n[0] = v[0] / norm; n[1] = v[1] / norm; n[2] = v[2] / norm;
}

/* closest two-sided Moller-Trumbore hit of segment o->e; returns triangle or -1 */
static int ray_closest(const OrPart *p, const double *o, const double *e, double *t_out, double *hit) {
    double d[3] = {e[0] - o[0], e[1] - o[1], e[2] - o[2]};
    double best_t = INFINITY;
    int best = -1;
    for (int i = 0; i < p->n_collision; ++i) {
        const double *v0 = p->col_v0 + 3 * i, *e1 = p->col_e1 + 3 * i, *e2 = p->col_e2 + 3 * i;
        double p0 = d[1] * e2[2] - d[2] * e2[1];
        double p1 = d[2] * e2[0] - d[0] * e2[2];
        double p2 = d[0] * e2[1] - d[1] * e2[0];
        double det = (e1[0] * p0 + e1[1] * p1) + e1[2] * p2;
        if (!(fabs(det) >= RAY_EPS_DET)) continue;
        double inv = 1.0 / det;
        double s0 = o[0] - v0[0], s1 = o[1] - v0[1], s2 = o[2] - v0[2];
        double u = ((s0 * p0 + s1 * p1) + s2 * p2) * inv;
        double q0 = s1 * e1[2] - s2 * e1[1];
        double q1 = s2 * e1[0] - s0 * e1[2];
        double q2 = s0 * e1[1] - s1 * e1[0];
        double v = ((d[0] * q0 + d[1] * q1) + d[2] * q2) * inv;
        double t = ((e2[0] * q0 + e2[1] * q1) + e2[2] * q2) * inv;
        if (u >= -RAY_EPS_BARY && v >= -RAY_EPS_BARY && (u + v) <= 1.0 + RAY_EPS_BARY && t >= 0.0 && t <= 1.0) {
            if (t < best_t) { best_t = t; best = i; }
        }
    }
    if (best >= 0) {
        hit[0] = o[0] + best_t * d[0];
        hit[1] = o[1] + best_t * d[1];
        hit[2] = o[2] + best_t * d[2];
    }
    *t_out = best_t;
    return best;
}

void or_ray_batch(const OrPart *p, int n, const double *from, const double *to, int32_t *idx, double *t, double *pos) {
    for (int i = 0; i < n; ++i) {
        double h[3] = {0, 0, 0};
        idx[i] = ray_closest(p, from + 3 * i, to + 3 * i, t + i, h);
        pos[3 * i] = h[0]; pos[3 * i + 1] = h[1]; pos[3 * i + 2] = h[2];
    }
}

/* bpw:154-163 _get_bary_coordinate */
static void bary_coord(const OrPart *p, int ti, const double *pt, double *u, double *v, double *w) {
    const double *a = p->tri_a + 3 * ti;
    double v2[3] = {pt[0] - a[0], pt[1] - a[1], pt[2] - a[2]};
    double d20 = dot3_np(v2, p->tri_v0 + 3 * ti);
    double d21 = dot3_np(v2, p->tri_v1 + 3 * ti);
    double inv = p->tri_inv[ti];
    *v = (p->tri_d11[ti] * d20 - p->tri_d01[ti] * d21) * inv;
    *w = (p->tri_d00[ti] * d21 - p->tri_d01[ti] * d20) * inv;
    *u = 1.0 - *v - *w;
    if (inv == 0) { *u = -1; *v = -1; *w = -1; }
}

/* scipy cKDTree.query(k=1) (ckdtree/src/query.cxx, p = 2, eps = 0) on the stale tree, leaf distances from the current
 * (moved) rows: descend to the near child, queue a far child whose lower bound is <= the best distance, scan a leaf in
 * tree order keeping strictly smaller distances, stop when the queue is empty or its nearest cell is beyond the best */
typedef struct { double mind, side[3]; int node; } KdItem;
static int stale_kd_query(const OrPart *p, const double *x) {
    KdItem heap[128];
    int n_heap = 0, best = -1;
    double dub = INFINITY;
    KdItem cur;
    cur.node = 0;
    for (int k = 0; k < 3; ++k) {
        double a = x[k] - p->kd_box[3 + k], b = p->kd_box[k] - x[k];
        double s = a > b ? a : b;
        s = s > 0 ? s : 0;
        cur.side[k] = s * s;
    }
    cur.mind = (cur.side[0] + cur.side[1]) + cur.side[2];
    for (;;) {
        int sd = p->kd_split_dim[cur.node];
        if (sd < 0) {
            for (int i = p->kd_start[cur.node]; i < p->kd_end[cur.node]; ++i) {
                int v = p->kd_points[i];
                if (v < 0) continue;                       /* a row parked at (10, 10, 10): never the nearest */
                const double *q = p->vertex_pos + 3 * v;
                double d0 = q[0] - x[0], d1 = q[1] - x[1], d2 = q[2] - x[2];
                double d = (d0 * d0 + d1 * d1) + d2 * d2;
                if (d < dub) { dub = d; best = v; }
            }
            if (n_heap == 0) break;
            int m = 0;                                     /* pop the nearest queued cell */
            for (int i = 1; i < n_heap; ++i) if (heap[i].mind < heap[m].mind) m = i;
            cur = heap[m];
            heap[m] = heap[--n_heap];
        } else {
            if (cur.mind > dub) break;
            double sp = p->kd_split[cur.node];
            KdItem near = cur, far = cur;
            if (x[sd] < sp) { near.node = p->kd_less[cur.node]; far.node = p->kd_greater[cur.node]; }
            else { near.node = p->kd_greater[cur.node]; far.node = p->kd_less[cur.node]; }
            double tmp = sp - x[sd], nw = tmp * tmp;
            far.mind = cur.mind + (nw - far.side[sd]);
            far.side[sd] = nw;
            if (near.mind > far.mind) { KdItem t = near; near = far; far = t; }
            if (far.mind <= dub && n_heap < 128) heap[n_heap++] = far;
            cur = near;
        }
    }
    return best;
}

/* test hook: the vertex bpw:526 would return for `x` (compact side-vertex id) */
int or_nearest_vertex(const OrPart *p, const double *x) {
    if (p->n_kd_nodes > 0) return stale_kd_query(p, x);
    int best_v = -1;
    double best_d = INFINITY;
    for (int i = 0; i < p->n_vertices; ++i) {
        const double *q = p->vertex_pos + 3 * i;
        double dx = q[0] - x[0], dy = q[1] - x[1], dz = q[2] - x[2];
        double d2 = (dx * dx + dy * dy) + dz * dz;
        if (d2 < best_d || (best_v >= 0 && d2 == best_d && p->vertex_rank[i] < p->vertex_rank[best_v])) { best_d = d2; best_v = i; }
    }
    return best_v;
}

/* bpw:525-534 _get_hook_point (+ 508-523 _get_closest_bary); returns 0 if no triangle */
static int hook_point(const OrPart *p, const double *pt, double *pose, double *orn) {
    int best_v = 0;
    double best_d = INFINITY;
    if (p->n_kd_nodes > 0) {                               /* rows were moved under the reference's tree */
        best_v = stale_kd_query(p, pt);
        if (best_v < 0) return 0;
    } else
    for (int i = 0; i < p->n_vertices; ++i) {
        const double *x = p->vertex_pos + 3 * i;
        double dx = x[0] - pt[0], dy = x[1] - pt[1], dz = x[2] - pt[2];
        double d2 = (dx * dx + dy * dy) + dz * dz;
        if (d2 < best_d || (d2 == best_d && p->vertex_rank[i] < p->vertex_rank[best_v])) { best_d = d2; best_v = i; }
    }
    int closest = -1;
    double closest_uvw = -1;
    for (int k = p->adj_off[best_v]; k < p->adj_off[best_v + 1]; ++k) {
        int ti = p->adj_tri[k];
        double u, v, w;
        bary_coord(p, ti, pt, &u, &v, &w);
        if (0 <= u && u <= 1 && 0 <= v && v <= 1 && 0 <= w && w <= 1) { closest = ti; break; }
        if (closest < 0) closest = ti;
        double m = v < u ? v : u;           /* python min(): keeps the first unless a later one is smaller */
        m = w < m ? w : m;
        if (m >= closest_uvw) { closest_uvw = m; closest = ti; }
    }
    if (closest < 0) return 0;
    const double *n = p->tri_normal + 3 * closest;
    for (int k = 0; k < 3; ++k) {
        pose[k] = pt[k] + n[k] * HOOK_DISTANCE;
        orn[k] = -n[k];
    }
    return 1;
}

/* bpw:865-880 get_guided_point; returns 1 on hit (pos/orn set), 0 on miss (orn = normal) */
static int guided_point(const OrPart *p, const double *pose, const double *normal, double d1, double d2,
                        double *pos, double *orn) {
    double pt[3] = {pose[0], pose[1], pose[2]};
    double delta_2 = d2 * p->lwr;
    pt[p->a1] += d1;
    pt[p->a2] += delta_2;
    double end[3] = {pt[0] + normal[0], pt[1] + normal[1], pt[2] + normal[2]};
    double t, hit[3];
    if (ray_closest(p, pt, end, &t, hit) < 0 || !hook_point(p, hit, pos, orn)) {
        orn[0] = normal[0]; orn[1] = normal[1]; orn[2] = normal[2];
        return 0;
    }
    return 1;
}

/* rob:292-300 */
static void count_not_on_part(OrEnv *e) {
    if (e->last_on_part) { e->last_on_part = 0; return; }
    e->terminate_counter += 1;
    e->last_on_part = 0;
    if (e->terminate_counter > NOT_ON_PART_TERMINATE) e->terminate = 1;
}

static inline int popcount64(uint64_t x) { return __builtin_popcountll(x); }

/* bpw:572-577 _paint on a hit mask `cur` (affected set); returns newly painted count */
static int apply_paint(int words, uint64_t *painted, uint64_t *last, const uint64_t *cur, uint64_t *uni) {
    int succeeded = 0;
    for (int w = 0; w < words; ++w) {
        succeeded += popcount64(cur[w] & ~painted[w]);
        painted[w] |= cur[w];
        uni[w] |= cur[w] & ~last[w];          /* valid = affected \ last shot */
        last[w] = cur[w];
    }
    return succeeded;
}

/* COLOR_MODE = 'HSI' (bpw:384-434 HSIColorHandler.change_pixels + bpw:572-577 _paint), as the reference behaves:
 * every front texel carries a uint8 (255 after the label pass, bpw:586); a shot subtracts from each hit texel
 *     quantity = int(TARGET_MAX * (1 - (d / r)**2) ** (BETA - 1)) + 1,   TARGET_MAX = 25, BETA = 2,
 * d = distance of the sample to the shot centre (scipy minkowski_distance: sqrt((dx^2 + dy^2) + dz^2)), r = the
 * largest d of the shot, unless the byte is already 0 (is_changed, bpw:392-394); the subtraction is numpy uint8
 * arithmetic and wraps below zero (the "bugs here" of bpw:393).  The shot's "succeed counter" is the float sum of
 * quantity / 255 over the texels it changed (summed here in ascending sample order: the reference sums in cKDTree
 * traversal order, so the last bits of rewards are not pinned -- tests compare them to 1e-12).  What the
 * observation and get_job_status call "painted" stays RGBColorHandler.is_changed, byte == 255 (bpw:723-725): every
 * texel reads painted until its first deposit.  `painted` holds that status bit, `thick` the bytes.
 * A shot that hits no sample raises ValueError in the reference (max of an empty array); here it deposits nothing. */
static double apply_paint_hsi(const OrPart *p, int words, uint64_t *painted, uint64_t *last, const uint64_t *cur,
                              uint64_t *uni, uint8_t *thick, const double *c) {
    double r = -1.0, succeeded = 0.0;
    for (int s = 0; s < p->n_samples; ++s)
        if ((cur[s >> 6] >> (s & 63)) & 1) {
            const double *x = p->sample_pos + 3 * s;
            double dx = c[0] - x[0], dy = c[1] - x[1], dz = c[2] - x[2];
            double d = sqrt((dx * dx + dy * dy) + dz * dz);
            if (d > r) r = d;
        }
    for (int s = 0; s < p->n_samples; ++s)
        if ((cur[s >> 6] >> (s & 63)) & 1) {
            const double *x = p->sample_pos + 3 * s;
            double dx = c[0] - x[0], dy = c[1] - x[1], dz = c[2] - x[2];
            double d = sqrt((dx * dx + dy * dy) + dz * dz);
            double q = d / r;
            int quantity = (int)(25 * (1 - q * q)) + 1;
            if (thick[s] != 0) {
                thick[s] = (uint8_t)(thick[s] - quantity);
                succeeded += quantity / 255.0;
            }
            if (thick[s] == 255) painted[s >> 6] |= (uint64_t)1 << (s & 63);
            else painted[s >> 6] &= ~((uint64_t)1 << (s & 63));
        }
    for (int w = 0; w < words; ++w) {
        uni[w] |= cur[w] & ~last[w];
        last[w] = cur[w];
    }
    return succeeded;
}

/* bpw:568-570 fast_paint: all samples with |x - c|^2 <= r^2 */
static void ball_query(const OrPart *p, double radius, const double *c, uint64_t *cur) {
    const double r2 = radius * radius;
    for (int s = 0; s < p->n_samples; ++s) {
        const double *x = p->sample_pos + 3 * s;
        double dx = x[0] - c[0], dy = x[1] - c[1], dz = x[2] - c[2];
        double d2 = (dx * dx + dy * dy) + dz * dz;
        if (d2 <= r2) cur[s >> 6] |= (uint64_t)1 << (s & 63);
    }
}

/* rob:280-285 _paint + bpw:562-566 paint: cone beams -> nearest sample per hit, in beam order (`list`, one entry per
 * beam that hits: a sample under several beams appears several times, as in pixel_kd_tree.query(points, k=1)[1]).
 * returns number of beam hits (0 => the reference returns early without touching last-shot state) */
static int cone_query(const OrPart *p, const double *pose, const double *quat, uint64_t *cur, int *list) {
    int hits = 0;
    for (int b = 0; b < p->n_beams; ++b) {
        double dst[3], t, hit[3];
        transform_point(pose, quat, p->beams + 3 * b, dst);
        if (ray_closest(p, pose, dst, &t, hit) < 0) continue;
        int best = 0;
        double best_d = INFINITY;
        for (int s = 0; s < p->n_samples; ++s) {
            const double *x = p->sample_pos + 3 * s;
            double dx = x[0] - hit[0], dy = x[1] - hit[1], dz = x[2] - hit[2];
            double d2 = (dx * dx + dy * dy) + dz * dz;
            if (d2 < best_d || (d2 == best_d && p->sample_rank[s] < p->sample_rank[best])) { best_d = d2; best = s; }
        }
        cur[best >> 6] |= (uint64_t)1 << (best & 63);
        if (list) list[hits] = best;
        ++hits;
    }
    return hits;
}

/* COLOR_MODE = 'HSI' under the cone beams (bpw:562-566 paint -> bpw:419-434 change_pixels with the list of nearest
 * samples, duplicates and all): r = the largest distance of a listed sample to the shot centre; then IN LIST ORDER every
 * entry deposits quantity = int(25 (1 - (d / r)^2)) + 1 on its sample unless the byte is 0 at that moment (uint8
 * arithmetic, wrapping) and adds quantity / 255 to the succeed counter -- a sample under k beams receives k deposits.
 * The float sum runs in beam order here, the reference's own order. */
static double apply_paint_hsi_list(const OrPart *p, int words, uint64_t *painted, uint64_t *last, const uint64_t *cur,
                                   uint64_t *uni, uint8_t *thick, const double *c, const int *list, int n) {
    double r = -1.0, succeeded = 0.0;
    for (int k = 0; k < n; ++k) {
        const double *x = p->sample_pos + 3 * list[k];
        double dx = c[0] - x[0], dy = c[1] - x[1], dz = c[2] - x[2];
        double d = sqrt((dx * dx + dy * dy) + dz * dz);
        if (d > r) r = d;
    }
    for (int k = 0; k < n; ++k) {
        const int s = list[k];
        const double *x = p->sample_pos + 3 * s;
        double dx = c[0] - x[0], dy = c[1] - x[1], dz = c[2] - x[2];
        double d = sqrt((dx * dx + dy * dy) + dz * dz);
        double q = d / r;
        int quantity = (int)(25 * (1 - q * q)) + 1;
        if (thick[s] != 0) {
            thick[s] = (uint8_t)(thick[s] - quantity);
            succeeded += quantity / 255.0;
        }
    }
    for (int k = 0; k < n; ++k) {
        const int s = list[k];
        if (thick[s] == 255) painted[s >> 6] |= (uint64_t)1 << (s & 63);
        else painted[s >> 6] &= ~((uint64_t)1 << (s & 63));
    }
    for (int w = 0; w < words; ++w) {
        uni[w] |= cur[w] & ~last[w];
        last[w] = cur[w];
    }
    return succeeded;
}

/* bpw:844-851 */
static int grid_index_2(const OrPart *p, double val) {
    double rel = (val - p->range2_min) / (p->range2_max - p->range2_min);
    double g = rel * GRID_GRANULARITY;
    int gi = (int)g;
    if (!(g > -2147483648.0 && g < 2147483648.0)) gi = g > 0 ? GRID_GRANULARITY - 1 : 0;
    if (gi < 0) return 0;
    if (gi > GRID_GRANULARITY - 1) return GRID_GRANULARITY - 1;
    return gi;
}

static double clip01(double v) { return v < 0 ? 0.0 : (v > 1 ? 1.0 : v); }

/* bpw:965-978 get_normalized_pose */
static void normalized_pose(const OrPart *p, double radius, const double *pose, double *out) {
    const double r = radius;
    double x1 = pose[p->a1], x2 = pose[p->a2];
    double in2 = (x2 - p->range2_min + r) / (p->range2_max - p->range2_min + 2 * r);
    int gi = grid_index_2(p, x2);
    double lo = p->grid_lo[gi], hi = p->grid_hi[gi];
    double in1;
    if (hi - lo == 0) in1 = 0;
    else in1 = (x1 - lo + r) / (hi - lo + 2 * r);
    out[0] = clip01(in1);
    out[1] = clip01(in2);
}

/* rge:92-103 */
static int handle_pos(double v) {
    if (v == 0) return 0;
    if (v == 1) return 21;
    return (int)(v * 20) + 1;
}

/* CPython float_floor_div (Objects/floatobject.c): the `//` of bpw:1030 */
static double py_floor_div(double vx, double wx) {
    double mod = fmod(vx, wx);
    double div = (vx - mod) / wx;
    if (mod) {
        if ((wx < 0) != (mod < 0)) div -= 1.0;
    }
    if (div) {
        double fl = floor(div);
        if (div - fl > 0.5) fl += 1.0;
        return fl;
    }
    return copysign(0.0, vx / wx);
}

/* rge:306-319 _augmented_observation (+ bpw:1045-1061, 1126-1139) */
static void observation(const OrPart *p, const OrConfig *c, const OrEnv *e, const uint64_t *painted, double *obs) {
    double npose[2];
    normalized_pose(p, c->paint_radius, e->pose, npose);
    if (c->obs_mode == OBS_SIMPLE) { obs[0] = npose[0]; obs[1] = npose[1]; return; }
    if (c->obs_mode == OBS_GRID) {
        int h = c->obs_grad;
        int cells = h * h;
        int num[1024], done[1024];
        memset(num, 0, sizeof(int) * cells);
        memset(done, 0, sizeof(int) * cells);
        for (int s = 0; s < p->n_samples; ++s) {
            int cl = p->sample_cell[s];
            num[cl] += 1;
            done[cl] += (int)((painted[s >> 6] >> (s & 63)) & 1);
        }
        for (int k = 0; k < cells; ++k) obs[k] = num[k] == 0 ? 0.0 : 1.0 - (double)done[k] / (double)num[k];
        return;
    }
    /* section (4-sector rule, bpw:1034-1043) and discrete */
    int g = c->obs_grad;
    long total[64], undone[64];
    memset(total, 0, sizeof(long) * g);
    memset(undone, 0, sizeof(long) * g);
    double px = e->pose[p->a1], py = e->pose[p->a2];
    const double basis = 2 * M_PI / g;
    for (int s = 0; s < p->n_samples; ++s) {
        const double *x = p->sample_pos + 3 * s;
        double rx = x[p->a1] - px, ry = x[p->a2] - py;
        if (rx == 0 && ry == 0) continue;
        int idx;
        if (g == 4) {
            if (rx > 0 && ry > 0) idx = 0;
            else if (rx < 0 && 0 < ry) idx = 1;
            else if (rx < 0 && ry < 0) idx = 2;
            else idx = 3;
        } else {                                   /* bpw:1026-1031 */
            double ang = atan2(ry, rx);
            if (ang < 0) ang = 2 * M_PI + ang;
            idx = (int)py_floor_div(ang, basis);
            if (idx > g - 1) idx = g - 1;       /* the reference raises IndexError here (angle rounds to 2*pi) */
        }
        total[idx] += 1;
        undone[idx] += 1 - (int)((painted[s >> 6] >> (s & 63)) & 1);
    }
    for (int k = 0; k < g; ++k) obs[k] = total[k] == 0 ? 0.0 : (double)undone[k] / (double)total[k];
    if (c->obs_mode == OBS_SECTION) { obs[g] = npose[0]; obs[g + 1] = npose[1]; }
    else {
        int position = (handle_pos(npose[0]) + 1) * 22 + handle_pos(npose[1]);
        obs[g] = 1.0 / (double)position;
    }
}

/* rge:370-387 reset + rob:366-372 + bpw:706-712 */
void or_reset(const OrPart *p, const OrConfig *c, OrEnv *env, uint64_t *painted, uint64_t *last,
              int n, const uint8_t *mask, const int32_t *start_idx, double *obs, uint8_t *thick) {
    int words = or_mask_words(p), od = or_obs_dim(c);
    for (int i = 0; i < n; ++i) {
        if (mask && !mask[i]) continue;
        OrEnv *e = env + i;
        int si = start_idx[i];
        memset(painted + (size_t)i * words, 0, sizeof(uint64_t) * words);
        memset(last + (size_t)i * words, 0, sizeof(uint64_t) * words);
        if (c->color_mode == COLOR_HSI) {               /* bpw:586 front label (1,1,1): every byte 255 = "painted" */
            memset(thick + (size_t)i * p->n_samples, 255, (size_t)p->n_samples);
            for (int s = 0; s < p->n_samples; ++s) painted[(size_t)i * words + (s >> 6)] |= (uint64_t)1 << (s & 63);
        }
        memcpy(e->pose, p->start_pos + 3 * si, sizeof(double) * 3);
        memcpy(e->quat, p->start_quat + 4 * si, sizeof(double) * 4);
        e->terminate = 0; e->terminate_counter = 0; e->last_on_part = 1; e->last_turning_angle = 0;
        e->step_counter = 0; e->total_reward = 0; e->total_return = 0;
        if (obs) observation(p, c, e, painted + (size_t)i * words, obs + (size_t)i * od);
    }
}

/* rge:306-319 _augmented_observation of the current state (test hook: poses set from outside) */
void or_observe(const OrPart *p, const OrConfig *c, const OrEnv *env, const uint64_t *painted, int n, double *obs) {
    int words = or_mask_words(p), od = or_obs_dim(c);
    for (int i = 0; i < n; ++i) observation(p, c, env + i, painted + (size_t)i * words, obs + (size_t)i * od);
}

/* rob:151-160 direction_normalize for continuous actions (libm; tolerance-level parity) */
static void direction(const OrConfig *c, const double *a, double *x, double *y) {
    if (c->action_dim == 1) {
        double phi = (a[0] + 1) * M_PI;
        *x = 1 * cos(phi); *y = 1 * sin(phi);
        return;
    }
    double phi = atan2(a[1], a[0]);
    double ax = fabs(a[0]), ay = fabs(a[1]);
    if (ax == 0 && ay == 0) { *x = ax; *y = ay; return; }
    double m = ax > ay ? ax : ay;
    *x = m * cos(phi); *y = m * sin(phi);
}

/* rge:349-368 step, rob:383-433 apply_action, rob:302-329 _get_actions */
void or_step(const OrPart *p, const OrConfig *c, OrEnv *env, uint64_t *painted_all, uint64_t *last_all,
             int n, const void *actions, double *obs, double *reward, uint8_t *done, double *info, uint8_t *thick_all) {
    int words = or_mask_words(p), od = or_obs_dim(c);
#ifdef _OPENMP
#pragma omp parallel for schedule(dynamic, 8) num_threads(or_threads)
#endif
    for (int i = 0; i < n; ++i) {
        uint64_t cur[words], uni[words];
        OrEnv *e = env + i;
        uint64_t *painted = painted_all + (size_t)i * words, *last = last_all + (size_t)i * words;
        double delta1, delta2, new_angle;
        if (c->action_mode == ACT_DISCRETE) {                   /* rge:342-347 via the host table */
            int a = ((const int32_t *)actions)[i];
            delta1 = c->act_delta1[a]; delta2 = c->act_delta2[a]; new_angle = c->act_angle[a];
        } else {
            double a[2] = {0, 0}, dx, dy;
            for (int k = 0; k < c->action_dim; ++k) {
                double v = ((const double *)actions)[(size_t)i * c->action_dim + k];
                if (!(-1 <= v && v <= 1)) v = v < -1 ? -1 : (v > 1 ? 1 : v);
                a[k] = v;
            }
            direction(c, a, &dx, &dy);
            delta1 = dx * c->step_size; delta2 = dy * c->step_size;
            new_angle = delta1 != 0 ? atan(fabs(delta2 / delta1)) : M_PI / 2;
        }
        e->angle_diff = fabs(new_angle - e->last_turning_angle);   /* rob:352-358 */
        e->last_turning_angle = new_angle;
        int counter_before = e->terminate_counter;

        double cur_pose[3] = {e->pose[0], e->pose[1], e->pose[2]}, cur_norm[3];
        tcp_orn_norm(e->pose, e->quat, cur_norm);
        double d1 = delta1 / PAINT_PER_ACTION, d2 = delta2 / PAINT_PER_ACTION;
        memset(uni, 0, sizeof(uint64_t) * words);
        double succeeded = 0;                                   /* an integer count in RGB mode, a float sum in HSI mode */
        for (int k = 0; k < PAINT_PER_ACTION; ++k) {
            double pos[3], orn[3], quat[4];
            int on = guided_point(p, cur_pose, cur_norm, d1, d2, pos, orn);
            pose_orn_quat(orn, quat);
            if (!on) {
                double mv[3] = {d2, d1, 0.0};
                transform_point(cur_pose, quat, mv, pos);
                count_not_on_part(e);
            } else {
                e->last_on_part = 1;
            }
            memcpy(cur_pose, pos, sizeof pos);
            memcpy(cur_norm, orn, sizeof orn);
            /* paint at this sub-pose (rob:403-424; order-independent of the pose chain) */
            memcpy(e->pose, pos, sizeof pos);
            memcpy(e->quat, quat, sizeof quat);
            memset(cur, 0, sizeof(uint64_t) * words);
            if (c->paint_method == PAINT_FAST) {
                static const double tip[3] = {0.0, 0.0, 0.1};
                double center[3];
                transform_point(pos, quat, tip, center);        /* rob:277-278 */
                ball_query(p, c->paint_radius, center, cur);
                if (c->color_mode == COLOR_HSI)
                    succeeded += apply_paint_hsi(p, words, painted, last, cur, uni, thick_all + (size_t)i * p->n_samples, center);
                else
                    succeeded += apply_paint(words, painted, last, cur, uni);
            } else {
                int *list = c->color_mode == COLOR_HSI ? (int *)malloc(sizeof(int) * (size_t)(p->n_beams > 0 ? p->n_beams : 1)) : NULL;
                const int hits = cone_query(p, pos, quat, cur, list);
                if (hits > 0) {
                    if (c->color_mode == COLOR_HSI) {
                        static const double tip[3] = {0.0, 0.0, 0.1};
                        double center[3];
                        transform_point(pos, quat, tip, center);    /* rob:277-278, 285 */
                        succeeded += apply_paint_hsi_list(p, words, painted, last, cur, uni, thick_all + (size_t)i * p->n_samples,
                                                          center, list, hits);
                    } else {
                        succeeded += apply_paint(words, painted, last, cur, uni);
                    }
                }
                free(list);
            }
        }
        int pixel_counter = 0;
        for (int w = 0; w < words; ++w) pixel_counter += popcount64(uni[w]);
        double rate = pixel_counter ? succeeded / (double)pixel_counter : 0.0;
        if (e->terminate_counter - counter_before >= PAINT_PER_ACTION && pixel_counter == 0) e->terminate = 1;

        double rew = succeeded / 100;                           /* rge:321-325 */
        e->total_reward += rew;
        double pen = 0.2;                                       /* rge:327-340 */
        if (c->overlap_penalty) pen += 0.1 * (1 - rate);
        if (c->turning_penalty) pen += 0.1 * (e->angle_diff / M_PI);
        double actual = rew - pen;
        /* rge:289-304 _termination */
        e->step_counter += 1;
        int finished = c->max_possible_point > e->total_reward * 100 ? 0 : 1;
        double avg = e->total_reward / e->step_counter;
        double expected = c->max_possible_point / (c->expected_episode_len * 100);
        int dn;
        if (avg < expected && c->termination_mode != TERM_LATE &&
            (c->termination_mode == TERM_EARLY ||
             e->total_reward < c->switch_threshold * c->max_possible_point / 100))
            dn = 1;
        else
            dn = finished || e->terminate || e->step_counter > c->max_episode_len - 1;
        observation(p, c, e, painted, obs + (size_t)i * od);
        if (!dn) e->total_return += actual;
        reward[i] = actual;
        done[i] = (uint8_t)dn;
        info[2 * i] = rew;
        info[2 * i + 1] = pen;
    }
}
