// prl_cone.hpp -- PAINT_METHOD 'normal': the cone beams of one shot, one beam per lane (rob:251-285, bpw:562-566).
// Device code of the cone-beam kernels (k_cone_beams.hip describes the five launches of a step); anonymous namespace.
// Compile with -ffp-contract=off.
//
// A shot casts n_beams (104-140 on the reference's parts, 450 with COLOR_MODE 'HSI') rays from the tool position to the
// points of a table on the plane 0.2 ahead (rob:23-35, 38-69); every ray that hits the part paints the sample nearest to its
// hit point.  A wave takes 64 beams at a time (a "trip"), one per lane:
//   * rays: on a convex collision set (hull mode) a lane WALKS the hull surface (cone_walk_lanes): it tests its ray against
//     one facet -- starting from the facet the front-facet grid names for the middle of the beam --, and while the ray meets
//     that facet's plane outside the triangle it steps across the violated edge to the neighbouring facet (three edge
//     neighbours per facet, derived on upload).  A facet that is ENTERED at a point clear of its edges ends the walk with the
//     exact closest hit of the whole set -- the criterion of ray_closest_wave's single-facet path, same arithmetic.  A walk
//     that reaches the hull's horizon ends with a PROVEN miss (cone_walk_step: both ends beyond a facet plane, or the
//     silhouette-edge certificate).  What is left over (edge and vertex hits, grazing entries, walks that run out of steps)
//     is searched by the general code: cone_rays_lanes, for the rest kernel.
//   * nearest samples: one hit point per lane on the fine sample grid -- nearest_sample_lane_f32 (the 2 x 2 cells around the
//     point, float records, the contenders confirmed in float64) in the beams kernel, nearest_sample_lane (float64 records,
//     rings of cells) in the general code; hit points these do not settle go down the box pyramid, a few lanes each
//     (nearest_sample_bfs): exact at any distance.
#pragma once

namespace {

// -DPRL_CONE_TRACE (tools/cone_stats.py): how many beams take which path, summed over a run (global counters, read back
// through prl_debug_cone_stats of the k_cone_beams unit).  The product build defines none of this.
#ifdef PRL_CONE_TRACE
__device__ unsigned long long g_cone_stat[32];
#if PRL_CONE_TRACE >= 2                          // the wave times alone (the counters' atomics stretch them tenfold)
#define CONE_STAT(k, v)
#else
#define CONE_STAT(k, v)                                                              \
    do {                                                                             \
        const unsigned long long v_ = (unsigned long long)(v);                       \
        if ((threadIdx.x & 63) == 0 && v_) atomicAdd(&g_cone_stat[k], v_);           \
    } while (0)
#endif
// wave times (s_memrealtime ticks of 10 ns) as a histogram: slot k / 3 (0 far-list, 1 trip-list waves of the rest
// kernel, 2 beams kernel), bucket = floor(log2(ticks)); ONE atomic per wave
__device__ unsigned long long g_cone_hist[5 * 32];
__device__ unsigned g_far_trace[4 * 16384];         // PRL_CONE_TRACE == 4: per wave of the far kernel, ticks of its phases (plain stores)
__device__ unsigned g_far_stamp[4];
#if PRL_CONE_TRACE == 4                          // (the far kernel's per-wave trace alone: no atomics anywhere)
#define CONE_TIME_BEGIN() \
    do {                  \
    } while (0)
#define CONE_TIME_END(k) \
    do {                 \
    } while (0)
#define CONE_HIST(k, bucket)
#else
#define CONE_TIME_BEGIN() const unsigned long long cone_t0_ = __builtin_amdgcn_s_memrealtime()
#define CONE_TIME_END(k)                                                             \
    do {                                                                             \
        const unsigned long long dt_ = __builtin_amdgcn_s_memrealtime() - cone_t0_;  \
        if ((threadIdx.x & 63) == 0) atomicAdd(&g_cone_hist[(k) * 32 + (63 - __builtin_clzll(dt_ | 1))], 1ull); \
    } while (0)
#define CONE_HIST(k, bucket)                                                         \
    do {                                                                             \
        const int b_ = (bucket);                                                     \
        if ((threadIdx.x & 63) == 0) atomicAdd(&g_cone_hist[(k) * 32 + (b_ < 31 ? b_ : 31)], 1ull); \
    } while (0)
#endif
#else
#define CONE_STAT(k, v)
#define CONE_HIST(k, bucket)
#define CONE_TIME_BEGIN() \
    do {                  \
    } while (0)
#define CONE_TIME_END(k) \
    do {                 \
    } while (0)
#endif

#ifndef PRL_WALK_STEPS
#define PRL_WALK_STEPS 12
#endif
constexpr int CONE_WALK_STEPS = PRL_WALK_STEPS;
#ifndef PRL_CONE_JOINT_FROM
#define PRL_CONE_JOINT_FROM 5
#endif
constexpr int CONE_JOINT_FROM = PRL_CONE_JOINT_FROM;
// (CONE_MISS_MARGIN: prl_ray.hpp)

// A beam ENTERS a facet for the purposes of the walk from this squared cosine on.  The single-facet criterion (prl_ray.hpp)
// needs the entry point FACET_EDGE_MARGIN = 1e-6 m clear of the facet's edges and the other facets' tolerance fringes (1e-9 of
// a triangle height: under 1e-9 m) out of the beam's way before it: a beam that comes down on the facet at |cos| c passes
// c x 1e-6 m above the facet's plane where it crosses the edge, so any c well above 1e-3 will do; 0.032 here (the tool's
// own rays keep prl_ray.hpp's 0.1).
#define CONE_MIN_COS2 1.0e-3
#define CONE_SIL_MIN_COS2 1.0e-6     // squared cosine between beam and the normal of a facet counted as facing away from it

// One step of the walk for this lane's ray (origin o, direction d, |d|^2 = dd) on facet i (>= 0), reached across an
// edge of facet `prev` (-1: the walk starts here): Moller-Trumbore on the facet record, arithmetic as in mt_rec.
// Returns 1 = entered at an interior point (t, exact closest hit), 2 = proven miss, 0 = keep walking (`next` = the
// facet across the most violated edge, or -1), -1 = give up (degenerate / behind).
struct WalkStep {
    int code, next;               // see cone_walk_step
    bool solid;                   // the facet is entered at more than a grazing angle
    double t;                     // code 1: the hit's parameter
    int rank;                     // code 1: the facet's reference index
};
__device__ __forceinline__ WalkStep cone_walk_step(PartRef P, int i, int prev, const double o[3], double d0, double d1, double d2,
                                                   double dd) {
    WalkStep out{-1, -1, false, INFINITY, 0};
    const f64x2 GAS *r = reinterpret_cast<const f64x2 GAS *>(P.col_rec);
    const int i6 = i * 6;
    const f64x2 r0 = ldg(r, i6), r1 = ldg(r, i6 + 1), r2 = ldg(r, i6 + 2), r3 = ldg(r, i6 + 3), r4 = ldg(r, i6 + 4),
                r5 = ldg(r, i6 + 5);
    const i32x4 nb = ldg(reinterpret_cast<const i32x4 GAS *>(P.col_enbr), i);      // neighbours | rank: the same round trip
    const double v00 = r0.x, v01 = r0.y, v02 = r1.x, e10 = r1.y, e11 = r2.x, e12 = r2.y;
    const double e20 = r3.x, e21 = r3.y, e22 = r4.x, m = r4.y, nn = r5.x, orient = r5.y;
    const double p0 = d1 * e22 - d2 * e21;
    const double p1 = d2 * e20 - d0 * e22;
    const double p2 = d0 * e21 - d1 * e20;
    const double det = (e10 * p0 + e11 * p1) + e12 * p2;
    if (!(fabs(det) >= RAY_EPS_DET)) return out;
    const double inv = rcp_det(det);
    const double s0 = o[0] - v00, s1 = o[1] - v01, s2 = o[2] - v02;
    const double u = ((s0 * p0 + s1 * p1) + s2 * p2) * inv;
    const double q0 = s1 * e12 - s2 * e11;
    const double q1 = s2 * e10 - s0 * e12;
    const double q2 = s0 * e11 - s1 * e10;
    const double v = ((d0 * q0 + d1 * q1) + d2 * q2) * inv;
    const double t = ((e20 * q0 + e21 * q1) + e22 * q2) * inv;
    const bool front = orient * det > 0;                      // the beam meets this facet's plane from outside
    const bool solid = front & (det * det >= CONE_MIN_COS2 * dd * nn);        // (bitwise between lane predicates: no branch)
    out.solid = solid;    // ... at more than a grazing angle
    if (!solid) {
        // The walk has reached the hull's horizon (or grazes).  Two certificates that the beam misses the whole set, by
        // more than any tolerance of the triangle tests -- then no search is needed:
        // (a) the hull lies on the inner side of this facet's plane: both ends of the beam clear of it on the OUTER side;
        const double n0 = e11 * e22 - e12 * e21, n1 = e12 * e20 - e10 * e22, n2 = e10 * e21 - e11 * e20;   // e1 x e2
        const double so = ((s0 * n0 + s1 * n1) + s2 * n2) * orient;                     // origin, outward if > 0
        const double se = so + ((d0 * n0 + d1 * n1) + d2 * n2) * orient;                // end point
        const double clear = CONE_MISS_MARGIN * CONE_MISS_MARGIN * nn;                  // (distance margin)^2 |n|^2
        if ((so > 0) & (se > 0) & (so * so > clear) & (se * se > clear)) {
            out.code = 2;
            return out;
        }
        // (b) the edge the walk just crossed is a SILHOUETTE edge of the hull as seen along the beam: the facet it left
        // faces the beam (`prev` is only passed for a facet the beam enters at more than a grazing angle), this one
        // faces away.  The plane through such an edge that contains the beam's direction supports the hull (its
        // normal lies between the two facets' outward normals), so a LINE that runs on its outer side, more than the
        // margin away, misses the hull -- most beams that miss pass beside the part and end exactly like this.
        if (!front & (prev >= 0) & (orient * det < 0) & (det * det >= CONE_SIL_MIN_COS2 * dd * nn)) {
            const int k = nb.x == prev ? 0 : (nb.y == prev ? 1 : (nb.z == prev ? 2 : -1));
            if (k >= 0) {
                // edge k of this facet (part_fill: 0 = v0 .. v0 + e2, 1 = v0 .. v0 + e1, 2 = v0 + e1 .. v0 + e2): a point
                // of it relative to the beam's origin, its direction, and the facet's third corner relative to that point
                const double a0 = k == 2 ? s0 - e10 : s0, a1 = k == 2 ? s1 - e11 : s1, a2 = k == 2 ? s2 - e12 : s2;     // o - point
                const double g0 = k == 0 ? e20 : (k == 1 ? e10 : e20 - e10), g1 = k == 0 ? e21 : (k == 1 ? e11 : e21 - e11),
                             g2 = k == 0 ? e22 : (k == 1 ? e12 : e22 - e12);
                const double c0 = k == 0 ? e10 : (k == 1 ? e20 : -e10), c1 = k == 0 ? e11 : (k == 1 ? e21 : -e11),
                             c2 = k == 0 ? e12 : (k == 1 ? e22 : -e12);
                const double m0 = g1 * d2 - g2 * d1, m1 = g2 * d0 - g0 * d2, m2 = g0 * d1 - g1 * d0;               // edge x beam
                const double side_o = (a0 * m0 + a1 * m1) + a2 * m2, side_p = (c0 * m0 + c1 * m1) + c2 * m2;
                const double mm = (m0 * m0 + m1 * m1) + m2 * m2;
                if ((side_o * side_p < 0) & (side_o * side_o > CONE_MISS_MARGIN * CONE_MISS_MARGIN * mm)) {
                    out.code = 2;
                    return out;
                }
            }
        }
        if (!front) {
            out.code = prev < 0 ? -2 : -3;                   // (the codes only matter to the statistics of diagnostic builds)
            return out;
        }
        // a facet met at a grazing angle decides nothing, but the walk may pass through it
    }
    if (!(t >= 0.0)) {                                       // behind the origin
        out.code = -4;
        return out;
    }
    if ((u >= m) & (v >= m) & ((u + v) <= 1.0 - m)) {
        if (!solid) {                                        // a grazing entry decides nothing (the walk may pass through such facets)
            out.code = -5;
            return out;
        }
        // entered at an interior point: the closest hit of the whole set (ray_closest_wave's single-facet criterion);
        // beyond the beam's end point it means the beam stops short of the part: a miss
        out.code = (t <= 1.0) ? 1 : 2;
        out.t = t;
        out.rank = nb.w;
        return out;
    }
    // outside the triangle (or within the edge margin): cross the edge that is violated most
    const double w = 1.0 - u - v;
    const int e = ((u <= v) & (u <= w)) ? 0 : ((v <= w) ? 1 : 2);         // 0: u smallest, 1: v, 2: w = 1 - u - v
    out.next = e == 0 ? nb.x : (e == 1 ? nb.y : nb.z);
    out.code = 0;
    return out;
}

// The general closest-hit search for up to 64 rays at once, one per lane (`need`), all from the same origin o: what
// ray_closest_wave's general search returns for each of them -- the closest two-sided hit over ALL collision triangles,
// equal t resolved to the lowest reference index.  The triangle loop is wave-uniform (every lane looks at the same
// triangle, read once for the wave), culled twice with the float boxes: a chunk of 64 is visited if ANY lane's segment
// box meets the chunk's box, a triangle is tested by the lanes whose segment box meets its box.  No candidate lists,
// no reductions: a lane keeps its own best.  (The wave-wide search costs ~3 us per ray; a shot of an env that has
// wandered off the part has ~130 rays that all miss -- 2 ms per step, and the launch waits for that env.)
__device__ __forceinline__ float bcast_f(float v, int src) {
    return __int_as_float(__builtin_amdgcn_readlane(__float_as_int(v), src));
}

__device__ void rays_general_lanes(PartRef P, const double o[3], const double dst[3], bool need, double &t_out, int &tri_out) {
    const double d0 = dst[0] - o[0], d1 = dst[1] - o[1], d2 = dst[2] - o[2];
    double best_t = INFINITY, best_det = 0;
    int best_r = 0x7fffffff, best_i = -1;
    const double o3[3] = {sel3(o[0], o[1], o[2], P.a1), sel3(o[0], o[1], o[2], P.a2), sel3(o[0], o[1], o[2], P.a0)};
    const double d3[3] = {sel3(d0, d1, d2, P.a1), sel3(d0, d1, d2, P.a2), sel3(d0, d1, d2, P.a0)};
    const SegBox sb = seg_box(o3, d3, 1.0);
    const f32x4 GAS *boxes = reinterpret_cast<const f32x4 GAS *>(P.col_bbox);
    const f32x4 GAS *chunk_boxes = reinterpret_cast<const f32x4 GAS *>(P.col_chunk_bbox);
    // The boxes are read 64 at a time, one per lane, and handed round by lane broadcasts: a read per loop trip at a
    // wave-uniform address would put a memory round trip into every one of the ~64 x chunks trips.
    const int lane = threadIdx.x & 63;
    for (int cbase = 0; cbase < P.n_col_chunks; cbase += 64) {                 // (table padded to 64 with empty boxes)
        const f32x4 cla = ldg(chunk_boxes, 2 * (cbase + lane)), clb = ldg(chunk_boxes, 2 * (cbase + lane) + 1);
        const int nc = P.n_col_chunks - cbase < 64 ? P.n_col_chunks - cbase : 64;
        for (int cl = 0; cl < nc; ++cl) {
            const f32x4 ca = {bcast_f(cla.x, cl), bcast_f(cla.y, cl), bcast_f(cla.z, cl), bcast_f(cla.w, cl)};
            const f32x4 cb = {bcast_f(clb.x, cl), bcast_f(clb.y, cl), 0.0f, 0.0f};
            const bool ov = need && box_overlap(sb, ca, cb);
            if (ballot64(ov) == 0) continue;
            const int i0 = (cbase + cl) << 6;
            const f32x4 tla = ldg(boxes, 2 * (i0 + lane)), tlb = ldg(boxes, 2 * (i0 + lane) + 1);
            for (int tl = 0; tl < 64; ++tl) {
                const f32x4 ba = {bcast_f(tla.x, tl), bcast_f(tla.y, tl), bcast_f(tla.z, tl), bcast_f(tla.w, tl)};
                const f32x4 bb = {bcast_f(tlb.x, tl), bcast_f(tlb.y, tl), 0.0f, 0.0f};
                const bool pass = ov && box_overlap(sb, ba, bb);
                if (ballot64(pass) == 0) continue;
                mt_one(P, pass ? i0 + tl : -1, o, d0, d1, d2, 1.0, best_t, best_r, best_i, best_det);
            }
        }
    }
    t_out = best_t;
    tri_out = best_i;
}

// The beams b0 + lane of the cone at tool pose (pos, quat) (rob:251-258): end point `dst` of this lane's beam, and its walk
// over the hull.  Returns the lane's state: 1 = hit at parameter t (exact closest hit of the whole set), 2 = no beam, or
// a proven miss, 3 = not settled (the caller searches).
__device__ __forceinline__ int cone_walk_lanes(PartRef P, const double pos[3], const double quat[4], int b0, int lane,
                                               double dst[3], double &t, int &last_facet) {
    last_facet = -1;                                // state 3: the facet the walk gave up on (where a search of the ray may start)
    const int bm = b0 + lane;
    const bool have = bm < P.n_beams;
    dst[0] = pos[0], dst[1] = pos[1], dst[2] = pos[2];
    if (have) transform_point(pos, quat, ldg(P.beams, 3 * bm), ldg(P.beams, 3 * bm + 1), ldg(P.beams, 3 * bm + 2), dst);
    const double d0 = dst[0] - pos[0], d1 = dst[1] - pos[1], d2 = dst[2] - pos[2];
    const double dd = (d0 * d0 + d1 * d1) + d2 * d2;
    t = INFINITY;
    int state = have ? 0 : 2;                       // 0 walking
    if (P.col_convex && P.hg_nx > 0) {
        // every lane starts on the facet the front-facet grid (part_fill) names for the middle of its beam -- the tool hovers
        // one hook distance above the part and the beams are two long, so that is about where it lands: one to three steps
        // instead of a walk from the facet under the tool to the rim of the cone, and a start for the beams of a tool
        // that has left the part (whose own ray missed)
        const double m1 = sel3(pos[0], pos[1], pos[2], P.a1) + 0.5 * sel3(d0, d1, d2, P.a1);
        const double m2 = sel3(pos[0], pos[1], pos[2], P.a2) + 0.5 * sel3(d0, d1, d2, P.a2);
        int cx = cell_coord(m1, P.hg_o1, P.hg_inv, P.hg_nx), cy = cell_coord(m2, P.hg_o2, P.hg_inv, P.hg_ny);
        cx = cx < 0 ? 0 : (cx > P.hg_nx - 1 ? P.hg_nx - 1 : cx);
        cy = cy < 0 ? 0 : (cy > P.hg_ny - 1 ? P.hg_ny - 1 : cy);
        int f = have ? ldg(P.hg_facet, cy * P.hg_nx + cx) : -1, prev = -1, rk = 0;
        if (have && f < 0) state = 3;
        CONE_STAT(1, __popcll(ballot64(have)));
        for (int step = 0; step < CONE_WALK_STEPS; ++step) {
            if (ballot64(state == 0) == 0) break;
            CONE_STAT(13, 1);
            CONE_STAT(15, __popcll(ballot64(state == 0)));
            if (state == 0) {
                const WalkStep ws = cone_walk_step(P, f, prev, pos, d0, d1, d2, dd);
                const int r = ws.code, next = ws.next;
                const bool solid = ws.solid;
                if (r == 1) {
                    t = ws.t;
                    rk = ws.rank;
                }
                CONE_STAT(5, __popcll(ballot64(r == -1)));
                CONE_STAT(6, __popcll(ballot64(r == -2)));
                CONE_STAT(7, __popcll(ballot64(r == -3)));
                CONE_STAT(8, __popcll(ballot64(r == -4)));
                CONE_STAT(9, __popcll(ballot64(r == 0 && next < 0)));
                if (r == 1) state = 1;
                else if (r == 2) state = 2;
                else if (r < 0 || next < 0) state = 3;
                else {
                    prev = solid ? f : -1;                   // (the silhouette certificate needs a facet that is entered for sure)
                    f = next;
                }
            }
        }
        CONE_STAT(10, __popcll(ballot64(state == 0)));
        if (state == 0) state = 3;
        last_facet = f;
        CONE_STAT(2, __popcll(ballot64(state == 1)));
        CONE_STAT(3, __popcll(ballot64(state == 2 && have)));
        CONE_STAT(4, __popcll(ballot64(state == 3)));
    } else if (have) {
        state = 3;
    }
    return state;
}

// Beams beside the part, one per lane (`state` 3 -> 2 where proven).  The collision set lies inside the slab [slab_lo,
// slab_hi] of the third axis and, projected to the principal plane, inside its outline polygon: if the stretch of the beam
// inside the slab (widened by the margin) lies, projected, more than the margin outside one edge of the outline, the beam
// misses every triangle.
__device__ __forceinline__ void beams_outside_outline_lanes(PartRef P, const double pos[3], double d0, double d1, double d2, int &state) {
    const int lane = threadIdx.x & 63;
    if (P.n_outline > 0 && ballot64(state == 3) != 0) {
        const double oz = sel3(pos[0], pos[1], pos[2], P.a0), dz = sel3(d0, d1, d2, P.a0);
        const double lo = P.slab_lo - CONE_MISS_MARGIN, hi = P.slab_hi + CONE_MISS_MARGIN;
        double ta = 0.0, tb = 1.0;
        bool clear = false;                                   // never inside the slab
        if (dz != 0.0) {
            const double t0 = (lo - oz) / dz, t1 = (hi - oz) / dz;
            ta = fmax(0.0, fmin(t0, t1) - 1e-9);
            tb = fmin(1.0, fmax(t0, t1) + 1e-9);
            clear = ta > tb;
        } else {
            clear = oz < lo || oz > hi;
        }
        const double o1 = sel3(pos[0], pos[1], pos[2], P.a1), o2 = sel3(pos[0], pos[1], pos[2], P.a2);
        const double e1 = sel3(d0, d1, d2, P.a1), e2 = sel3(d0, d1, d2, P.a2);
        const double ax = o1 + ta * e1, ay = o2 + ta * e2, bx = o1 + tb * e1, by = o2 + tb * e2;
        const f64x2 GAS *ol = reinterpret_cast<const f64x2 GAS *>(P.outline);
        for (int base = 0; base < P.n_outline && ballot64(state == 3 && !clear) != 0; base += 64) {
            const f64x2 pq = ldg(ol, 2 * (base + lane)), nq = ldg(ol, 2 * (base + lane) + 1);
            const int ne = P.n_outline - base < 64 ? P.n_outline - base : 64;
            for (int j = 0; j < ne; ++j) {
                const double px = bcast_d(pq.x, j), py = bcast_d(pq.y, j), nx = bcast_d(nq.x, j), ny = bcast_d(nq.y, j);
                const double sa = nx * (ax - px) + ny * (ay - py), sb = nx * (bx - px) + ny * (by - py);
                clear = clear || (sa > CONE_MISS_MARGIN && sb > CONE_MISS_MARGIN);
            }
        }
        if (state == 3 && clear) state = 2;
    }
}

// The rays of beams b0 + lane: hit[3] of this lane's beam, returns whether it hit.  `hint`: facet of the tool's ray (where
// the wave-wide searches of leftover rays start).
__device__ bool cone_rays_lanes(PartRef P, const double pos[3], const double quat[4], int b0, int hint, int lane,
                                int *cand_lds, double hit[3]) {
    const bool have = b0 + lane < P.n_beams;
    double dst[3], t;
    int last_facet;
    int state = cone_walk_lanes(P, pos, quat, b0, lane, dst, t, last_facet);
    const double d0 = dst[0] - pos[0], d1 = dst[1] - pos[1], d2 = dst[2] - pos[2];
#ifdef PRL_FORCE_GENERAL_RAY
    if (have) state = 3;
#endif
    hit[0] = pos[0] + t * d0;
    hit[1] = pos[1] + t * d1;
    hit[2] = pos[2] + t * d2;
    beams_outside_outline_lanes(P, pos, d0, d1, d2, state);
    // the stragglers (edge and vertex hits, and every miss), together
    const uint64_t todo = ballot64(state == 3);
    if (__popcll(todo) > CONE_JOINT_FROM) {
        double tw;
        int tri;
        rays_general_lanes(P, pos, dst, state == 3, tw, tri);
        if (state == 3) {
            state = tri >= 0 ? 1 : 2;
            hit[0] = pos[0] + tw * d0;
            hit[1] = pos[1] + tw * d1;
            hit[2] = pos[2] + tw * d2;
        }
    } else {                                        // a few: one wave-wide search each is cheaper than the joint loop
        uint64_t left = todo;
        int wide_hint = hint;
        while (left) {
            const int L = __builtin_ctzll(left);
            left &= left - 1;
            const double e3[3] = {bcast_d(dst[0], L), bcast_d(dst[1], L), bcast_d(dst[2], L)};
            double tw, hw[3];
            const int idx = ray_closest_wave(P, pos, e3, lane, tw, hw, wide_hint, cand_lds);
            if (lane == L) {
                state = idx >= 0 ? 1 : 2;
                hit[0] = hw[0];
                hit[1] = hw[1];
                hit[2] = hw[2];
            }
        }
    }
    return state == 1;
}

// ---------------------------------------------------------------- bpw:565 pixel_kd_tree.query(k=1), one hit point per lane
// The nearest sample of this lane's point `pt` (if `want`), exact, equal distances resolved to the lowest reference
// index -- as nearest_sample_wave, but 64 queries at once: each lane scans the (2r+1)^2 block of FINE grid cells around
// its point (PartDev::fg_*: ~2.6 samples a cell, so a 3 x 3 block holds ~23 candidates where the painter's own sample
// grid holds ~1 350), rows as contiguous record ranges, four records per trip so that their reads travel together.
// A sample outside the block is more than r cells away in the principal plane: the best of the block is the answer
// once it lies within r * 0.99 * cell; a block whose best does not settle the query names the radius that will.  Returns
// the device position of the sample, -1 if not `want`, or -2 if three rings do not settle it (a hit point centimetres off
// the sampled surface): the caller asks nearest_sample_far.
__device__ __forceinline__ int nearest_sample_lane(PartRef P, const double pt[3], bool want) {
    const double h1 = sel3(pt[0], pt[1], pt[2], P.a1), h2 = sel3(pt[0], pt[1], pt[2], P.a2);
    const int icx = cell_coord(h1, P.fg_o1, P.fg_inv, P.fg_nx), icy = cell_coord(h2, P.fg_o2, P.fg_inv, P.fg_ny);
    const f64x2 GAS *rec = reinterpret_cast<const f64x2 GAS *>(P.fg_rec);
    int result = want ? -2 : -1;
    bool open = want;
    int r = 1;
    for (int pass = 0; pass < 3; ++pass) {
        if (open && r > 3) open = false;                              // (stays -2)
        if (ballot64(open) == 0) break;
        const int rmax = -wave_min_i(open ? -r : 0);                  // wave-uniform trip count, per-lane ranges
        double best_d = INFINITY;
        int best_rank = 0x7fffffff, best_pos = -1;
        const int cx0 = icx - r < 0 ? 0 : icx - r, cx1 = icx + r > P.fg_nx - 1 ? P.fg_nx - 1 : icx + r;
        for (int dy = -rmax; dy <= rmax; ++dy) {
            const int cy = icy + dy;
            const bool row_ok = open && dy >= -r && dy <= r && cy >= 0 && cy < P.fg_ny && cx0 <= cx1;
            const int b = row_ok ? ldg(P.fg_start, cy * P.fg_nx + cx0) : 0;
            const int e = row_ok ? ldg(P.fg_start, cy * P.fg_nx + cx1 + 1) : 0;
            for (int i0 = b; ballot64(i0 < e) != 0; i0 += 4) {
                f64x2 ra[4], rb[4];
#pragma unroll
                for (int q = 0; q < 4; ++q) {
                    const int i = i0 + q < e ? i0 + q : (e > b ? e - 1 : 0);     // (in range: the fold repeats a record)
                    ra[q] = ldg(rec, 2 * i);
                    rb[q] = ldg(rec, 2 * i + 1);
                }
#pragma unroll
                for (int q = 0; q < 4; ++q) {
                    if (i0 + q < e) {
                        const double dx = ra[q].x - pt[0], dy2 = ra[q].y - pt[1], dz = rb[q].x - pt[2];
                        const double dd = (dx * dx + dy2 * dy2) + dz * dz;
                        const int rk = __double2loint(rb[q].y);
                        if (dd < best_d || (dd == best_d && rk < best_rank)) {
                            best_d = dd;
                            best_rank = rk;
                            best_pos = __double2hiint(rb[q].y);
                        }
                    }
                }
            }
        }
        const double lim = r * P.fg_accept;
        if (open && best_pos >= 0 && best_d <= lim * lim) {
            result = best_pos;
            open = false;
        } else if (open) {
            // the block that will settle it: its radius times 0.99 cells reaches the best found so far
            const int need = best_pos >= 0 ? (int)fmin(ceil(sqrt(best_d) / P.fg_accept), 1.0e6) : 2 * r + 2;
            r = need > r ? need : r + 1;
        }
    }
#ifdef PRL_FORCE_FULL_SCANS                          // diagnostic build: every query through the tree walk
    if (want) result = -2;
#endif
    return result;
}

// ---------------------------------------------------------------- the same query on float records (the beams kernel)
// The beams kernel is bound by the loads its lanes gather through the CU's L1 (rocprofv3: TA busy 73-85 %) and by vector
// issue; most of both went into this query.  Three measures, the answer stays the float64 one bit for bit:
//  * the block is scanned on 16-byte records (x y z rounded to float) keeping the three nearest in float; only the ones
//    that float arithmetic cannot tell from the nearest are then measured in float64 (one, as a rule):
//      a coordinate c, |c| <= M, rounds to float with error <= 2^-24 M; a difference of two such floats is exact before its
//      own rounding, so each float difference is off by E <= 2^-23 M (1 + 2^-24); the float squared distance (three
//      products, two sums, each rounded) is then off by at most  band(d) = 3.5 d E + 3 E^2 + 2^-22 d^2  for a distance d.
//    A record is a contender if its float distance minus its band does not exceed the nearest's plus its band.  Three
//    contenders: there may be a fourth -- the lane reports -2 and the rest kernel's tree walk answers exactly.
//  * the three nearest are kept as KEYS: the float distance's bit pattern (non-negative floats order like unsigned
//    integers) with its ten low mantissa bits replaced by the record's place in the scan (row of the block, offset in
//    the row) -- three integer min / median instructions per record instead of three compares and six selects.  A key
//    stands for a distance in [t, t (1 + 2^-13)), t = the key with the ten bits cleared; the contender test allows for it.
//  * the block is the 2 x 2 cells around the point (the cell's quadrant picks them): every sample outside it is more than
//    half a cell away, so a nearest sample within its reach is final -- ~10 records.  Anything else (-2) is the far
//    kernel's, with the distance found here as its first bound: rings of cells scanned here, by one or two lanes of 64,
//    cost the wave more than the few-lane search there (beams 139 + far 38 us with three rings, 99 + 66 with none).
__device__ __forceinline__ float nn_band(float d2, float E) {
    const float d = __builtin_sqrtf(d2) * 1.0001f + 1e-12f;
    return 3.5f * d * E + 3.0f * E * E + 2.4e-7f * d2 + 1e-30f;
}
__device__ __forceinline__ unsigned umed3(unsigned a, unsigned b, unsigned c) {
    unsigned r;
    asm("v_med3_u32 %0, %1, %2, %3" : "=v"(r) : "v"(a), "v"(b), "v"(c));
    return r;
}
#define NN_KEY_INF 0x7f800000u                  // +inf as a key
#define NN_KEY_PLACE 0x3ffu                     // place in the scan: row << 7 | offset in the row
#define NN_KEY_SLACK 1.000123f                  // > 1 + 2^-13

// the float64 distance of the record a key names (row `row` of the block starts at record b_row), with rank and position
__device__ __forceinline__ void nn_key_measure(PartRef P, unsigned key, int b_row, const double pt[3], double &dd, int &rank, int &pos) {
    const int i = b_row + (int)(key & 127u);
    const f64x2 GAS *rec = reinterpret_cast<const f64x2 GAS *>(P.fg_rec);
    const f64x2 ra = ldg(rec, 2 * i), rb = ldg(rec, 2 * i + 1);
    const double dx = ra.x - pt[0], dy = ra.y - pt[1], dz = rb.x - pt[2];
    dd = (dx * dx + dy * dy) + dz * dz;
    rank = __double2loint(rb.y);
    pos = __double2hiint(rb.y);
}

__device__ __forceinline__ int nearest_sample_lane_f32(PartRef P, const double pt[3], bool want, float &far_bound) {
    far_bound = INFINITY;                                             // -2: the squared distance of a sample seen on the way, rounded up
    const double h1 = sel3(pt[0], pt[1], pt[2], P.a1), h2 = sel3(pt[0], pt[1], pt[2], P.a2);
    const int icx = cell_coord(h1, P.fg_o1, P.fg_inv, P.fg_nx), icy = cell_coord(h2, P.fg_o2, P.fg_inv, P.fg_ny);
    const double f1 = (h1 - P.fg_o1) * P.fg_inv - (double)icx, f2 = (h2 - P.fg_o2) * P.fg_inv - (double)icy;   // place in the cell
    const bool in_cell = (f1 >= 0.0) & (f1 < 1.0) & (f2 >= 0.0) & (f2 < 1.0);      // (cell_coord clamps far outside the grid)
    const f32x4 GAS *rec = reinterpret_cast<const f32x4 GAS *>(P.fg_rec32);
    const float qx = (float)pt[0], qy = (float)pt[1], qz = (float)pt[2];
    const double mq = fmax(fmax(fabs(pt[0]), fabs(pt[1])), fmax(fabs(pt[2]), P.samp_absmax));
    const float E = (float)(mq * 1.1920929e-7 * 1.001);              // 2^-23 M, rounded up
    int result = want ? -2 : -1;
    const bool open = want & (mq < 1.0e6) & in_cell;
    if (ballot64(open) == 0) return result;
    int cx0 = f1 < 0.5 ? icx - 1 : icx, cx1 = cx0 + 1, cy0 = f2 < 0.5 ? icy - 1 : icy, cy1 = cy0 + 1;
    // the block's border is this far from the point, in cells, wherever it is not the grid's own (0.5 .. 1): every sample
    // outside the block is farther than that
    const double reach = fmin(fmin(f1 < 0.5 ? f1 + 1.0 : f1, f1 < 0.5 ? 1.0 - f1 : 2.0 - f1), fmin(f2 < 0.5 ? f2 + 1.0 : f2, f2 < 0.5 ? 1.0 - f2 : 2.0 - f2));
    cx0 = cx0 < 0 ? 0 : cx0, cx1 = cx1 > P.fg_nx - 1 ? P.fg_nx - 1 : cx1;
    cy0 = cy0 < 0 ? 0 : cy0, cy1 = cy1 > P.fg_ny - 1 ? P.fg_ny - 1 : cy1;
    const int rows = (open & (cx0 <= cx1) & (cy0 <= cy1)) ? cy1 - cy0 + 1 : 0;
    // the record ranges of the two rows travel together
    const int b0 = rows > 0 ? ldg(P.fg_start, cy0 * P.fg_nx + cx0) : 0, e0 = rows > 0 ? ldg(P.fg_start, cy0 * P.fg_nx + cx1 + 1) : 0;
    const int b1 = rows > 1 ? ldg(P.fg_start, (cy0 + 1) * P.fg_nx + cx0) : 0, e1 = rows > 1 ? ldg(P.fg_start, (cy0 + 1) * P.fg_nx + cx1 + 1) : 0;
    unsigned k1 = NN_KEY_INF, k2 = NN_KEY_INF, k3 = NN_KEY_INF;
    bool wide = false;
#pragma unroll
    for (int j = 0; j < 2; ++j) {
        const int b = j == 0 ? b0 : b1;
        int e = j == 0 ? e0 : e1;
        if (e - b > 128) wide = true, e = b;                          // (more than a key can place: the pyramid decides)
        unsigned place = (unsigned)j << 7;
        for (int i0 = b; ballot64(i0 < e) != 0; i0 += 4, place += 4) {
            const int ib = i0 < e ? i0 : b;                           // (a lane that is through stays in range)
            f32x4 rc[4];
#pragma unroll
            for (int q = 0; q < 4; ++q) rc[q] = ldg(rec, ib + q);     // (the table is padded by four records)
#pragma unroll
            for (int q = 0; q < 4; ++q) {
                const float ex = rc[q].x - qx, ey = rc[q].y - qy, ez = rc[q].z - qz;
                const float dd = __builtin_fmaf(ez, ez, __builtin_fmaf(ey, ey, ex * ex));
                const unsigned key = i0 + q < e ? ((__float_as_uint(dd) & ~NN_KEY_PLACE) | (place + q)) : NN_KEY_INF;
                k3 = umed3(k2, k3, key);                              // keep the three smallest (k1 <= k2 <= k3)
                k2 = umed3(k1, k2, key);
                k1 = k1 < key ? k1 : key;
            }
        }
    }
    // float64 for the contenders
    double best_d = INFINITY;
    int best_rank = 0x7fffffff, best_pos = -1;
    const float t1 = __uint_as_float(k1 & ~NN_KEY_PLACE) * NN_KEY_SLACK;
    const float lim1 = t1 + nn_band(t1, E);
    const float t2 = __uint_as_float(k2 & ~NN_KEY_PLACE), t3 = __uint_as_float(k3 & ~NN_KEY_PLACE);
    const bool c2 = open & (k2 < NN_KEY_INF) & (t2 - nn_band(t2 * NN_KEY_SLACK, E) <= lim1);
    const bool c3 = open & (k3 < NN_KEY_INF) & (t3 - nn_band(t3 * NN_KEY_SLACK, E) <= lim1);
    const bool m1 = open & (k1 < NN_KEY_INF), m2 = c2 & !c3;
    if (m1) nn_key_measure(P, k1, ((k1 >> 7) & 1u) ? b1 : b0, pt, best_d, best_rank, best_pos);
    if (ballot64(m2) != 0) {
        if (m2) {
            double dd;
            int rk, ps;
            nn_key_measure(P, k2, ((k2 >> 7) & 1u) ? b1 : b0, pt, dd, rk, ps);
            if ((dd < best_d) | ((dd == best_d) & (rk < best_rank))) {
                best_d = dd;
                best_pos = ps;
            }
        }
    }
    const double lim = reach * P.fg_accept;
    if (open & (best_pos >= 0)) far_bound = __double2float_ru(best_d);
    // three the float distances cannot order, a row beyond a key's places, a nearest sample beyond the block's reach: -2
    if (open & !(c3 | wide) & (best_pos >= 0) & (best_d <= lim * lim)) result = best_pos;
    // a point that has seen no sample at all (its block is empty: the hull over a window of the part): the distance to the
    // seed sample of its cell (fg_seed) is the bound the far kernel starts with -- two dependent reads that cost a wave of
    // this kernel nothing it does not hide, and were a fifth of a search's chain there
    const bool blind = want & (result == -2) & !(far_bound < INFINITY);
    if (ballot64(blind) != 0) {
        if (blind) {
            const int gx = icx < 0 ? 0 : (icx > P.fg_nx - 1 ? P.fg_nx - 1 : icx), gy = icy < 0 ? 0 : (icy > P.fg_ny - 1 ? P.fg_ny - 1 : icy);
            const int i = ldg(P.fg_seed, gy * P.fg_nx + gx);
            if (i >= 0) {
                const f64x2 GAS *rec64 = reinterpret_cast<const f64x2 GAS *>(P.fg_rec);
                const f64x2 ra = ldg(rec64, 2 * i), rb = ldg(rec64, 2 * i + 1);
                const double dx = ra.x - pt[0], dy = ra.y - pt[1], dz = rb.x - pt[2];
                const double dd = (dx * dx + dy * dy) + dz * dz;
                if (dd == dd) far_bound = __double2float_ru(dd);      // (+inf stays: the search there finds nothing either)
            }
        }
    }
#ifdef PRL_FORCE_FULL_SCANS                          // diagnostic build: every query through the far kernel's search
    if (want) result = -2;
#endif
    return result;
}

// ---------------------------------------------------------------- nearest sample by branch and bound (any distance)
// The same query where the ring argument has no grip: the collision hull spans the windows and recesses of a part, and
// stands decimetres above a curved panel, so a hit point can lie 3 - 30 cm from the nearest sample; a block of cells
// whose radius reaches that far holds thousands of samples.  The box pyramid over the fine grid (PartDev::py_*) answers
// at any distance: only nodes whose bounding box (rounded outward, so never farther than any sample inside) lies within
// the best distance known are opened; the samples of the cells that remain are measured as in nearest_sample_lane
// (float64, equal distances to the lowest reference index).
//
// BFS_G LANES PER POINT, level by level (BFS_N points a wave, group g = lanes BFS_G g .. BFS_G g + BFS_G - 1).  The walk of a
// point is a chain of dependent reads and a lone wave issues an instruction every five cycles, so what counts is the number
// of steps, not the lanes they keep busy: a depth-first walk with one point per lane took 20 loop trips of ~1.7 us for
// the slowest of its 64 lanes (rest kernel 84 us); level by level a point takes one trip per level:
//   * the first bound is a sample of the point's own cell column, or of the nearest column that has one (fg_seed);
//   * per level, the lanes of the group share the children of the group's open nodes (its FRONTIER, in LDS); a child
//     within the bound joins the next frontier; the farthest corner of a child that holds samples is a bound too
//     (there is a sample at least that near), taken from the next level on;
//   * the cells that remain carry their record range in the spare floats of their box; the lanes share them and the
//     group's best is reduced over its lanes.
// `fr`: the wave's BFS_LDS_INTS ints of LDS.  A frontier that outgrows BFS_CAP gives up: -2, the caller asks
// nearest_sample_wave (exact as well).  Returns (in every lane of the group) the device position of the sample; -1 if not
// `want`.
#ifndef PRL_BFS_LANES
#define PRL_BFS_LANES 4                        // (8: far kernel 66 us, 4: 55, 2: 86)
#endif
constexpr int BFS_G = PRL_BFS_LANES;              // lanes a point (a power of two, at most 8), BFS_N points a wave
constexpr int BFS_N = 64 / BFS_G;
constexpr int BFS_SHIFT = BFS_G == 8 ? 3 : (BFS_G == 4 ? 2 : 1);
constexpr int BFS_CAP = 32;
constexpr int BFS_LDS_INTS = 2 * BFS_N * BFS_CAP;
__device__ __forceinline__ void lds_wave_sync() {
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
    __builtin_amdgcn_wave_barrier();
    __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
}
__device__ __forceinline__ float group8_min(float v) {
#pragma unroll
    for (int o = 1; o < BFS_G; o <<= 1) v = fminf(v, __shfl_xor(v, o));
    return v;
}

template <bool SEED = true>
__device__ __forceinline__ int nearest_sample_bfs(PartRef P, const double pt[3], bool want, float hint, int lane, int *fr) {
    if (P.py_levels <= 0) return want ? -2 : -1;
    const int g = lane >> BFS_SHIFT, m = lane & (BFS_G - 1);
    int *cur = fr + g * BFS_CAP, *nxt = fr + (BFS_N + g) * BFS_CAP;
    const f32x4 GAS *boxes = reinterpret_cast<const f32x4 GAS *>(P.py_box);
    const f64x2 GAS *rec = reinterpret_cast<const f64x2 GAS *>(P.fg_rec);
    double best_d = INFINITY;
    int best_rank = 0x7fffffff, best_pos = -1;
    // the first bound: `hint` (the caller has seen a sample at that squared distance), or a sample of the point's own cell
    // column, or of the nearest column that has one (fg_seed)
    // (SEED = false: the caller always has a hint -- the far kernel, whose entries carry one; without, the search still ends)
    if (SEED && ballot64(want && !(hint < INFINITY)) != 0) {
        if (want && !(hint < INFINITY)) {
            int cx = cell_coord(sel3(pt[0], pt[1], pt[2], P.a1), P.fg_o1, P.fg_inv, P.fg_nx);
            int cy = cell_coord(sel3(pt[0], pt[1], pt[2], P.a2), P.fg_o2, P.fg_inv, P.fg_ny);
            cx = cx < 0 ? 0 : (cx > P.fg_nx - 1 ? P.fg_nx - 1 : cx);
            cy = cy < 0 ? 0 : (cy > P.fg_ny - 1 ? P.fg_ny - 1 : cy);
            const int i = ldg(P.fg_seed, cy * P.fg_nx + cx);
            if (i >= 0) {
                const f64x2 ra = ldg(rec, 2 * i), rb = ldg(rec, 2 * i + 1);
                const double dx = ra.x - pt[0], dy = ra.y - pt[1], dz = rb.x - pt[2];
                const double dd = (dx * dx + dy * dy) + dz * dz;
                if (dd == dd) {                          // (a NaN point finds nothing)
                    best_d = dd;
                    best_rank = __double2loint(rb.y);
                    best_pos = __double2hiint(rb.y);
                }
            }
        }
    }
    double bound = hint < INFINITY ? (double)hint : best_d;
    bool over = false;
#ifdef PRL_CONE_TRACE
    int rounds_ = 0, widest_ = 0;
#endif
    // the levels' dimensions and first nodes, level k in lane k, read once: a level's own are lane reads at a wave-uniform index
    const int tl = lane < PY_MAX_LEVELS ? lane : PY_MAX_LEVELS - 1;
    const int lv_nx = P.py_nx[tl], lv_ny = P.py_ny[tl], lv_off = P.py_off[tl];
    // the walk starts with ALL nodes of the highest level that has no more than eight (the levels above it: 1 .. 8 nodes,
    // a round trip each for nothing)
    int top = P.py_levels - 1;
    while (top > 0 && P.py_nx[top - 1] * P.py_ny[top - 1] <= 8) --top;
    const int top_nx = __builtin_amdgcn_readlane(lv_nx, top), top_n = top_nx * __builtin_amdgcn_readlane(lv_ny, top);
    int nf = want ? top_n : 0;
    // [-DPRL_FAR_HINT_START, the far kernel: every entry carries a bound]  A sample within the bound lies within r = sqrt(bound) of
    // the point in the principal plane, so in the 3 x 3 nodes around the point's own at the lowest level whose nodes are at least r
    // wide: the walk of this group can start THERE (nine nodes, the ones outside the grid marked so that their children test as
    // outside) instead of at the top -- two to four levels lower on the door, more on a large part's deeper pyramid.
    int lg = top, scx = 0, scy = 0;
#ifdef PRL_FAR_HINT_START
    if constexpr (!SEED) {
        if (want) {
            const double rc = sqrt(bound) * P.fg_inv;                 // r in cells
            const int k = (int)fmin(rc * 1.000001 + 1.001, 1.0e9);    // >= ceil(r in cells), a hair more
            int l = k <= 1 ? 0 : 32 - __builtin_clz((unsigned)(k - 1));     // the lowest level whose nodes span k cells
            l = l < 1 ? 1 : l;
            if (l < top) {
                int cx = cell_coord(sel3(pt[0], pt[1], pt[2], P.a1), P.fg_o1, P.fg_inv, P.fg_nx);
                int cy = cell_coord(sel3(pt[0], pt[1], pt[2], P.a2), P.fg_o2, P.fg_inv, P.fg_ny);
                cx = cx < 0 ? 0 : (cx > P.fg_nx - 1 ? P.fg_nx - 1 : cx);
                cy = cy < 0 ? 0 : (cy > P.fg_ny - 1 ? P.fg_ny - 1 : cy);
                lg = l;
                scx = cx >> l;
                scy = cy >> l;
                nf = 0;
            }
        }
    }
    const int level_from = -wave_min_i(-(want ? lg : 1));
#else
    const int level_from = top;
#endif
    for (int i = m; want && lg == top && i < top_n; i += BFS_G) cur[i] = (top << 24) | ((i / top_nx) << 12) | (i % top_nx);      // (node: level << 24 | cy << 12 | cx)
    lds_wave_sync();
    for (int level = level_from; level >= 1; --level) {              // (wave-uniform)
        const int cl = level - 1, cnx = __builtin_amdgcn_readlane(lv_nx, cl), cny = __builtin_amdgcn_readlane(lv_ny, cl),
                  off = __builtin_amdgcn_readlane(lv_off, cl);
#ifdef PRL_FAR_HINT_START
        if constexpr (!SEED) {
            if (ballot64(want && lg == level && lg != top) != 0) {   // the groups whose walk starts at this level
                const int lnx = __builtin_amdgcn_readlane(lv_nx, level), lny = __builtin_amdgcn_readlane(lv_ny, level);
                if (want && lg == level && lg != top) {
                    for (int i = m; i < 9; i += BFS_G) {
                        const int nx_ = scx + (i % 3) - 1, ny_ = scy + (i / 3) - 1;
                        const bool ok = (nx_ >= 0) & (nx_ < lnx) & (ny_ >= 0) & (ny_ < lny);
                        cur[i] = (level << 24) | ((ok ? ny_ : 0xfff) << 12) | (ok ? nx_ : 0xfff);
                    }
                    nf = 9;
                }
                lds_wave_sync();
            }
        }
#endif
        int nn = 0;
        float tight = INFINITY;
        const int rounds = -wave_min_i(-((nf * 4 + BFS_G - 1) >> BFS_SHIFT));
#ifdef PRL_CONE_TRACE
        rounds_ += rounds;
        widest_ = widest_ > nf ? widest_ : nf;
#endif
        for (int r = 0; r < rounds; ++r) {
            CONE_STAT(19, 1);
            const int c = r * BFS_G + m;
            const bool has = c < nf * 4;
            const int node = cur[has ? c >> 2 : 0];
            const int q = c & 3, px = 2 * (node & 0xfff) + (q & 1), py = 2 * ((node >> 12) & 0xfff) + (q >> 1);
            const bool in = has & (px < cnx) & (py < cny);
            const int n = off + (in ? py * cnx + px : 0);
            const f32x4 lo = ldg(boxes, 2 * n), hi = ldg(boxes, 2 * n + 1);      // (a copy of the upper levels in LDS: no faster)
            const double ax = (double)lo.x - pt[0], bx = pt[0] - (double)hi.x, ay = (double)lo.y - pt[1], by = pt[1] - (double)hi.y,
                         az = (double)lo.z - pt[2], bz = pt[2] - (double)hi.z;
            const double ex = fmax(fmax(ax, bx), 0.0), ey = fmax(fmax(ay, by), 0.0), ez = fmax(fmax(az, bz), 0.0);
            const double d2 = (ex * ex + ey * ey) + ez * ez;         // (an empty node: +inf)
            const bool keep = in & (d2 <= bound);
            if (keep & (lo.x <= hi.x)) {                              // the farthest corner of a box that holds samples
                const double fx = fmax(fabs(ax), fabs(bx)), fy = fmax(fabs(ay), fabs(by)), fz = fmax(fabs(az), fabs(bz));
                tight = fminf(tight, __double2float_ru((fx * fx + fy * fy) + fz * fz));
            }
            int word = (cl << 24) | (py << 12) | px;
            if (cl == 0) {                                           // a cell travels as its record range (sign bit | count << 22 | first)
                const int b = __float_as_int(lo.w), cnt = __float_as_int(hi.w) - b;
                if ((b < (1 << 22)) & (cnt < 512)) word = (int)(0x80000000u | ((unsigned)cnt << 22) | (unsigned)b);
            }
            const unsigned bits = (unsigned)(ballot64(keep) >> (BFS_G * g)) & ((1u << BFS_G) - 1u);
            const int slot = nn + __popc(bits & ((1u << m) - 1u));
            if (keep & (slot < BFS_CAP)) nxt[slot] = word;
            nn += __popc(bits);
        }
        over = over || nn > BFS_CAP;
        lds_wave_sync();
        int *t = cur;
        cur = nxt;
        nxt = t;
        nf = over ? 0 : nn;
        bound = fmin(bound, (double)group8_min(tight));
    }
#if defined(PRL_CONE_TRACE) && PRL_CONE_TRACE != 3
    CONE_HIST(3, rounds_);                               // rounds of the levels, widest frontier (cells included) of the wave
    widest_ = -wave_min_i(-(widest_ > nf ? widest_ : nf));
    CONE_HIST(4, (int)__popcll(ballot64(want && !(hint < INFINITY))) / 8 + 10 * (widest_ >= 12));     // searches without a hint; + 10: a wide wave
#endif
    // the cells that remain (a grid of no more than eight cells: all of them), shared by the lanes of the group
    {
        const int rounds = -wave_min_i(-((nf + BFS_G - 1) >> BFS_SHIFT));
        for (int r = 0; r < rounds; ++r) {
            const int c = r * BFS_G + m;
            const bool has = c < nf;
            const int word = cur[has ? c : 0];
            int b = word & 0x3fffff, cnt = has ? (word >> 22) & 0x1ff : 0;
            if (ballot64(has && word >= 0) != 0) {                    // (a range that did not fit its entry: looked up)
                if (has && word >= 0) {
                    const int cell = ((word >> 12) & 0xfff) * P.fg_nx + (word & 0xfff);
                    b = ldg(P.fg_start, cell);
                    cnt = ldg(P.fg_start, cell + 1) - b;
                }
            }
            for (int i0 = 0; ballot64(i0 < cnt) != 0; i0 += 4) {
                CONE_STAT(19, 1);
                f64x2 ra[4], rb[4];
#pragma unroll
                for (int q = 0; q < 4; ++q) {
                    const int i = b + (i0 + q < cnt ? i0 + q : (cnt > 0 ? cnt - 1 : 0));
                    ra[q] = ldg(rec, 2 * (cnt > 0 ? i : 0));
                    rb[q] = ldg(rec, 2 * (cnt > 0 ? i : 0) + 1);
                }
#pragma unroll
                for (int q = 0; q < 4; ++q) {
                    {   // (straight-line: a lane whose range is through measures its clamped record and is masked in the predicate)
                        const double dx = ra[q].x - pt[0], dy = ra[q].y - pt[1], dz = rb[q].x - pt[2];
                        const double dd = (dx * dx + dy * dy) + dz * dz;
                        const int rk = __double2loint(rb[q].y);
                        if ((i0 + q < cnt) & ((dd < best_d) | ((dd == best_d) & (rk < best_rank)))) {
                            best_d = dd;
                            best_rank = rk;
                            best_pos = __double2hiint(rb[q].y);
                        }
                    }
                }
            }
        }
    }
#pragma unroll
    for (int o = 1; o < BFS_G; o <<= 1) {                            // the group's best
        const double od = __shfl_xor(best_d, o);
        const int ork = __shfl_xor(best_rank, o), ops = __shfl_xor(best_pos, o);
        if ((od < best_d) | ((od == best_d) & (ork < best_rank))) {
            best_d = od;
            best_rank = ork;
            best_pos = ops;
        }
    }
    return want ? ((over || (best_pos < 0 && hint < INFINITY)) ? -2 : best_pos) : -1;       // (a hint that found nothing: never, but exact anyway)
}

// The search above for the points of a wave's groups, and nearest_sample_wave for a group whose frontier outgrew
// its list (or a part without the pyramid): the position in every lane of the group.
template <bool SEED = true>
__device__ __forceinline__ int nearest_sample_groups(PartRef P, const double pt[3], bool want, float hint, int lane, int *fr) {
    int pos = nearest_sample_bfs<SEED>(P, pt, want, hint, lane, fr);
    uint64_t ov = ballot64(want && pos == -2 && (lane & (BFS_G - 1)) == 0);
    while (ov) {
        const int L = __builtin_ctzll(ov);
        ov &= ov - 1;
        const double h3[3] = {bcast_d(pt[0], L), bcast_d(pt[1], L), bcast_d(pt[2], L)};
        const int s2 = nearest_sample_wave(P, h3, lane);
        if ((lane >> BFS_SHIFT) == (L >> BFS_SHIFT)) pos = s2;
    }
    return pos;
}

// What is left of the hit points of a wave's lanes (one per lane, sidx == -2) after the rings of the fine grid: BFS_N
// of them at a time through nearest_sample_groups.
__device__ __forceinline__ void nearest_sample_far(PartRef P, const double pt[3], int lane, int &sidx, int *fr) {
    uint64_t rest = ballot64(sidx == -2);
    const int g = lane >> BFS_SHIFT;
    while (rest) {
        uint64_t t = rest;
        int src = -1;                                                // the g-th of the lanes left is this group's
#pragma unroll
        for (int k = 0; k < BFS_N; ++k) {
            const int L = t ? (int)__builtin_ctzll(t) : -1;
            src = k == g ? L : src;
            t &= t - 1;                                              // (0 stays 0)
        }
        const uint64_t taken = rest & ~t;
        rest = t;
        const bool want = src >= 0;
        const int s = want ? src : 0;
        const double q3[3] = {__shfl(pt[0], s), __shfl(pt[1], s), __shfl(pt[2], s)};
        const int pos = nearest_sample_groups(P, q3, want, INFINITY, lane, fr);
        const bool me = (taken >> lane) & 1;
        const int got = __shfl(pos, me ? BFS_G * (int)__popcll(taken & ((1ull << lane) - 1)) : 0);
        if (me) sidx = got;
    }
}

// One trip of a shot: the beams b0 + lane of the cone at tool pose (pos, quat) -- the device position of the sample each
// lane's beam paints (bpw:562-566: the sample nearest to the hit point), or -1 (no such beam, or it misses the part).
// The general code: whatever the walk leaves over is searched (cone_rays_lanes), whatever the rings do not settle goes
// down the box pyramid (`fr`: the wave's LDS for nearest_sample_bfs).
__device__ __forceinline__ int cone_trip(PartRef P, const double pos[3], const double quat[4], int b0, int hint, int lane,
                                         int *cand_lds, int *fr) {
    double bh[3];
    const bool hit = cone_rays_lanes(P, pos, quat, b0, hint, lane, cand_lds, bh);
    int sidx = nearest_sample_lane(P, bh, hit);
    nearest_sample_far(P, bh, lane, sidx, fr);
    return sidx;
}

// The common case of cone_trip alone, for the beams kernel (k_cone_beams.hip): walk + the fine grid's three rings, nothing
// wave-wide and nothing long.  Per lane: `state` as cone_walk_lanes returns it (3: a ray the walk left over), `bh` the
// hit point if state = 1, `sidx` as cone_trip returns it where the lane is settled, or -2: a hit point centimetres from
// every sample (the collision hull spans a hole or a recess of the part there) -- `far_bound` then bounds its squared
// distance to the nearest sample from above (a sample the rings did see), or is +inf; `last_facet` (state 3): the facet
// the walk gave up on.
__device__ __forceinline__ void cone_trip_fast(PartRef P, const double pos[3], const double quat[4], int b0, int lane,
                                               int &state, double bh[3], int &sidx, float &far_bound, int &last_facet) {
    far_bound = INFINITY;
    double dst[3], t;
    state = cone_walk_lanes(P, pos, quat, b0, lane, dst, t, last_facet);
#ifdef PRL_CONE_OUTLINE_IN_BEAMS                      // (A/B switch: settles 40 % of the leftover rays here; beams kernel 138 -> 144 us, rest kernel the same)
    beams_outside_outline_lanes(P, pos, dst[0] - pos[0], dst[1] - pos[1], dst[2] - pos[2], state);
#endif
#ifdef PRL_FORCE_GENERAL_RAY                          // diagnostic build: every trip through the rest kernel's general code
    if (b0 + lane < P.n_beams) state = 3;
#endif
    bh[0] = pos[0] + t * (dst[0] - pos[0]);
    bh[1] = pos[1] + t * (dst[1] - pos[1]);
    bh[2] = pos[2] + t * (dst[2] - pos[2]);
#if defined(PRL_CONE_CUT_NN)                          // (timing build, wrong results: what the beams kernel takes without the search)
    sidx = state == 1 ? 0 : -1;
#elif defined(PRL_CONE_F64_RECORDS)                   // (A/B switch: the float64 records in the beams kernel too)
    sidx = nearest_sample_lane(P, bh, state == 1);
#else
    sidx = nearest_sample_lane_f32(P, bh, state == 1, far_bound);
#endif
    CONE_STAT(0, 1);
    CONE_STAT(11, __popcll(ballot64(sidx == -2)));
}

}  // namespace
