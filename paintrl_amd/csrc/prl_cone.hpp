// prl_cone.hpp -- PAINT_METHOD 'normal': the cone beams of one shot, one beam per lane (rob:251-285, bpw:562-566).
// Part of the single translation unit paintrl_hip.hip (device code, anonymous namespace); see that file for the
// overall design.  Compile with -ffp-contract=off.
//
// A shot casts n_beams (104-140) rays from the tool position to the points of a lattice on the plane 0.2 ahead
// (rob:23-35); every ray that hits the part paints the sample nearest to its hit point.  The first version ran the
// wave-wide ray search and the wave-wide nearest-sample search once per beam, one after the other (3.1 ms per batched
// step).  Here a wave takes 64 beams at a time, one per lane:
//   * rays: on a convex collision set (hull mode) a lane WALKS the hull surface: it tests its ray against one
//     facet (starting from the facet the tool's own ray hit), and while the ray meets that facet's plane outside the
//     triangle it steps across the violated edge to the neighbouring facet (three edge neighbours per facet, derived
//     on upload).  A facet that is ENTERED at a point clear of its edges ends the walk with the exact closest hit of
//     the whole set -- the criterion of ray_closest_wave's single-facet path, same arithmetic.  Lanes that do not
//     get there in a few steps (edge and vertex hits, misses, rays leaving the hull) take the wave-wide search.
//   * nearest samples: see nearest_samples_lanes below.
#pragma once

namespace {

constexpr int CONE_WALK_STEPS = 6;

// One step of the walk for this lane's ray (origin o, direction d, |d|^2 = dd) on facet i (>= 0): Moller-Trumbore on
// the facet record, arithmetic as in mt_rec.  Returns 1 = entered at an interior point (t, exact closest hit),
// 0 = keep walking (`next` = the facet across the most violated edge, or -1), -1 = give up (degenerate / behind).
__device__ __forceinline__ int cone_walk_step(PartRef P, int i, const double o[3], double d0, double d1, double d2,
                                              double dd, double &t_out, int &rank_out, int &next) {
    const f64x2 GAS *r = reinterpret_cast<const f64x2 GAS *>(P.col_rec);
    const int i6 = i * 6;
    const f64x2 r0 = ldg(r, i6), r1 = ldg(r, i6 + 1), r2 = ldg(r, i6 + 2), r3 = ldg(r, i6 + 3), r4 = ldg(r, i6 + 4),
                r5 = ldg(r, i6 + 5);
    const double v00 = r0.x, v01 = r0.y, v02 = r1.x, e10 = r1.y, e11 = r2.x, e12 = r2.y;
    const double e20 = r3.x, e21 = r3.y, e22 = r4.x, m = r4.y, nn = r5.x, orient = r5.y;
    const double p0 = d1 * e22 - d2 * e21;
    const double p1 = d2 * e20 - d0 * e22;
    const double p2 = d0 * e21 - d1 * e20;
    const double det = (e10 * p0 + e11 * p1) + e12 * p2;
    next = -1;
    if (!(fabs(det) >= RAY_EPS_DET)) return -1;
    const double inv = 1.0 / det;
    const double s0 = o[0] - v00, s1 = o[1] - v01, s2 = o[2] - v02;
    const double u = ((s0 * p0 + s1 * p1) + s2 * p2) * inv;
    const double q0 = s1 * e12 - s2 * e11;
    const double q1 = s2 * e10 - s0 * e12;
    const double q2 = s0 * e11 - s1 * e10;
    const double v = ((d0 * q0 + d1 * q1) + d2 * q2) * inv;
    const double t = ((e20 * q0 + e21 * q1) + e22 * q2) * inv;
    const bool entering = orient * det > 0 && det * det >= FACET_MIN_COS2 * dd * nn;
    if (!entering || !(t >= 0.0)) return -1;               // leaving through this facet, grazing, or behind the origin
    if (u >= m && v >= m && (u + v) <= 1.0 - m) {
        if (!(t <= 1.0)) return -1;                          // the facet lies beyond the beam's end point: a miss, let
        t_out = t;                                           // the general search confirm it
        rank_out = ldg(P.col_rank, i);
        return 1;
    }
    // outside the triangle (or within the edge margin): cross the edge that is violated most
    const double w = 1.0 - u - v;
    const int e = (u <= v && u <= w) ? 0 : ((v <= w) ? 1 : 2);         // 0: u smallest, 1: v, 2: w = 1 - u - v
    next = ldg(P.col_enbr, 3 * i + e);
    return 0;
}

// The rays of beams b0 + lane: hit[3] / t of this lane's beam, returns whether it hit.  `hint`: facet of the tool's ray.
__device__ bool cone_rays_lanes(PartRef P, const double pos[3], const double quat[4], int b0, int hint, int lane,
                                int *cand_lds, double hit[3]) {
    const int bm = b0 + lane;
    const bool have = bm < P.n_beams;
    double dst[3] = {pos[0], pos[1], pos[2]};
    if (have) transform_point(pos, quat, ldg(P.beams, 3 * bm), ldg(P.beams, 3 * bm + 1), ldg(P.beams, 3 * bm + 2), dst);
    const double d0 = dst[0] - pos[0], d1 = dst[1] - pos[1], d2 = dst[2] - pos[2];
    const double dd = (d0 * d0 + d1 * d1) + d2 * d2;
    double t = INFINITY;
    int state = have ? 0 : 2;                       // 0 walking, 1 hit, 2 finished without a hit (no beam), 3 -> wave-wide search
    if (P.col_convex && hint >= 0) {
        int f = hint, rk = 0;
        for (int step = 0; step < CONE_WALK_STEPS; ++step) {
            if (__ballot(state == 0) == 0) break;
            if (state == 0) {
                int next;
                const int r = cone_walk_step(P, f, pos, d0, d1, d2, dd, t, rk, next);
                if (r == 1) state = 1;
                else if (r < 0 || next < 0) state = 3;
                else f = next;
            }
        }
        if (state == 0) state = 3;
    } else if (have) {
        state = 3;
    }
#ifdef PRL_FORCE_GENERAL_RAY
    if (have) state = 3;
#endif
    hit[0] = pos[0] + t * d0;
    hit[1] = pos[1] + t * d1;
    hit[2] = pos[2] + t * d2;
    // the stragglers, one wave-wide search each
    uint64_t todo = __ballot(state == 3);
    int wide_hint = hint;
    while (todo) {
        const int L = __builtin_ctzll(todo);
        todo &= todo - 1;
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
    return state == 1;
}

}  // namespace
