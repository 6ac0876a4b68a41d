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
//   * nearest samples: nearest_sample_lane below, one hit point per lane; the hit bits meet in the wave's LDS mask row.
#pragma once

namespace {

#ifndef PRL_WALK_STEPS
#define PRL_WALK_STEPS 8
#endif
constexpr int CONE_WALK_STEPS = PRL_WALK_STEPS;
#ifndef PRL_CONE_JOINT_FROM
#define PRL_CONE_JOINT_FROM 5
#endif
constexpr int CONE_JOINT_FROM = PRL_CONE_JOINT_FROM;
#ifndef PRL_CONE_FAR_K0
#define PRL_CONE_FAR_K0 4
#endif
constexpr int CONE_FAR_K0 = PRL_CONE_FAR_K0;
#define FAR_BAND 4.0e-6f              // m^2, see nearest_samples_shared             // first widening (fine cells) of the shared far scan     // more stragglers than this in a trip: searched together
#define CONE_MISS_MARGIN 1.0e-6      // metres clear of a separating facet plane (triangle tolerances are ~1e-9 of an edge)

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
    if (!entering) {
        // The walk has gone over the hull's horizon (or grazes).  The hull lies on the inner side of this facet's
        // plane: if both ends of the beam are clear of the plane on the OUTER side, by more than any tolerance of the
        // triangle tests, the beam misses the whole set -- no search needed.  (Most beams that miss pass beside the
        // part; the walk ends on a rim facet whose plane separates them.)
        const double n0 = e11 * e22 - e12 * e21, n1 = e12 * e20 - e10 * e22, n2 = e10 * e21 - e11 * e20;   // e1 x e2
        const double so = ((s0 * n0 + s1 * n1) + s2 * n2) * orient;                     // origin, outward if > 0
        const double se = so + ((d0 * n0 + d1 * n1) + d2 * n2) * orient;                // end point
        const double clear = CONE_MISS_MARGIN * CONE_MISS_MARGIN * nn;                  // (distance margin)^2 |n|^2
        if (so > 0 && se > 0 && so * so > clear && se * se > clear) return 2;
        return -1;
    }
    if (!(t >= 0.0)) return -1;                              // behind the origin
    if (u >= m && v >= m && (u + v) <= 1.0 - m) {
        // entered at an interior point: the closest hit of the whole set (ray_closest_wave's single-facet criterion);
        // beyond the beam's end point it means the beam stops short of the part: a miss
        if (!(t <= 1.0)) return 2;
        t_out = t;
        rank_out = ldg(P.col_rank, i);
        return 1;
    }
    // outside the triangle (or within the edge margin): cross the edge that is violated most
    const double w = 1.0 - u - v;
    const int e = (u <= v && u <= w) ? 0 : ((v <= w) ? 1 : 2);         // 0: u smallest, 1: v, 2: w = 1 - u - v
    next = ldg(P.col_enbr, 3 * i + e);
    return 0;
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
            if (ballot64(state == 0) == 0) break;
            if (state == 0) {
                int next;
                const int r = cone_walk_step(P, f, pos, d0, d1, d2, dd, t, rk, next);
                if (r == 1) state = 1;
                else if (r == 2) state = 2;
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
    // Beams beside the part.  The collision set lies inside the slab [slab_lo, slab_hi] of the third axis and, projected
    // to the principal plane, inside its outline polygon: if the stretch of the beam inside the slab (widened by the
    // margin) lies, projected, more than the margin outside one edge of the outline, the beam misses every triangle.
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
    // the stragglers (edge and vertex hits, and every miss), together
    const uint64_t todo = ballot64(state == 3);
    WCNT16(0, __popcll(todo));
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
// index -- as nearest_sample_wave, but 64 queries at once: each lane scans the (2k+1)^2 block of FINE grid cells around
// its point (PartDev::fg_*: ~4 samples a cell, so a 3 x 3 block holds ~36 candidates where the painter's own sample
// grid holds ~1 350), rows as contiguous record ranges, four records per trip so that their reads travel together.
// A sample outside the block is more than k cells away in the principal plane: the best of the block is the answer
// once it lies within k * 0.99 * cell.  Returns the device position of the sample, -1 if not `want`, or -2 if three
// rings did not settle it (a hit point centimetres off the sampled surface): the caller asks nearest_sample_wave.
__device__ int nearest_sample_lane(PartRef P, const double pt[3], bool want) {
    const double h1 = sel3(pt[0], pt[1], pt[2], P.a1), h2 = sel3(pt[0], pt[1], pt[2], P.a2);
    const int icx = cell_coord(h1, P.fg_o1, P.fg_inv, P.fg_nx), icy = cell_coord(h2, P.fg_o2, P.fg_inv, P.fg_ny);
    const f64x2 GAS *rec = reinterpret_cast<const f64x2 GAS *>(P.fg_rec);
    int result = want ? -2 : -1;
    bool open = want;
    for (int ring = 1; ring <= 3; ++ring) {
        if (ballot64(open) == 0) break;
        double best_d = INFINITY;
        int best_rank = 0x7fffffff, best_pos = -1;
        const int cx0 = icx - ring < 0 ? 0 : icx - ring, cx1 = icx + ring > P.fg_nx - 1 ? P.fg_nx - 1 : icx + ring;
        for (int dy = -ring; dy <= ring; ++dy) {                    // wave-uniform trip count, per-lane ranges
            const int cy = icy + dy;
            const bool row_ok = open && cy >= 0 && cy < P.fg_ny && cx0 <= cx1;
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
        const double lim = ring * P.fg_accept;
        if (open && best_pos >= 0 && best_d <= lim * lim) {
            result = best_pos;
            open = false;
        }
    }
#ifdef PRL_FORCE_FULL_SCANS                          // diagnostic build: every query through the wave-wide search
    if (want) result = -2;
#endif
    return result;
}

// ---------------------------------------------------------------- nearest samples of points FAR from the sampled surface
// Where the collision hull spans a recess of the part, every hit of a shot lies centimetres above the samples: three
// rings of the fine grid do not settle any of them, and a wave-wide search per hit (4 us each, ~520 a step) made such
// an env five times slower than the rest -- and the launch waits for it.  Here the needy lanes share one scan: the block
// of fine cells around ALL their points, widened by `k` cells, is read once (64 records at a time, one per lane, handed
// round by lane broadcast) and every needy lane measures every record against its own point.  A lane whose cell is at
// least r cells inside the block (a block side on the grid's own border counts as infinitely far) has seen every sample
// closer than r cells: its best is exact once within r * 0.99 * cell.  k = 4, 8, 16 (measured: 664 steps/s; from 8: 641; from 6: 340), then the wave-wide search.
// (Tried: a small first block whose best distance sizes the second -- slower, 468 vs 696 steps/s: over a hole in the
// part the bound is loose and the sized block larger than the one doubling reaches.)
__device__ __attribute__((noinline)) void nearest_samples_shared(PartRef P, const double pt[3], int lane, int &sidx) {
    const double h1 = sel3(pt[0], pt[1], pt[2], P.a1), h2 = sel3(pt[0], pt[1], pt[2], P.a2);
    const int icx = cell_coord(h1, P.fg_o1, P.fg_inv, P.fg_nx), icy = cell_coord(h2, P.fg_o2, P.fg_inv, P.fg_ny);
    const f64x2 GAS *rec = reinterpret_cast<const f64x2 GAS *>(P.fg_rec);
    for (int k = CONE_FAR_K0; k <= 4 * CONE_FAR_K0; k *= 2) {
        const bool need = sidx == -2;
        if (ballot64(need) == 0) return;
        const int ccx = icx < 0 ? 0 : (icx > P.fg_nx - 1 ? P.fg_nx - 1 : icx), ccy = icy < 0 ? 0 : (icy > P.fg_ny - 1 ? P.fg_ny - 1 : icy);
        int bx0 = wave_min_i(need ? ccx : 0x7fffffff) - k, bx1 = -wave_min_i(need ? -ccx : 0x7fffffff) + k;
        int by0 = wave_min_i(need ? ccy : 0x7fffffff) - k, by1 = -wave_min_i(need ? -ccy : 0x7fffffff) + k;
        const bool open_x0 = bx0 <= 0, open_x1 = bx1 >= P.fg_nx - 1, open_y0 = by0 <= 0, open_y1 = by1 >= P.fg_ny - 1;
        bx0 = bx0 < 0 ? 0 : bx0;
        by0 = by0 < 0 ? 0 : by0;
        bx1 = bx1 > P.fg_nx - 1 ? P.fg_nx - 1 : bx1;
        by1 = by1 > P.fg_ny - 1 ? P.fg_ny - 1 : by1;
        double best_d = INFINITY;
        int best_rank = 0x7fffffff, best_pos = -1;
        // float pre-test: a record is measured in float64 only if, in float, it comes within FAR_BAND of some needy lane's
        // best so far.  Coordinates are below 2 m and differences below 1 m here, so a float squared distance is within
        // 1e-6 m^2 of the exact one (rounding the six coordinates: 6 x 1.2e-7 x 2 x 1; the arithmetic: 1e-7): a record
        // the test drops is farther than the lane's best and could neither replace nor tie it.
        const float pfx = (float)pt[0], pfy = (float)pt[1], pfz = (float)pt[2];
        float bestf = INFINITY;                                                  // >= best_d
        for (int cy = by0; cy <= by1; ++cy) {                                   // wave-uniform loops
            const int b = P.fg_start[cy * P.fg_nx + bx0], e = P.fg_start[cy * P.fg_nx + bx1 + 1];
            for (int i0 = b; i0 < e; i0 += 64) {
                const int i = i0 + lane < e ? i0 + lane : e - 1;
                const f64x2 ra = ldg(rec, 2 * i), rb = ldg(rec, 2 * i + 1);
                const float fx = (float)ra.x, fy = (float)ra.y, fz = (float)rb.x;
                const int n = e - i0 < 64 ? e - i0 : 64;
                for (int j = 0; j < n; ++j) {
                    const float ex = bcast_f(fx, j) - pfx, ey = bcast_f(fy, j) - pfy, ez = bcast_f(fz, j) - pfz;
                    const float ddf = ex * ex + ey * ey + ez * ez;
                    if (ballot64(need && ddf <= bestf + FAR_BAND) == 0) continue;
                    const double x = bcast_d(ra.x, j), y = bcast_d(ra.y, j), z = bcast_d(rb.x, j);
                    const int rk = __builtin_amdgcn_readlane(__double2loint(rb.y), j);
                    const int ps = __builtin_amdgcn_readlane(__double2hiint(rb.y), j);
                    const double dx = x - pt[0], dy = y - pt[1], dz = z - pt[2];
                    const double dd = (dx * dx + dy * dy) + dz * dz;
                    if (dd < best_d || (dd == best_d && rk < best_rank)) {
                        best_d = dd;
                        best_rank = rk;
                        best_pos = ps;
                        bestf = nextafterf((float)dd, INFINITY);
                    }
                }
            }
        }
        // cells between this lane's (clamped) cell and the nearest CLOSED side of the block
        int r = 0x7fffffff;
        if (!open_x0) r = ccx - bx0 < r ? ccx - bx0 : r;
        if (!open_x1) r = bx1 - ccx < r ? bx1 - ccx : r;
        if (!open_y0) r = ccy - by0 < r ? ccy - by0 : r;
        if (!open_y1) r = by1 - ccy < r ? by1 - ccy : r;
        // (a point outside the grid is farther from every sample beyond the block than its clamped cell is)
        const double lim = (r == 0x7fffffff ? 1.0e30 : (double)r * P.fg_accept);
        if (need && best_pos >= 0 && best_d <= lim * lim) sidx = best_pos;
    }
}

// One trip of a shot: the beams b0 + lane of the cone at tool pose (pos, quat) -- the device position of the sample each
// lane's beam paints (bpw:562-566: the sample nearest to the hit point), or -1 (no such beam, or it misses the part).
__device__ __forceinline__ int cone_trip(PartRef P, const double pos[3], const double quat[4], int b0, int hint, int lane,
                                         int *cand_lds) {
    double bh[3];
    const bool hit = cone_rays_lanes(P, pos, quat, b0, hint, lane, cand_lds, bh);
    WCNT16(3, __popcll(ballot64(hit)));
    // nearest sample of every hit point: one query per lane; the few that the fine grid does not settle go through
    // the wave-wide search
    int sidx = nearest_sample_lane(P, bh, hit);
    if (__popcll(ballot64(sidx == -2)) > 3) nearest_samples_shared(P, bh, lane, sidx);   // (a recess of the part)
    uint64_t rest = ballot64(sidx == -2);
    WCNT16(1, __popcll(rest));
    while (rest) {
        const int L = __builtin_ctzll(rest);
        rest &= rest - 1;
        const double h3[3] = {bcast_d(bh[0], L), bcast_d(bh[1], L), bcast_d(bh[2], L)};
        const int s2 = nearest_sample_wave(P, h3, lane);
        if (lane == L) sidx = s2;
    }
    return sidx;
}

}  // namespace
