// prl_ray.hpp -- closest two-sided ray hit against the collision triangles (rayTestBatch semantics).
// Part of the single translation unit paintrl_hip.hip (device code, anonymous namespace); see that
// file for the overall design.  Compile with -ffp-contract=off.
#pragma once

namespace {

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

// A float at or below / at or above a double, for culling boxes and thresholds that only have to be CAUTIOUS: the nearest
// float moved outward by 2^-22 of its magnitude (its own rounding error is at most 2^-24 of it) and a tiny absolute step for
// zero and denormals.  Four instructions; nextafterf is ~43, and a ray that takes the general search paid 12 of them (a tool
// that has left the part: ten searches a step, every one a miss).  Infinite or NaN input gives +-FLT_MAX: nothing is culled.
// (-DPRL_EXACT_OUTWARD: the nextafterf form, A/B switch.)
__device__ __forceinline__ float f32_at_or_below(double x) {
#ifdef PRL_EXACT_OUTWARD
    return nextafterf((float)x, -INFINITY);
#else
    const float f = (float)x;
    return fmaxf(f - __builtin_fmaf(fabsf(f), 2.384185791015625e-07f, 1.0e-37f), -3.402823466e+38f);
#endif
}
__device__ __forceinline__ float f32_at_or_above(double x) {
#ifdef PRL_EXACT_OUTWARD
    return nextafterf((float)x, INFINITY);
#else
    const float f = (float)x;
    return fminf(f + __builtin_fmaf(fabsf(f), 2.384185791015625e-07f, 1.0e-37f), 3.402823466e+38f);
#endif
}

__device__ __forceinline__ SegBox seg_box(const double o3[3], const double d3[3], double tmax) {
    SegBox b;
#pragma unroll
    for (int k = 0; k < 3; ++k) {
        const double e = o3[k] + tmax * d3[k];
        b.lo[k] = f32_at_or_below(fmin(o3[k], e));
        b.hi[k] = f32_at_or_above(fmax(o3[k], e));
    }
    return b;
}

__device__ __forceinline__ bool box_overlap(const SegBox &s, const f32x4 a, const f32x4 b) {
    return (s.lo[0] <= a.y) && (s.hi[0] >= a.x) && (s.lo[1] <= a.w) && (s.hi[1] >= a.z) && (s.lo[2] <= b.y) &&
           (s.hi[2] >= b.x);
}

// 1.0 / d for a determinant that passed |d| >= RAY_EPS_DET (and is far below 1e300): the compiler's float64 division is
// v_div_scale x 2, v_rcp, four fma, v_mul, fma, v_div_fmas, v_div_fixup; the scale / fixup steps only rescale operands whose
// quotient would leave the normal range and are the identity here, so reciprocal + two Newton steps + the correcting fma pair
// is the same sequence without them: bit for bit the same quotient in 7 instructions instead of 13 (-DPRL_PLAIN_DIVISION: A/B
// and parity switch; the oracle's C division is the reference either way).
__device__ __forceinline__ double rcp_det(double d) {
#ifdef PRL_PLAIN_DIVISION
    return 1.0 / d;
#else
    double r = __builtin_amdgcn_rcp(d);
    double e = __builtin_fma(-d, r, 1.0);
    r = __builtin_fma(r, e, r);
    e = __builtin_fma(-d, r, 1.0);
    r = __builtin_fma(r, e, r);
    const double rem = __builtin_fma(-d, r, 1.0);
    return __builtin_fma(rem, r, r);
#endif
}

#define CONE_MISS_MARGIN 1.0e-6      // metres clear of a separating facet plane / outline edge (triangle tolerances are ~1e-9 of an edge)
// The outline test for ONE segment (pos -> dst, the same on every lane), the edges of the outline over the lanes: whether the
// segment passes beside the collision set (then it misses every triangle).  The set lies inside the slab [slab_lo, slab_hi] of
// the third axis and, projected to the principal plane, inside its outline polygon (PartDev::outline, derived on upload): if
// the stretch of the segment inside the slab (widened by the margin) lies, projected, more than the margin outside one edge
// of the outline, no triangle can report a hit (their tolerances are ~1e-9 of an edge).  Used by the cone beams' leftover
// rays (prl_cone.hpp has the one-beam-per-lane form) and by the general search below.
__device__ __forceinline__ bool beam_outside_outline_wave(PartRef P, const double pos[3], const double dst[3], int lane) {
    if (P.n_outline <= 0) return false;
    const double d0 = dst[0] - pos[0], d1 = dst[1] - pos[1], d2 = dst[2] - pos[2];
    const double oz = sel3(pos[0], pos[1], pos[2], P.a0), dz = sel3(d0, d1, d2, P.a0);
    const double lo = P.slab_lo - CONE_MISS_MARGIN, hi = P.slab_hi + CONE_MISS_MARGIN;
    double ta = 0.0, tb = 1.0;
    if (dz != 0.0) {
        const double t0 = (lo - oz) / dz, t1 = (hi - oz) / dz;
        ta = fmax(0.0, fmin(t0, t1) - 1e-9);
        tb = fmin(1.0, fmax(t0, t1) + 1e-9);
        if (ta > tb) return true;                             // never inside the slab
    } else if (oz < lo || oz > hi) {
        return true;
    }
    const double o1 = sel3(pos[0], pos[1], pos[2], P.a1), o2 = sel3(pos[0], pos[1], pos[2], P.a2);
    const double e1 = sel3(d0, d1, d2, P.a1), e2 = sel3(d0, d1, d2, P.a2);
    const double ax = o1 + ta * e1, ay = o2 + ta * e2, bx = o1 + tb * e1, by = o2 + tb * e2;
    const f64x2 GAS *ol = reinterpret_cast<const f64x2 GAS *>(P.outline);
    for (int base = 0; base < P.n_outline; base += 64) {      // (the table is padded to a multiple of 64 rows)
        const f64x2 pq = ldg(ol, 2 * (base + lane)), nq = ldg(ol, 2 * (base + lane) + 1);
        const double sa = nq.x * (ax - pq.x) + nq.y * (ay - pq.y), sb = nq.x * (bx - pq.x) + nq.y * (by - pq.y);
        if (ballot64(base + lane < P.n_outline && sa > CONE_MISS_MARGIN && sb > CONE_MISS_MARGIN) != 0) return true;
    }
    return false;
}

// The triangles whose own box passes are first compacted (their ids go to a per-wave LDS list, slot =
// running count + number of passing lanes below), then the float64 test runs ONCE over the list
// with one candidate per lane, instead of once per visited chunk with a handful of active lanes.
// One float64 Moller-Trumbore test per lane (triangle i, or none if i < 0); keeps the lane's best
// (t, reference rank) and remembers which triangle and which determinant produced it.
__device__ __forceinline__ void mt_one(PartRef P, int i, const double o[3], double d0, double d1, double d2,
                                       double tmax, double &best_t, int &best_r, int &best_i, double &best_det) {
    if (i >= 0) {
        const double v00 = ldg(P.col[0], i), v01 = ldg(P.col[1], i), v02 = ldg(P.col[2], i);
        const double e10 = ldg(P.col[3], i), e11 = ldg(P.col[4], i), e12 = ldg(P.col[5], i);
        const double e20 = ldg(P.col[6], i), e21 = ldg(P.col[7], i), e22 = ldg(P.col[8], i);
        const int rk = ldg(P.col_rank, i);
        const double p0 = d1 * e22 - d2 * e21;
        const double p1 = d2 * e20 - d0 * e22;
        const double p2 = d0 * e21 - d1 * e20;
        const double det = (e10 * p0 + e11 * p1) + e12 * p2;
        if (fabs(det) >= RAY_EPS_DET) {
            const double inv = rcp_det(det);
            const double s0 = o[0] - v00, s1 = o[1] - v01, s2 = o[2] - v02;
            const double u = ((s0 * p0 + s1 * p1) + s2 * p2) * inv;
            const double q0 = s1 * e12 - s2 * e11;
            const double q1 = s2 * e10 - s0 * e12;
            const double q2 = s0 * e11 - s1 * e10;
            const double v = ((d0 * q0 + d1 * q1) + d2 * q2) * inv;
            const double t = ((e20 * q0 + e21 * q1) + e22 * q2) * inv;
            if ((u >= -RAY_EPS_BARY) & (v >= -RAY_EPS_BARY) & ((u + v) <= 1.0 + RAY_EPS_BARY) & (t >= 0.0) & (t <= tmax) &
                ((t < best_t) | ((t == best_t) & (rk < best_r)))) {
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
__device__ __forceinline__ void mt_rec_core(int i, const f64x2 r0, const f64x2 r1, const f64x2 r2, const f64x2 r3, const f64x2 r4,
                                            const f64x2 r5, int rk, const double o[3], double d0, double d1, double d2, double dd,
                                            double &best_t, int &best_r, int &best_i, double &best_det, bool &interior) {
    // Straight-line on purpose: every lane evaluates the whole test (a lane without a facet on whatever record it was handed, a
    // degenerate determinant on infinities and NaNs that fail every comparison) and the conditions meet in ONE lane predicate
    // at the end -- nested per-lane `if`s are regions of their own (exec mask saved, branch, restored: three per round here).
    const double v00 = r0.x, v01 = r0.y, v02 = r1.x, e10 = r1.y, e11 = r2.x, e12 = r2.y;
    const double e20 = r3.x, e21 = r3.y, e22 = r4.x, m = r4.y, nn = r5.x, orient = r5.y;
    const double p0 = d1 * e22 - d2 * e21;
    const double p1 = d2 * e20 - d0 * e22;
    const double p2 = d0 * e21 - d1 * e20;
    const double det = (e10 * p0 + e11 * p1) + e12 * p2;
    const double inv = rcp_det(det);
    const double s0 = o[0] - v00, s1 = o[1] - v01, s2 = o[2] - v02;
    const double u = ((s0 * p0 + s1 * p1) + s2 * p2) * inv;
    const double q0 = s1 * e12 - s2 * e11;
    const double q1 = s2 * e10 - s0 * e12;
    const double q2 = s0 * e11 - s1 * e10;
    const double v = ((d0 * q0 + d1 * q1) + d2 * q2) * inv;
    const double t = ((e20 * q0 + e21 * q1) + e22 * q2) * inv;
    const bool hit = (i >= 0) & (fabs(det) >= RAY_EPS_DET) & (u >= -RAY_EPS_BARY) & (v >= -RAY_EPS_BARY) & ((u + v) <= 1.0 + RAY_EPS_BARY) &
                     (t >= 0.0) & (t <= 1.0) & ((t < best_t) | ((t == best_t) & (rk < best_r)));
    interior = hit & (u >= m) & (v >= m) & ((u + v) <= 1.0 - m) & (orient * det > 0) & (det * det >= FACET_MIN_COS2 * dd * nn);
    if (hit) {
        best_t = t;
        best_r = rk;
        best_i = i;
        best_det = det;
    }
}

__device__ __forceinline__ void mt_rec(PartRef P, int i, const double o[3], double d0, double d1, double d2, double dd,
                                       double &best_t, int &best_r, int &best_i, double &best_det, bool &interior) {
    // (a lane without a facet reads record 0: one load sequence for the wave, no region under a per-lane condition)
    const f64x2 GAS *r = reinterpret_cast<const f64x2 GAS *>(P.col_rec);
    const int ic = i >= 0 ? i : 0, i6 = ic * 6;
    const f64x2 r0 = ldg(r, i6), r1 = ldg(r, i6 + 1), r2 = ldg(r, i6 + 2), r3 = ldg(r, i6 + 3), r4 = ldg(r, i6 + 4), r5 = ldg(r, i6 + 5);
    const int rk = ldg(P.col_rank, ic);
    mt_rec_core(i, r0, r1, r2, r3, r4, r5, rk, o, d0, d1, d2, dd, best_t, best_r, best_i, best_det, interior);
}

// ---------------------------------------------------------------- the facet tile (prl_device.hpp FacetTile)
typedef __attribute__((address_space(3))) void *lds_void_p;

// lane l's entry of the vertex neighbourhood of `facet` (itself in lane 0): the one load the records' addresses depend on
__device__ __forceinline__ int tile_ids_load(PartRef P, int facet, int lane) {
    return lane < P.nbr_width ? ldg(P.col_nbr, facet * P.nbr_width + lane) : -1;
}

// Starts the copy of the records (and reference indices) of `ids` into the tile: seven LDS-DMA instructions that write
// wave-uniform base + lane * size, no vector register in between; the tile may be read after `s_waitcnt vmcnt(0)`
// (tile_wait).  The caller guarantees P.nbr_width <= TILE_LANES.
__device__ __forceinline__ void tile_fill(PartRef P, FacetTile *T, int facet, int ids, int lane) {
    if (lane < TILE_LANES) {
        T->id[lane] = ids;
        if (ids >= 0) {
            const f64x2 GAS *src = reinterpret_cast<const f64x2 GAS *>(P.col_rec) + (uint32_t)ids * 6u;
#pragma unroll
            for (int c = 0; c < 6; ++c)
                __builtin_amdgcn_global_load_lds(reinterpret_cast<const GAS void *>(src + c), (lds_void_p)&T->rec[c][0], 16, 0, 0);
            __builtin_amdgcn_global_load_lds(reinterpret_cast<const GAS void *>(P.col_rank + (uint32_t)ids), (lds_void_p)&T->rank[0], 4, 0, 0);
        }
    }
    if (lane == 0) T->facet = facet;
}

__device__ __forceinline__ void tile_wait() {
    asm volatile("s_waitcnt vmcnt(0) lgkmcnt(0)" ::: "memory");
    __builtin_amdgcn_wave_barrier();
}

// what a ray leaves for the hook-point search to start (sub_shot): the next tile's facet and its neighbour ids
struct TilePrefetch {
    int facet;       // -1: nothing to fetch
    int ids;         // per lane
};

// Lane holding the wave's best (t, rank); -1 if no lane has a hit.
__device__ __forceinline__ int ray_winner_lane(double best_t, int best_r, double &tmin) {
    if (ballot64(best_t < INFINITY) == 0) return -1;
    tmin = wave_min_nonneg_d(best_t + 0.0);        // t >= 0 or +inf (no hit in this lane); + 0.0 turns a -0.0 into +0.0
    const uint64_t tie = ballot64(best_t == tmin);
    if ((tie & (tie - 1)) == 0) return __builtin_ctzll(tie);
    const int rmin = wave_min_i(best_t == tmin ? best_r : 0x7fffffff);        // equal t: lowest reference index
    return __builtin_ctzll(ballot64((best_t == tmin) & (best_r == rmin)));
}

// Returns the collision-set position of the facet hit (= the new `hint`; its reference index is PartDev::col_rank of it), or -1.
// `hint` (in/out): collision-set position of the facet hit by the previous ray of this env, or -1.
//
// Convex fast path (collision set = boundary of a convex polytope, i.e. hull mode): a segment that
// starts outside enters the polytope at one point, so every facet with a valid hit at or before the
// entry parameter contains that point and therefore shares a vertex with any one of them.  If the
// vertex-neighbourhood of `hint` holds a valid hit whose facet is ENTERED (orient * det > 0), the
// closest hit of the whole set is the best over that facet's own neighbourhood.  Anything else (no
// hit there, an exit hit, a facet without a neighbour list) takes the general search below.
// `cand`: this wave's 64-int LDS row (WaveLds::cand).
// `tile` / `pf` (optional): this wave's facet tile and where to leave the next one's prefetch.  With a tile the facet of `hint`
// and the facets around it are tested in ONE pass on LDS-resident records -- lane 0 holds `hint` itself, so the pass
// subsumes test (1) below: an interior entry in any lane is the closest hit of the whole set, same arithmetic.
__device__ int ray_closest_wave(PartRef P, const double o[3], const double e[3], int lane, double &t_out,
                                double hit[3], int &hint, int *cand, FacetTile *tile = nullptr, TilePrefetch *pf = nullptr) {
    const double d0 = e[0] - o[0], d1 = e[1] - o[1], d2 = e[2] - o[2];
    double best_t = INFINITY, best_det = 0, tmin = INFINITY;
    int best_r = 0x7fffffff, best_i = -1, win = -1;
    // (the ray before this one missed: this one probably does too -- the general search then takes the whole segment at once)
    bool after_a_miss = hint < 0;
#ifdef PRL_FORCE_GENERAL_RAY                         // diagnostic build: never take the convex fast path
    hint = -1;
    after_a_miss = false;                            // (... and both stages of the general search)
#endif
#ifndef PRL_FACET_TILE                               // A/B switch: the LDS tile lost to the scalar test + global rounds
    tile = nullptr;                                  // (profiles/r04_ab_log.txt: 42.0 against 40.2 us), so it is off
#endif
    if (tile && P.nbr_width > TILE_LANES) tile = nullptr;
    if (pf) pf->facet = -1;
    if (P.col_convex && hint >= 0 && tile) {
        // (lane 0 wrote `facet` and the tile's ids in tile_fill, every lane reads them: order the two within the wave)
        __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
        __builtin_amdgcn_wave_barrier();
        __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
        if (rfl(tile->facet) != hint) tile_fill(P, tile, hint, tile_ids_load(P, hint, lane), lane);       // (not prefetched)
        tile_wait();
        WCNT(4, 1);
        const double dd = (d0 * d0 + d1 * d1) + d2 * d2;
        int i1 = -1, rk = 0;
        f64x2 r0 = {0, 0}, r1 = r0, r2 = r0, r3 = r0, r4 = r0, r5 = r0;
        if (lane < TILE_LANES) {
            i1 = tile->id[lane];
            rk = tile->rank[lane];
            r0 = tile->rec[0][lane], r1 = tile->rec[1][lane], r2 = tile->rec[2][lane], r3 = tile->rec[3][lane], r4 = tile->rec[4][lane],
            r5 = tile->rec[5][lane];
        }
        bool interior;
        mt_rec_core(i1, r0, r1, r2, r3, r4, r5, rk, o, d0, d1, d2, dd, best_t, best_r, best_i, best_det, interior);
        const uint64_t im = ballot64(interior);
        if (im) {
            win = __builtin_ctzll(im);
            tmin = bcast_d(best_t, win);
            if (win == 0) WCNT(7, 1);
        } else {
            win = ray_winner_lane(best_t, best_r, tmin);
            if (win >= 0) {
                const int f = __builtin_amdgcn_readlane(best_i, rfl(win));
                const double fdet = bcast_d(best_det, win);
                const int i2 = lane < P.nbr_width ? ldg(P.col_nbr, f * P.nbr_width + lane) : -1;
                const bool entering = (double)P.col_orient[f] * fdet > 0;
                if (entering && ballot64(i2 >= 0) != 0) {
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
    // (the record table's address is read together with the flag, not behind it: one scalar round trip less per ray)
    const uint64_t col_rec_addr = (uint64_t)P.col_rec;
#ifdef PRL_WIDE_FACET_LOAD
    asm volatile("" ::"s"(col_rec_addr), "s"(P.col_convex));       // (pins both loads here: the compiler sinks the address' otherwise)
#endif
    if (tile) {
    } else if (P.col_convex && hint >= 0) {
#ifndef PRL_WIDE_FACET_LOAD                         // (default: the round-3 form; the whole-record form below measured +0.4 us)
        // (1) The previous facet alone, wave-uniform on scalar-loaded data, early exits.
        {
            const int h = rfl(hint);
            const double CAS *r = reinterpret_cast<const double CAS *>((uint64_t)P.col_rec) + (size_t)h * 12;
#ifndef PRL_FACET_FIELD_LOADS                         // the record in two scalar round trips (v0 e1 e2 m | nn orient): 36.42 -> 36.31 us;
                                                      // -DPRL_FACET_FIELD_LOADS: every field where first needed (five round trips)
            typedef double dv8 __attribute__((ext_vector_type(8), aligned(8)));
            typedef double dv2 __attribute__((ext_vector_type(2), aligned(8)));
            const dv8 ra = *reinterpret_cast<const dv8 CAS *>(r);
            const dv2 rb = *reinterpret_cast<const dv2 CAS *>(r + 8);
            const double r0 = ra[0], r1 = ra[1], r2 = ra[2], e10 = ra[3], e11 = ra[4], e12 = ra[5], e20 = ra[6], e21 = ra[7], e22 = rb[0], m = rb[1];
#else
            const double e10 = r[3], e11 = r[4], e12 = r[5], e20 = r[6], e21 = r[7], e22 = r[8];
#endif
            const double p0 = d1 * e22 - d2 * e21;
            const double p1 = d2 * e20 - d0 * e22;
            const double p2 = d0 * e21 - d1 * e20;
            const double det = (e10 * p0 + e11 * p1) + e12 * p2;
            bool inside = false;
            double t = 0;
            if (fabs(det) >= RAY_EPS_DET) {
#ifndef PRL_FACET_FIELD_LOADS
                const dv2 rc = *reinterpret_cast<const dv2 CAS *>(r + 10);
                asm volatile("" ::"s"(rc));                 // (requested here, used at the end of the block)
                const double nn = rc[0], orient = rc[1];
#else
                const double r0 = r[0], r1 = r[1], r2 = r[2];
#endif
                const double inv = rcp_det(det);
                const double s0 = o[0] - r0, s1 = o[1] - r1, s2 = o[2] - r2;
                const double u = ((s0 * p0 + s1 * p1) + s2 * p2) * inv;
                const double q0 = s1 * e12 - s2 * e11;
                const double q1 = s2 * e10 - s0 * e12;
                const double q2 = s0 * e11 - s1 * e10;
                const double v = ((d0 * q0 + d1 * q1) + d2 * q2) * inv;
                t = ((e20 * q0 + e21 * q1) + e22 * q2) * inv;
                const double dd = (d0 * d0 + d1 * d1) + d2 * d2;
#ifndef PRL_FACET_FIELD_LOADS
                inside = u >= m && v >= m && (u + v) <= 1.0 - m && t >= 0.0 && t <= 1.0 && orient * det > 0 &&
                         det * det >= FACET_MIN_COS2 * dd * nn;
#else
                const double m = r[9];
                inside = u >= m && v >= m && (u + v) <= 1.0 - m && t >= 0.0 && t <= 1.0 && r[11] * det > 0 &&
                         det * det >= FACET_MIN_COS2 * dd * r[10];
#endif
            }
            if (rfl(inside)) {
                WCNT(7, 1);
                t_out = t;
                hit[0] = o[0] + t * d0;
                hit[1] = o[1] + t * d1;
                hit[2] = o[2] + t * d2;
                return h;
            }
        }
#else
        // (1) The previous facet alone, wave-uniform on scalar-loaded data.  If the segment ENTERS the hull
        // through it at a point at least FACET_EDGE_MARGIN away from its edges, and not at a grazing
        // angle, no other facet can report a hit at or before that point: a second hit there would lie
        // in the other facet's 1e-9 tolerance fringe, i.e. within nanometres of an edge of the entered
        // facet.  The result is then this facet's own Moller-Trumbore value, arithmetic as in mt_one.
        // [-DPRL_WIDE_FACET_LOAD, A/B] The record is fetched whole (two scalar loads, one wait) and the test evaluated without
        // early exits: six dependent scalar round trips fewer per ray -- and eight more spilled scalar registers:
        // 40.46 against 40.07 us (profiles/r04_ab_log.txt), so the form above stays.
        {
            const int h = rfl(hint);
            typedef double d8 __attribute__((ext_vector_type(8), aligned(8)));
            typedef double d4 __attribute__((ext_vector_type(4), aligned(8)));
            const double CAS *r = reinterpret_cast<const double CAS *>(col_rec_addr) + (size_t)h * 12;
            const d8 ra = *reinterpret_cast<const d8 CAS *>(r);
            const d4 rb = *reinterpret_cast<const d4 CAS *>(r + 8);
            const double e10 = ra[3], e11 = ra[4], e12 = ra[5], e20 = ra[6], e21 = ra[7], e22 = rb[0];
            const double p0 = d1 * e22 - d2 * e21;
            const double p1 = d2 * e20 - d0 * e22;
            const double p2 = d0 * e21 - d1 * e20;
            const double det = (e10 * p0 + e11 * p1) + e12 * p2;
            const double inv = 1.0 / det;                       // (a degenerate det fails the first condition below)
            const double s0 = o[0] - ra[0], s1 = o[1] - ra[1], s2 = o[2] - ra[2];
            const double u = ((s0 * p0 + s1 * p1) + s2 * p2) * inv;
            const double q0 = s1 * e12 - s2 * e11;
            const double q1 = s2 * e10 - s0 * e12;
            const double q2 = s0 * e11 - s1 * e10;
            const double v = ((d0 * q0 + d1 * q1) + d2 * q2) * inv;
            const double t = ((e20 * q0 + e21 * q1) + e22 * q2) * inv;
            const double m = rb[1], dd = (d0 * d0 + d1 * d1) + d2 * d2;
            const bool inside = (fabs(det) >= RAY_EPS_DET) & (u >= m) & (v >= m) & ((u + v) <= 1.0 - m) & (t >= 0.0) & (t <= 1.0) &
                                (rb[3] * det > 0) & (det * det >= FACET_MIN_COS2 * dd * rb[2]);
            if (rfl(inside)) {
                WCNT(7, 1);
                t_out = t;
                hit[0] = o[0] + t * d0;
                hit[1] = o[1] + t * d1;
                hit[2] = o[2] + t * d2;
                return h;
            }
        }
#endif
        // (2) The facets that share a vertex with it, one per lane.  A lane whose facet is entered at an
        // interior point holds the closest hit of the whole set by the same argument (there is at most
        // one such lane): no reduction, no second round.
        WCNT(4, 1);
        const double dd = (d0 * d0 + d1 * d1) + d2 * d2;
        const int i1 = lane < P.nbr_width ? ldg(P.col_nbr, hint * P.nbr_width + lane) : -1;
        bool interior;
        mt_rec(P, i1, o, d0, d1, d2, dd, best_t, best_r, best_i, best_det, interior);
        const uint64_t im = ballot64(interior);
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
                const int i2 = lane < P.nbr_width ? ldg(P.col_nbr, f * P.nbr_width + lane) : -1;
                const bool entering = (double)P.col_orient[f] * fdet > 0;
                if (entering && ballot64(i2 >= 0) != 0) {
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
        // The tool beside the part -- a fifth of all env-steps of a random walk, every ray of theirs -- is certified a miss by
        // the set's outline (one read, ~70 instructions) instead of by two stages of box and triangle tests that find nothing
        // (-DPRL_NO_OUTLINE_MISS: A/B and parity switch).  Then: a segment whose box meets no chunk of the set misses it too.
#ifndef PRL_NO_OUTLINE_MISS
        if (beam_outside_outline_wave(P, o, e, lane)) {
            WCNT(1, 1);
            t_out = INFINITY;
            hint = -1;
            return -1;
        }
#endif
        const SegBox sb_all = seg_box(o3, d3, 1.0);
        uint64_t any_chunk = 0;
        for (int cbase = 0; cbase < P.n_col_chunks; cbase += 64)
            any_chunk |= ballot64(box_overlap(sb_all, ldg(chunk_boxes, 2 * (cbase + lane)), ldg(chunk_boxes, 2 * (cbase + lane) + 1)));
        // Two stages -- the near part of the segment first: a tool on the part hits at t ~ 0.1 -- unless the previous ray missed
        // (no facet hint): a tool within millimetres of the rim, tilted so that the outline does not certify its rays, casts five
        // that pass over the rim and hit nothing; both stages were 4 - 8 chunks each, every chunk a dependent round trip: the
        // one 42 us wave that a launch of 4 096 envs waited for in three launches of five (tools/wave_trace.py, the LAST wave
        // of each launch).  The full stage alone finds the same closest hit if there is one (a hit at t <= 0.125 is in both).
        for (int stage = any_chunk ? (after_a_miss ? 1 : 0) : 2; stage < 2; ++stage) {
            const double tmax = stage == 0 ? 0.125 : 1.0;
            if (stage == 1) WCNT(1, 1);
            const SegBox sb = stage == 0 ? seg_box(o3, d3, tmax) : sb_all;
            int n_cand = 0;
            for (int cbase = 0; cbase < P.n_col_chunks; cbase += 64) {
                const f32x4 ca = ldg(chunk_boxes, 2 * (cbase + lane)), cb = ldg(chunk_boxes, 2 * (cbase + lane) + 1);
                uint64_t cm = ballot64(box_overlap(sb, ca, cb));   // table is padded to 64 with empty boxes
                // (a chunk's triangle boxes are a dependent round trip: the next chunk's are asked for before this one's are
                // looked at -- a ray that misses near the rim visits five of them)
                if (cm) {
                    int i = ((cbase + __builtin_ctzll(cm)) << 6) + lane;
                    cm &= cm - 1;
                    f32x4 ba = ldg(boxes, 2 * i), bb = ldg(boxes, 2 * i + 1);
                    for (;;) {
                        WCNT(2, 1);
                        const bool more = cm != 0;
                        int i_n = i;
                        f32x4 na = ba, nb = bb;
                        if (more) {
                            i_n = ((cbase + __builtin_ctzll(cm)) << 6) + lane;
                            cm &= cm - 1;
                            na = ldg(boxes, 2 * i_n);
                            nb = ldg(boxes, 2 * i_n + 1);
                        }
                        const bool pass = box_overlap(sb, ba, bb);
                        const uint64_t pm = ballot64(pass);
                        if (pm != 0) {
                            const int np = __popcll(pm);
                            if (n_cand + np > 64) {                    // list full: test what is queued first
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
                        if (!more) break;
                        i = i_n;
                        ba = na;
                        bb = nb;
                    }
                }
            }
            if (n_cand) {
                __builtin_amdgcn_wave_barrier();
                mt_one(P, lane < n_cand ? cand[lane] : -1, o, d0, d1, d2, tmax, best_t, best_r, best_i, best_det);
                __builtin_amdgcn_wave_barrier();
            }
            if (ballot64(best_t < INFINITY)) break;
        }
        win = ray_winner_lane(best_t, best_r, tmin);
    }
    if (win < 0) {
        t_out = INFINITY;
        hint = -1;
        return -1;
    }
    hint = __builtin_amdgcn_readlane(best_i, rfl(win));
    if (tile && pf && P.col_convex && rfl(tile->facet) != hint) {     // the next ray starts from this facet: its tile, early
        pf->facet = hint;
        pf->ids = tile_ids_load(P, hint, lane);
    }
    t_out = tmin;
    hit[0] = o[0] + tmin * d0;
    hit[1] = o[1] + tmin * d1;
    hit[2] = o[2] + tmin * d2;
    return hint;
}

}  // namespace
