// prl_search.hpp -- nearest same-side vertex, nearest sample, hook point (bpw:508-534, 565).
// Part of the single translation unit paintrl_hip.hip (device code, anonymous namespace); see that
// file for the overall design.  Compile with -ffp-contract=off.
#pragma once

namespace {

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
    const int v = ok ? ldg(start, cy * nx + ((lane & 1) ? cx1 + 1 : cx0)) : 0;
    Rows3 out;
#pragma unroll
    for (int k = 0; k < 3; ++k) {
        const int b = __builtin_amdgcn_readlane(v, 2 * k), e = __builtin_amdgcn_readlane(v, 2 * k + 1);
        out.begin[k] = b;
        out.count[k] = e - b;
    }
    return out;
}

// One vertex record (x y z | rank) as two 16-byte loads.
__device__ __forceinline__ void load_vertex(PartRef P, int v, double &x, double &y, double &z, int &rank) {
    const f64x2 GAS *r = reinterpret_cast<const f64x2 GAS *>(P.vert4);
    const f64x2 a = ldg(r, 2 * v), b = ldg(r, 2 * v + 1);
    x = a.x;
    y = a.y;
    z = b.x;
    rank = __double2loint(b.y);
}

__device__ __forceinline__ void nv_scan(PartRef P, int begin, int end, const double pt[3], int lane,
                                        double &best_d, int &best_rank, int &best_idx) {
    for (int b = begin; b < end; b += 128) {                   // two batches per trip: four loads in flight
        const int v0 = b + lane, v1 = v0 + 64;
        const bool k0 = v0 < end, k1 = v1 < end;
        double x0 = 0, y0 = 0, z0 = 0, x1 = 0, y1 = 0, z1 = 0;
        int r0 = 0, r1 = 0;
        if (k0) load_vertex(P, v0, x0, y0, z0, r0);
        if (k1) load_vertex(P, v1, x1, y1, z1, r1);
        if (k0) {
            const double dx = x0 - pt[0], dy = y0 - pt[1], dz = z0 - pt[2];
            const double dd = (dx * dx + dy * dy) + dz * dz;
            if ((dd < best_d) | ((dd == best_d) & (r0 < best_rank))) {
                best_d = dd;
                best_rank = r0;
                best_idx = v0;
            }
        }
        if (k1) {
            const double dx = x1 - pt[0], dy = y1 - pt[1], dz = z1 - pt[2];
            const double dd = (dx * dx + dy * dy) + dz * dz;
            if ((dd < best_d) | ((dd == best_d) & (r1 < best_rank))) {
                best_d = dd;
                best_rank = r1;
                best_idx = v1;
            }
        }
    }
}

// The vertices of a (2k+1)-row cell block, rows as index ranges (lane 2r / 2r+1 of `bound` = begin / end of row r),
// flattened into one candidate list and scanned two candidates per lane and trip.
template <int NROWS>
__device__ __forceinline__ void ring_block_scan(PartRef P, int bound, const double pt[3], int lane, double &best_d,
                                                int &best_rank, int &best_idx) {
    // per-row begin and exclusive prefix of counts, wave-uniform
    int rbeg[NROWS], rpre[NROWS + 1];
    rpre[0] = 0;
#pragma unroll
    for (int r = 0; r < NROWS; ++r) {
        const int b0 = __builtin_amdgcn_readlane(bound, 2 * r), e0 = __builtin_amdgcn_readlane(bound, 2 * r + 1);
        rbeg[r] = b0;
        rpre[r + 1] = rpre[r] + (e0 - b0);
    }
    const int total = rpre[NROWS];
    for (int c0 = 0; c0 < total; c0 += 128) {           // two candidates per lane and trip: one round trip
        WCNT(3, 1);                                     // covers the 60-90 vertices of a typical 3 x 3 block
        const int ca = c0 + lane, cb = ca + 64;
        const bool ka = ca < total, kb = cb < total;
        int va = rbeg[0] + ca, vb = rbeg[0] + cb;
#pragma unroll
        for (int r = 1; r < NROWS; ++r) {
            if (ca >= rpre[r]) va = rbeg[r] + (ca - rpre[r]);
            if (cb >= rpre[r]) vb = rbeg[r] + (cb - rpre[r]);
        }
        // (lanes without a candidate read vertex 0 and are masked in the comparison: a load or a distance under a per-lane
        // condition is a region of its own -- exec mask saved, branch, restored -- and there were six of them per trip)
        double ax, ay, az, bx = 0, by = 0, bz = 0;
        int ra, rb = 0;
        const bool second = c0 + 64 < total;            // wave-uniform: anyone with a second candidate?
        load_vertex(P, ka ? va : 0, ax, ay, az, ra);
        if (second) load_vertex(P, kb ? vb : 0, bx, by, bz, rb);
        {
            const double dx = ax - pt[0], dy = ay - pt[1], dz = az - pt[2];
            const double dd = (dx * dx + dy * dy) + dz * dz;
            if (ka & ((dd < best_d) | ((dd == best_d) & (ra < best_rank)))) {
                best_d = dd;
                best_rank = ra;
                best_idx = va;
            }
        }
        if (second) {
            const double dx = bx - pt[0], dy = by - pt[1], dz = bz - pt[2];
            const double dd = (dx * dx + dy * dy) + dz * dz;
            if (kb & ((dd < best_d) | ((dd == best_d) & (rb < best_rank)))) {
                best_d = dd;
                best_rank = rb;
                best_idx = vb;
            }
        }
    }
}

// Exact nearest neighbour by expanding rings: the (2k+1)^2 cell block around the query's cell is
// scanned (its rows are contiguous index ranges, flattened into one candidate list); every vertex
// outside the block is at least k cells away in the principal plane, so the result is exact once the
// best distance is within k * 0.99 * cell.  After ring 3 the whole table is scanned.
__device__ int nearest_vertex_wave(PartRef P, const double pt[3], int lane, const int *vg_lds = nullptr) {
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
#ifdef PRL_GRID_LDS                                  // (A/B switch, k_step.hip: the table's LDS copy where the kernel has one)
        const int bidx = okr ? rcy * P.vg_nx + ((lane & 1) ? cx1 + 1 : cx0) : 0;
        const int bound = okr ? (vg_lds ? vg_lds[bidx] : ldg(P.vg_start, bidx)) : 0;
#else
        (void)vg_lds;
        const int bound = okr ? ldg(P.vg_start, rcy * P.vg_nx + ((lane & 1) ? cx1 + 1 : cx0)) : 0;
#endif
        best_d = INFINITY;
        best_rank = 0x7fffffff;
        best_idx = -1;
        // (one copy of the scan per block height: the first ring -- nearly every query ends there -- then maps a
        // candidate to its row with two selects instead of six)
        if (ring == 1) ring_block_scan<3>(P, bound, pt, lane, best_d, best_rank, best_idx);
        else if (ring == 2) ring_block_scan<5>(P, bound, pt, lane, best_d, best_rank, best_idx);
        else ring_block_scan<7>(P, bound, pt, lane, best_d, best_rank, best_idx);
        dmin = wave_min_nonneg_d(best_d);
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
        dmin = wave_min_nonneg_d(best_d);
    }
    const uint64_t tie = ballot64(best_d == dmin);
    if (tie == 0) return -1;                                        // NaN query point
    if ((tie & (tie - 1)) == 0) return __builtin_amdgcn_readlane(best_idx, rfl(__builtin_ctzll(tie)));
    const int rmin = wave_min_i(best_d == dmin ? best_rank : 0x7fffffff);
    const uint64_t win = ballot64(best_d == dmin && best_rank == rmin);
    return __builtin_amdgcn_readlane(best_idx, rfl(__builtin_ctzll(win)));
}

// ---------------------------------------------------------------- bpw:526 on a part whose vertex rows the reference moved
// scipy's cKDTree.query(k = 1) on the STALE tree (include/paintrl.h kd_node; oracle/paint_oracle.c stale_kd_query is
// the scalar statement): descend to the near child, queue the far child if its lower bound is <= the best squared
// distance, scan a leaf's points in tree order keeping strictly smaller distances (distances to the rows as they
// are NOW), continue with the nearest queued cell until none is left or it lies beyond the best.  The walk is
// wave-uniform; the (at most 16) points of a leaf are one per lane; the queue lives in this wave's LDS rows.
// The tree of a coarse part is tiny (the reference's sheet: 33 nodes) and every query walks it node by node, two dependent
// global round trips a level.  kd_stage copies trees of up to KD_LDS_NODES nodes into this wave's LDS once per step (behind
// the queue in `heap`); the walk then reads its nodes from there.
__device__ __forceinline__ bool kd_lanes_on(PartRef P) {
#ifdef PRL_KD_GENERAL_WALK                          // (parity and A-B switch: the general walk for every tree)
    return false;
#else
    return P.n_kd_leaves > 0;
#endif
}
__device__ __forceinline__ void kd_stage(PartRef P, double *heap, int lane) {
    if (P.n_kd_nodes > 0 && P.n_kd_nodes <= KD_LDS_NODES) {
        if (lane < P.n_kd_nodes) {
            if (kd_lanes_on(P)) {                                   // (the lane-parallel query reads its own tables, and nothing else)
                reinterpret_cast<i32x4 *>(heap + KD_HEAP * 5)[lane] = ldg(reinterpret_cast<const i32x4 GAS *>(P.kd_lane), lane);
                reinterpret_cast<uint64_t *>(heap)[lane] = ldg(P.kd_anc, lane);       // (where the general walk keeps its queue)
            } else {
                reinterpret_cast<i32x4 *>(heap + KD_HEAP * 5)[lane] = ldg(reinterpret_cast<const i32x4 GAS *>(P.kd_node), lane);
                heap[KD_HEAP * 5 + 2 * KD_LDS_NODES + lane] = ldg(P.kd_split, lane);
            }
        }
        __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
        __builtin_amdgcn_wave_barrier();
        __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
    }
}

// Round 5: the whole query one lane per NODE.  What the general walk below does node by node -- read the node, the
// bound of its far child, push, pop the nearest queued cell with a wave-wide minimum, each a dependent LDS round trip behind a
// fence: 2.6 us a query on the coarse sheet, 13 of the step's 45 us -- depends on the query only through values every node can
// compute for itself:
//   * whether the walk, arriving at its parent, would go to this node FIRST (x[dim] < split) and the squared distance to the
//     parent's plane if not (query.cxx: the far child's side distance);
//   * its lower bound: the parent's, plus (new side - old side) if it is the far child -- the additions in the order of the
//     path from the root, one tree level per trip, the parent's four values read across lanes;
//   * a leaf's answer: its points are a row of 16 lanes (PartDev::kd_rec16), smallest distance by row rotations on the bit
//     patterns, first point reaching it from the ballot.
// The walk that remains is scalar: the node's children and flags from one v_readlane, `bound <= best` / `bound > best` /
// `leaf's distance < best` as three ballots renewed when the best distance changes, the queue a 64-bit mask of nodes plus each
// queued lane's position in scipy's list (equal bounds pop in list order; the last entry fills the hole).  Same visiting order,
// same strict comparisons as oracle/paint_oracle.c stale_kd_query.  `node_tab`: the wave's copy of PartDev::kd_lane (kd_stage).
__device__ int nearest_vertex_kd_lanes(PartRef P, const double pt[3], int lane, const i32x4 *node_tab, const uint64_t *anc_tab PROF_ARG) {
    const int n_nodes = P.n_kd_nodes;
    const bool isn = lane < n_nodes;
    // the leaves' points: all loads in flight before anything else
    const f64x2 GAS *r16 = reinterpret_cast<const f64x2 GAS *>(P.kd_rec16);
    const int n_slots = 16 * P.n_kd_leaves;                       // (the table itself: whole trips of 64)
    const i32x4 st = node_tab[isn ? lane : 0];
    const int pk = st.x, ordinal = isn ? st.y : -1;
    const int sd = pk & 3, psd = (pk >> 2) & 3, depth = (pk >> 5) & 63, parent = (pk >> 11) & 63, lesser = (pk >> 17) & 63,
              greater = (pk >> 23) & 63;
    const bool is_lesser = (pk >> 4) & 1;
    const double psp = __hiloint2double(st.w, st.z);
    // the edge into this node
    const double xp = sel3(pt[0], pt[1], pt[2], psd);
    const bool i_far = isn && lane != 0 && ((xp < psp) != is_lesser);
    const double tmp = psp - xp, nw = tmp * tmp;
    // the root's bound and side distances (every lane)
    double s0, s1, s2;
    {
        const double a0 = pt[0] - P.kd_box[3], b0 = P.kd_box[0] - pt[0], a1 = pt[1] - P.kd_box[4], b1 = P.kd_box[1] - pt[1],
                     a2 = pt[2] - P.kd_box[5], b2 = P.kd_box[2] - pt[2];
        // (max(a, b, 0) as two v_max_f64: for a NaN query both differences are NaN and either form gives 0)
        s0 = fmax(fmax(a0, b0), 0.0);
        s1 = fmax(fmax(a1, b1), 0.0);
        s2 = fmax(fmax(a2, b2), 0.0);
        s0 = s0 * s0;
        s1 = s1 * s1;
        s2 = s2 * s2;
    }
    const uint64_t farmask = ballot64(i_far);
    // the far child's bound is its parent's + (new side distance - old): the old one is the deepest far step across a plane of
    // the same dimension on the way down (children follow their parents: the highest node index), or the root's
    const uint64_t set_by = anc_tab[isn ? lane : 0] & farmask;
    const double nw_set = __shfl(nw, 63 - __clzll((long long)(set_by | 1)));
    const double delta = nw - (set_by ? nw_set : sel3(s0, s1, s2, psd));
    double mind = (s0 + s1) + s2;
    for (int d = 1; d <= P.kd_depth; ++d) {                         // level by level: the additions in the order of the path
        const double pm = __shfl(mind, parent);
        if (depth == d) mind = i_far ? pm + delta : pm;
    }
    const bool lesser_far = (farmask >> lesser) & 1;
    // (a leaf's `near` field: the lane its answer ends up in, below)
    const int walk = sd == 3 ? (0x10000 | (16 * (ordinal & 3) + (ordinal >> 2))) : ((lesser_far ? greater : lesser) | ((lesser_far ? lesser : greater) << 8));
    if (!isn) mind = __longlong_as_double(0x7ff8000000000000ll);   // (no node: every comparison below is false)
    // the bound as a pair of unsigned words that order like the doubles (-0 -> +0 first; a negative bound -- rounding of
    // old + (new - old) -- orders below the positive ones): the nearest queued cell is two 32-bit minima, not a float64 one
    const double mz = mind + 0.0;
    const uint32_t mneg = (uint32_t)(__double2hiint(mz) >> 31);
    const uint32_t khi = (uint32_t)__double2hiint(mz) ^ (mneg | 0x80000000u), klo = (uint32_t)__double2loint(mz) ^ mneg;
    // the leaves: row r of trip t is the leaf of ordinal 4 t + r; its answer goes to lane 16 r + t
    uint32_t lmin_hi = 0x7ff00000u, lmin_lo = 0;
    int lvert = -1;
    for (int t = 0; 64 * t < n_slots; ++t) {
        // (the table is padded to whole trips with points at +inf: their distance is +inf, or NaN for a NaN query -- above
        // every real one either way -- so nothing here asks whether a slot is real)
        // (the records of the next trips asked for ahead of their use: measured slower, the registers cost more than the wait)
        const f64x2 a = ldg(r16, 2 * (64 * t + lane)), b = ldg(r16, 2 * (64 * t + lane) + 1);
        const double d0 = a.x - pt[0], d1 = a.y - pt[1], d2 = b.x - pt[2];
        const double dd = (d0 * d0 + d1 * d1) + d2 * d2;            // (a sum of squares: never -0)
        const uint32_t hi = (uint32_t)__double2hiint(dd), lo = (uint32_t)__double2loint(dd);
        uint32_t mh = hi;                                           // (>= +0 or NaN: ordered like their bit patterns)
        mh = dpp_umin<0x121, 0xf>(mh);                              // row_ror 1, 2, 4, 8: every lane of the row has the row's minimum
        mh = dpp_umin<0x122, 0xf>(mh);
        mh = dpp_umin<0x124, 0xf>(mh);
        mh = dpp_umin<0x128, 0xf>(mh);
        uint32_t ml = hi == mh ? lo : 0xffffffffu;
        ml = dpp_umin<0x121, 0xf>(ml);
        ml = dpp_umin<0x122, 0xf>(ml);
        ml = dpp_umin<0x124, 0xf>(ml);
        ml = dpp_umin<0x128, 0xf>(ml);
        // the first point (tree order = lane order) reaching it: a third minimum, over lane-in-row << 28 | vertex
        uint32_t mv = (hi == mh) & (lo == ml) ? ((uint32_t)(lane & 15) << 28) | (uint32_t)__double2loint(b.y) : 0xffffffffu;
        mv = dpp_umin<0x121, 0xf>(mv);
        mv = dpp_umin<0x122, 0xf>(mv);
        mv = dpp_umin<0x124, 0xf>(mv);
        mv = dpp_umin<0x128, 0xf>(mv);
        if ((lane & 15) == t) {
            lmin_hi = mh;
            lmin_lo = ml;
            lvert = (int)(mv & 0x0fffffffu);
        }
    }
    const double lminv = __hiloint2double((int)lmin_hi, (int)lmin_lo);
    STAMP(9);
    double dub = INFINITY;
    int best = -1, node = 0, n_heap = 0, pos = -1;
    uint64_t queued = 0;
    uint64_t le = ballot64(mind <= dub), gt = 0, lt = ballot64(lminv < dub);
    const uint64_t leaves = ballot64(isn & (sd == 3));
    for (int pops = 0; pops <= KD_LDS_NODES; ++pops) {               // (a node is queued at most once; a bound all the same)
        // down to a leaf, the far children queued (a child's index is above its parent's -- part_fill checks it: this ends)
        int w = __builtin_amdgcn_readlane(walk, node);
        while (!(w & 0x10000)) {
            if ((gt >> node) & 1) return best;                      // this cell lies beyond the best: scipy's query ends here
            const int far = (w >> 8) & 0xff;
            if ((le >> far) & 1) {
                queued |= 1ull << far;
                pos = lane == far ? n_heap : pos;
                ++n_heap;
            }
            node = w & 0xff;
            w = __builtin_amdgcn_readlane(walk, node);
        }
        const int ll = w & 63;
        if ((lt >> ll) & 1) {
            dub = bcast_d(lminv, ll);
            best = __builtin_amdgcn_readlane(lvert, ll);
            le = ballot64(mind <= dub);
            gt = ballot64(mind > dub);
            lt = ballot64(lminv < dub);
        }
        // nothing queued, or only cells (no leaf) that all lie beyond the best: whichever is nearest ends the query
        if ((queued & (leaves | ~gt)) == 0) break;
        int m = (int)__builtin_ctzll(queued);
        if (queued & (queued - 1)) {                                // more than one queued: the nearest
            const bool mine = (queued >> lane) & 1;
            const uint32_t mh = wave_min_u32(mine ? khi : 0xffffffffu);
            const bool top = mine & (khi == mh);
            uint64_t tie = ballot64(top);
            if (tie & (tie - 1)) {                                  // (the bounds' high words equal: rare)
                const uint32_t ml = wave_min_u32(top ? klo : 0xffffffffu);
                tie = ballot64(top & (klo == ml));
            }
            m = (int)__builtin_ctzll(tie);
            if (tie & (tie - 1)) {                                  // equal bounds: the first in the list
                const bool tied = (tie >> lane) & 1;
                const int pmin = wave_min_i(tied ? pos : 0x7fffffff);
                m = (int)__builtin_ctzll(ballot64(tied & (pos == pmin)));
            }
        }
        m = rfl(m);
        const int pos_m = __builtin_amdgcn_readlane(pos, m);
        queued &= ~(1ull << m);
        --n_heap;
        pos = pos == n_heap ? pos_m : pos;                          // the last entry fills the hole
        node = m;
    }
    return best;
}

__device__ int nearest_vertex_kd(PartRef P, const double pt[3], int lane, double *heap, bool has_copy PROF_ARG) {
    const bool staged = has_copy && P.n_kd_nodes <= KD_LDS_NODES;           // (kd_stage ran in shots_begin)
    const i32x4 *lds_node = reinterpret_cast<const i32x4 *>(heap + KD_HEAP * 5);
    if (staged && kd_lanes_on(P)) return nearest_vertex_kd_lanes(P, pt, lane, lds_node, reinterpret_cast<const uint64_t *>(heap) PROF_PASS);
    const double *lds_split = heap + KD_HEAP * 5 + 2 * KD_LDS_NODES;
    double side0, side1, side2;
    {
        const double a0 = pt[0] - P.kd_box[3], b0 = P.kd_box[0] - pt[0], a1 = pt[1] - P.kd_box[4], b1 = P.kd_box[1] - pt[1],
                     a2 = pt[2] - P.kd_box[5], b2 = P.kd_box[2] - pt[2];
        double s0 = a0 > b0 ? a0 : b0, s1 = a1 > b1 ? a1 : b1, s2 = a2 > b2 ? a2 : b2;
        s0 = s0 > 0 ? s0 : 0;
        s1 = s1 > 0 ? s1 : 0;
        s2 = s2 > 0 ? s2 : 0;
        side0 = s0 * s0;
        side1 = s1 * s1;
        side2 = s2 * s2;
    }
    double mind = (side0 + side1) + side2, dub = INFINITY;
    int node = 0, best = -1, n_heap = 0;
    for (int guard = 0; guard < 4 * 4096; ++guard) {                // every path ends far earlier; a bound all the same
        const i32x4 nd = staged ? lds_node[node] : ldg(reinterpret_cast<const i32x4 GAS *>(P.kd_node), node);
        const int nd0 = rfl(nd.x), nd1 = rfl(nd.y), nd2 = rfl(nd.z);
        if (nd0 < 0) {                                              // leaf: points nd1 .. nd2 - 1
            for (int i0 = nd1; i0 < nd2; i0 += 64) {
                const int i = i0 + lane;
                int v = -1;                                         // -1: a row parked at (10, 10, 10)
                double d = INFINITY;
                if (i < nd2) {
                    const f64x2 GAS *r = reinterpret_cast<const f64x2 GAS *>(P.kd_rec);
                    const f64x2 a = ldg(r, 2 * i), b = ldg(r, 2 * i + 1);      // the point's record, in tree order (PartDev::kd_rec)
                    v = __double2loint(b.y);
                    if (v >= 0) {
                        const double d0 = a.x - pt[0], d1 = a.y - pt[1], d2 = b.x - pt[2];
                        d = (d0 * d0 + d1 * d1) + d2 * d2;
                    }
                }
                const double dmin = wave_min_d(d);
                if (dmin < dub) {                                   // the first point (tree order) reaching the minimum
                    dub = dmin;
                    best = __builtin_amdgcn_readlane(v, rfl(__builtin_ctzll(ballot64(d == dmin))));
                }
            }
            if (n_heap == 0) break;
            // pop the nearest queued cell: lane l looks at entry l
            const double key = lane < n_heap ? heap[5 * lane] : INFINITY;
            const double kmin = wave_min_d(key);
            const int m = rfl(__builtin_ctzll(ballot64(key == kmin)));
            mind = heap[5 * m];
            side0 = heap[5 * m + 1];
            side1 = heap[5 * m + 2];
            side2 = heap[5 * m + 3];
            node = rfl((int)heap[5 * m + 4]);
            --n_heap;
            __builtin_amdgcn_wave_barrier();
            if (lane < 5 && m != n_heap) heap[5 * m + lane] = heap[5 * n_heap + lane];      // the last entry fills the hole
            __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
            __builtin_amdgcn_wave_barrier();
            __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
        } else {
            if (mind > dub) break;
            const double sp = staged ? lds_split[node] : ldg(P.kd_split, node);
            const double xs = sel3(pt[0], pt[1], pt[2], nd0), old = sel3(side0, side1, side2, nd0);
            const bool low = xs < sp;
            const int near = low ? nd1 : nd2, far = low ? nd2 : nd1;
            const double tmp = sp - xs, nw = tmp * tmp;
            const double far_mind = mind + (nw - old);
            // (query.cxx also swaps the two if the near child came out farther; with new >= old that needs a NaN)
            if (far_mind <= dub && n_heap < KD_HEAP) {
                if (lane == 0) {
                    heap[5 * n_heap] = far_mind;
                    heap[5 * n_heap + 1] = nd0 == 0 ? nw : side0;
                    heap[5 * n_heap + 2] = nd0 == 1 ? nw : side1;
                    heap[5 * n_heap + 3] = nd0 == 2 ? nw : side2;
                    heap[5 * n_heap + 4] = (double)far;
                }
                ++n_heap;
                __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
                __builtin_amdgcn_wave_barrier();
                __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
            }
            node = rfl(near);
        }
    }
    return best;
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
        const int bound = okr ? ldg(P.sg_start, rcy * P.sg_nx + ((lane & 1) ? cx1 + 1 : cx0)) : 0;
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
                    const double dx = ldg(P.samp[0], sidx) - pt[0], dy = ldg(P.samp[1], sidx) - pt[1], dz = ldg(P.samp[2], sidx) - pt[2];
                    const double dd = (dx * dx + dy * dy) + dz * dz;
                    const int rk = ldg(P.samp_rank, sidx);
                    if (dd < best_d || (dd == best_d && rk < best_rank)) {
                        best_d = dd;
                        best_rank = rk;
                        best_idx = sidx;
                    }
                }
            }
        }
        dmin = wave_min_nonneg_d(best_d);
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
            const double dx = ldg(P.samp[0], sidx) - pt[0], dy = ldg(P.samp[1], sidx) - pt[1], dz = ldg(P.samp[2], sidx) - pt[2];
            const double dd = (dx * dx + dy * dy) + dz * dz;
            const int rk = ldg(P.samp_rank, sidx);
            if (rk != 0x7fffffff && (dd < best_d || (dd == best_d && rk < best_rank))) {
                best_d = dd;
                best_rank = rk;
                best_idx = sidx;
            }
        }
        dmin = wave_min_nonneg_d(best_d);
    }
    const int rmin = wave_min_i(best_d == dmin ? best_rank : 0x7fffffff);
    const uint64_t win = ballot64((best_d == dmin) & (best_rank == rmin) & (best_idx >= 0));
    if (win == 0) return -1;
    return __builtin_amdgcn_readlane(best_idx, rfl(__builtin_ctzll(win)));
}

// ---------------------------------------------------------------- records of a few lanes, fetched by all (prl_device.hpp)
// Lane j < n holds record id[j] >= 0 of a table of C-chunk records (16-byte chunks).  All n * C chunks are fetched in
// ceil(n C / 64) instructions -- lane l of trip i takes chunk f = 64 i + l: piece f % C of record f / C -- and stored at
// buf[f]; record j then lies at buf[j C .. j C + C - 1].  n * C <= GATHER_CHUNKS.  Ends with the wave-scope fence that
// orders the stores before other lanes' reads.
template <int C>
__device__ __forceinline__ void record_gather(const f64x2 GAS *table, int id, int n, int lane, f64x2 *buf) {
    static_assert(C == 12 || C == 6, "division constants below");
    const int total = n * C;
    constexpr int TRIPS = GATHER_CHUNKS / 64;
    f64x2 v[TRIPS];
#pragma unroll
    for (int i = 0; i < TRIPS; ++i) {
        const int f = 64 * i + lane;
        const int j = C == 12 ? (f * 43691) >> 19 : (f * 43691) >> 18;          // f / C for f < 2048
        const int piece = f - C * j;
        const int rec = __shfl(id, j);                                          // (lanes beyond the records: some id, unused)
        if (64 * i < total && f < total) v[i] = ldg(table, rec * C + piece);
    }
#pragma unroll
    for (int i = 0; i < TRIPS; ++i) {
        const int f = 64 * i + lane;
        if (64 * i < total && f < total) buf[f] = v[i];
    }
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
    __builtin_amdgcn_wave_barrier();
    __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
}

// ---------------------------------------------------------------- bpw:525-534 _get_hook_point (+508-523)
// Also returns what the chosen triangle's normal implies for the tool: the quaternion of rob:93-100 and the shot
// centre of rob:277-278 (pose + R(quat)(0, 0, 0.1)), read from the triangle record's precomputed tail.
// `tri` receives the chosen triangle (index into the triangle records).  KD: the part carries the reference's stale
// vertex kd-tree (kd_heap = this wave's LDS rows for its walk); otherwise the exact nearest vertex.
// `tile` / `pf`: the ray's facet tile and the prefetch the last ray left (prl_ray.hpp): its records are sent on their way to
// LDS here, between the two searches -- the neighbour ids have arrived under the vertex search, the copy lands under the
// triangle search.
// `gather`: this wave's record_gather buffer, or nullptr (then every candidate lane fetches its own record).
template <bool KD>
__device__ bool hook_point_wave(PartRef P, const double pt[3], int lane, double pose[3], double orn[3], double quat[4],
                                double center[3], int &tri, const WaveLds &wl, const TilePrefetch &pf PROF_ARG) {
    double *kd_heap = wl.kd_heap;
    FacetTile *tile = wl.tile;
    f64x2 *gather = wl.gather;
    int vidx;
    if constexpr (KD) vidx = P.n_kd_nodes > 0 ? nearest_vertex_kd(P, pt, lane, kd_heap, wl.kd_staged != 0 PROF_PASS) : nearest_vertex_wave(P, pt, lane, wl.vg_lds);
    else vidx = nearest_vertex_wave(P, pt, lane, wl.vg_lds);
    if (tile && pf.facet >= 0) tile_fill(P, tile, pf.facet, pf.ids, lane);
    STAMP(PH_VERTEX);
    if (vidx < 0) return false;
    const int ti = lane < P.adj_width ? ldg(P.vadj, vidx * P.adj_width + lane) : -1;   // file order, -1 = pad
    const uint64_t cmask = ballot64(ti >= 0);
    if (cmask == 0) return false;
    bool inside = false, ok = false;
    double m = -INFINITY, n0, n1, n2;
    const f64x2 GAS *r2 = reinterpret_cast<const f64x2 GAS *>(P.tri_rec);
    // the candidates' records through LDS (record_gather) where the kernel has the buffer, the candidates are the first
    // lanes (rows of vadj are padded at the end) and fit it
    const int n_cand = __popcll(cmask);
#ifdef PRL_RECORD_GATHER                              // A/B switch; OFF: 13 -> 3 vector-memory instructions and one dependent round trip
    const bool coop = gather != nullptr && (cmask & (cmask + 1)) == 0 && n_cand * (TRI_REC / 2) <= GATHER_CHUNKS;     // less per hook point, but
#else                                                 // +212 vector / +107 scalar / +60 LDS instructions per env-step: 40.22 against 39.75 us
    const bool coop = false;                          // (profiles/r04_ab_log.txt) -- the step is bound by instruction issue
#endif
    if (coop) record_gather<TRI_REC / 2>(r2, ti, n_cand, lane, gather);
    {
        // (straight-line: a lane without a candidate evaluates triangle 0 and is masked in the predicates at the end)
        const bool cand = ti >= 0;
        f64x2 q0, q1, q2, q3, q4, q5, q6;
        if (coop) {
            const f64x2 *g = gather + (cand ? lane : 0) * (TRI_REC / 2);
            q0 = g[0], q1 = g[1], q2 = g[2], q3 = g[3], q4 = g[4], q5 = g[5], q6 = g[6];
        } else {
            const int t8 = (cand ? ti : 0) * (TRI_REC / 2);
            q0 = ldg(r2, t8), q1 = ldg(r2, t8 + 1), q2 = ldg(r2, t8 + 2), q3 = ldg(r2, t8 + 3), q4 = ldg(r2, t8 + 4), q5 = ldg(r2, t8 + 5),
            q6 = ldg(r2, t8 + 6);
        }
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
        inside = cand & (0 <= u) & (u <= 1) & (0 <= v) & (v <= 1) & (0 <= w) & (w <= 1);
        double mm = v < u ? v : u;
        mm = w < mm ? w : mm;
        m = cand ? mm : -INFINITY;
        ok = cand & (mm >= -1.0);
    }
    int j;
    const uint64_t in_mask = ballot64(inside);
    if (in_mask) {
        j = __builtin_ctzll(in_mask);                               // first triangle containing the point
    } else {
        const uint64_t ok_mask = ballot64(ok);
        if (ok_mask == 0) {
            j = 0;                                                   // nothing beat -1: the first candidate stays
        } else {
            const double mx = wave_max_d(ok ? m : -INFINITY);
            j = 63 - __builtin_clzll(ballot64(ok & (m == mx)));       // last one reaching the maximum
        }
    }
    // the chosen triangle's normal, quaternion and centre offset: one wave-uniform read of its record's tail
    // (measured: every candidate lane fetching its whole record and broadcasting the winner's tail saves the
    // dependent read but costs more than it gains, 48.0 -> 48.8 us)
    const int tj = __builtin_amdgcn_readlane(ti, rfl(j));
    tri = tj;
    f64x2 t6, t7, t8, t9, t10, t11;
    if (coop) {                                     // it came with the candidates: no further round trip
        const f64x2 *g = gather + rfl(j) * (TRI_REC / 2);
        t6 = g[6], t7 = g[7], t8 = g[8], t9 = g[9], t10 = g[10], t11 = g[11];
    } else {
        const f64x2 GAS *rj = reinterpret_cast<const f64x2 GAS *>(P.tri_rec) + (uint32_t)tj * (TRI_REC / 2);
        t6 = rj[6], t7 = rj[7], t8 = rj[8], t9 = rj[9], t10 = rj[10], t11 = rj[11];
    }
    n0 = t6.y;
    n1 = t7.x;
    n2 = t7.y;
    pose[0] = pt[0] + n0 * HOOK_DISTANCE;
    pose[1] = pt[1] + n1 * HOOK_DISTANCE;
    pose[2] = pt[2] + n2 * HOOK_DISTANCE;
    orn[0] = -n0;
    orn[1] = -n1;
    orn[2] = -n2;
    quat[0] = t8.x;
    quat[1] = t8.y;
    quat[2] = t9.x;
    quat[3] = t9.y;
    center[0] = pose[0] + t10.x;
    center[1] = pose[1] + t10.y;
    center[2] = pose[2] + t11.x;
    STAMP(PH_BARY);
    return true;
}

}  // namespace
