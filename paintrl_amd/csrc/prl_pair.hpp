// prl_pair.hpp -- TWO envs per wavefront for the five sub-shots of a step (rob:302-329 + bpw:865-880, 508-534): lanes 0-31 carry
// env A, lanes 32-63 env B.  Round-5 experiment (DESIGN "two envs per wave"), built where the sub-shots
// stand alone: the cone-beam step's path kernel (k_cone_beams.hip cone_path_pair_kernel, -DPRL_CONE_PATH_PAIRS).
//
// With one env per wave the sub-shots' float64 arithmetic is wave-uniform -- one env's values computed in all 64 lanes -- and
// their lane-parallel passes use at most 32 lanes (the <= 32 facets around a facet, the <= 12 triangles around a vertex).
// Here every "uniform" quantity is a per-lane value that is equal within a half, lane predicates are read by halves of the
// ballot, reductions stop at the half (DPP row shifts + one row broadcast), and a value is handed round a half with
// ds_bpermute.  The common paths run for both envs in the same instructions:
//   ray:     the facet the previous ray hit (mt_rec on the half's own record), then its vertex neighbours one per lane with
//            the interior-entry exit -- the arithmetic of ray_closest_wave's steps (1) and (2), same values;
//   vertex:  ring 1 of the vertex grid, 32 candidates a trip per env;
//   triangle choice and the winner's record tail.
// Whatever leaves them -- no hint or no interior entry (then the closest hit needs a second neighbourhood or the general
// search), a ring that is not exact, the stale kd-tree, parts that differ between the two envs, neighbour or adjacency rows
// wider than 32 -- runs through the one-env code of prl_ray.hpp / prl_search.hpp for that env alone, with its inputs made
// wave-uniform: same results by construction, the other half idles meanwhile.
#pragma once

namespace {

// ---------------------------------------------------------------- half-wave helpers
// minimum of a 32-bit unsigned value over each half (lanes 0-31 / 32-63), returned to every lane of its half
__device__ __forceinline__ uint32_t half_min_u32(uint32_t v, bool upper) {
    v = dpp_umin<0x111, 0xf>(v);
    v = dpp_umin<0x112, 0xf>(v);
    v = dpp_umin<0x114, 0xf>(v);
    v = dpp_umin<0x118, 0xf>(v);
    v = dpp_umin<0x142, 0xa>(v);                      // row 0 -> row 1, row 2 -> row 3: lanes 31 / 63 hold their half's minimum
    const uint32_t a = (uint32_t)__builtin_amdgcn_readlane((int)v, 31), b = (uint32_t)__builtin_amdgcn_readlane((int)v, 63);
    return upper ? b : a;
}

// the same for doubles known to be >= +0.0 (or NaN): they order like their bit patterns (prl_device.hpp wave_min_nonneg_d)
__device__ __forceinline__ double half_min_nonneg_d(double v, bool upper) {
    const uint32_t hi = (uint32_t)__double2hiint(v), lo = (uint32_t)__double2loint(v);
    const uint32_t mh = half_min_u32(hi, upper);
    const uint32_t ml = half_min_u32(hi == mh ? lo : 0xffffffffu, upper);
    return __hiloint2double((int)mh, (int)ml);
}

__device__ __forceinline__ int half_min_i(int v, bool upper) {      // signed: flip the sign bit, take the unsigned minimum
    return (int)(half_min_u32((uint32_t)v ^ 0x80000000u, upper) ^ 0x80000000u);
}

// maximum of any doubles over each half (values may be negative: no bit-pattern shortcut)
__device__ __forceinline__ double half_max_d(double v, bool upper) {
    double x;
    x = dpp_d<0x111, 0xf>(v); v = OP_MAX(x, v);
    x = dpp_d<0x112, 0xf>(v); v = OP_MAX(x, v);
    x = dpp_d<0x114, 0xf>(v); v = OP_MAX(x, v);
    x = dpp_d<0x118, 0xf>(v); v = OP_MAX(x, v);
    x = dpp_d<0x142, 0xa>(v); v = OP_MAX(x, v);
    const double a = bcast_d(v, 31), b = bcast_d(v, 63);
    return upper ? b : a;
}

// this half's 32 bits of a lane mask (a per-lane value: equal within a half)
__device__ __forceinline__ uint32_t half_bits(uint64_t m, bool upper) { return upper ? (uint32_t)(m >> 32) : (uint32_t)m; }

__device__ __forceinline__ double shfl_d(double v, int src) {        // ds_bpermute, two 32-bit halves
    return __hiloint2double(__shfl(__double2hiint(v), src), __shfl(__double2loint(v), src));
}

// what the five sub-shots of a step carry from one to the next, per lane (equal within a half): prl_step.hpp ShotCtx + the
// off-part bookkeeping of rob:292-300
struct PairCtx {
    double cur_pose[3], cur_norm[3], dvec[3];
    double d1, d2;
    int facet_hint, last_tri;
    int last_on_part, terminate_counter, terminate;
};

// One sub-shot of both envs (prl_step.hpp sub_shot states the one-env form).  `quat`: the tool quaternion at the new pose.
template <bool KD>
__device__ __forceinline__ void pair_sub_shot(PartRef P, int lane, PairCtx &X, const WaveLds &wl, double quat[4]) {
    const bool upper = lane >= 32;
    const int l32 = lane & 31, base = lane & 32;
    // bpw:865-880 get_guided_point
    const double pt[3] = {X.cur_pose[0] + X.dvec[0], X.cur_pose[1] + X.dvec[1], X.cur_pose[2] + X.dvec[2]};
    const double end[3] = {pt[0] + X.cur_norm[0], pt[1] + X.cur_norm[1], pt[2] + X.cur_norm[2]};
    const double d0 = end[0] - pt[0], d1 = end[1] - pt[1], d2 = end[2] - pt[2];
    // ---- the ray (prl_ray.hpp ray_closest_wave steps (1), (2))
    bool resolved = false, on = false;
    double t = INFINITY;
    int new_hint = -1;
    {
        const bool has = X.facet_hint >= 0;
        const double dd = (d0 * d0 + d1 * d1) + d2 * d2;
        double bt = INFINITY, bdet = 0;
        int br = 0x7fffffff, bi = -1;
        bool inside;
        mt_rec(P, has ? X.facet_hint : -1, pt, d0, d1, d2, dd, bt, br, bi, bdet, inside);      // (1) the previous facet alone
        if (inside) {
            resolved = true;
            on = true;
            t = bt;
            new_hint = X.facet_hint;
        }
        const bool need2 = has & !inside;
        if (ballot64(need2)) {                                       // (2) the facets that share a vertex with it, one per lane
            const int i1 = (need2 & (l32 < P.nbr_width)) ? ldg(P.col_nbr, (need2 ? X.facet_hint : 0) * P.nbr_width + l32) : -1;
            double bt2 = INFINITY, bdet2 = 0;
            int br2 = 0x7fffffff, bi2 = -1;
            bool interior;
            mt_rec(P, i1, pt, d0, d1, d2, dd, bt2, br2, bi2, bdet2, interior);
            const uint32_t imh = half_bits(ballot64(interior), upper);
            const int src = base + (imh ? __builtin_ctz(imh) : 0);
            const double tw = shfl_d(bt2, src);
            const int fw = __shfl(bi2, src);
            if (need2 & (imh != 0)) {
                resolved = true;
                on = true;
                t = tw;
                new_hint = fw;
            }
        }
    }
    double hit[3] = {pt[0] + t * d0, pt[1] + t * d1, pt[2] + t * d2};
    // (3) everything else -- no hint, the closest hit of a neighbourhood that is not entered at an interior point, the general
    // search -- by the one-env code, one env at a time
    {
        const uint64_t open = ballot64(!resolved);
#pragma unroll
        for (int hh = 0; hh < 2; ++hh) {
            if (!((open >> (32 * hh)) & 1)) continue;
            const double o_u[3] = {bcast_d(pt[0], 32 * hh), bcast_d(pt[1], 32 * hh), bcast_d(pt[2], 32 * hh)};
            const double e_u[3] = {bcast_d(end[0], 32 * hh), bcast_d(end[1], 32 * hh), bcast_d(end[2], 32 * hh)};
            int hint_u = __builtin_amdgcn_readlane(X.facet_hint, 32 * hh);
            double t_u, hit_u[3] = {0, 0, 0};
            const int f = ray_closest_wave(P, o_u, e_u, lane, t_u, hit_u, hint_u, wl.cand);
            if (upper == (hh == 1)) {
                on = f >= 0;
                new_hint = hint_u;
                hit[0] = hit_u[0];
                hit[1] = hit_u[1];
                hit[2] = hit_u[2];
            }
        }
    }
    X.facet_hint = new_hint;
    // a half without a hit runs the hook point's passes on its guided point (finite) and discards them
    if (!on) {
        hit[0] = pt[0];
        hit[1] = pt[1];
        hit[2] = pt[2];
    }
    // ---- bpw:526 nearest same-side vertex
    int vidx = -1;
    bool vdone = false;
    if constexpr (KD) {
        if (P.n_kd_nodes > 0) {                                      // the stale tree: the serial walk, one env at a time
            const uint64_t want = ballot64(on);
#pragma unroll
            for (int hh = 0; hh < 2; ++hh) {
                if (!((want >> (32 * hh)) & 1)) continue;
                const double p_u[3] = {bcast_d(hit[0], 32 * hh), bcast_d(hit[1], 32 * hh), bcast_d(hit[2], 32 * hh)};
#if defined(PRL_WAVE_TRACE) || defined(PRL_PHASE_TIMING)
                Prof prof = {};                                      // (the query's own stamps are the step kernel's; unused here)
#endif
                const int v = nearest_vertex_kd(P, p_u, lane, wl.kd_heap, wl.kd_staged != 0 PROF_PASS);
                if (upper == (hh == 1)) vidx = v;
            }
            vdone = true;
        }
    }
    if (!vdone) {
        const double h1 = sel3(hit[0], hit[1], hit[2], P.a1), h2 = sel3(hit[0], hit[1], hit[2], P.a2);
        const int icx = cell_coord(h1, P.vg_o1, P.vg_inv, P.vg_nx), icy = cell_coord(h2, P.vg_o2, P.vg_inv, P.vg_ny);
        const int cx0 = icx - 1 < 0 ? 0 : icx - 1, cx1 = icx + 1 > P.vg_nx - 1 ? P.vg_nx - 1 : icx + 1;
        int rb[3], rn[3];
#pragma unroll
        for (int r = 0; r < 3; ++r) {                                 // the rows' ranges: every lane reads its own half's six bounds
            const int rcy = icy - 1 + r;
            const bool okr = (rcy >= 0) & (rcy < P.vg_ny) & (cx0 <= cx1);
            const int at = okr ? rcy * P.vg_nx : 0;
            const int b = ldg(P.vg_start, at + (okr ? cx0 : 0)), e = ldg(P.vg_start, at + (okr ? cx1 + 1 : 0));
            rb[r] = b;
            rn[r] = okr ? e - b : 0;
        }
        const int total = rn[0] + rn[1] + rn[2];
        double best_d = INFINITY;
        int best_rank = 0x7fffffff, best_idx = -1;
        for (int c0 = 0; ballot64(c0 < total) != 0; c0 += 32) {
            const int c = c0 + l32;
            const bool k = c < total;
            int v = rb[0] + c;
            if (c >= rn[0]) v = rb[1] + (c - rn[0]);
            if (c >= rn[0] + rn[1]) v = rb[2] + (c - rn[0] - rn[1]);
            double x, y, z;
            int rk;
            load_vertex(P, k ? v : 0, x, y, z, rk);
            const double dx = x - hit[0], dy = y - hit[1], dz = z - hit[2];
            const double dd = (dx * dx + dy * dy) + dz * dz;
            if (k & ((dd < best_d) | ((dd == best_d) & (rk < best_rank)))) {
                best_d = dd;
                best_rank = rk;
                best_idx = v;
            }
        }
        const double dmin = half_min_nonneg_d(best_d, upper);
        const double lim = P.vg_accept;                              // ring 1: 0.99 * cell
        bool exact = dmin <= lim * lim;
#ifdef PRL_FORCE_FULL_SCANS
        exact = false;
#endif
        const uint32_t tie = half_bits(ballot64(best_d == dmin), upper);
        int src = base + (tie ? __builtin_ctz(tie) : 0);
        if (ballot64((tie & (tie - 1)) != 0)) {                       // equally distant vertices somewhere: the lowest reference rank
            const int rmin = half_min_i(best_d == dmin ? best_rank : 0x7fffffff, upper);
            const uint32_t win = half_bits(ballot64((best_d == dmin) & (best_rank == rmin)), upper);
            src = base + (win ? __builtin_ctz(win) : 0);
        }
        const int vw = __shfl(best_idx, src);
        vidx = tie ? vw : -1;                                         // (no lane reaches the minimum: a NaN query point)
        // a ring that does not settle the query: the expanding search of the one-env code
        const uint64_t more = ballot64(on & !exact);
#pragma unroll
        for (int hh = 0; hh < 2; ++hh) {
            if (!((more >> (32 * hh)) & 1)) continue;
            const double p_u[3] = {bcast_d(hit[0], 32 * hh), bcast_d(hit[1], 32 * hh), bcast_d(hit[2], 32 * hh)};
            const int v = nearest_vertex_wave(P, p_u, lane);
            if (upper == (hh == 1)) vidx = v;
        }
    }
    on = on & (vidx >= 0);
    // ---- bpw:508-523 _get_closest_bary: the triangles around the vertex, one per lane
    double pos[3], orn[3];
    {
        const int vc = vidx >= 0 ? vidx : 0;
        const int ti = (on & (l32 < P.adj_width)) ? ldg(P.vadj, vc * P.adj_width + l32) : -1;      // file order, -1 = pad
        const bool cand = ti >= 0;
        const uint32_t cmask = half_bits(ballot64(cand), upper);
        on = on & (cmask != 0);
        const f64x2 GAS *r2 = reinterpret_cast<const f64x2 GAS *>(P.tri_rec);
        const int t8 = (cand ? ti : 0) * (TRI_REC / 2);
        const f64x2 q0 = ldg(r2, t8), q1 = ldg(r2, t8 + 1), q2 = ldg(r2, t8 + 2), q3 = ldg(r2, t8 + 3), q4 = ldg(r2, t8 + 4), q5 = ldg(r2, t8 + 5),
                    q6 = ldg(r2, t8 + 6);
        const double x0 = hit[0] - q0.x, x1 = hit[1] - q0.y, x2 = hit[2] - q1.x;
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
        const bool inside = cand & (0 <= u) & (u <= 1) & (0 <= v) & (v <= 1) & (0 <= w) & (w <= 1);
        double mm = v < u ? v : u;
        mm = w < mm ? w : mm;
        const double m = cand ? mm : -INFINITY;
        const bool ok = cand & (mm >= -1.0);
        const uint32_t in_mask = half_bits(ballot64(inside), upper);
        int j = in_mask ? __builtin_ctz(in_mask) : 0;                 // first triangle containing the point
        if (ballot64(on & (in_mask == 0))) {                          // none does: the one maximising min(u, v, w), last wins
            const uint32_t ok_mask = half_bits(ballot64(ok), upper);
            const double mx = half_max_d(ok ? m : -INFINITY, upper);
            const uint32_t top = half_bits(ballot64(ok & (m == mx)), upper);
            const int jl = (ok_mask && top) ? 31 - __builtin_clz(top) : 0;      // nothing beat -1: the first candidate stays
            if (in_mask == 0) j = jl;
        }
        const int tj = __shfl(ti, base + j);
        const int tjc = (on & (tj >= 0)) ? tj : 0;
        if (on) X.last_tri = tj;
        const f64x2 GAS *rj = r2 + (uint32_t)tjc * (TRI_REC / 2);
        const f64x2 t6 = rj[6], t7 = rj[7], t8q = rj[8], t9 = rj[9];
        const double n0 = t6.y, n1 = t7.x, n2 = t7.y;
        pos[0] = hit[0] + n0 * HOOK_DISTANCE;
        pos[1] = hit[1] + n1 * HOOK_DISTANCE;
        pos[2] = hit[2] + n2 * HOOK_DISTANCE;
        orn[0] = -n0;
        orn[1] = -n1;
        orn[2] = -n2;
        quat[0] = t8q.x;
        quat[1] = t8q.y;
        quat[2] = t9.x;
        quat[3] = t9.y;
    }
    // ---- rob:313-320: a miss moves the tool in its own frame; the off-part bookkeeping of rob:292-300
    if (!on) {
        X.last_tri = -1;
        double q[4], moved[3];
        orn[0] = X.cur_norm[0];
        orn[1] = X.cur_norm[1];
        orn[2] = X.cur_norm[2];
        pose_orn_quat(orn, q);
        transform_point(X.cur_pose, q, X.d2, X.d1, 0.0, moved);      // rob:317, tool frame [delta2, delta1, 0]
        pos[0] = moved[0];
        pos[1] = moved[1];
        pos[2] = moved[2];
        quat[0] = q[0];
        quat[1] = q[1];
        quat[2] = q[2];
        quat[3] = q[3];
        if (X.last_on_part) {
            X.last_on_part = 0;
        } else {
            X.terminate_counter += 1;
            if (X.terminate_counter > NOT_ON_PART_TERMINATE) X.terminate = 1;
        }
    } else {
        X.last_on_part = 1;
    }
#pragma unroll
    for (int k = 0; k < 3; ++k) {
        X.cur_pose[k] = pos[k];
        X.cur_norm[k] = orn[k];
    }
}

}  // namespace
