// prl_step.hpp -- one PaintGymEnv.step() of one environment by one wavefront (rge:349-368), as a device function
// shared by the per-step kernel (step_kernel) and the persistent rollout-fragment kernel (rollout_fragment_kernel).
// Part of the single translation unit paintrl_hip.hip (device code, anonymous namespace); see that file for the
// overall design.  Compile with -ffp-contract=off.
#pragma once

namespace {

// rge:342-347 _preprocess_action + rob:390-398, 403-409 + rob:352-358: action -> (move per sub-shot along axis 1, along
// axis 2, turning angle).  The moves are the action's deltas over PAINT_PER_ACTION: from the batch's tables for discrete
// actions (CfgDev), divided here for continuous ones.
__device__ __forceinline__ void decode_discrete_action(CfgRef C, int act, double &d1, double &d2, double &new_angle) {
    act = rfl(act);                                  // wave-uniform: the table is read with scalar loads
    act = act < 0 ? 0 : (act >= C.n_discrete ? C.n_discrete - 1 : act);
    d1 = C.act_d1[act];
    d2 = C.act_d2[act];
    new_angle = C.act_angle[act];
}

__device__ __forceinline__ void decode_action(CfgRef C, const void *actions, int env, double &d1, double &d2, double &new_angle) {
    if (C.action_mode == PRL_ACT_DISCRETE) {
        decode_discrete_action(C, reinterpret_cast<const int *>(actions)[env], d1, d2, new_angle);
    } else {
        const double *av = reinterpret_cast<const double *>(actions) + (size_t)env * C.action_dim;
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
        const double delta1 = uni_d(dx * C.step_size), delta2 = uni_d(dy * C.step_size);
        new_angle = uni_d(delta1 != 0 ? atan(fabs(delta2 / delta1)) : PI / 2);
        d1 = uni_d(delta1 / PAINT_PER_ACTION);
        d2 = uni_d(delta2 / PAINT_PER_ACTION);
    }
}

// COLOR_MODE 'HSI' reset (bpw:586 front label (1,1,1), bpw:706-707): every byte 255, every real sample "painted".
template <int KW>
__device__ __forceinline__ void reset_thickness(PartRef P, uint8_t *thick, int lane, uint64_t painted[KW_MAX]) {
    uint64_t *t8 = reinterpret_cast<uint64_t *>(thick);
    for (int i = lane; i < 8 * P.n_words; i += 64) t8[i] = ~0ull;
#pragma unroll
    for (int k = 0; k < KW; ++k) {
        const int w = lane + 64 * k;
        painted[k] = w < P.n_words ? ldg(P.word_valid, w) : 0;
    }
}

// What the five sub-shots of a step carry from one to the next (rob:302-329): tool position and normal (vector
// registers holding wave-uniform values: every consumer is a vector instruction, and scalar registers are the
// scarce kind), the per-shot move, the facet the last ray hit and the triangle the last shot hooked to.
struct ShotCtx {
    double cur_pose[3], cur_norm[3];
    double d1, d2;                 // the action's move per sub-shot along axis a1 / a2 (rob:403-409)
    double dvec[3];                // the guided point's offset from the pose (bpw:865-880), see shots_begin
    int facet_hint, last_tri;
};

template <bool KD = false>
__device__ __forceinline__ void shots_begin(PartRef P, const EnvState &S, double d1, double d2, ShotCtx &X, int lane, const WaveLds &wl) {
    FacetTile *tile = wl.tile;
    if constexpr (KD)
        if (wl.kd_staged) kd_stage(P, wl.kd_heap, lane);
    X.cur_pose[0] = S.pose[0];
    X.cur_pose[1] = S.pose[1];
    X.cur_pose[2] = S.pose[2];
    tcp_orn_norm(S.pose, S.quat, X.cur_norm);
    // (the per-shot deltas and the turning angle are read rarely: those do live in scalar registers)
    X.d1 = uni_d(d1);
    X.d2 = uni_d(d2);
    // facet hit by the previous ray, also across steps (convex fast path); only a cache, but it indexes a table
    X.facet_hint = (S.facet_hint >= 0 && S.facet_hint < P.n_col_pad) ? S.facet_hint : -1;
#ifdef PRL_FACET_TILE
    if (tile) {          // the ray's facet tile (prl_ray.hpp): nothing resident at the start of a launch; the hint's goes on its way
        if (lane == 0) tile->facet = -1;
        if (P.col_convex && X.facet_hint >= 0 && P.nbr_width <= TILE_LANES)
            tile_fill(P, tile, X.facet_hint, tile_ids_load(P, X.facet_hint, lane), lane);
    }
#endif
    // The tool quaternion is a function of the tool normal alone (rob:93-100), so it is not carried through the
    // shots (eight vector registers): after the last one it is read from the record of the triangle that shot
    // hooked to, or recomputed from the normal after a miss -- the same arithmetic either way.
    X.last_tri = -1;
    // the guided point's offset from the pose (bpw:865-880: d1 along axis a1, d2 * lwr along a2) is the same for the
    // five shots; the third component is -0.0, the one addend that leaves every double (either zero too) unchanged
    // (held in scalar registers: as three vector-register pairs it was what the act-and-step kernel spilled to scratch
    // and fetched back in every shot)
    const double delta_2 = X.d2 * P.lwr;
    X.dvec[0] = uni_d(P.a1 == 0 ? X.d1 : (P.a2 == 0 ? delta_2 : -0.0));
    X.dvec[1] = uni_d(P.a1 == 1 ? X.d1 : (P.a2 == 1 ? delta_2 : -0.0));
    X.dvec[2] = uni_d(P.a1 == 2 ? X.d1 : (P.a2 == 2 ? delta_2 : -0.0));
}

// One sub-shot (rob:302-329 + bpw:865-880 + 525-534): guided point, ray, hook point -- or the tool-frame move after a
// miss with the off-part bookkeeping of rob:292-300.  Advances X and S.pose; `center` = the shot centre (rob:277-278),
// `quat` = the tool quaternion at the new pose (the ball painter never reads it: dead code there).
template <bool KD>
__device__ __forceinline__ void sub_shot(PartRef P, int lane, EnvState &S, ShotCtx &X, const WaveLds &wl, double center[3],
                                         double quat[4] PROF_ARG) {
    // bpw:865-880 get_guided_point
    const double pt[3] = {X.cur_pose[0] + X.dvec[0], X.cur_pose[1] + X.dvec[1], X.cur_pose[2] + X.dvec[2]};
    const double end[3] = {pt[0] + X.cur_norm[0], pt[1] + X.cur_norm[1], pt[2] + X.cur_norm[2]};
    double t, hit[3], pos[3], orn[3];
    STAMP(PH_MATH);
#if defined(PRL_CUT) && PRL_CUT >= 5
    TilePrefetch pf;
    pf.facet = -1;
    pf.ids = -1;
    bool on = true;
    t = 0;
    hit[0] = end[0] * 0.1 + pt[0] * 0.9, hit[1] = end[1] * 0.1 + pt[1] * 0.9, hit[2] = end[2] * 0.1 + pt[2] * 0.9;
#else
    TilePrefetch pf;
    pf.facet = -1;
    pf.ids = -1;
    bool on = ray_closest_wave(P, pt, end, lane, t, hit, X.facet_hint, wl.cand, wl.tile, &pf) >= 0;
#endif
    STAMP(PH_RAY);
#if defined(PRL_CUT) && PRL_CUT >= 4
    if (on) {
        for (int k = 0; k < 3; ++k) pos[k] = hit[k] - 0.1 * X.cur_norm[k], orn[k] = X.cur_norm[k], center[k] = hit[k];
        quat[0] = quat[1] = quat[2] = 0, quat[3] = 1;
    }
#else
    if (on) on = hook_point_wave<KD>(P, hit, lane, pos, orn, quat, center, X.last_tri, wl, pf PROF_PASS);
#endif
    if (!on) {
        X.last_tri = -1;
        orn[0] = X.cur_norm[0];
        orn[1] = X.cur_norm[1];
        orn[2] = X.cur_norm[2];
        pose_orn_quat(orn, quat);
        transform_point(X.cur_pose, quat, X.d2, X.d1, 0.0, pos);      // rob:317, tool frame [delta2, delta1, 0]
        transform_point(pos, quat, 0.0, 0.0, SHOT_CENTRE_OFFSET, center);
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
        X.cur_pose[k] = pos[k];
        X.cur_norm[k] = orn[k];
        S.pose[k] = pos[k];
    }
    STAMP(PH_MATH);
}

// After the last sub-shot: the tool quaternion of the final pose (rob:93-100), see shots_begin.
__device__ __forceinline__ void shots_end(PartRef P, EnvState &S, const ShotCtx &X) {
    if (X.last_tri >= 0) {
        const f64x2 GAS *rj = reinterpret_cast<const f64x2 GAS *>(P.tri_rec) + (uint32_t)X.last_tri * (TRI_REC / 2);
        const f64x2 qa = rj[8], qb = rj[9];
        S.quat[0] = qa.x;
        S.quat[1] = qa.y;
        S.quat[2] = qb.x;
        S.quat[3] = qb.y;
    } else {
        pose_orn_quat(X.cur_norm, S.quat);
    }
}

// The rest of a step once the shots are painted: robot bookkeeping (rob:352-358, 425-431), reward / penalty /
// termination (rge:321-340, 289-304), observation (rge:306-319), episode statistics, in-kernel auto-reset
// (rge:370-387), masks back through `masks`.  `succeeded_f` = newly painted samples of the five shots (HSI: the float
// sum of deposited fractions), `pixel_counter` = size of the union of the shots' valid sets (rob:425).  Returns done.
template <int KW, bool GENSEC, bool LATE_ACC, bool HSI, int OBSM = -1, typename MaskIO, typename RowIO>
__device__ __forceinline__ int finish_step(PartRef P, CfgRef C, int part_id, int env, int lane, EnvState &S,
                                           const double *state_rec, const MaskIO &masks, uint64_t painted[KW_MAX],
                                           uint64_t last[KW_MAX], double succeeded_f, int pixel_counter, int counter_before,
                                           double new_angle, int facet_hint, const RowIO &a, const WaveLds &wl,
                                           const uint64_t *last_row PROF_ARG) {
    // last_row: where the masks' last-shot words wait in LDS instead of `last` (step_env with WaveLds::lastrow), or nullptr
    constexpr bool BIG = KW == 0;
    PRIO_YOUNG_DECL();
    if constexpr (LATE_ACC) load_state_accumulators(state_rec, S);
    const double angle_diff = fabs(new_angle - S.last_angle);        // rob:357
    S.last_angle = new_angle;
    S.facet_hint = facet_hint;
    if (S.terminate_counter - counter_before >= PAINT_PER_ACTION && pixel_counter == 0) S.terminate = 1;

    // ---- reward, penalty, termination   rge:321-340, 289-304
    const double rew = succeeded_f / 100;
    S.total_reward += rew;
    double pen = 0.2;
    if (C.overlap_penalty) pen += 0.1 * (1 - (pixel_counter ? succeeded_f / (double)pixel_counter : 0.0));           // rob:425-426
    if (C.turning_penalty) pen += 0.1 * (angle_diff / PI);
    const double actual = rew - pen;
    S.step_counter += 1;
    const double max_pts = C.max_possible_point[part_id & 7];
    const int finished = max_pts > S.total_reward * 100 ? 0 : 1;
    int dn = 0;
    if (C.termination_mode != PRL_TERM_LATE) {       // (the average is only looked at in the 'early' / 'hybrid' modes)
        const double avg = S.total_reward / S.step_counter;
        dn = avg < C.expected_reward[part_id & 7] &&
             (C.termination_mode == PRL_TERM_EARLY || S.total_reward < C.switch_points[part_id & 7]);
    }
    if (!dn) dn = finished || S.terminate || S.step_counter > C.max_episode_len - 1;
    if (!dn) S.total_return += actual;
    STAMP(PH_APPLY);

    PRIO_YOUNG_OLD(1, 0, 6);
    const bool do_reset = dn && C.auto_reset;
    const int od = obs_dim_of(C.obs_mode, C.obs_grad);
    double *obs_row = a.obs() + (size_t)env * od;
    double *term_row = do_reset ? (a.final_obs() ? a.final_obs() + (size_t)env * od : nullptr) : obs_row;
#if defined(PRL_CUT) && PRL_CUT >= 1
    term_row = nullptr;
#endif
    if (term_row) {
        if constexpr (BIG) observation_big<GENSEC, OBSM>(P, C, S.pose, masks.painted, lane, term_row, wl.cnt, wl.cand);
        else observation_wave<KW, GENSEC, OBSM>(P, C, S.pose, painted, lane, term_row, wl.cnt);
    }
    if (lane == 0) {
        a.reward()[env] = actual;
        a.done()[env] = (uint8_t)dn;
        a.info()[2 * (size_t)env] = rew;
        a.info()[2 * (size_t)env + 1] = pen;
    }
    if (dn) {                                   // episode statistics (the RCCL gather payload)
        uint32_t cnt_l = 0;
#pragma unroll
        for (int k = 0; k < KW; ++k) cnt_l += __popcll(painted[k]);
        if constexpr (BIG)
            for (int w = lane; w < masks.n_words; w += 64) cnt_l += __popcll(masks.painted[w]);
        S.last_ep_painted = (int)wave_sum_u64(cnt_l);
        S.last_ep_return = S.total_return;
        S.last_ep_reward = S.total_reward;
        S.last_ep_len = S.step_counter;
    }
    if (do_reset) {
        int start = a.start_idx() ? a.start_idx()[env] : draw_start(C.seed, env, S.episode, P.n_start);
        start = start < 0 ? 0 : (start >= P.n_start ? P.n_start - 1 : start);
        reset_state(P, S, start);
#pragma unroll
        for (int k = 0; k < KW; ++k) {
            painted[k] = 0;
            last[k] = 0;
        }
        // every byte 255 again, every status bit "painted" (bpw:586, 706-707) / the LDS copies cleared
        if constexpr (HSI) reset_thickness<KW>(P, a.thick() + (size_t)env * 64 * a.mask_stride(), lane, painted);
        if constexpr (BIG) masks.clear();
        if constexpr (BIG && HSI) masks.all_painted(P);
        for (int k = lane; k < od; k += 64) obs_row[k] = ldg(P.reset_obs, start * od + k);     // see PartDev::reset_obs
    }
    STAMP(PH_OBS);
    if (last_row && !do_reset) {
#pragma unroll
        for (int k = 0; k < KW; ++k) last[k] = last_row[lane + 64 * k];
    }
    masks.template store<KW>(painted, last);
    return dn;
}

// Advances env `env` by one step (PAINT_METHOD 'fast': the ball query of bpw:568-570; the cone beams of
// PAINT_METHOD 'normal' have their own kernel, prl_cone_step.hpp, around the same sub_shot / finish_step).  In: the
// motion part of S (load_state_motion); with LATE_ACC the episode accumulators are read from `state_rec` only after
// the five sub-shots (the per-step kernel: their scalar registers are then free during the shots), otherwise S is
// complete on entry.  Out: S and the masks advanced (reset if the episode ended and C.auto_reset), the observation in
// row `env` of a.obs (the post-reset one after an auto-reset, the terminal one then goes to a.final_obs if that is not
// null), reward / done / info rows through lane 0; the output addresses are formed where they are used, so that they
// hold no registers during the shots.  Returns done.
//
// The coverage masks are not touched by the five sub-shots, so they are fetched through `masks` (GlobalMasks: HBM,
// LdsMasks: the fragment kernel's LDS copy) only when painting starts and put back after the observation: twelve
// vector registers less during the shots.
//
// `a` (RowIO) names this step's output rows: obs / final_obs / reward / info / done / start_idx members that are
// evaluated where they are used.  Both implementations read the kernel arguments through the constant address
// space at that point; handing the kernel's by-value argument struct down by reference instead makes the compiler
// copy all of it into registers at kernel entry.
// HSI = COLOR_MODE 'HSI' (bpw:384-434): thickness bytes in a.thick, float "succeed counter" (prl_paint.hpp).
// KD: parts of the batch may carry the reference's stale vertex kd-tree (prl_search.hpp nearest_vertex_kd).
template <int KW, bool GENSEC, bool LATE_ACC, bool HSI, bool KD, int OBSM = -1, typename MaskIO, typename RowIO>
__device__ __forceinline__ int step_env(PartRef P, CfgRef C, int part_id, int env, int lane, EnvState &S,
                                        const double *state_rec, const MaskIO &masks, double d1, double d2,
                                        double new_angle, const RowIO &a, const WaveLds &wl PROF_ARG) {
    // KW = 0: a part with more than 16 384 samples; its masks stay in LDS (MaskIO = BigMasks) for the whole step
    constexpr bool BIG = KW == 0;
    PRIO_YOUNG_DECL();
    uint64_t painted[KW_MAX] = {0, 0, 0, 0}, last[KW_MAX] = {0, 0, 0, 0};
    if constexpr (BIG) masks.template load<KW>(painted, last);
    const int counter_before = S.terminate_counter;

    // ---- five chained sub-shots   rob:302-329 + 403-424
    ShotCtx X;
    shots_begin<KD>(P, S, d1, d2, X, lane, wl);
    new_angle = uni_d(new_angle);
    double *cen = wl.cen;
#if defined(PRL_CUT) && PRL_CUT >= 6              // diagnostic instruction-count builds (prl_diag.hpp): phases cut away
    for (int shot = 0; shot < 0; ++shot) {
#else
    for (int shot = 0; shot < PAINT_PER_ACTION; ++shot) {
#endif
        // Issue priority by progress (s_setprio 3: shots 0-1, 2: shots 2-4, 1: painting, 0: observation; the two youngest
        // waves of a SIMD one level higher from shot 2 on).  The SIMD's
        // arbiter serves its oldest wave first: of the four envs that share a SIMD the youngest then ends 14 us after
        // the oldest (27.7 / 31.3 / 36.0 / 41.7 us, tools/wave_trace.py) and the launch waits for it.  With the wave
        // that is behind served first the four end within 6 us of each other: 47.9 -> 42.8 us per step.
#ifndef PRL_PRIO_HI
#define PRL_PRIO_HI 1                      // (shots 0 .. PRL_PRIO_HI at level 3; A/B: 0, 2, 3, and -DPRL_PRIO_C -- profiles/r04_ab_log.txt)
#endif
#ifdef PRL_PRIO_C
        if (shot == 0) PRIO_BY_PROGRESS(3);
        else if (shot <= 2) PRIO_BY_PROGRESS(2);
        else PRIO_BY_PROGRESS(1);
#else
        if (shot <= PRL_PRIO_HI) PRIO_BY_PROGRESS(3);
        else PRIO_YOUNG_OLD(3, 2, shot);
#endif
        double center[3], quat[4];                         // rob:277-278 shot centre
#if defined(PRL_UNIT_STEP) && !defined(PRL_KEEP_PART)
        // (k_step.hip: the part's table pointers are read again in every shot -- scalar loads from the constant cache --
        // instead of kept across the shots, which spilled them to vector lanes and fetched them back with vector
        // instructions: 65 -> 46 spilled scalar registers, 40.2 -> 39.7 us)
        const PartDev CAS *pp = &P;
        asm volatile("" : "+s"(pp));
        sub_shot<KD>(*pp, lane, S, X, wl, center, quat PROF_PASS);
#else
        sub_shot<KD>(P, lane, S, X, wl, center, quat PROF_PASS);
#endif
        // painting is deferred until all five centres are known
#pragma unroll
        for (int k = 0; k < 3; ++k)
            if (lane == 0) cen[3 * shot + k] = center[k];
    }
    // lane 0 wrote the shot centres to LDS, every lane reads them below: order the two within the wave
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
    __builtin_amdgcn_wave_barrier();
    __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
    shots_end(P, S, X);
#ifdef PRL_PRIO_C
    PRIO_BY_PROGRESS(0);
#else
    PRIO_YOUNG_OLD(2, 1, 5);
#endif
    // the last-shot mask waits in this wave's LDS rows where the kernel provides them (wl.lastrow), not in registers
    const bool rows = !BIG && !HSI && wl.lastrow != nullptr;
    if constexpr (!BIG) {
        masks.template load<KW>(painted, last);
        if (rows) {
#pragma unroll
            for (int k = 0; k < KW; ++k) {
                wl.lastrow[lane + 64 * k] = last[k];
                wl.lastrow[64 * KW + lane + 64 * k] = 0;
            }
            __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
            __builtin_amdgcn_wave_barrier();
            __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
        }
    }
    STAMP(PH_LOAD);
    // bpw:568-577 fast_paint + _paint for the five shots
    int succeeded = 0, pixel_counter = 0;
    double succeeded_f = 0.0;                      // HSI: the float sum of deposited fractions
    if constexpr (HSI && BIG) {
        masks.paint_hsi(P, C.paint_radius, cen, a.thick() + (size_t)env * 64 * a.mask_stride(), succeeded_f, pixel_counter,
                        reinterpret_cast<double *>(wl.cand));      // (the candidate list's LDS is free while the painter runs)
    } else if constexpr (HSI) {
        paint_shots_hsi<KW>(P, C.paint_radius, cen, lane, painted, last, a.thick() + (size_t)env * 64 * a.mask_stride(),
                            succeeded_f, pixel_counter, reinterpret_cast<double *>(wl.cand));
    } else {
        if constexpr (BIG) {
            masks.paint(P, C.paint_radius, cen, succeeded, pixel_counter);      // (LDS copies: BigMasks, rows in HBM: HbmMasks)
        } else if (rows) {
#if !defined(PRL_CUT) || PRL_CUT < 2
            paint_shots_union(P, C.paint_radius, cen, lane, RowWords<KW>{painted, wl.lastrow, wl.lastrow + 64 * KW, lane}, succeeded,
                              pixel_counter);
#endif
            __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");      // lane 0 wrote the row, its owners read it back later
            __builtin_amdgcn_wave_barrier();
            __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
        } else {
            uint64_t new_last[KW_MAX] = {0, 0, 0, 0};
#if !defined(PRL_CUT) || PRL_CUT < 2
            paint_shots_union(P, C.paint_radius, cen, lane, RegWords<KW>{painted, last, new_last, lane}, succeeded,
                              pixel_counter, wl.sg_lds);
#endif
#pragma unroll
            for (int k = 0; k < KW; ++k) last[k] = new_last[k];
        }
        succeeded_f = (double)succeeded;
    }
    STAMP(PH_BALL);
    return finish_step<KW, GENSEC, LATE_ACC, HSI, OBSM>(P, C, part_id, env, lane, S, state_rec, masks, painted, last, succeeded_f,
                                                  pixel_counter, counter_before, new_angle, X.facet_hint, a, wl,
                                                  rows ? wl.lastrow + 64 * KW : nullptr PROF_PASS);
}

// Output rows of the per-step kernel: the launch's StepArgs, read from the kernel-argument segment when used.
struct StepRows {
    const StepArgs CAS *k;
    __device__ __forceinline__ uint8_t *thick() const { return k->thick; }
    __device__ __forceinline__ int mask_stride() const { return k->mask_stride; }
    __device__ __forceinline__ double *obs() const { return k->obs; }
    __device__ __forceinline__ double *final_obs() const { return k->final_obs; }
    __device__ __forceinline__ double *reward() const { return k->reward; }
    __device__ __forceinline__ double *info() const { return k->info; }
    __device__ __forceinline__ uint8_t *done() const { return k->done; }
    __device__ __forceinline__ const int *start_idx() const { return k->start_idx; }
};

// The masks of an env of a LARGE part: rows in HBM, working copies in LDS for the length of the kernel
// (painted, last = the previous shot's set, new_last = what this step's last shot affected, zero on entry).
struct BigMasks {
    uint64_t *g_painted, *g_last;                 // HBM rows of this env
    uint64_t *painted, *last, *new_last;          // LDS, n_words each
    int n_words, lane;
    uint64_t *extra;                              // a fourth LDS copy for the painters that need one (HSI: the union of valid sets)
    template <int KW>
    __device__ __forceinline__ void load(uint64_t *, uint64_t *) const {
        for (int w = lane; w < n_words; w += 64) {
            painted[w] = g_painted[w];
            last[w] = g_last[w];
            new_last[w] = 0;
        }
        __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
        __builtin_amdgcn_wave_barrier();
        __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
    }
    __device__ __forceinline__ void clear() const {               // reset: nothing painted, no last shot
        for (int w = lane; w < n_words; w += 64) {
            painted[w] = 0;
            new_last[w] = 0;
        }
        __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
        __builtin_amdgcn_wave_barrier();
        __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
    }
    __device__ __forceinline__ void all_painted(PartRef P) const {    // COLOR_MODE 'HSI' after a reset: every real sample reads painted
        for (int w = lane; w < n_words; w += 64) painted[w] = ldg(P.word_valid, w);
        __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
        __builtin_amdgcn_wave_barrier();
        __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
    }
    template <int KW>
    __device__ __forceinline__ void store(const uint64_t *, const uint64_t *) const {
        for (int w = lane; w < n_words; w += 64) {
            g_painted[w] = painted[w];
            g_last[w] = new_last[w];
        }
    }
    // bpw:568-577 fast_paint + _paint for the five shots on the LDS copies
    __device__ __forceinline__ void paint(PartRef P, double radius, const double *cen, int &succeeded, int &pixel_counter) const {
        paint_shots_union(P, radius, cen, lane, LdsWords{painted, last, new_last, lane}, succeeded, pixel_counter);
        __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");      // lane 0 wrote the words, every lane reads them
        __builtin_amdgcn_wave_barrier();
        __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
    }
};

// The masks of an env of a LARGE part left where they are -- rows in HBM / L2, no LDS copy (round 5: step_kernel_big, the
// fused rollout kernels of large parts).  Three LDS copies of a 70 654-sample part's rows (door_rr_big, rge:116) were 26.6 KB
// per env: four waves a CU, four generations of waves for a 4 096-env launch, and 2 x 17.7 KB of copying per env-step for a
// step that paints a few dozen words.  Here the painter reads and writes the words of its cell block in place (HbmWords,
// prl_paint.hpp), the observation streams the painted row with coalesced 8-byte loads, and the last-shot row is kept zero
// outside the last shot's words by clearing exactly the words that were not zero before and were not visited now:
//   nz: this env's row of StepArgs::last_nz, one bit per word of the last-shot row that may be non-zero (lane k holds the
//       bits of words 64 k .. 64 k + 63; rows of up to 64 x 64 words).  Every writer of `last` of a large part keeps it exact
//       or sets it to all ones (reset_kernel_big: zero, it clears the row).
struct HbmMasks {
    gu64_rw_p painted, last;                      // HBM rows of this env (the names finish_step reads)
    gu64_rw_p nz;
    int n_words, lane;
    mutable uint64_t old_nz = 0, vis = 0, nzn = 0;    // this lane's words of: the set on entry | visited by the painter | non-zero after it
    __device__ __forceinline__ int nz_words() const { return (n_words + 63) >> 6; }
    template <int KW>
    __device__ __forceinline__ void load(uint64_t *, uint64_t *) const {      // (called when the kernel starts: the set arrives under the shots)
        old_nz = lane < nz_words() ? nz[lane] : 0;
        vis = 0;
        nzn = 0;
    }
    // zeroes the words of the last-shot row named by this lane's `set` bits (lane k: words 64 k ..): one store instruction per
    // lane that names any, its 64 words over the 64 lanes
    __device__ __forceinline__ void zero_last(uint64_t set) const {
        uint64_t m = ballot64(set != 0);
        while (m) {
            const int k = __builtin_ctzll(m);
            m &= m - 1;
            const uint64_t sk = bcast_u64(set, k);
            if ((sk >> lane) & 1) last[64 * k + lane] = 0;
        }
    }
    __device__ __forceinline__ void paint(PartRef P, double radius, const double *cen, int &succeeded, int &pixel_counter) const {
        paint_shots_union(P, radius, cen, lane, HbmWords{painted, last, lane, &vis, &nzn}, succeeded, pixel_counter);
        // bpw:575-576: the last shot's set replaces the previous one -- what was set before and not visited now is cleared
        zero_last(old_nz & ~vis);
        if (lane < nz_words() && nzn != old_nz) nz[lane] = nzn;
        old_nz = nzn;
        // lane 0 wrote the words, the observation's lanes read them: same CU, same L1 -- ordered once the stores have left
        __builtin_amdgcn_fence(__ATOMIC_RELEASE, "workgroup");
        __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "workgroup");
    }
    // COLOR_MODE 'HSI' (bpw:384-434) for the five shots, rows in place (paint_shots_hsi_words)
    __device__ __forceinline__ void paint_hsi(PartRef P, double radius, const double *cen, uint8_t *thick, double &succeeded,
                                              int &pixel_counter, double *scratch) const {
        paint_shots_hsi_words(P, radius, cen, lane, HbmWords{painted, last, lane, &vis, &nzn}, thick, succeeded, pixel_counter, scratch);
        zero_last(old_nz & ~vis);
        if (lane < nz_words() && nzn != old_nz) nz[lane] = nzn;
        old_nz = nzn;
        __builtin_amdgcn_fence(__ATOMIC_RELEASE, "workgroup");
        __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "workgroup");
    }
    __device__ __forceinline__ void clear() const {               // reset: nothing painted, no last shot
        for (int w = lane; w < n_words; w += 64) painted[w] = 0;
        zero_last(old_nz);
        if (lane < nz_words() && old_nz != 0) nz[lane] = 0;
        old_nz = 0;
    }
    // COLOR_MODE 'HSI' after a reset: every real sample reads painted (only the thickness kernel calls it)
    __device__ __forceinline__ void all_painted(PartRef P) const {
        for (int w = lane; w < n_words; w += 64) painted[w] = ldg(P.word_valid, w);
    }
    template <int KW>
    __device__ __forceinline__ void store(const uint64_t *, const uint64_t *) const {}
};

// The masks of env `env` in HBM: word w of a mask is read / written by lane w & 63 into slot w >> 6.
// Every launch reads each env's two mask rows once and writes them once -- 21 MB at 4 096 envs of the door, through 8 x 4 MB of
// L2 that also have to hold the 2 MB of part tables every sub-shot reads through a chain of dependent loads.  Plain loads
// and stores leave the rows in L2 as most recently used lines and push tables out: measured, each XCD fetches
// 1.2 MB of tables again in every launch (profiles/hbm_traffic.json, round 3), each first touch at the fabric's latency
// (tools/microbench/l2_cold: 358 ns a dependent hop against 90 ns from L2).  Moving the rows with the
// non-temporal hint (`nt`: allocated for streaming, first to be evicted) -- was the idea: measured, the step took 41.3 us
// with it against 39.9 without (profiles/r04_ab_log.txt), so it is OFF; -DPRL_NT_MASKS is the A/B switch.
template <typename T>
__device__ __forceinline__ T stream_load(const T *p) {
#ifdef PRL_NT_MASKS
    return __builtin_nontemporal_load(p);
#else
    return *p;
#endif
}
template <typename T>
__device__ __forceinline__ void stream_store(T *p, T v) {
#ifdef PRL_NT_MASKS
    __builtin_nontemporal_store(v, p);
#else
    *p = v;
#endif
}

// the mask rows of env `env` of the launch, left in HBM (HbmMasks)
__device__ __forceinline__ HbmMasks hbm_masks(const StepArgs CAS &a, int env, int n_words, int lane) {
    return HbmMasks{(gu64_rw_p)(a.painted + (size_t)env * a.mask_stride), (gu64_rw_p)(a.last + (size_t)env * a.mask_stride),
                    (gu64_rw_p)(a.last_nz + (size_t)env * a.nz_stride), n_words, lane};
}

template <bool TRACK>
struct GlobalMasksT {
    uint64_t *painted, *last;     // rows of this env
    int n_words, lane;
    // StepArgs::last_nz row of this env (or nullptr): which words of the last-shot row are not zero -- a dozen of the door's 158.
    // TRACK kernels read only those (the row is otherwise 1.2 KB of zeros per env-step) and record the new set; the others
    // mark every word (the row is then read whole next time).
    uint64_t *nz;
    // this lane's bits of `nz` (bit k: word lane + 64 k), fetched by prefetch() when the kernel starts -- with the state
    // record, not in front of the masks' own loads (read there, the set cost a dependent round trip: 39.7 against 39.3 us)
    mutable uint32_t mine = ~0u;
    __device__ __forceinline__ void prefetch() const {
#ifndef PRL_LOAD_ALL_WORDS                           // (A/B switch)
        if constexpr (TRACK) {
            if (nz) {
                uint32_t m = 0;
#pragma unroll
                for (int k = 0; k < KW_MAX; ++k) m |= (uint32_t)((nz[k] >> lane) & 1) << k;
                mine = m;
            }
        }
#endif
    }
    // TRACK: what load() read, so that store() writes back only the words a step changed (a step paints ~7 of the door's 158
    // words and clears / sets a dozen of the last-shot mask: the rest of the two rows would be rewritten with what it holds --
    // 2.5 KB per env-step, most of the launch's write traffic; -DPRL_STORE_ALL_WORDS is the A/B switch).  Costs 4 KW vector
    // registers: the step kernels track (except the atan2-sector variants), the other kernels do not.
    mutable uint64_t p0[TRACK ? KW_MAX : 1], l0[TRACK ? KW_MAX : 1];
    template <int KW>
    __device__ __forceinline__ void load(uint64_t p[KW_MAX], uint64_t l[KW_MAX]) const {
#pragma unroll
        for (int k = 0; k < KW; ++k) {
            const uint32_t w = lane + 64 * k;
            const bool in = (int)w < n_words;
            p[k] = in ? stream_load(painted + w) : 0;
            const bool lin = in && ((mine >> k) & 1);
            l[k] = lin ? stream_load(last + w) : 0;
            if constexpr (TRACK) {
                p0[k] = p[k];
                l0[k] = l[k];
            }
        }
    }
    template <int KW>
    __device__ __forceinline__ void store(const uint64_t p[KW_MAX], const uint64_t l[KW_MAX]) const {
#pragma unroll
        for (int k = 0; k < KW; ++k) {
            const uint32_t w = lane + 64 * k;
            if constexpr (TRACK) {
                if (nz) {
                    const uint64_t set = ballot64((int)w < n_words && l[k] != 0);
                    if (lane == 0) nz[k] = set;
                }
            }
            if ((int)w < n_words) {
#ifdef PRL_STORE_ALL_WORDS
                constexpr bool changed_only = false;
#else
                constexpr bool changed_only = TRACK;
#endif
                if constexpr (changed_only) {
                    if (p[k] != p0[k]) stream_store(painted + w, p[k]);
                    if (l[k] != l0[k]) stream_store(last + w, l[k]);
                } else {
                    stream_store(painted + w, p[k]);
                    stream_store(last + w, l[k]);
                }
            }
        }
    }
};
typedef GlobalMasksT<false> GlobalMasks;

// A kernel that writes last-shot rows without tracking them (the persistent rollout kernels) says so ONCE, before its first
// store: every word of env `env`'s row counts as non-zero until a tracking kernel has read it whole and recorded the real set.
__device__ __forceinline__ void last_row_untracked(const StepArgs CAS &a, int env, int lane) {
    if (a.last_nz && lane < KW_MAX) a.last_nz[(size_t)env * KW_MAX + lane] = ~0ull;
}

// the mask rows of env `env` of the launch (GlobalMasksT above)
template <bool TRACK = false>
__device__ __forceinline__ GlobalMasksT<TRACK> global_masks(const StepArgs CAS &a, int env, int n_words, int lane) {
    uint64_t *nz = a.last_nz ? a.last_nz + (size_t)env * KW_MAX : nullptr;
    return GlobalMasksT<TRACK>{a.painted + (size_t)env * a.mask_stride, a.last + (size_t)env * a.mask_stride, n_words, lane, nz};
}

}  // namespace
