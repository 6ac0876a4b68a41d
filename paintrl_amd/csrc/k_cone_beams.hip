// k_cone_beams.hip -- PAINT_METHOD 'normal' (rob:251-285 + bpw:562-566): every sub-shot casts the part's cone beams
// (104-140 rays on the reference's parts) from the tool and paints the sample nearest to each hit.
//
// A batched step is FIVE launches on the caller's stream, handing over through HBM buffers of the batch:
//   cone_path_kernel    one wave per env: the five sub-shots of the tool (prl_step.hpp sub_shot: ray, hook point); the
//                       five tool poses go to cone_shots.  The tool path of a step does not depend on what the beams paint.
//   cone_beams_kernel   one wave per beam TRIP (env, shot, 64 beams), one beam per lane: walk over the hull, nearest
//                       sample in the 2 x 2 cells of the fine grid around the hit point (prl_cone.hpp cone_trip_fast) --
//                       the common case only, in few registers.  The 5 x 2 trips of an env are independent of each other
//                       and the hardware dispatcher balances them over the chip: with one wave per env a launch was as
//                       long as its slowest env (an env at the rim of the part, or over a recess of it, takes several
//                       times the work of one in the middle).  What a lane does not settle goes to a work list.
//   cone_rest_kernel    the rays the walk left over, one per wave through the wave-wide closest-hit search (their hit
//                       points: the 2 x 2 cells, or the far list); whole trips of a set that is not convex.
//   cone_far_kernel     the hit points not settled so far (8 % of the hits on the synthetic door: the hull spans windows
//                       and recesses of the part and stands above a curved panel -- centimetres to decimetres from every
//                       sample), sixteen per wave, four lanes each, level by level down the box pyramid over the samples
//                       (nearest_sample_bfs).
//   cone_finish_kernel  (k_cone.hip, per mask width) one wave per env: the five hit lists folded shot by shot into the
//                       coverage masks (bpw:572-577), reward, termination, observation, auto-reset.
// The extra HBM traffic (10 MB of hit lists written and read per 4 096-env step) is 3 us at HBM speed.
#include "prl_all.hpp"
#include "prl_cone.hpp"
#include "prl_pair.hpp"

namespace {

#ifndef PRL_BEAM_WAVES
#define PRL_BEAM_WAVES 2
#endif
constexpr int BEAM_WAVES = PRL_BEAM_WAVES;   // waves (= beam trips) per workgroup of the beams kernel
#ifndef PRL_REST_WGS
#define PRL_REST_WGS 512
#endif
#ifndef PRL_FAR_WGS
#define PRL_FAR_WGS 2048
#endif
constexpr int FAR_WGS = PRL_FAR_WGS;     // workgroups of the far kernel, grid-stride over the far list: what the chip holds at once
constexpr int REST_WGS = PRL_REST_WGS;   // workgroups of the rest kernel, grid-stride over its lists

// The work lists the beams kernel fills (StepArgs::cone_work, cone_far).  Every trip with a far hit point or a leftover ray
// reserves its entries with an atomic add that returns; on ONE counter those ~20 000 adds a step, from eight XCDs, queue
// up behind each other (75 of the beams kernel's 207 us).  So the far list and the ray list are WORK_LISTS sub-lists each,
// chosen by a hash of the trip, with counters on cache lines of their own:
//   cone_work: [0] trips in the trip list, [2] capacity of a far sub-list, then the trip list (capacity: every trip), the
//   ray sub-lists (item << 6 | lane, last facet of the walk; capacity: prl_cone_ray_sub_cap), the far counters, the ray counters
//   (one every 16 ints); cone_far: WORK_LISTS x capacity entries of 32 bytes.
// The kernels that empty them map work chunks (BFS_N far entries, one ray) to sub-lists by a prefix sum over the counters
// (SubLists).
constexpr int WORK_LISTS = PRL_CONE_WORK_LISTS;      // (prl_device.hpp: the host sizes the lists)
__device__ __forceinline__ int cone_items(const StepArgs CAS &a) { return a.n_envs * PAINT_PER_ACTION * (a.cone_nb >> 6); }
__device__ __forceinline__ int ray_sub_cap(const StepArgs CAS &a) {
    return prl_cone_ray_sub_cap(cone_items(a));
}
__device__ __forceinline__ int *ray_list(const StepArgs CAS &a) { return a.cone_work + 4 + (size_t)cone_items(a); }
__device__ __forceinline__ int *far_counters(const StepArgs CAS &a) { return ray_list(a) + 2 * (size_t)WORK_LISTS * ray_sub_cap(a); }
__device__ __forceinline__ int *ray_counters(const StepArgs CAS &a) { return far_counters(a) + 16 * WORK_LISTS; }

__device__ __forceinline__ int wave_scan_incl(int v, int lane) {
#pragma unroll
    for (int o = 1; o < 64; o <<= 1) {
        const int t = __shfl_up(v, o);
        if (lane >= o) v += t;
    }
    return v;
}
// the chunks of `unit` entries of the WORK_LISTS sub-lists in one numbering: lane l holds sub-lists 4 l .. 4 l + 3
struct SubLists {
    int c[4], incl, excl, total, unit;
    __device__ __forceinline__ void load(const int *counters, int cap, int unit_, int lane) {
        static_assert(WORK_LISTS == 256, "four sub-lists a lane");
        unit = unit_;
        int mine = 0;
#pragma unroll
        for (int k = 0; k < 4; ++k) {
            const int n = counters[16 * (4 * lane + k)];
            c[k] = n < cap ? n : cap;                     // (reservations beyond the capacity went to the trip list)
            mine += (c[k] + unit - 1) / unit;
        }
        incl = wave_scan_incl(mine, lane);
        excl = incl - mine;
        total = __builtin_amdgcn_readlane(incl, 63);
    }
    // chunk (wave-uniform, < total) -> its sub-list, its number within it, the sub-list's entry count
    __device__ __forceinline__ void find(int chunk, int &sub, int &j, int &count) const {
        const int L = (int)__builtin_ctzll(ballot64(chunk < incl));
        int rem = chunk - __builtin_amdgcn_readlane(excl, L), k = 0;
        const int c0 = __builtin_amdgcn_readlane(c[0], L), c1 = __builtin_amdgcn_readlane(c[1], L), c2 = __builtin_amdgcn_readlane(c[2], L),
                  c3 = __builtin_amdgcn_readlane(c[3], L);
        const int h0 = (c0 + unit - 1) / unit, h1 = (c1 + unit - 1) / unit, h2 = (c2 + unit - 1) / unit;
        if (rem >= h0) {
            rem -= h0, k = 1;
            if (rem >= h1) {
                rem -= h1, k = 2;
                if (rem >= h2) rem -= h2, k = 3;
            }
        }
        sub = 4 * L + k;
        j = rem;
        count = k == 0 ? c0 : (k == 1 ? c1 : (k == 2 ? c2 : c3));
    }
};

template <bool KD, int WAVES>
__global__ __launch_bounds__(64 * WAVES, 4) void cone_path_kernel(StepArgs) {
    const StepArgs CAS &a = *(const StepArgs CAS *)__builtin_amdgcn_kernarg_segment_ptr();
    const int lane = threadIdx.x & 63;
    const int env = rfl(blockIdx.x * WAVES + (threadIdx.x >> 6));
    if (blockIdx.x == 0) {                                                 // the work lists of this step start empty
        if (threadIdx.x == 0) a.cone_work[0] = 0;
        for (int i = threadIdx.x; i < 2 * WORK_LISTS; i += 64 * WAVES) far_counters(a)[16 * i] = 0;      // (far, then ray counters)
    }
    if (env >= a.n_envs) return;
    const WaveLds wl = wave_lds<false, KD>();
    const int part_id = a.env_part ? a.env_part[env] : 0;
    PartRef P = *(const PartDev CAS *)(a.parts + part_id);
    CfgRef C = *(const CfgDev CAS *)a.cfg;
    double *state_rec = a.state + (size_t)env * PRL_STATE_DOUBLES;
    EnvState S;
    load_state_motion(state_rec, S);
    double delta1, delta2, new_angle;
    decode_action(C, a.actions, env, delta1, delta2, new_angle);
    const int counter_before = S.terminate_counter;
    ShotCtx X;
    shots_begin<KD>(P, S, delta1, delta2, X, lane, wl);
    double *shots = a.cone_shots + (size_t)env * PAINT_PER_ACTION * 8;
    PROF_BEGIN();
#ifndef PRL_NO_PATH_PRIO                   // (A/B switch; issue priority by progress as in step_env: 30.7 -> 29.9 us)
    PRIO_YOUNG_DECL();
#endif
    for (int shot = 0; shot < PAINT_PER_ACTION; ++shot) {
#ifndef PRL_NO_PATH_PRIO
        // the wave that is behind is served first (prl_step.hpp): 3 for the first shots, falling to 0 with progress
        if (shot <= 1) PRIO_BY_PROGRESS(3);
        else if (shot <= 3) PRIO_YOUNG_OLD(3, 2, shot);
        else PRIO_YOUNG_OLD(2, 1, shot);
#endif
        double center[3], quat[4];
        sub_shot<KD>(P, lane, S, X, wl, center, quat PROF_PASS);
        // lanes 0..7 write the record: pos, quat, {facet hint, 0}
        double v = __hiloint2double(0, X.facet_hint);
#pragma unroll
        for (int k = 0; k < 3; ++k) v = lane == k ? X.cur_pose[k] : v;
#pragma unroll
        for (int k = 0; k < 4; ++k) v = lane == 3 + k ? quat[k] : v;
        if (lane < 8) shots[8 * shot + lane] = v;
    }
    shots_end(P, S, X);
    // the motion part of the record (pose, quaternion, off-part bookkeeping) goes back now; the rest of it is the finish
    // kernel's: doubles 0..6 and the int pairs at 10, 11
    {
        const double *src = reinterpret_cast<const double *>(&S);
        double v = 0;
#pragma unroll
        for (int k = 0; k < 7; ++k) v = lane == k ? src[k] : v;
        v = lane == 10 ? src[10] : v;
        v = lane == 11 ? src[11] : v;
        if (lane < 7 || lane == 10 || lane == 11) state_rec[lane] = v;
    }
    if (lane == 0) {
        a.cone_aux[2 * (size_t)env] = new_angle;
        a.cone_aux[2 * (size_t)env + 1] = __hiloint2double(X.facet_hint, counter_before);
    }
}

// The tool path of ONE env by the whole wave (the body of cone_path_kernel): also what the pair kernel below falls back to.
template <bool KD>
__device__ void cone_path_env(const StepArgs CAS &a, int env, int lane, const WaveLds &wl) {
    const int part_id = a.env_part ? a.env_part[env] : 0;
    PartRef P = *(const PartDev CAS *)(a.parts + part_id);
    CfgRef C = *(const CfgDev CAS *)a.cfg;
    double *state_rec = a.state + (size_t)env * PRL_STATE_DOUBLES;
    EnvState S;
    load_state_motion(state_rec, S);
    double delta1, delta2, new_angle;
    decode_action(C, a.actions, env, delta1, delta2, new_angle);
    const int counter_before = S.terminate_counter;
    ShotCtx X;
    shots_begin<KD>(P, S, delta1, delta2, X, lane, wl);
    double *shots = a.cone_shots + (size_t)env * PAINT_PER_ACTION * 8;
    for (int shot = 0; shot < PAINT_PER_ACTION; ++shot) {
        double center[3], quat[4];
        sub_shot<KD>(P, lane, S, X, wl, center, quat);
        double v = __hiloint2double(0, X.facet_hint);
#pragma unroll
        for (int k = 0; k < 3; ++k) v = lane == k ? X.cur_pose[k] : v;
#pragma unroll
        for (int k = 0; k < 4; ++k) v = lane == 3 + k ? quat[k] : v;
        if (lane < 8) shots[8 * shot + lane] = v;
    }
    shots_end(P, S, X);
    {
        const double *src = reinterpret_cast<const double *>(&S);
        double v = 0;
#pragma unroll
        for (int k = 0; k < 7; ++k) v = lane == k ? src[k] : v;
        v = lane == 10 ? src[10] : v;
        v = lane == 11 ? src[11] : v;
        if (lane < 7 || lane == 10 || lane == 11) state_rec[lane] = v;
    }
    if (lane == 0) {
        a.cone_aux[2 * (size_t)env] = new_angle;
        a.cone_aux[2 * (size_t)env + 1] = __hiloint2double(X.facet_hint, counter_before);
    }
}

// [-DPRL_CONE_PATH_PAIRS: round-5 experiment, prl_pair.hpp]  Two envs per wave: lanes 0-31 walk env 2 j, lanes 32-63 env 2 j + 1.
template <bool KD, int WAVES>
__global__ __launch_bounds__(64 * WAVES, 2) void cone_path_pair_kernel(StepArgs) {
    const StepArgs CAS &a = *(const StepArgs CAS *)__builtin_amdgcn_kernarg_segment_ptr();
    const int lane = threadIdx.x & 63;
    const int env0 = rfl(2 * (blockIdx.x * WAVES + (threadIdx.x >> 6)));
    if (blockIdx.x == 0) {                                                 // the work lists of this step start empty
        if (threadIdx.x == 0) a.cone_work[0] = 0;
        for (int i = threadIdx.x; i < 2 * WORK_LISTS; i += 64 * WAVES) far_counters(a)[16 * i] = 0;      // (far, then ray counters)
    }
    if (env0 >= a.n_envs) return;
    const WaveLds wl = wave_lds<false, KD, 0, WAVES>();
    const int pa = a.env_part ? a.env_part[env0] : 0, pb = (a.env_part && env0 + 1 < a.n_envs) ? a.env_part[env0 + 1] : pa;
    PartRef P = *(const PartDev CAS *)(a.parts + pa);
    const bool pairs = env0 + 1 < a.n_envs && pa == pb && P.col_convex && P.nbr_width <= 32 && P.adj_width <= 32;
    if (!pairs) {                                                          // (a last odd env, two parts in one wave, wide tables)
        cone_path_env<KD>(a, env0, lane, wl);
        if (env0 + 1 < a.n_envs) cone_path_env<KD>(a, env0 + 1, lane, wl);
        return;
    }
    CfgRef C = *(const CfgDev CAS *)a.cfg;
    const bool upper = lane >= 32;
    const int l32 = lane & 31, env = env0 + (upper ? 1 : 0);
    // both envs' entry state by the one-env code (scalar loads, wave-uniform arithmetic), then one value per lane
    PairCtx X;
    double new_angle_l;
    int counter_before_l, step_pair_l;                                      // (record double 11 = {last_on_part, step_counter}: the counter goes back as read)
    uint32_t episode_l;
    double quat_l[4];
    {
        EnvState SA, SB;
        load_state_motion(a.state + (size_t)env0 * PRL_STATE_DOUBLES, SA);
        load_state_motion(a.state + (size_t)(env0 + 1) * PRL_STATE_DOUBLES, SB);
        double d1a, d2a, naa, d1b, d2b, nab;
        decode_action(C, a.actions, env0, d1a, d2a, naa);
        decode_action(C, a.actions, env0 + 1, d1b, d2b, nab);
        double na[3], nb[3];
        tcp_orn_norm(SA.pose, SA.quat, na);
        tcp_orn_norm(SB.pose, SB.quat, nb);
#pragma unroll
        for (int k = 0; k < 3; ++k) {
            X.cur_pose[k] = upper ? SB.pose[k] : SA.pose[k];
            X.cur_norm[k] = upper ? nb[k] : na[k];
        }
#pragma unroll
        for (int k = 0; k < 4; ++k) quat_l[k] = upper ? SB.quat[k] : SA.quat[k];
        X.d1 = upper ? uni_d(d1b) : uni_d(d1a);
        X.d2 = upper ? uni_d(d2b) : uni_d(d2a);
        new_angle_l = upper ? uni_d(nab) : uni_d(naa);
        const double delta_2 = X.d2 * P.lwr;                              // (prl_step.hpp shots_begin)
        X.dvec[0] = P.a1 == 0 ? X.d1 : (P.a2 == 0 ? delta_2 : -0.0);
        X.dvec[1] = P.a1 == 1 ? X.d1 : (P.a2 == 1 ? delta_2 : -0.0);
        X.dvec[2] = P.a1 == 2 ? X.d1 : (P.a2 == 2 ? delta_2 : -0.0);
        const int ha = (SA.facet_hint >= 0 && SA.facet_hint < P.n_col_pad) ? SA.facet_hint : -1;
        const int hb = (SB.facet_hint >= 0 && SB.facet_hint < P.n_col_pad) ? SB.facet_hint : -1;
        X.facet_hint = upper ? hb : ha;
        X.last_tri = -1;
        X.last_on_part = upper ? SB.last_on_part : SA.last_on_part;
        X.terminate_counter = upper ? SB.terminate_counter : SA.terminate_counter;
        X.terminate = upper ? SB.terminate : SA.terminate;
        counter_before_l = X.terminate_counter;
        step_pair_l = upper ? SB.step_counter : SA.step_counter;
        episode_l = upper ? SB.episode : SA.episode;
        if constexpr (KD)
            if (wl.kd_staged) kd_stage(P, wl.kd_heap, lane);
    }
    double *shots = a.cone_shots + (size_t)env * PAINT_PER_ACTION * 8;
    for (int shot = 0; shot < PAINT_PER_ACTION; ++shot) {
        pair_sub_shot<KD>(P, lane, X, wl, quat_l);
        // lanes 0..7 of each half write their env's record: pos, quat, {facet hint, 0}
        double v = __hiloint2double(0, X.facet_hint);
#pragma unroll
        for (int k = 0; k < 3; ++k) v = l32 == k ? X.cur_pose[k] : v;
#pragma unroll
        for (int k = 0; k < 4; ++k) v = l32 == 3 + k ? quat_l[k] : v;
        if (l32 < 8) shots[8 * shot + l32] = v;
    }
    // the tool quaternion of the final pose (prl_step.hpp shots_end): quat_l holds it -- the hooked triangle's record tail, or
    // get_pose_orn of the normal after a miss, the same arithmetic either way.  The motion part of the record goes back:
    // doubles 0..6 and the int pairs at 10, 11
    {
        double v = 0;
#pragma unroll
        for (int k = 0; k < 3; ++k) v = l32 == k ? X.cur_pose[k] : v;
#pragma unroll
        for (int k = 0; k < 4; ++k) v = l32 == 3 + k ? quat_l[k] : v;
        v = l32 == 10 ? __hiloint2double(X.terminate_counter, X.terminate) : v;
        v = l32 == 11 ? __hiloint2double(step_pair_l, X.last_on_part) : v;
        (void)episode_l;
        double *state_rec = a.state + (size_t)env * PRL_STATE_DOUBLES;
        if (l32 < 7 || l32 == 10 || l32 == 11) state_rec[l32] = v;
    }
    if (l32 == 0) {
        a.cone_aux[2 * (size_t)env] = new_angle_l;
        a.cone_aux[2 * (size_t)env + 1] = __hiloint2double(X.facet_hint, counter_before_l);
    }
}

// item = (env * 5 + shot) * trips_per_shot + trip
__device__ __forceinline__ bool beam_item(const StepArgs CAS &a, int item, int &env, int &shot, int &b0) {
    const int tps = a.cone_nb >> 6;
    const int es = item / tps;
    b0 = (item - es * tps) << 6;
    env = es / PAINT_PER_ACTION;
    shot = es - env * PAINT_PER_ACTION;
    return env < a.n_envs;
}

#ifndef PRL_BEAM_OCC
#define PRL_BEAM_OCC 7
#endif
__global__ __launch_bounds__(64 * BEAM_WAVES, PRL_BEAM_OCC) void cone_beams_kernel(StepArgs) {
    const StepArgs CAS &a = *(const StepArgs CAS *)__builtin_amdgcn_kernarg_segment_ptr();
    const int lane = threadIdx.x & 63;
    const int item = rfl(blockIdx.x * BEAM_WAVES + (threadIdx.x >> 6));
    int env, shot, b0;
    if (!beam_item(a, item, env, shot, b0)) return;
    const int part_id = a.env_part ? a.env_part[env] : 0;
    PartRef P = *(const PartDev CAS *)(a.parts + part_id);
    if (b0 >= P.n_beams) return;
    CONE_TIME_BEGIN();
    // the shot's record was written by the previous launch: constant here, fetched with scalar loads
    const double CAS *sh = (const double CAS *)(a.cone_shots + ((size_t)env * PAINT_PER_ACTION + shot) * 8);
    const double pos[3] = {sh[0], sh[1], sh[2]}, quat[4] = {sh[3], sh[4], sh[5], sh[6]};
    int state, sidx;
    double bh[3];
    float far_bound;
    int last_facet;
    cone_trip_fast(P, pos, quat, b0, lane, state, bh, sidx, far_bound, last_facet);
    const int dest = (env * PAINT_PER_ACTION + shot) * a.cone_nb + b0 + lane;
#ifdef PRL_CONE_CUT_FAR                               // (timing build, wrong results: far hit points and leftover rays count as misses)
    if (sidx == -2) sidx = -1;
    if (state == 3) state = 2;
#endif
    const uint64_t far = ballot64(sidx == -2), left = ballot64(state == 3);
    const int n_far = (int)__popcll(far), n_left = (int)__popcll(left);
    // (sub-list by trip, scattered: the ten trips of an env that hovers over a window of the part -- every hit point far -- go to
    // ten sub-lists)
    const int sub = (int)(((unsigned)item * 0x9E3779B1u) >> 24) & (WORK_LISTS - 1), far_cap = a.cone_work[2], ray_cap = ray_sub_cap(a);
    int far_base = 0, ray_base = 0;
    bool far_reserved = false, ray_reserved = false;
    // A collision set that is not convex leaves every ray over: such trips go through the general code whole (the trip
    // list).  Otherwise the leftover rays go to the ray list one by one (a wave each in the rest kernel: microseconds -- the
    // 64 rays of a tool inside the hull as 64 waves, not as one wave's joint search of 50 us), the far hit points to the
    // far list; a sub-list that is full sends the trip to the trip list as well.
    bool redo = left != 0 && !P.col_convex;
    CONE_STAT(17, redo);
    if (!redo && far) {
        if (lane == 0) far_base = atomicAdd(far_counters(a) + 16 * sub, n_far);
        far_base = rfl(far_base);
        far_reserved = true;
        redo = far_base + n_far > far_cap;
        CONE_STAT(14, redo);
    }
    if (!redo && left) {
        if (lane == 0) ray_base = atomicAdd(ray_counters(a) + 16 * sub, n_left);
        ray_base = rfl(ray_base);
        ray_reserved = true;
        redo = ray_base + n_left > ray_cap;
        CONE_STAT(16, redo);
    }
    if (redo) {
        CONE_STAT(12, 1);
        if (lane == 0) a.cone_work[4 + atomicAdd(a.cone_work, 1)] = item;
        // a reservation that ran over the end of its sub-list: the entries of it that do lie inside are marked void (the
        // kernels that empty the lists walk a sub-list up to its capacity)
        if (far_reserved && lane < n_far && far_base + lane < far_cap)
            reinterpret_cast<f64x2 *>(a.cone_far)[2 * ((size_t)sub * far_cap + far_base + lane) + 1] = f64x2{0.0, __hiloint2double(0, -1)};
        if (ray_reserved && lane < n_left && ray_base + lane < ray_cap) ray_list(a)[2 * ((size_t)sub * ray_cap + ray_base + lane)] = -1;
        CONE_TIME_END(2);
        return;
    }
    if (state == 3) {                                                    // item << 6 | lane, the facet its walk gave up on
        int *e = ray_list(a) + 2 * ((size_t)sub * ray_cap + ray_base + (int)__popcll(left & ((1ull << lane) - 1)));
        e[0] = (item << 6) | lane;
        e[1] = last_facet;
    }
    if (sidx == -2) {
        const int slot = far_base + (int)__popcll(far & ((1ull << lane) - 1));
        f64x2 *e = reinterpret_cast<f64x2 *>(a.cone_far) + 2 * ((size_t)sub * far_cap + slot);
        e[0] = f64x2{bh[0], bh[1]};
        e[1] = f64x2{bh[2], __hiloint2double(__float_as_int(far_bound), dest)};       // (bound on the squared distance | where the answer goes)
    } else if (state != 3 && b0 + lane < P.n_beams) {
        a.cone_hits[dest] = sidx;
    }
    CONE_TIME_END(2);
}

// What the beams kernel left: three work lists, short, their items chains of dependent reads -- what counts is that every
// item finds a wave at once.  Two kernels, one after the other:
//   cone_rest_kernel  the ray list: single leftover rays, one per wave: the wave-wide closest-hit search of the tool's own
//                     ray; a hit point that one ring of the fine grid does not settle joins the far list.  And the trip list:
//                     trips of a collision set that is not convex (every ray left over) and trips a full sub-list turned
//                     away, one per wave, through the general code (prl_cone.hpp cone_trip).
//   cone_far_kernel   the far list: hit points the rings of the fine grid did not settle, whatever trip, shot and env they
//                     come from: BFS_N per wave, BFS_G lanes each, level by level down the box pyramid over the samples
//                     (prl_cone.hpp nearest_sample_bfs).  Few registers: six waves a SIMD, every entry of a 4 096-env step
//                     has its wave at once.
// (Side by side on two streams they took as long as the longer one plus ~25 us of fork and join; the rays' own far points
// searched inside the rest kernel made that the longer one.)
template <bool PUSH>
__device__ __forceinline__ void cone_rest_work(const StepArgs CAS &a, int first, int stride, int lane, int *fr, const WaveLds &wl);

#if defined(PRL_CONE_TAIL_MERGED) && !defined(PRL_FAR_OCC)
#define PRL_FAR_OCC 4                 // (the rays' wave-wide search needs the registers; the far search runs as fast at four waves a SIMD)
#endif
#ifndef PRL_FAR_OCC
#define PRL_FAR_OCC 6                 // (80 registers: at 8 waves a SIMD the search spills; 45 -> 40 us)
#endif
#ifndef PRL_FAR_WAVES
#define PRL_FAR_WAVES 4
#endif
constexpr int FAR_WAVES = PRL_FAR_WAVES;         // waves a workgroup of the far kernel (a workgroup's slot is free when its last wave is through)
__global__ __launch_bounds__(64 * FAR_WAVES, PRL_FAR_OCC) void cone_far_kernel(StepArgs) {
    __shared__ int s_bfs[FAR_WAVES * BFS_LDS_INTS];
    const StepArgs CAS &a = *(const StepArgs CAS *)__builtin_amdgcn_kernarg_segment_ptr();
    const int lane = threadIdx.x & 63;
    int *fr = s_bfs + (threadIdx.x >> 6) * BFS_LDS_INTS;
    const int wave = rfl(blockIdx.x * FAR_WAVES + (threadIdx.x >> 6)), n_waves = 4 * FAR_WGS;
    const int far_cap = a.cone_work[2];
#ifdef PRL_CONE_TAIL_MERGED
    // [A/B switch] the rest kernel's work inside this launch (no separate launch): leftover rays and trips go to the waves from
    // the TOP of the grid down -- the launch's second generation of waves, which have one far chunk each -- and a ray's hit
    // point that needs the far search gets it at once (nothing is pushed: the far counters are final when this kernel starts)
    cone_rest_work<false>(a, n_waves - 1 - wave, n_waves, lane, fr, wave_lds<false, false, 0, FAR_WAVES>());
#endif
    SubLists lists;
    lists.load(far_counters(a), far_cap, BFS_N, lane);
#if defined(PRL_CONE_TRACE) && PRL_CONE_TRACE == 4
    const unsigned long long far_t0_ = __builtin_amdgcn_s_memrealtime();
#endif
#ifdef PRL_FAR_LOCAL_QUEUE
    // [A/B switch] The chunks of 16 points handed out dynamically INSIDE a workgroup: workgroup g owns the contiguous chunk range
    // [total g / G, total (g + 1) / G) and its four waves draw from it through an LDS counter -- a wave that drew a long search
    // (a point far from every sample: ten rounds down the pyramid instead of five) does not also get a second one by its index.
    __shared__ int s_next;
    if (threadIdx.x == 0) s_next = 0;
    __syncthreads();
    const int c_lo = (int)((long long)lists.total * blockIdx.x / gridDim.x), c_hi = (int)((long long)lists.total * (blockIdx.x + 1) / gridDim.x);
    (void)wave;
    (void)n_waves;
    for (;;) {
        int k_ = 0;
        if (lane == 0) k_ = atomicAdd(&s_next, 1);
        const int chunk = c_lo + rfl(k_);
        if (chunk >= c_hi) break;
#else
    for (int chunk = wave; chunk < lists.total; chunk += n_waves) {
#endif
        CONE_TIME_BEGIN();
        int sub, j, count;
        lists.find(chunk, sub, j, count);
#if defined(PRL_CONE_TRACE) && PRL_CONE_TRACE == 4
        const unsigned long long far_t1_ = __builtin_amdgcn_s_memrealtime();
#endif
        const bool in = BFS_N * j + (lane >> BFS_SHIFT) < count;
        const f64x2 *e = reinterpret_cast<const f64x2 *>(a.cone_far) + 2 * ((size_t)sub * far_cap + BFS_N * j + (in ? lane >> BFS_SHIFT : 0));
        const f64x2 e0 = e[0], e1 = e[1];
        const double pt[3] = {e0.x, e0.y, e1.x};
        const int dest = __double2loint(e1.y);
        const float hint = __int_as_float(__double2hiint(e1.y));
        const bool have = in && dest >= 0;                               // (void entries: see cone_beams_kernel)
        const int part = (have && a.env_part) ? a.env_part[dest / (PAINT_PER_ACTION * a.cone_nb)] : 0;
        uint64_t todo = ballot64(have);
        while (todo) {                                                   // (one trip unless the batch mixes parts)
            const int p = __builtin_amdgcn_readlane(part, __builtin_ctzll(todo));
            const bool mine = have && part == p;
            todo &= ~ballot64(mine);
            PartRef P = *(const PartDev CAS *)(a.parts + p);
#if defined(PRL_CONE_TRACE) && PRL_CONE_TRACE == 4
            const unsigned long long far_t2_ = __builtin_amdgcn_s_memrealtime();
#endif
            const int sidx = nearest_sample_groups<false>(P, pt, mine, hint, lane, fr);      // (every entry carries a bound)
            if (mine && (lane & (BFS_G - 1)) == 0) a.cone_hits[dest] = sidx;
#if defined(PRL_CONE_TRACE) && PRL_CONE_TRACE == 4
            if (lane == 0 && chunk < 16384) {                             // ticks: kernel start -> this chunk, entry read, search
                g_far_trace[4 * chunk] = (unsigned)(far_t1_ - far_t0_);
                g_far_trace[4 * chunk + 1] = (unsigned)(far_t2_ - far_t1_);
                g_far_trace[4 * chunk + 2] = (unsigned)(__builtin_amdgcn_s_memrealtime() - far_t2_);
                g_far_trace[4 * chunk + 3] = (unsigned)(far_t0_ & 0xffffffffu);
            }
#endif
        }
        CONE_TIME_END(0);
    }
}

// The leftover work of the beams kernel for the waves [0, n_waves): whole trips of the trip list through the general code, single
// leftover rays through the wave-wide closest-hit search.  PUSH: a ray's hit point that one ring of the fine grid does not
// settle joins the far list (the rest kernel, which runs BEFORE the far kernel); otherwise it is searched here at once (the
// merged tail kernel: the far list is being emptied by other waves of the same launch).  This wave takes items first, first +
// stride, ... of both lists.
template <bool PUSH>
__device__ __forceinline__ void cone_rest_work(const StepArgs CAS &a, int first, int stride, int lane, int *fr, const WaveLds &wl) {
    const int n_work = rfl(a.cone_work[0]);
    for (int i = first; i < n_work; i += stride) {
        const int item = rfl(a.cone_work[4 + i]);
        int env, shot, b0;
        if (!beam_item(a, item, env, shot, b0)) continue;
        CONE_TIME_BEGIN();
        PartRef P = *(const PartDev CAS *)(a.parts + (a.env_part ? a.env_part[env] : 0));
        const double *sh = a.cone_shots + ((size_t)env * PAINT_PER_ACTION + shot) * 8;
        const double pos[3] = {uni_d(sh[0]), uni_d(sh[1]), uni_d(sh[2])};
        const double quat[4] = {uni_d(sh[3]), uni_d(sh[4]), uni_d(sh[5]), uni_d(sh[6])};
        const int hint = rfl(__double2loint(sh[7]));
        const int sidx = cone_trip(P, pos, quat, b0, (hint >= 0 && hint < P.n_col_pad) ? hint : -1, lane, wl.cand, fr);
        if (b0 + lane < P.n_beams) a.cone_hits[((size_t)env * PAINT_PER_ACTION + shot) * a.cone_nb + b0 + lane] = sidx;
        CONE_TIME_END(1);
    }
    SubLists lists;
    lists.load(ray_counters(a), ray_sub_cap(a), 1, lane);
    const int *rays = ray_list(a);
    for (int i = first; i < lists.total; i += stride) {
        int sub, j, count;
        lists.find(i, sub, j, count);
        const int entry = rfl(rays[2 * ((size_t)sub * ray_sub_cap(a) + j)]), walk_facet = rfl(rays[2 * ((size_t)sub * ray_sub_cap(a) + j) + 1]);
        if (entry < 0) continue;                                         // (void: its trip went to the trip list)
        const int item = entry >> 6, L = entry & 63;
        int env, shot, b0;
        if (!beam_item(a, item, env, shot, b0)) continue;
        CONE_TIME_BEGIN();
        PartRef P = *(const PartDev CAS *)(a.parts + (a.env_part ? a.env_part[env] : 0));
        const double *sh = a.cone_shots + ((size_t)env * PAINT_PER_ACTION + shot) * 8;
        const double pos[3] = {uni_d(sh[0]), uni_d(sh[1]), uni_d(sh[2])};
        const double quat[4] = {uni_d(sh[3]), uni_d(sh[4]), uni_d(sh[5]), uni_d(sh[6])};
        int hint = walk_facet >= 0 ? walk_facet : rfl(__double2loint(sh[7]));       // where the ray's walk ended, or the tool's own facet
        hint = (hint >= 0 && hint < P.n_col_pad) ? hint : -1;
        const int bm = b0 + L;
        double dst[3] = {pos[0], pos[1], pos[2]};
        transform_point(pos, quat, ldg(P.beams, 3 * bm), ldg(P.beams, 3 * bm + 1), ldg(P.beams, 3 * bm + 2), dst);
        int sidx = -1;
        bool pushed = false;
        if (!beam_outside_outline_wave(P, pos, dst, lane)) {
            double tw, hw[3];
            if (ray_closest_wave(P, pos, dst, lane, tw, hw, hint, wl.cand) >= 0) {
                float hintf;
                sidx = nearest_sample_lane_f32(P, hw, true, hintf);      // (every lane the same query)
                if (sidx == -2) {
                    bool search_here = !PUSH;
                    if constexpr (PUSH) {
                        // a hit point far from every sample: one more entry of the far list (the far kernel runs after this one)
                        const int fsub = blockIdx.x & (WORK_LISTS - 1), far_cap = a.cone_work[2];
                        int slot = 0;
                        if (lane == 0) slot = atomicAdd(far_counters(a) + 16 * fsub, 1);
                        slot = rfl(slot);
                        if (slot < far_cap) {
                            if (lane == 0) {
                                f64x2 *e = reinterpret_cast<f64x2 *>(a.cone_far) + 2 * ((size_t)fsub * far_cap + slot);
                                e[0] = f64x2{hw[0], hw[1]};
                                e[1] = f64x2{hw[2], __hiloint2double(__float_as_int(hintf), (int)(((size_t)env * PAINT_PER_ACTION + shot) * a.cone_nb + bm))};
                            }
                            pushed = true;
                        } else {
                            search_here = true;                          // (the sub-list is full)
                        }
                    }
                    if (search_here) sidx = nearest_sample_groups(P, hw, lane < BFS_G, hintf, lane, fr);
                }
            }
        }
        if (lane == 0 && !pushed) a.cone_hits[((size_t)env * PAINT_PER_ACTION + shot) * a.cone_nb + bm] = sidx;
        CONE_TIME_END(1);
    }
}

__global__ __launch_bounds__(256, 4) void cone_rest_kernel(StepArgs) {
    __shared__ int s_bfs[4 * BFS_LDS_INTS];
    const StepArgs CAS &a = *(const StepArgs CAS *)__builtin_amdgcn_kernarg_segment_ptr();
    const int lane = threadIdx.x & 63;
    int *fr = s_bfs + (threadIdx.x >> 6) * BFS_LDS_INTS;
    const int wave = rfl(blockIdx.x * 4 + (threadIdx.x >> 6)), n_waves = 4 * REST_WGS;
    const WaveLds wl = wave_lds<false, false>();
    cone_rest_work<true>(a, wave, n_waves, lane, fr, wl);
}

}  // namespace

PRL_HIDDEN int prl_kc_path(const void *step_args, int kd, int wide, void *stream) {
    const StepArgs &a = *static_cast<const StepArgs *>(step_args);
    hipStream_t s = static_cast<hipStream_t>(stream);
#ifdef PRL_CONE_PATH_PAIRS
    {
        constexpr int W = PRL_CONE_PATH_PAIRS;                             // waves (= env pairs) per workgroup
        const dim3 grid(((a.n_envs + 1) / 2 + W - 1) / W), block(64 * W);
        if (kd) hipLaunchKernelGGL((cone_path_pair_kernel<true, W>), grid, block, 0, s, a);
        else hipLaunchKernelGGL((cone_path_pair_kernel<false, W>), grid, block, 0, s, a);
        (void)wide;
        return (int)hipGetLastError();
    }
#endif
    if (wide) {
        const dim3 grid((a.n_envs + STEP_WAVES_WIDE - 1) / STEP_WAVES_WIDE), block(64 * STEP_WAVES_WIDE);
        if (kd) hipLaunchKernelGGL((cone_path_kernel<true, STEP_WAVES_WIDE>), grid, block, 0, s, a);
        else hipLaunchKernelGGL((cone_path_kernel<false, STEP_WAVES_WIDE>), grid, block, 0, s, a);
    } else {
        const dim3 grid((a.n_envs + STEP_WAVES_NARROW - 1) / STEP_WAVES_NARROW), block(64 * STEP_WAVES_NARROW);
        if (kd) hipLaunchKernelGGL((cone_path_kernel<true, STEP_WAVES_NARROW>), grid, block, 0, s, a);
        else hipLaunchKernelGGL((cone_path_kernel<false, STEP_WAVES_NARROW>), grid, block, 0, s, a);
    }
    return (int)hipGetLastError();
}

// beams, the rest kernel (its rays' far hit points join the far list), the far kernel
PRL_HIDDEN int prl_kc_beams(const void *step_args, void *stream) {
    const StepArgs &a = *static_cast<const StepArgs *>(step_args);
    hipStream_t s = static_cast<hipStream_t>(stream);
    const long long items = (long long)a.n_envs * PAINT_PER_ACTION * (a.cone_nb >> 6);
    hipLaunchKernelGGL(cone_beams_kernel, dim3((unsigned)((items + BEAM_WAVES - 1) / BEAM_WAVES)), dim3(64 * BEAM_WAVES), 0, s, a);
#ifndef PRL_CONE_TAIL_MERGED
    hipLaunchKernelGGL(cone_rest_kernel, dim3(REST_WGS), dim3(256), 0, s, a);
#endif
#ifdef PRL_FAR_LOCAL_QUEUE
    {   // as many workgroups as the chip holds at once (PRL_FAR_OCC waves a SIMD): each works its own slice of the chunks off
        static int resident_wgs = 0;
        if (!resident_wgs) {
            int dev = 0, cus = 256;
            (void)hipGetDevice(&dev);
            (void)hipDeviceGetAttribute(&cus, hipDeviceAttributeMultiprocessorCount, dev);
            resident_wgs = cus * 4 * PRL_FAR_OCC / FAR_WAVES;
        }
        hipLaunchKernelGGL(cone_far_kernel, dim3(resident_wgs), dim3(64 * FAR_WAVES), 0, s, a);
    }
#else
    hipLaunchKernelGGL(cone_far_kernel, dim3(FAR_WGS * 4 / FAR_WAVES), dim3(64 * FAR_WAVES), 0, s, a);
#endif
    return (int)hipGetLastError();
}

#if defined(PRL_CONE_TRACE) && defined(PRL_DIAG_EXPORT)
// diagnostic build only: read and clear the beams kernel's path counters (prl_cone.hpp CONE_STAT)
extern "C" int prl_debug_far_trace(unsigned *out) {                      // out[4 * 16384]
    return hipMemcpyFromSymbol(out, HIP_SYMBOL(g_far_trace), sizeof(unsigned) * 4 * 16384) == hipSuccess ? PRL_OK : PRL_E_HIP;
}
extern "C" int prl_debug_cone_stats(unsigned long long *out) {       // out[32 + 160]: counters, wave-time histograms
    if (hipMemcpyFromSymbol(out, HIP_SYMBOL(g_cone_stat), sizeof(unsigned long long) * 32) != hipSuccess) return PRL_E_HIP;
    if (hipMemcpyFromSymbol(out + 32, HIP_SYMBOL(g_cone_hist), sizeof(unsigned long long) * 160) != hipSuccess) return PRL_E_HIP;
    unsigned long long zero[160] = {0};
    if (hipMemcpyToSymbol(HIP_SYMBOL(g_cone_stat), zero, sizeof(unsigned long long) * 32) != hipSuccess) return PRL_E_HIP;
    if (hipMemcpyToSymbol(HIP_SYMBOL(g_cone_hist), zero, sizeof zero) != hipSuccess) return PRL_E_HIP;
    return PRL_OK;
}
#endif

#include "prl_diag_export.hpp"
