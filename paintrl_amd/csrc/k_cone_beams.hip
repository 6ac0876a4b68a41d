// k_cone_beams.hip -- PAINT_METHOD 'normal' (rob:251-285 + bpw:562-566): every sub-shot casts the part's cone beams
// (104-140 rays on the reference's parts) from the tool and paints the sample nearest to each hit.
//
// A batched step is FOUR launches on the caller's stream, handing over through HBM buffers of the batch:
//   cone_path_kernel    one wave per env: the five sub-shots of the tool (prl_step.hpp sub_shot: ray, hook point); the
//                       five tool poses go to cone_shots.  The tool path of a step does not depend on what the beams paint.
//   cone_beams_kernel   one wave per beam TRIP (env, shot, 64 beams), one beam per lane: walk over the hull, nearest
//                       sample on the fine grid (prl_cone.hpp cone_trip_fast) -- the common case only, in few registers,
//                       so that eight waves share a SIMD and hide each other's dependent table reads.  The 5 x 2 trips of
//                       an env are independent of each other and the hardware dispatcher balances them over the chip:
//                       with one wave per env a launch was as long as its slowest env (an env at the rim of the part, or
//                       over a recess of it, takes several times the work of one in the middle).
//   cone_rest_kernel    two work lists side by side: the hit points the beams kernel found centimetres to decimetres from
//                       every sample (the hull spans windows and recesses of the part, and stands above a curved panel),
//                       64 per wave, each lane walking the box pyramid over the samples (nearest_sample_tree); and the
//                       few trips with a ray the walk left over, through the general searches (cone_trip).
//   cone_finish_kernel  (k_cone.hip, per mask width) one wave per env: the five hit lists folded shot by shot into the
//                       coverage masks (bpw:572-577), reward, termination, observation, auto-reset.
// The extra HBM traffic (10 MB of hit lists written and read per 4 096-env step) is 3 us at HBM speed.
#include "prl_all.hpp"
#include "prl_cone.hpp"

namespace {

constexpr int CONE_RAY_LIST_MAX = PRL_CONE_RAY_LIST_MAX;   // (prl_device.hpp: the host sizes the list)
constexpr int BEAM_WAVES = 4;        // waves (= beam trips) per workgroup of the beams kernel
#ifndef PRL_REST_WGS
#define PRL_REST_WGS 384
#endif
// Workgroups of the rest kernel per role, grid-stride over their lists, the roles interleaved (even / odd workgroups): with
// the lanes' tree stacks in LDS (47 KB a workgroup on the door) a CU holds three workgroups, the chip 768 -- a larger grid
// queues behind itself and the second role would only start when the first is through (2 x 1024: 100 us, of waves that
// take 20-40 us each).
constexpr int REST_WGS = PRL_REST_WGS, FAR_WGS = PRL_REST_WGS;

template <bool KD, int WAVES>
__global__ __launch_bounds__(64 * WAVES, 4) void cone_path_kernel(StepArgs) {
    const StepArgs CAS &a = *(const StepArgs CAS *)__builtin_amdgcn_kernarg_segment_ptr();
    const int lane = threadIdx.x & 63;
    const int env = rfl(blockIdx.x * WAVES + (threadIdx.x >> 6));
    if (blockIdx.x == 0 && threadIdx.x == 0) {                             // the work lists of this step start empty
        a.cone_work[0] = 0;
        a.cone_work[1] = 0;
        a.cone_work[3] = 0;
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
    shots_begin(P, S, delta1, delta2, X);
    double *shots = a.cone_shots + (size_t)env * PAINT_PER_ACTION * 8;
    PROF_BEGIN();
    for (int shot = 0; shot < PAINT_PER_ACTION; ++shot) {
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

// item = (env * 5 + shot) * trips_per_shot + trip
__device__ __forceinline__ bool beam_item(const StepArgs CAS &a, int item, int &env, int &shot, int &b0) {
    const int tps = a.cone_nb >> 6;
    const int es = item / tps;
    b0 = (item - es * tps) << 6;
    env = es / PAINT_PER_ACTION;
    shot = es - env * PAINT_PER_ACTION;
    return env < a.n_envs;
}

// cone_work: [0] trips in the trip list, [1] entries of the far list, [2] its capacity, [3] rays in the ray list, then the
// trip list (capacity: every trip) and the ray list (item << 6 | lane; CONE_RAY_LIST_MAX per trip at most)
__device__ __forceinline__ int *ray_list(const StepArgs CAS &a) {
    return a.cone_work + 4 + (size_t)a.n_envs * PAINT_PER_ACTION * (a.cone_nb >> 6);
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
    cone_trip_fast(P, pos, quat, b0, lane, state, bh, sidx);
    const int dest = (env * PAINT_PER_ACTION + shot) * a.cone_nb + b0 + lane;
    const uint64_t far = ballot64(sidx == -2), left = ballot64(state == 3);
    int far_base = 0;
    // rays the walk left over go to the ray list one by one (a wave each in the rest kernel: microseconds); a trip where more
    // than half the lanes are left over (a collision set that is not convex: all of them) goes through the general code whole
    bool redo = (int)__popcll(left) > CONE_RAY_LIST_MAX;
    if (!redo && far) {
        // hit points far from every sample: one entry each in the far list (the far role takes 64 of them per wave)
        if (lane == 0) far_base = atomicAdd(a.cone_work + 1, (int)__popcll(far));
        far_base = rfl(far_base);
        redo = far_base + (int)__popcll(far) > a.cone_work[2];          // the list is full: the general code settles them too
    }
    if (redo) {
        CONE_STAT(12, 1);
        if (lane == 0) a.cone_work[4 + atomicAdd(a.cone_work, 1)] = item;
        // a reservation that ran over the end of the far list: the entries of it that do lie inside are marked void (the
        // far role of cone_rest_kernel walks the list up to its capacity)
        if (far && (int)__popcll(left) <= CONE_RAY_LIST_MAX) {
            const int slot = far_base + lane;
            if (lane < (int)__popcll(far) && slot < a.cone_work[2])
                reinterpret_cast<f64x2 *>(a.cone_far)[2 * (size_t)slot + 1] = f64x2{0.0, __hiloint2double(0, -1)};
        }
        CONE_TIME_END(2);
        return;
    }
    if (left) {
        int ray_base = 0;
        if (lane == 0) ray_base = atomicAdd(a.cone_work + 3, (int)__popcll(left));
        ray_base = rfl(ray_base);
        if (state == 3) ray_list(a)[ray_base + (int)__popcll(left & ((1ull << lane) - 1))] = (item << 6) | lane;
    }
    if (sidx == -2) {
        const int slot = far_base + (int)__popcll(far & ((1ull << lane) - 1));
        f64x2 *e = reinterpret_cast<f64x2 *>(a.cone_far) + 2 * (size_t)slot;
        e[0] = f64x2{bh[0], bh[1]};
        e[1] = f64x2{bh[2], __hiloint2double(part_id, dest)};
    } else if (state != 3 && b0 + lane < P.n_beams) {
        a.cone_hits[dest] = sidx;
    }
    CONE_TIME_END(2);
}

// The hit points the beams kernel could not settle within three rings of the fine grid, 64 per wave whatever trip, shot
// and env they come from: every lane runs the block search at the radius its own cell asks for (prl_cone.hpp
// nearest_sample_lane<true>: 10-20 cells over the window of a door, where a lane of the beams kernel would drag the 60
// settled lanes of its trip through as many rows).
// What the beams kernel left, in ONE launch (the two lists are short and their items long chains of dependent reads: side
// by side they take as long as the slower of the two):
//   workgroups [0, FAR_WGS): the hit points three rings of the fine grid did not settle, 64 per wave whatever trip, shot
//     and env they come from -- every lane walks the box pyramid over the samples for its own point (prl_cone.hpp
//     nearest_sample_tree);
//   workgroups [FAR_WGS, FAR_WGS + REST_WGS): the trips with a ray the walk left over, one per wave, through the general
//     code (prl_cone.hpp cone_trip).
// Dynamic LDS: the lanes' tree stacks, 2 x tree_cap ints each.
__global__ __launch_bounds__(256) void cone_rest_kernel(StepArgs, int tree_cap) {
    extern __shared__ int s_tree[];
    const StepArgs CAS &a = *(const StepArgs CAS *)__builtin_amdgcn_kernarg_segment_ptr();
    const int lane = threadIdx.x & 63;
    int *stack = s_tree + threadIdx.x;                                  // entry k of this lane: stack[256 k]
    const int role_wg = rfl(blockIdx.x >> 1);
    if ((blockIdx.x & 1) == 0) {
        int n_far = rfl(a.cone_work[1]);
        n_far = n_far < a.cone_work[2] ? n_far : rfl(a.cone_work[2]);  // (entries beyond the capacity went to the trip list)
        for (int i0 = rfl(role_wg * 4 + (threadIdx.x >> 6)) * 64; i0 < n_far; i0 += 64 * 4 * FAR_WGS) {
            CONE_TIME_BEGIN();
            const bool in = i0 + lane < n_far;
            const f64x2 *e = reinterpret_cast<const f64x2 *>(a.cone_far) + 2 * (size_t)(in ? i0 + lane : i0);
            const f64x2 e0 = e[0], e1 = e[1];
            const double pt[3] = {e0.x, e0.y, e1.x};
            const int dest = __double2loint(e1.y), part = __double2hiint(e1.y);
            const bool have = in && dest >= 0;                           // (void entries: see cone_beams_kernel)
            uint64_t todo = ballot64(have);
            while (todo) {                                               // (one trip unless the batch mixes parts)
                const int p = __builtin_amdgcn_readlane(part, __builtin_ctzll(todo));
                const bool mine = have && part == p;
                todo &= ~ballot64(mine);
                PartRef P = *(const PartDev CAS *)(a.parts + p);
                int sidx = mine ? -2 : -1;
                nearest_sample_far<256>(P, pt, lane, sidx, stack, tree_cap);
                if (mine) a.cone_hits[dest] = sidx;
            }
            CONE_TIME_END(0);
        }
        return;
    }
    const WaveLds wl = wave_lds<false, false>();
    const int n_work = rfl(a.cone_work[0]);
    for (int i = rfl(role_wg * 4 + (threadIdx.x >> 6)); i < n_work; i += 4 * REST_WGS) {
        const int item = rfl(a.cone_work[4 + i]);
        int env, shot, b0;
        if (!beam_item(a, item, env, shot, b0)) continue;
        CONE_TIME_BEGIN();
        PartRef P = *(const PartDev CAS *)(a.parts + (a.env_part ? a.env_part[env] : 0));
        const double *sh = a.cone_shots + ((size_t)env * PAINT_PER_ACTION + shot) * 8;
        const double pos[3] = {uni_d(sh[0]), uni_d(sh[1]), uni_d(sh[2])};
        const double quat[4] = {uni_d(sh[3]), uni_d(sh[4]), uni_d(sh[5]), uni_d(sh[6])};
        const int hint = rfl(__double2loint(sh[7]));
        const int sidx = cone_trip<256>(P, pos, quat, b0, (hint >= 0 && hint < P.n_col_pad) ? hint : -1, lane, wl.cand, stack, tree_cap);
        if (b0 + lane < P.n_beams) a.cone_hits[((size_t)env * PAINT_PER_ACTION + shot) * a.cone_nb + b0 + lane] = sidx;
        CONE_TIME_END(1);
    }
    // the ray list: one leftover ray per wave -- the wave-wide closest-hit search of the tool's own ray, then the nearest
    // sample of its hit point (every lane the same query; lane 0's answer counts)
    const int n_rays = rfl(a.cone_work[3]);
    const int *rays = ray_list(a);
    for (int i = rfl(role_wg * 4 + (threadIdx.x >> 6)); i < n_rays; i += 4 * REST_WGS) {
        const int entry = rfl(rays[i]);
        const int item = entry >> 6, L = entry & 63;
        int env, shot, b0;
        if (!beam_item(a, item, env, shot, b0)) continue;
        CONE_TIME_BEGIN();
        PartRef P = *(const PartDev CAS *)(a.parts + (a.env_part ? a.env_part[env] : 0));
        const double *sh = a.cone_shots + ((size_t)env * PAINT_PER_ACTION + shot) * 8;
        const double pos[3] = {uni_d(sh[0]), uni_d(sh[1]), uni_d(sh[2])};
        const double quat[4] = {uni_d(sh[3]), uni_d(sh[4]), uni_d(sh[5]), uni_d(sh[6])};
        int hint = rfl(__double2loint(sh[7]));
        hint = (hint >= 0 && hint < P.n_col_pad) ? hint : -1;
        const int bm = b0 + L;
        double dst[3] = {pos[0], pos[1], pos[2]};
        transform_point(pos, quat, ldg(P.beams, 3 * bm), ldg(P.beams, 3 * bm + 1), ldg(P.beams, 3 * bm + 2), dst);
        int sidx = -1;
        if (!beam_outside_outline_wave(P, pos, dst, lane)) {
            double tw, hw[3];
            if (ray_closest_wave(P, pos, dst, lane, tw, hw, hint, wl.cand) >= 0) {
                sidx = nearest_sample_lane_f32(P, hw, true);
                if (lane != 0) sidx = -1;
                nearest_sample_far<256>(P, hw, lane, sidx, stack, tree_cap);
            }
        }
        if (lane == 0) a.cone_hits[((size_t)env * PAINT_PER_ACTION + shot) * a.cone_nb + bm] = sidx;
        CONE_TIME_END(1);
    }
}

}  // namespace

PRL_HIDDEN int prl_kc_path(const void *step_args, int kd, int wide, void *stream) {
    const StepArgs &a = *static_cast<const StepArgs *>(step_args);
    hipStream_t s = static_cast<hipStream_t>(stream);
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

PRL_HIDDEN int prl_kc_beams(const void *step_args, void *stream) {
    const StepArgs &a = *static_cast<const StepArgs *>(step_args);
    hipStream_t s = static_cast<hipStream_t>(stream);
    const long long items = (long long)a.n_envs * PAINT_PER_ACTION * (a.cone_nb >> 6);
    hipLaunchKernelGGL(cone_beams_kernel, dim3((unsigned)((items + BEAM_WAVES - 1) / BEAM_WAVES)), dim3(64 * BEAM_WAVES), 0, s, a);
    const size_t lds = sizeof(int) * 2 * 256 * (size_t)a.cone_tree_cap;
    if (lds > 48 * 1024) {
        const hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void *>(cone_rest_kernel), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
        if (e != hipSuccess) return (int)e;
    }
    hipLaunchKernelGGL(cone_rest_kernel, dim3(FAR_WGS + REST_WGS), dim3(256), lds, s, a, a.cone_tree_cap);
    return (int)hipGetLastError();
}

#if defined(PRL_CONE_TRACE) && defined(PRL_DIAG_EXPORT)
// diagnostic build only: read and clear the beams kernel's path counters (prl_cone.hpp CONE_STAT)
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
