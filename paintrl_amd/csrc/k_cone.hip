// k_cone.hip -- PAINT_METHOD 'normal' (rob:251-285 + bpw:562-566): every sub-shot casts the part's cone beams (104-140
// rays) from the tool and paints the sample nearest to each hit.  Compiled once per mask width (-DPRL_KW=1..4), see
// prl_launch.hpp.
//
// A workgroup of WAVES waves owns WAVES envs for one step, in three phases that hand over through LDS only:
//   A  every wave takes its own env through the five sub-shots (prl_step.hpp sub_shot: ray, hook point) and leaves the
//      five tool poses in LDS.  The tool path of a step does not depend on what the beams paint.
//   B  the 5 x ceil(n_beams / 64) beam trips of ALL the workgroup's envs form one queue that the waves drain together,
//      a trip at a time, one beam per lane (prl_cone.hpp).  The beams of a step are independent of each other until their
//      hit bits meet, and envs differ a lot in what their beams cost (an env at the rim of the part, or over a recess of
//      it, takes several times the work of one in the middle): with one wave per env the launch was as long as its
//      slowest env (mean wave life 490 us, launch 1 290 us).  Each beam leaves the sample it paints in LDS.
//   C  every wave folds its env's five hit lists shot by shot into the coverage masks (bpw:572-577), then reward,
//      termination, observation, auto-reset (prl_step.hpp finish_step).
// Each phase starts from laundered lane / wave numbers and the kernel-argument segment, so that none holds another's
// registers (the single-phase kernel of round 2 spilled 70-110 vector registers, in loops).
#include "prl_all.hpp"
#include "prl_cone.hpp"

#ifndef PRL_KW
#error "compile with -DPRL_KW=1..4 (paintrl_amd/build.py)"
#endif

namespace {

__device__ __forceinline__ int opaque_v(int v) {
    asm volatile("" : "+v"(v));
    return v;
}
__device__ __forceinline__ int opaque_s(int v) {
    asm volatile("" : "+s"(v));
    return v;
}

// What phase A leaves for phases B and C, per env of the workgroup.
struct ConeShot {
    double pos[3], quat[4];       // tool pose after the sub-shot: the beams' origin and frame (rob:251-258)
    int hint, pad;                // collision facet the tool's own ray hit (-1: it missed): where the beams' walks start
};
constexpr int CONE_ENV_DOUBLES = PRL_STATE_DOUBLES + 2;      // the state record + {new turning angle, {counter before, facet hint}}

template <int KW, bool GENSEC, bool KD, int WAVES>
__global__ __launch_bounds__(64 * WAVES, 4) void cone_step_kernel(StepArgs, int nb_pad) {
    extern __shared__ int s_hits[];                                   // [WAVES][5][nb_pad]: sample painted by each beam, or -1
    __shared__ ConeShot s_shot[WAVES][PAINT_PER_ACTION];
    __shared__ double s_env[WAVES][CONE_ENV_DOUBLES];
    __shared__ uint64_t s_row[WAVES][64 * KW];                        // phase C: the hit bits of the shot being folded
    __shared__ int s_cand[WAVES][64];
    __shared__ double s_centres[WAVES][PAINT_PER_ACTION * 3 + 1];
    __shared__ int s_cnt[GENSEC ? WAVES : 1][128];
    __shared__ double s_kd[KD ? WAVES : 1][KD ? KD_HEAP * 5 : 1];
    __shared__ int s_trips[WAVES + 1], s_next;
    const int wave0 = rfl((int)(threadIdx.x >> 6));
    // ---------------------------------------------------------------- A: the tool path
    {
        const StepArgs CAS &a = *(const StepArgs CAS *)__builtin_amdgcn_kernarg_segment_ptr();
        const int lane = opaque_v((int)(threadIdx.x & 63)), wave = opaque_s(wave0);
        const int env = opaque_s((int)blockIdx.x) * WAVES + wave;
        int trips = 0;
        if (env < a.n_envs) {
            const int part_id = a.env_part ? a.env_part[env] : 0;
            PartRef P = *(const PartDev CAS *)(a.parts + part_id);
            CfgRef C = *(const PrlConfig CAS *)a.cfg;
            const double *state_rec = a.state + (size_t)env * PRL_STATE_DOUBLES;
            EnvState S;
            load_state_motion(state_rec, S);
            load_state_accumulators(state_rec, S);
            S.last_ep_return = state_rec[13];
            S.last_ep_reward = state_rec[14];
            S.last_ep_len = reinterpret_cast<const int *>(state_rec)[30];
            S.last_ep_painted = reinterpret_cast<const int *>(state_rec)[31];
            double delta1, delta2, new_angle;
            decode_action(C, a.actions, env, delta1, delta2, new_angle);
            const int counter_before = S.terminate_counter;
            const WaveLds wl{s_cand[wave], s_centres[wave], nullptr, s_kd[KD ? wave : 0], nullptr};
            ShotCtx X;
            shots_begin(P, S, delta1, delta2, X);
            PROF_BEGIN();
            for (int shot = 0; shot < PAINT_PER_ACTION; ++shot) {
                double center[3], quat[4];
                sub_shot<KD>(P, lane, S, X, wl, center, quat PROF_PASS);
                if (lane == 0) {
                    ConeShot &o = s_shot[wave][shot];
#pragma unroll
                    for (int k = 0; k < 3; ++k) o.pos[k] = X.cur_pose[k];
#pragma unroll
                    for (int k = 0; k < 4; ++k) o.quat[k] = quat[k];
                    o.hint = X.facet_hint;
                }
            }
            shots_end(P, S, X);
            store_state(s_env[wave], S, lane);
            if (lane == 0) {
                s_env[wave][PRL_STATE_DOUBLES] = new_angle;
                const int pair[2] = {counter_before, X.facet_hint};
                s_env[wave][PRL_STATE_DOUBLES + 1] = __hiloint2double(pair[1], pair[0]);
            }
            trips = PAINT_PER_ACTION * ((P.n_beams + 63) >> 6);
        }
        if (lane == 0) s_trips[wave + 1] = trips;
        if (threadIdx.x == 0) {
            s_trips[0] = 0;
            s_next = 0;
        }
    }
    __syncthreads();
    // ---------------------------------------------------------------- B: all beams of the workgroup's envs, one queue
    {
        const StepArgs CAS &a = *(const StepArgs CAS *)__builtin_amdgcn_kernarg_segment_ptr();
        const int lane = opaque_v((int)(threadIdx.x & 63)), wave = opaque_s(wave0);
        const int env0 = opaque_s((int)blockIdx.x) * WAVES;
        for (;;) {
            int q = 0;
            if (lane == 0) q = atomicAdd(&s_next, 1);
            q = rfl(q);
            // which env, shot and trip is item q?  (at most WAVES envs: a short scalar scan)
            int e = 0, base = 0;
            for (; e < WAVES; ++e) {
                const int t = rfl(s_trips[e + 1]);
                if (q < base + t) break;
                base += t;
            }
            if (e >= WAVES) break;
            const int env = env0 + e;
            PartRef P = *(const PartDev CAS *)(a.parts + (a.env_part ? a.env_part[env] : 0));
            const int tps = (P.n_beams + 63) >> 6, item = q - base;
            const int shot = item / tps, b0 = (item - shot * tps) << 6;
            const ConeShot &sh = s_shot[e][shot];
            const double pos[3] = {uni_d(sh.pos[0]), uni_d(sh.pos[1]), uni_d(sh.pos[2])};
            const double quat[4] = {uni_d(sh.quat[0]), uni_d(sh.quat[1]), uni_d(sh.quat[2]), uni_d(sh.quat[3])};
            const int hint = rfl(sh.hint);
            const int sidx = cone_trip(P, pos, quat, b0, hint, lane, s_cand[wave]);
            if (b0 + lane < P.n_beams) s_hits[(e * PAINT_PER_ACTION + shot) * nb_pad + b0 + lane] = sidx;
        }
    }
    __syncthreads();
    // ---------------------------------------------------------------- C: fold the hit lists, finish the step
    {
        const StepArgs CAS &a = *(const StepArgs CAS *)__builtin_amdgcn_kernarg_segment_ptr();
        const int lane = opaque_v((int)(threadIdx.x & 63)), wave = opaque_s(wave0);
        const int env = opaque_s((int)blockIdx.x) * WAVES + wave;
        if (env >= a.n_envs) return;
        const int part_id = a.env_part ? a.env_part[env] : 0;
        PartRef P = *(const PartDev CAS *)(a.parts + part_id);
        CfgRef C = *(const PrlConfig CAS *)a.cfg;
        double *state_rec = a.state + (size_t)env * PRL_STATE_DOUBLES;
        EnvState S;
        {
            double *dst = reinterpret_cast<double *>(&S);
#pragma unroll
            for (int k = 0; k < PRL_STATE_DOUBLES; ++k) dst[k] = uni_d(s_env[wave][k]);
        }
        const double new_angle = uni_d(s_env[wave][PRL_STATE_DOUBLES]);
        const double pair = s_env[wave][PRL_STATE_DOUBLES + 1];
        const int counter_before = rfl(__double2loint(pair)), facet_hint = rfl(__double2hiint(pair));
        const GlobalMasks masks{a.painted + (size_t)env * a.mask_stride, a.last + (size_t)env * a.mask_stride, P.n_words, lane};
        uint64_t painted[KW_MAX] = {0, 0, 0, 0}, last[KW_MAX] = {0, 0, 0, 0}, valid[KW_MAX] = {0, 0, 0, 0};
        masks.template load<KW>(painted, last);
        uint64_t *row = s_row[wave];
        const int *hits = s_hits + (size_t)wave * PAINT_PER_ACTION * nb_pad;
        uint32_t n_succeeded_l = 0;
        for (int shot = 0; shot < PAINT_PER_ACTION; ++shot) {
            // bpw:562-566 + 572-577: the samples this shot's beams hit are its "affected" set; no hit at all: the
            // reference returns early and leaves the last-shot set untouched (rob:283-285)
#pragma unroll
            for (int k = 0; k < KW; ++k) row[lane + 64 * k] = 0;
            __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
            __builtin_amdgcn_wave_barrier();
            int beam_hits = 0;
            for (int b = 0; b < P.n_beams; b += 64) {
                const int sidx = b + lane < P.n_beams ? hits[shot * nb_pad + b + lane] : -1;
                if (sidx >= 0) atomicOr(reinterpret_cast<unsigned long long *>(&row[sidx >> 6]), 1ull << (sidx & 63));
                beam_hits += __popcll(ballot64(sidx >= 0));
            }
            __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
            __builtin_amdgcn_wave_barrier();
            __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
            if (beam_hits > 0) {
#pragma unroll
                for (int k = 0; k < KW; ++k) {
                    const uint64_t c = row[lane + 64 * k];
                    n_succeeded_l += __popcll(c & ~painted[k]);
                    painted[k] |= c;
                    valid[k] |= c & ~last[k];
                    last[k] = c;
                }
            }
            __builtin_amdgcn_wave_barrier();
        }
        uint32_t pix_l = 0;
#pragma unroll
        for (int k = 0; k < KW; ++k) pix_l += __popcll(valid[k]);
        const uint64_t sums = wave_sum_u64(((uint64_t)n_succeeded_l << 32) | pix_l);
        const int succeeded = (int)(sums >> 32), pixel_counter = (int)(sums & 0xffffffffu);
        const WaveLds wl{s_cand[wave], s_centres[wave], s_cnt[GENSEC ? wave : 0], nullptr, nullptr};
        PROF_BEGIN();
        const int dn = finish_step<KW, GENSEC, false, false>(P, C, part_id, env, lane, S, state_rec, masks, painted, last,
                                                             (double)succeeded, pixel_counter, counter_before, new_angle,
                                                             facet_hint, StepRows{&a}, wl PROF_PASS);
        store_state_live(state_rec, S, lane, dn != 0);
    }
}

template <int WAVES>
int launch_cone(const StepArgs &a, const PrlStepSel &sel, int nb_pad, hipStream_t s) {
    constexpr int KW = PRL_KW;
    void (*k)(StepArgs, int) =
        sel.kd ? (sel.gensec ? cone_step_kernel<KW, true, true, WAVES> : cone_step_kernel<KW, false, true, WAVES>)
               : (sel.gensec ? cone_step_kernel<KW, true, false, WAVES> : cone_step_kernel<KW, false, false, WAVES>);
    const size_t lds = (size_t)WAVES * PAINT_PER_ACTION * nb_pad * sizeof(int);
    if (lds > 32 * 1024) {
        const hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void *>(k), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
        if (e != hipSuccess) return (int)e;
    }
    hipLaunchKernelGGL(k, dim3((a.n_envs + WAVES - 1) / WAVES), dim3(64 * WAVES), lds, s, a, nb_pad);
    return (int)hipGetLastError();
}

}  // namespace

PRL_HIDDEN int KFN(cone)(const void *step_args, const PrlStepSel *sel, void *stream) {
    const StepArgs &a = *static_cast<const StepArgs *>(step_args);
    const int nb_pad = ((sel->max_beams + 63) / 64) * 64;
    return launch_cone<CONE_WAVES>(a, *sel, nb_pad, static_cast<hipStream_t>(stream));
}

#include "prl_diag_export.hpp"
