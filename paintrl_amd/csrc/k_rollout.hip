// k_rollout.hip -- the rollout paths: policy + env step in one launch (act_step_kernel), whole fragments in ONE
// persistent launch with the policy (rollout_policy_kernel) or with given actions (rollout_fragment_kernel).
// Compiled once per mask width (-DPRL_KW=1..4, and -DPRL_KW=0 for parts beyond 16 384 samples: their mask rows stay in HBM,
// prl_step.hpp HbmMasks, so these kernels carry no mask LDS for them), see prl_launch.hpp.  KD: parts of the batch may carry the reference's
// stale vertex kd-tree (bpw:943-946; the reference's own sheet `square.urdf` does), walked in per-wave LDS rows.
#define PRL_UNIT_STEP 1                    // (prl_step.hpp step_env: the part's table pointers re-read per sub-shot; fragment 21.4 -> 21.8 k,
                                           // given actions 29.5 -> 30.3 k steps/s)
#include "prl_all.hpp"
#include "prl_kargs.hpp"
#define PRL_HAVE_F32X4
#include "prl_policy.hpp"

#ifndef PRL_KW
#error "compile with -DPRL_KW=0..4 (paintrl_amd/build.py)"
#endif

namespace {

// ---------------------------------------------------------------- rollout fragment: policy + step, T times, one launch
// The caller of the step in BASELINE.json configs 3-4 is a rollout worker (paint_ppo.py:170-195, fragments of
// sample_batch_size = 100 steps).  One launch per step makes every step end with a grid-wide wait for the slowest
// of all waves, and costs two launches plus an observation round trip through HBM.  Here a four-wave workgroup owns
// four envs for the whole fragment:
//     repeat T times:  stage the 4 observations in LDS -> the policy's three layers on the matrix cores (rows 4..15
//                      of the 16-row MFMA tiles are zero) -> one draw per env -> workgroup barrier
//                      -> every wave steps its env (prl_step.hpp) -> workgroup barrier
// so a slow env delays its three neighbours, not the whole batch, there is nothing to launch, and while one
// workgroup of a CU runs its policy on the matrix pipe the other three step their envs on the vector pipe.  The
// coverage masks stay in LDS from the first step to the last (no mask traffic to HBM in between); trajectory rows
// are written straight into the caller's [T][N] buffers, bit for bit what T rounds of prl_policy_act +
// prl_batch_step write.  All workgroups must be resident together (4 per CU at 4096 envs): 23 KB of LDS each.
// Envs (= waves) per workgroup of the two kernels below = rows of the policy's MFMA tiles: one workgroup per CU at
// 4 096 envs, four waves per SIMD.
constexpr int FRAG_WAVES = POLICY_WAVES;

// Output rows of step t of a fragment (see StepRows in prl_step.hpp).
struct FragmentRows {
    const FragmentArgs CAS *f;
    int t, n, od;
    __device__ __forceinline__ uint8_t *thick() const { return f->s.thick; }          // (COLOR_MODE 'HSI')
    __device__ __forceinline__ int mask_stride() const { return f->s.mask_stride; }
    __device__ __forceinline__ double *obs() const { return f->obs + (size_t)(t + 1) * n * od; }
    __device__ __forceinline__ double *final_obs() const { return f->final_obs ? f->final_obs + (size_t)t * n * od : nullptr; }
    __device__ __forceinline__ double *reward() const { return f->reward + (size_t)t * n; }
    __device__ __forceinline__ double *info() const { return f->info + (size_t)t * n * 2; }
    __device__ __forceinline__ uint8_t *done() const { return f->done + (size_t)t * n; }
    __device__ __forceinline__ const int *start_idx() const { return nullptr; }
};

// The masks of one env in the workgroup's LDS (same word-to-lane mapping as GlobalMasks).
struct LdsMasks {
    uint64_t *painted, *last;
    int n_words, lane;
    template <int KW>
    __device__ __forceinline__ void load(uint64_t p[KW_MAX], uint64_t l[KW_MAX]) const {
#pragma unroll
        for (int k = 0; k < KW; ++k) {
            const int w = lane + 64 * k;
            const bool in = w < n_words;
            p[k] = in ? painted[w] : 0;
            l[k] = in ? last[w] : 0;
        }
    }
    template <int KW>
    __device__ __forceinline__ void store(const uint64_t p[KW_MAX], const uint64_t l[KW_MAX]) const {
#pragma unroll
        for (int k = 0; k < KW; ++k) {
            const int w = lane + 64 * k;
            if (w < n_words) {
                painted[w] = p[k];
                last[w] = l[k];
            }
        }
    }
};

// Every iteration of the fragment loop starts from this pointer: the compiler cannot tell that it is the same
// one each time, so nothing derived from the kernel arguments or the part descriptor is hoisted out of the loop
// and held in registers across both phases (that cost the first version of this kernel 950 spilled registers).
__device__ __forceinline__ const FragmentArgs CAS *opaque(const FragmentArgs CAS *p) {
    asm volatile("" : "+s"(p));
    return p;
}
// ... and the same for what is derived from the lane and wave numbers (per-lane offsets, lane predicates).
__device__ __forceinline__ int opaque_v(int v) {
    asm volatile("" : "+v"(v));
    return v;
}
__device__ __forceinline__ int opaque_s(int v) {
    asm volatile("" : "+s"(v));
    return v;
}

template <int KW, bool KD, bool GRID, bool HSI>
__global__ __launch_bounds__(64 * FRAG_WAVES, 4) void rollout_fragment_kernel(FragmentArgs) {
    extern __shared__ float lds[];
    const FragmentArgs CAS *f0 = (const FragmentArgs CAS *)__builtin_amdgcn_kernarg_segment_ptr();
    const int lane0 = threadIdx.x & 63, wave0 = rfl((int)(threadIdx.x >> 6));
    const int lane = lane0, wave = wave0, env0 = blockIdx.x * FRAG_WAVES, env = env0 + wave;
    if constexpr (KW != 0) {   // coverage masks: HBM -> LDS, once
        const FragmentArgs CAS &f = *opaque(f0);
        const StepArgs CAS &a = f.s;
        if (env < a.n_envs) {
            PartRef P = *(const PartDev CAS *)(a.parts + rfl(a.env_part ? a.env_part[env] : 0));
            uint64_t *mask_lds = reinterpret_cast<uint64_t *>(lds) + (size_t)wave * 2 * a.mask_stride;
            const GlobalMasks g = global_masks(a, env, P.n_words, lane);
            const LdsMasks m{mask_lds, mask_lds + a.mask_stride, P.n_words, lane};
            uint64_t p[KW_MAX] = {0, 0, 0, 0}, l[KW_MAX] = {0, 0, 0, 0};
            g.template load<KW>(p, l);
            m.template store<KW>(p, l);
            last_row_untracked(a, env, lane);                 // (the rows come back whole at the end of the fragment)
        }
    }
    FRAG_DECL();
    for (int t = 0;; ++t) {
        const FragmentArgs CAS &f = *opaque(f0);
        const StepArgs CAS &a = f.s;
        const int lane = opaque_v(lane0), wave = opaque_s(wave0);
        const int env0 = opaque_s((int)blockIdx.x) * FRAG_WAVES, env = env0 + wave;
        const int n_envs = a.n_envs, T = f.T;
        const size_t n = (size_t)n_envs;
        if (t >= T) break;
        FRAG_T(ft1);
        if (env < n_envs) {                                           // exactly the per-step kernel's body
            const int part_id = rfl(a.env_part ? a.env_part[env] : 0);       // (wave-uniform: the part's fields are scalar reads)
            PartRef P = *(const PartDev CAS *)(a.parts + part_id);
            CfgRef C = *(const CfgDev CAS *)a.cfg;
            uint64_t *mask_lds = reinterpret_cast<uint64_t *>(lds) + (size_t)wave * 2 * a.mask_stride;
            // (large parts: the rows stay in HBM and are worked on in place)
            const auto masks = [&] {
                if constexpr (KW == 0) return hbm_masks(a, env, P.n_words, lane);
                else return LdsMasks{mask_lds, mask_lds + a.mask_stride, P.n_words, lane};
            }();
            double *state_rec = a.state + (size_t)env * PRL_STATE_DOUBLES;
            EnvState S;
            load_state_motion(state_rec, S);
            double delta1, delta2, new_angle;
            decode_discrete_action(C, f.action[(size_t)t * n + env], delta1, delta2, new_angle);
            const FragmentRows row{&f, t, n_envs, obs_dim_of(C.obs_mode, C.obs_grad)};
            __shared__ int s_cand[FRAG_WAVES][64];
            __shared__ double s_centres[FRAG_WAVES][PAINT_PER_ACTION * 3 + 1];
            __shared__ double s_kd[KD ? FRAG_WAVES : 1][KD ? KD_HEAP * 5 : 1];
            const WaveLds wl{s_cand[wave], s_centres[wave], nullptr, s_kd[KD ? wave : 0], nullptr, nullptr, nullptr, 0};      // (sixteen waves' tree copies do not fit)
            PROF_BEGIN();
            const int dn = step_env<KW, false, true, HSI, KD, GRID ? 1 : 0>(P, C, part_id, env, lane, S, state_rec, masks, delta1, delta2,
                                                                new_angle, row, wl PROF_PASS);
            store_state_live(state_rec, S, lane, dn != 0);
        }
        FRAG_T(ft2);
        FRAG_ACC(1, ft1, ft2);
        FRAG_COUNT();
    }
    FRAG_FLUSH();
    if constexpr (KW != 0) {   // coverage masks: LDS -> HBM
        const FragmentArgs CAS &f = *opaque(f0);
        const StepArgs CAS &a = f.s;
        if (env < a.n_envs) {
            PartRef P = *(const PartDev CAS *)(a.parts + rfl(a.env_part ? a.env_part[env] : 0));
            uint64_t *mask_lds = reinterpret_cast<uint64_t *>(lds) + (size_t)wave * 2 * a.mask_stride;
            const GlobalMasks g = global_masks(a, env, P.n_words, lane);
            const LdsMasks m{mask_lds, mask_lds + a.mask_stride, P.n_words, lane};
            uint64_t p[KW_MAX] = {0, 0, 0, 0}, l[KW_MAX] = {0, 0, 0, 0};
            m.template load<KW>(p, l);
            g.template store<KW>(p, l);
        }
    }
}
// The env step of act_step_kernel: everything is derived afresh from laundered lane / wave numbers and from the
// kernel-argument segment, so that nothing of the policy phase is still held in registers (the step is at its ceiling).
template <int KW, bool KD, bool GRID, bool HSI>
__device__ __forceinline__ void act_step_env(int env, int lane, int wave, int act, int (*s_cand)[64],
                                             double (*s_centres)[PAINT_PER_ACTION * 3 + 1], double (*s_kd)[KD ? KD_HEAP * 5 : 1]) {
    const ActStepArgs CAS &f = *(const ActStepArgs CAS *)__builtin_amdgcn_kernarg_segment_ptr();
    const StepArgs CAS &a = f.s;
    const int part_id = rfl(a.env_part ? a.env_part[env] : 0);       // (wave-uniform: the part's fields are scalar reads)
    PartRef P = *(const PartDev CAS *)(a.parts + part_id);
    CfgRef C = *(const CfgDev CAS *)a.cfg;
    double *state_rec = a.state + (size_t)env * PRL_STATE_DOUBLES;
    EnvState S;
    load_state_motion(state_rec, S);
    const auto masks = [&] {
        if constexpr (KW == 0) return hbm_masks(a, env, P.n_words, lane);      // (large parts: rows worked on in place)
        else return global_masks<true>(a, env, P.n_words, lane);                // (changed words only: GlobalMasksT)
    }();
    if constexpr (KW != 0) masks.prefetch();
    double delta1, delta2, new_angle;
    decode_discrete_action(C, act, delta1, delta2, new_angle);
    const WaveLds wl{s_cand[wave], s_centres[wave], nullptr, s_kd[KD ? wave : 0], nullptr, nullptr, nullptr, 0};      // (sixteen waves' tree copies do not fit)
    PROF_BEGIN();
    const int dn = step_env<KW, false, true, HSI, KD, GRID ? 1 : 0>(P, C, part_id, env, lane, S, state_rec, masks, delta1, delta2, new_angle,
                                                        StepRows{&a}, wl PROF_PASS);
    store_state_live(state_rec, S, lane, dn != 0);
}

// ---------------------------------------------------------------- policy + env step in one launch
// What a rollout worker does per step (paint_ppo.py:170-195: policy forward, sample, env.step) as ONE kernel: the
// sixteen envs of a workgroup first run the policy on their observations together (prl_policy.hpp: three MFMA layers,
// ~3 us, bound by the weight reads it issues up front), each wave then steps its own env with the sampled action.
// No second launch and no ~2.5 us of dispatch gaps per step; rows as prl_policy_act + prl_batch_step write them.
template <int KW, bool KD, bool GRID, bool HSI>
__global__ __launch_bounds__(64 * POLICY_WAVES) void act_step_kernel(ActStepArgs) {
    extern __shared__ float lds[];
    __shared__ int s_cand[POLICY_WAVES][64];
    __shared__ double s_centres[POLICY_WAVES][PAINT_PER_ACTION * 3 + 1];
    __shared__ double s_kd[KD ? POLICY_WAVES : 1][KD ? KD_HEAP * 5 : 1];
    const ActStepArgs CAS &f = *(const ActStepArgs CAS *)__builtin_amdgcn_kernarg_segment_ptr();
    const StepArgs CAS &a = f.s;
    const int tid = threadIdx.x, lane = tid & 63, wave = rfl(tid >> 6);
    const int env0 = blockIdx.x * POLICY_WAVES, env = env0 + wave, n_envs = a.n_envs;
    FRAG_DECL();
    FRAG_T(ft0);
    int act = 0;
    {
        PrlPolicyWeights W;                                      // (no implicit copy out of the constant address space)
        W.in_dim = f.w.in_dim; W.h1 = f.w.h1; W.h2 = f.w.h2; W.n_actions = f.w.n_actions;
        W.w1 = f.w.w1; W.b1 = f.w.b1; W.w2 = f.w.w2; W.b2 = f.w.b2; W.w3 = f.w.w3; W.b3 = f.w.b3;
        const PolicyLds L = policy_lds_layout(W);
        const int rows_real = n_envs - env0 < POLICY_WAVES ? n_envs - env0 : POLICY_WAVES;
        SamplerPre sp;
        policy_forward(W, f.obs_in + (size_t)env0 * W.in_dim, rows_real, lds, L, tid, env0, nullptr, f.rng_count, sp);
        if (lane == 0 && wave < rows_real) {                      // every wave draws for its own env: no barrier after it
            const int e = env0 + wave, A = W.n_actions;
            const float u = policy_uniform(f.rng_seed, e, sp.count);
            float lse;
            float *Ow = lds + L.o_off;
            act = policy_sample_row(A, lds + L.b3_off, Ow, wave, u, lse);
            POL_STAMP(7);
            f.action[e] = act;
            f.logp[e] = Ow[wave * 17 + act] - lse;
            f.value[e] = Ow[wave * 17 + A];
        }
    }
    act = rfl(act);
    FRAG_T(ft1);
    FRAG_ACC(0, ft0, ft1);
    FRAG_COUNT();
    FRAG_FLUSH();
    if (env >= n_envs) return;
    act_step_env<KW, KD, GRID, HSI>(opaque_s((int)blockIdx.x) * POLICY_WAVES + opaque_s(wave), opaque_v((int)(threadIdx.x & 63)), opaque_s(wave),
                         act, s_cand, s_centres, s_kd);
}
// ---------------------------------------------------------------- a whole fragment WITH the policy in one persistent launch
// (prl_rollout_fragment with weights) -- the loop of act_step_kernel's two phases: the sixteen waves of a workgroup meet
// at the policy's barriers, workgroups never wait for each other, nothing is launched in between.  Each phase starts
// from laundered pointers and lane / wave numbers, so that neither holds the other's registers (5 spilled VGPRs, none in
// a loop; the round's first version of this kernel spilled 954).  47.1 us per step against 50.4 for T launches of
// act_step_kernel: no dispatch ramp, and only sixteen envs wait for their slowest.
__device__ __forceinline__ const PolicyFragmentArgs CAS *opaque(const PolicyFragmentArgs CAS *p) {
    asm volatile("" : "+s"(p));
    return p;
}

template <int KW, bool KD, bool GRID, bool HSI>
__global__ __launch_bounds__(64 * POLICY_WAVES) void rollout_policy_kernel(PolicyFragmentArgs) {
    extern __shared__ float lds[];
    __shared__ int s_cand[POLICY_WAVES][64];
    __shared__ double s_centres[POLICY_WAVES][PAINT_PER_ACTION * 3 + 1];
    __shared__ double s_kd[KD ? POLICY_WAVES : 1][KD ? KD_HEAP * 5 : 1];
    const PolicyFragmentArgs CAS *g0 = (const PolicyFragmentArgs CAS *)__builtin_amdgcn_kernarg_segment_ptr();
    const int wave0 = rfl((int)(threadIdx.x >> 6));
    {   // this kernel writes the last-shot rows whole, step by step: say so once (prl_step.hpp last_row_untracked)
        const StepArgs CAS &a0 = opaque(g0)->f.s;
        const int env_ = (int)blockIdx.x * POLICY_WAVES + wave0;
#ifndef PRL_POLICY_TRACKED
        if constexpr (KW != 0)
            if (env_ < a0.n_envs) last_row_untracked(a0, env_, (int)(threadIdx.x & 63));
#else
        (void)a0;
        (void)env_;
#endif
    }
    for (int t = 0;; ++t) {
        const PolicyFragmentArgs CAS &g = *opaque(g0);
        const int lane = opaque_v((int)(threadIdx.x & 63)), wave = opaque_s(wave0), tid = 64 * wave + lane;
        const int env0 = opaque_s((int)blockIdx.x) * POLICY_WAVES, env = env0 + wave;
        const int n_envs = g.f.s.n_envs, T = g.f.T;
        const size_t n = (size_t)n_envs;
        int act = 0;
        {
            PrlPolicyWeights W;
            W.in_dim = g.w.in_dim; W.h1 = g.w.h1; W.h2 = g.w.h2; W.n_actions = g.w.n_actions;
            W.w1 = g.w.w1; W.b1 = g.w.b1; W.w2 = g.w.w2; W.b2 = g.w.b2; W.w3 = g.w.w3; W.b3 = g.w.b3;
            const PolicyLds L = policy_lds_layout(W);
            const int rows_real = n_envs - env0 < POLICY_WAVES ? n_envs - env0 : POLICY_WAVES;
            SamplerPre sp;
            policy_forward(W, g.f.obs + ((size_t)t * n + env0) * W.in_dim, rows_real, lds, L, tid, env0, nullptr, g.rng_count, sp);
            if (lane == 0 && wave < rows_real) {
                const int e = env0 + wave, A = W.n_actions;
                const float u = policy_uniform(g.rng_seed, e, sp.count);
                float lse;
                float *Ow = lds + L.o_off;
                act = policy_sample_row(A, lds + L.b3_off, Ow, wave, u, lse);
                if (t < T) {
                    const_cast<int32_t *>(g.f.action)[(size_t)t * n + e] = act;
                    g.logp[(size_t)t * n + e] = Ow[wave * 17 + act] - lse;
                    g.value[(size_t)t * n + e] = Ow[wave * 17 + A];
                } else {
                    g.last_value[e] = Ow[wave * 17 + A];          // the bootstrap value; its draw is discarded
                }
            }
        }
        if (t >= T) break;
        act = rfl(act);
        if (env < n_envs) {
            const PolicyFragmentArgs CAS &h = *opaque(g0);
            const StepArgs CAS &a = h.f.s;
            const int lane = opaque_v((int)(threadIdx.x & 63)), wave = opaque_s(wave0);
            const int env = opaque_s((int)blockIdx.x) * POLICY_WAVES + wave;
            const int part_id = rfl(a.env_part ? a.env_part[env] : 0);       // (wave-uniform: the part's fields are scalar reads)
            PartRef P = *(const PartDev CAS *)(a.parts + part_id);
            CfgRef C = *(const CfgDev CAS *)a.cfg;
            double *state_rec = a.state + (size_t)env * PRL_STATE_DOUBLES;
            EnvState S;
            load_state_motion(state_rec, S);
#ifndef PRL_POLICY_TRACKED                           // (default: whole mask rows read and written every step)
            const auto masks = [&] {
                if constexpr (KW == 0) return hbm_masks(a, env, P.n_words, lane);      // (large parts: rows worked on in place)
                else return global_masks(a, env, P.n_words, lane);
            }();
#else
            // (A/B switch, OFF: changed words only, the last-shot row's non-zero words only -- GlobalMasksT<true>, as the per-step
            // kernels: twelve more vector registers, 19 spilled instead of 5: 22.0 k steps/s against 23.4 k)
            const auto masks = [&] {
                if constexpr (KW == 0) return hbm_masks(a, env, P.n_words, lane);      // (large parts: rows worked on in place)
                else return global_masks<true>(a, env, P.n_words, lane);
            }();
            if constexpr (KW != 0) masks.prefetch();
#endif
            double delta1, delta2, new_angle;
            decode_discrete_action(C, act, delta1, delta2, new_angle);
            const FragmentRows row{&h.f, t, a.n_envs, obs_dim_of(C.obs_mode, C.obs_grad)};
            const WaveLds wl{s_cand[wave], s_centres[wave], nullptr, s_kd[KD ? wave : 0], nullptr, nullptr, nullptr, 0};      // (sixteen waves' tree copies do not fit)
            PROF_BEGIN();
            const int dn = step_env<KW, false, true, HSI, KD, GRID ? 1 : 0>(P, C, part_id, env, lane, S, state_rec, masks, delta1, delta2,
                                                                new_angle, row, wl PROF_PASS);
            store_state_live(state_rec, S, lane, dn != 0);
        }
        __syncthreads();            // the observations of step t are written (workgroup-scope fences included)
    }
}

template <typename Args>
int launch_dyn(void (*kernel)(Args), const Args &args, int n_envs, int waves, size_t lds, hipStream_t s) {
    if (const hipError_t e = prl_grant_dyn_lds(reinterpret_cast<const void *>(kernel), lds)) return (int)e;      // (once per device, not per launch)
    hipLaunchKernelGGL(kernel, dim3((n_envs + waves - 1) / waves), dim3(64 * waves), lds, s, args);
    return (int)hipGetLastError();
}

}  // namespace

// flags: bit 0 = some part carries the stale kd-tree, bit 1 = OBS_MODE 'grid' (the kernels are built per observation family like
// step_kernel: prl_observe.hpp observation_wave OBSM)
#define PRL_ROLLOUT_PICK2(kernel, flags, H)                                                                  \
    (((flags)&2) ? (((flags)&1) ? kernel<PRL_KW, true, true, H> : kernel<PRL_KW, false, true, H>)            \
                 : (((flags)&1) ? kernel<PRL_KW, true, false, H> : kernel<PRL_KW, false, false, H>))
// (bit 2 = COLOR_MODE 'HSI': the thickness painter, round 5)
#ifdef PRL_ROLLOUT_NO_HSI                  // (the forced-path variant libraries: half the builds; their host side serves HSI launch by launch)
#define PRL_ROLLOUT_PICK(kernel, flags) PRL_ROLLOUT_PICK2(kernel, flags, false)
#else
#define PRL_ROLLOUT_PICK(kernel, flags) (((flags)&4) ? PRL_ROLLOUT_PICK2(kernel, flags, true) : PRL_ROLLOUT_PICK2(kernel, flags, false))
#endif

PRL_HIDDEN int KFN(act_step)(const void *act_step_args, size_t policy_lds, int flags, void *stream) {
    const ActStepArgs &f = *static_cast<const ActStepArgs *>(act_step_args);
    return launch_dyn(PRL_ROLLOUT_PICK(act_step_kernel, flags), f, f.s.n_envs, POLICY_WAVES, policy_lds, static_cast<hipStream_t>(stream));
}

PRL_HIDDEN int KFN(rollout_policy)(const void *policy_fragment_args, size_t policy_lds, int flags, void *stream) {
    const PolicyFragmentArgs &g = *static_cast<const PolicyFragmentArgs *>(policy_fragment_args);
    return launch_dyn(PRL_ROLLOUT_PICK(rollout_policy_kernel, flags), g, g.f.s.n_envs, POLICY_WAVES, policy_lds, static_cast<hipStream_t>(stream));
}

PRL_HIDDEN int KFN(rollout_fragment)(const void *fragment_args, int flags, void *stream) {
    const FragmentArgs &f = *static_cast<const FragmentArgs *>(fragment_args);
    const size_t lds = PRL_KW == 0 ? 0 : (size_t)FRAG_WAVES * 2 * f.s.mask_stride * sizeof(uint64_t);      // both masks of sixteen envs
    return launch_dyn(PRL_ROLLOUT_PICK(rollout_fragment_kernel, flags), f, f.s.n_envs, FRAG_WAVES, lds, static_cast<hipStream_t>(stream));
}

#include "prl_diag_export.hpp"
