// k_big.hip -- parts with more than 16 384 samples (door_lf ... door_rr_big, Part_Dict rge:106-117: 18 000 - 71 000
// front samples) do not fit four mask words per lane.  Their kernels keep the env's masks in LDS instead (three copies
// of n_words words per env in the step kernel, one in reset / observe), sized at launch; everything else is the same
// device code (prl_step.hpp with KW = 0).  See prl_launch.hpp for the translation-unit layout.
#include "prl_all.hpp"

#define PRL_KW 0

namespace {

// The observation a reset to start point s returns, for every s of one part (PartDev::reset_obs): one wave per start
// point, run once when a batch is created.  KW = 0: LDS-resident mask (large parts).
template <int KW, bool GENSEC>
__global__ __launch_bounds__(256) void reset_obs_kernel(const PartDev *part, const PrlConfig *cfg, double *out) {
    const int lane = threadIdx.x & 63;
    const int s = rfl(blockIdx.x * 4 + (threadIdx.x >> 6));
    PartRef P = *(const PartDev CAS *)part;
    CfgRef C = *(const PrlConfig CAS *)cfg;
    if (s >= P.n_start) return;
    const double pose[3] = {P.start_pos[3 * s], P.start_pos[3 * s + 1], P.start_pos[3 * s + 2]};
    const int od = obs_dim_of(C.obs_mode, C.obs_grad);
    const bool hsi = C.color_mode == PRL_COLOR_HSI;           // thickness mode: every texel reads "painted" after a reset
    if constexpr (KW == 0) {
        extern __shared__ uint64_t big_lds[];
        uint64_t *m = big_lds + (size_t)rfl((int)(threadIdx.x >> 6)) * P.n_words;
        for (int w = lane; w < P.n_words; w += 64) m[w] = 0;
        __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
        __builtin_amdgcn_wave_barrier();
        __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
        observation_big<GENSEC>(P, C, pose, m, lane, out + (size_t)s * od, wave_lds<GENSEC>().cnt);
    } else {
        uint64_t painted[KW_MAX] = {0, 0, 0, 0};
#pragma unroll
        for (int k = 0; k < KW; ++k) {
            const int w = lane + 64 * k;
            painted[k] = (hsi && w < P.n_words) ? ldg(P.word_valid, w) : 0;
        }
        observation_wave<KW, GENSEC>(P, C, pose, painted, lane, out + (size_t)s * od, wave_lds<GENSEC>().cnt);
    }
}

__device__ __forceinline__ BigMasks big_masks(const StepArgs CAS &a, int env, int n_words, int lane, int copies) {
    extern __shared__ uint64_t big_lds[];
    const int wave = rfl((int)(threadIdx.x >> 6));
    uint64_t *base = big_lds + (size_t)wave * copies * a.mask_stride;
    return BigMasks{a.painted + (size_t)env * a.mask_stride, a.last + (size_t)env * a.mask_stride, base,
                    base + (copies > 1 ? a.mask_stride : 0), base + (copies > 2 ? 2 * a.mask_stride : 0), n_words, lane};
}

template <bool GENSEC, bool KD>
__global__ __launch_bounds__(256, 2) void step_kernel_big(StepArgs) {
    const StepArgs CAS &a = *(const StepArgs CAS *)__builtin_amdgcn_kernarg_segment_ptr();
    const int lane = threadIdx.x & 63;
    const int env = rfl(blockIdx.x * 4 + (threadIdx.x >> 6));
    if (env >= a.n_envs) return;
    const WaveLds wl = wave_lds<GENSEC, KD>();
    const int part_id = a.env_part ? a.env_part[env] : 0;
    PartRef P = *(const PartDev CAS *)(a.parts + part_id);
    CfgRef C = *(const PrlConfig CAS *)a.cfg;
    double *state_rec = a.state + (size_t)env * PRL_STATE_DOUBLES;
    EnvState S;
    load_state_motion(state_rec, S);
    const BigMasks masks = big_masks(a, env, P.n_words, lane, 3);
    double delta1, delta2, new_angle;
    decode_action(C, a.actions, env, delta1, delta2, new_angle);
    PROF_BEGIN();                                    // (stamped builds time step_kernel; this one only has to compile)
    const int dn = step_env<0, GENSEC, true, false, KD>(P, C, part_id, env, lane, S, state_rec, masks, delta1, delta2,
                                                    new_angle, StepRows{&a}, wl PROF_PASS);
    store_state_live(state_rec, S, lane, dn != 0);
}

template <bool GENSEC>
__global__ __launch_bounds__(256, 2) void reset_kernel_big(StepArgs a) {
    const int lane = threadIdx.x & 63;
    const int env = rfl(blockIdx.x * 4 + (threadIdx.x >> 6));
    if (env >= a.n_envs) return;
    if (a.reset_mask && !a.reset_mask[env]) return;
    PartRef P = *(const PartDev CAS *)(a.parts + (a.env_part ? a.env_part[env] : 0));
    CfgRef C = *(const PrlConfig CAS *)a.cfg;
    EnvState S = *reinterpret_cast<const EnvState *>(a.state + (size_t)env * PRL_STATE_DOUBLES);
    int start = a.start_idx ? a.start_idx[env] : draw_start(C.seed, env, S.episode, P.n_start);
    start = start < 0 ? 0 : (start >= P.n_start ? P.n_start - 1 : start);
    reset_state(P, S, start);
    for (int w = lane; w < P.n_words; w += 64) {
        a.painted[(size_t)env * a.mask_stride + w] = 0;
        a.last[(size_t)env * a.mask_stride + w] = 0;
    }
    store_state(a.state + (size_t)env * PRL_STATE_DOUBLES, S, lane);
    if (a.obs) {
        const int od = obs_dim_of(C.obs_mode, C.obs_grad);
        for (int k = lane; k < od; k += 64) a.obs[(size_t)env * od + k] = ldg(P.reset_obs, start * od + k);
    }
}

template <bool GENSEC>
__global__ __launch_bounds__(256, 2) void observe_kernel_big(StepArgs a) {
    const int lane = threadIdx.x & 63;
    const int env = rfl(blockIdx.x * 4 + (threadIdx.x >> 6));
    if (env >= a.n_envs) return;
    PartRef P = *(const PartDev CAS *)(a.parts + (a.env_part ? a.env_part[env] : 0));
    CfgRef C = *(const PrlConfig CAS *)a.cfg;
    const EnvState S = *reinterpret_cast<const EnvState *>(a.state + (size_t)env * PRL_STATE_DOUBLES);
    extern __shared__ uint64_t big_lds[];
    uint64_t *painted = big_lds + (size_t)rfl((int)(threadIdx.x >> 6)) * a.mask_stride;
    for (int w = lane; w < P.n_words; w += 64) painted[w] = a.painted[(size_t)env * a.mask_stride + w];
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
    __builtin_amdgcn_wave_barrier();
    __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
    observation_big<GENSEC>(P, C, S.pose, painted, lane, a.obs + (size_t)env * obs_dim_of(C.obs_mode, C.obs_grad),
                            wave_lds<GENSEC>().cnt);
}

// Large parts: dynamic LDS = 4 waves x copies x mask_stride words.
int launch_big(void (*kernel)(StepArgs), const StepArgs &a, int copies, hipStream_t s) {
    const size_t lds = (size_t)4 * copies * a.mask_stride * sizeof(uint64_t);
    if (lds > 64 * 1024) {
        const hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void *>(kernel),
                                                 hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
        if (e != hipSuccess) return (int)e;
    }
    hipLaunchKernelGGL(kernel, dim3((a.n_envs + 3) / 4), dim3(256), lds, s, a);
    return (int)hipGetLastError();
}

}  // namespace

PRL_HIDDEN int KFN(step)(const void *step_args, const PrlStepSel *sel, void *stream) {
    const StepArgs &a = *static_cast<const StepArgs *>(step_args);
    const bool gs = sel->gensec != 0;
    return launch_big(sel->kd ? (gs ? step_kernel_big<true, true> : step_kernel_big<false, true>)
                              : (gs ? step_kernel_big<true, false> : step_kernel_big<false, false>), a, 3,
                      static_cast<hipStream_t>(stream));
}

PRL_HIDDEN int KFN(reset)(const void *step_args, int gensec, void *stream) {
    const StepArgs &a = *static_cast<const StepArgs *>(step_args);
    return launch_big(gensec ? reset_kernel_big<true> : reset_kernel_big<false>, a, 0, static_cast<hipStream_t>(stream));
}

PRL_HIDDEN int KFN(observe)(const void *step_args, int gensec, void *stream) {
    const StepArgs &a = *static_cast<const StepArgs *>(step_args);
    return launch_big(gensec ? observe_kernel_big<true> : observe_kernel_big<false>, a, 1, static_cast<hipStream_t>(stream));
}

PRL_HIDDEN int KFN(reset_obs)(const void *part_dev, const void *cfg_dev, double *out, int n_start, int n_words, int gensec) {
    void (*k)(const PartDev *, const PrlConfig *, double *) = gensec ? reset_obs_kernel<0, true> : reset_obs_kernel<0, false>;
    const size_t lds = (size_t)4 * n_words * sizeof(uint64_t);
    if (lds > 64 * 1024) {
        const hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void *>(k), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
        if (e != hipSuccess) return (int)e;
    }
    hipLaunchKernelGGL(k, dim3((n_start + 3) / 4), dim3(256), lds, 0, static_cast<const PartDev *>(part_dev),
                       static_cast<const PrlConfig *>(cfg_dev), out);
    return (int)hipGetLastError();
}

// Not built for large parts yet: the host side refuses these combinations before it gets here.
PRL_HIDDEN int KFN(cone)(const void *, const PrlStepSel *, void *) { return (int)hipErrorNotSupported; }
PRL_HIDDEN int KFN(act_step)(const void *, size_t, int, void *) { return (int)hipErrorNotSupported; }
PRL_HIDDEN int KFN(rollout_policy)(const void *, size_t, int, void *) { return (int)hipErrorNotSupported; }
PRL_HIDDEN int KFN(rollout_fragment)(const void *, int, void *) { return (int)hipErrorNotSupported; }

#include "prl_diag_export.hpp"
