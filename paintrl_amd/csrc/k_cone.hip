// k_cone.hip -- PAINT_METHOD 'normal': the last of a cone-beam step's four launches (k_cone_beams.hip describes all of
// them).  Compiled once per mask width (-DPRL_KW=1..4), see prl_launch.hpp.
//
// cone_finish_kernel, one wave per env: the five hit lists the beams kernels left in cone_hits are folded shot by shot
// into the coverage masks (bpw:562-566 + 572-577), then reward, termination, observation, auto-reset (prl_step.hpp
// finish_step) -- what the ball painter's step kernel does after its paint phase.
#include "prl_all.hpp"

#ifndef PRL_KW
#error "compile with -DPRL_KW=1..4 (paintrl_amd/build.py)"
#endif

namespace {

template <int KW, bool GENSEC, int WAVES>
__global__ __launch_bounds__(64 * WAVES) void cone_finish_kernel(StepArgs) {
    __shared__ uint64_t s_row[WAVES][64 * KW];                        // the hit bits of the shot being folded
    __shared__ int s_cnt[GENSEC ? WAVES : 1][128];
    const StepArgs CAS &a = *(const StepArgs CAS *)__builtin_amdgcn_kernarg_segment_ptr();
    const int lane = threadIdx.x & 63, wave = rfl((int)(threadIdx.x >> 6));
    const int env = blockIdx.x * WAVES + wave;
    if (env >= a.n_envs) return;
    const int part_id = a.env_part ? a.env_part[env] : 0;
    PartRef P = *(const PartDev CAS *)(a.parts + part_id);
    CfgRef C = *(const PrlConfig CAS *)a.cfg;
    double *state_rec = a.state + (size_t)env * PRL_STATE_DOUBLES;
    EnvState S;
    load_state_motion(state_rec, S);
    load_state_accumulators(state_rec, S);
    const double new_angle = uni_d(a.cone_aux[2 * (size_t)env]);
    const double pair = a.cone_aux[2 * (size_t)env + 1];
    const int counter_before = rfl(__double2loint(pair)), facet_hint = rfl(__double2hiint(pair));
    const GlobalMasks masks{a.painted + (size_t)env * a.mask_stride, a.last + (size_t)env * a.mask_stride, P.n_words, lane};
    uint64_t painted[KW_MAX] = {0, 0, 0, 0}, last[KW_MAX] = {0, 0, 0, 0}, valid[KW_MAX] = {0, 0, 0, 0};
    masks.template load<KW>(painted, last);
    uint64_t *row = s_row[wave];
    const int *hits = a.cone_hits + (size_t)env * PAINT_PER_ACTION * a.cone_nb;
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
            const int sidx = b + lane < P.n_beams ? hits[shot * a.cone_nb + b + lane] : -1;
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
    const WaveLds wl{nullptr, nullptr, s_cnt[GENSEC ? wave : 0], nullptr, nullptr};
    PROF_BEGIN();
    const int dn = finish_step<KW, GENSEC, false, false>(P, C, part_id, env, lane, S, state_rec, masks, painted, last,
                                                         (double)succeeded, pixel_counter, counter_before, new_angle, facet_hint,
                                                         StepRows{&a}, wl PROF_PASS);
    store_state_live(state_rec, S, lane, dn != 0);
}

}  // namespace

PRL_HIDDEN int KFN(cone)(const void *step_args, const PrlStepSel *sel, void *stream) {
    const StepArgs &a = *static_cast<const StepArgs *>(step_args);
    hipStream_t s = static_cast<hipStream_t>(stream);
    const dim3 grid((a.n_envs + 3) / 4), block(256);
    if (sel->gensec) hipLaunchKernelGGL((cone_finish_kernel<PRL_KW, true, 4>), grid, block, 0, s, a);
    else hipLaunchKernelGGL((cone_finish_kernel<PRL_KW, false, 4>), grid, block, 0, s, a);
    return (int)hipGetLastError();
}

#include "prl_diag_export.hpp"
