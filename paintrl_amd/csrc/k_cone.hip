// k_cone.hip -- PAINT_METHOD 'normal': the last of a cone-beam step's five launches (k_cone_beams.hip describes all of
// them).  Compiled once per mask width (-DPRL_KW=1..4), see prl_launch.hpp.
//
// cone_finish_kernel, one wave per env: the five hit lists the beams kernels left in cone_hits are folded shot by shot
// into the coverage masks (bpw:562-566 + 572-577), then reward, termination, observation, auto-reset (prl_step.hpp
// finish_step) -- what the ball painter's step kernel does after its paint phase.  HSI: COLOR_MODE 'HSI' (bpw:384-434) under
// the cone: thickness bytes, one deposit per beam.
#include "prl_all.hpp"

#ifndef PRL_KW
#error "compile with -DPRL_KW=1..4 (paintrl_amd/build.py)"
#endif

namespace {

template <int KW, bool GENSEC, bool HSI, int WAVES>
__global__ __launch_bounds__(64 * WAVES) void cone_finish_kernel(StepArgs) {
    extern __shared__ int s_list[];                                   // HSI: [WAVES][cone_nb] the shot's hit list, in beam order
    __shared__ uint64_t s_row[WAVES][HSI ? 2 : 1][64 * KW];           // the hit bits of the shot being folded (HSI: + status bits)
    __shared__ int s_cnt[GENSEC ? WAVES : 1][128];
    const StepArgs CAS &a = *(const StepArgs CAS *)__builtin_amdgcn_kernarg_segment_ptr();
    const int lane = threadIdx.x & 63, wave = rfl((int)(threadIdx.x >> 6));
    const int env = blockIdx.x * WAVES + wave;
    if (env >= a.n_envs) return;
    const int part_id = a.env_part ? a.env_part[env] : 0;
    PartRef P = *(const PartDev CAS *)(a.parts + part_id);
    CfgRef C = *(const CfgDev CAS *)a.cfg;
    double *state_rec = a.state + (size_t)env * PRL_STATE_DOUBLES;
    EnvState S;
    load_state_motion(state_rec, S);
    load_state_accumulators(state_rec, S);
    const double new_angle = uni_d(a.cone_aux[2 * (size_t)env]);
    const double pair = a.cone_aux[2 * (size_t)env + 1];
    const int counter_before = rfl(__double2loint(pair)), facet_hint = rfl(__double2hiint(pair));
    // (the rows' words this step leaves alone are not written back, the zero words of the last-shot row not read: GlobalMasksT)
    const GlobalMasksT<true> masks = global_masks<true>(a, env, P.n_words, lane);
    masks.prefetch();
    uint64_t painted[KW_MAX] = {0, 0, 0, 0}, last[KW_MAX] = {0, 0, 0, 0}, valid[KW_MAX] = {0, 0, 0, 0};
    masks.template load<KW>(painted, last);
    uint64_t *row = s_row[wave][0];
    const int *hits = a.cone_hits + (size_t)env * PAINT_PER_ACTION * a.cone_nb;
    uint32_t n_succeeded_l = 0;
    double succ_l = 0.0;                                              // HSI: this lane's share of the deposited fractions
    // the five shots' hit lists are read together, ahead of the loop that folds them one after the other (up to 192 beams:
    // three a lane; five dependent round trips otherwise)
    constexpr int PRE = 3;
    const bool prefetched = !HSI && (a.cone_nb >> 6) <= PRE;
    int pre[PAINT_PER_ACTION][PRE];
    if (prefetched) {
#pragma unroll
        for (int shot = 0; shot < PAINT_PER_ACTION; ++shot)
#pragma unroll
            for (int q = 0; q < PRE; ++q) pre[shot][q] = 64 * q + lane < P.n_beams ? hits[shot * a.cone_nb + 64 * q + lane] : -1;
    }
#pragma unroll
    for (int shot = 0; shot < PAINT_PER_ACTION; ++shot) {
        // bpw:562-566 + 572-577: the samples this shot's beams hit are its "affected" set; no hit at all: the
        // reference returns early and leaves the last-shot set untouched (rob:283-285)
#pragma unroll
        for (int k = 0; k < KW; ++k) row[lane + 64 * k] = 0;
        if constexpr (HSI) {
            // COLOR_MODE 'HSI' under the cone (bpw:419-434 with the list of nearest samples, duplicates and all;
            // oracle/paint_oracle.c apply_paint_hsi_list is the scalar statement): r = the largest distance of a listed
            // sample to the shot centre, then every ENTRY deposits int(25 (1 - (d / r)^2)) + 1 on its sample unless the
            // byte is 0 at that moment -- a sample under k beams receives k deposits.  The first entry of a sample makes
            // all of them; the float sum of quantity / 255 is reduced in lane order (rewards to 1e-12, bytes exactly).
            uint64_t *stat = s_row[wave][1];
            int *list = s_list + (size_t)wave * a.cone_nb;
            uint8_t *thick = a.thick + (size_t)env * 64 * a.mask_stride;
#pragma unroll
            for (int k = 0; k < KW; ++k) stat[lane + 64 * k] = 0;
            for (int b = lane; b < a.cone_nb; b += 64) list[b] = b < P.n_beams ? hits[shot * a.cone_nb + b] : -1;
            __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
            __builtin_amdgcn_wave_barrier();
            __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
            const double *sh = a.cone_shots + ((size_t)env * PAINT_PER_ACTION + shot) * 8;
            const double pos[3] = {uni_d(sh[0]), uni_d(sh[1]), uni_d(sh[2])};
            const double quat[4] = {uni_d(sh[3]), uni_d(sh[4]), uni_d(sh[5]), uni_d(sh[6])};
            double c[3];
            transform_point(pos, quat, 0.0, 0.0, SHOT_CENTRE_OFFSET, c);     // rob:277-278, 285
            double dmax_l = -1.0;
            int beam_hits = 0;
            for (int b0 = 0; b0 < P.n_beams; b0 += 64) {
                const int sidx = list[b0 + lane];
                if (sidx >= 0) {
                    const double dx = c[0] - ldg(P.samp[0], sidx), dy = c[1] - ldg(P.samp[1], sidx), dz = c[2] - ldg(P.samp[2], sidx);
                    const double dd = (dx * dx + dy * dy) + dz * dz;
                    dmax_l = dd > dmax_l ? dd : dmax_l;
                }
                beam_hits += __popcll(ballot64(sidx >= 0));
            }
            if (beam_hits == 0) continue;
            const double rmax = sqrt(wave_max_d(dmax_l));                   // = max of the sqrt's: sqrt is monotone
            if (P.n_beams > HSI_ROUNDS_FROM && P.n_beams <= 64 * 64) {
                hsi_list_deposits(P, list, P.n_beams, lane, c, rmax, thick, row, stat,
                                  reinterpret_cast<uint64_t *>(s_list + (size_t)WAVES * a.cone_nb) + (size_t)wave * HSI_HASH_WORDS, succ_l);
            } else
            for (int b0 = 0; b0 < P.n_beams; b0 += 64) {                     // (more than 4 096 beams a shot: the first entry counts the rest)
                const int sidx = list[b0 + lane];
                int mult = 0;
                bool first = sidx >= 0;
                for (int j = 0; j < P.n_beams; ++j) {                        // (wave-uniform reads: LDS broadcasts)
                    const bool same = list[j] == sidx;
                    mult += same ? 1 : 0;
                    first = first && !(same && j < b0 + lane);
                }
                if (first) {
                    const double dx = c[0] - ldg(P.samp[0], sidx), dy = c[1] - ldg(P.samp[1], sidx), dz = c[2] - ldg(P.samp[2], sidx);
                    const double dd = (dx * dx + dy * dy) + dz * dz;
                    const double q = sqrt(dd) / rmax;
                    const int quantity = (int)(25 * (1 - q * q)) + 1;
                    uint8_t v = thick[sidx];
                    for (int k = 0; k < mult; ++k)
                        if (v != 0) {
                            v = (uint8_t)(v - quantity);
                            succ_l += quantity / 255.0;
                        }
                    thick[sidx] = v;
                    atomicOr(reinterpret_cast<unsigned long long *>(&row[sidx >> 6]), 1ull << (sidx & 63));
                    if (v == 255) atomicOr(reinterpret_cast<unsigned long long *>(&stat[sidx >> 6]), 1ull << (sidx & 63));
                }
            }
            __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
            __builtin_amdgcn_wave_barrier();
            __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
#pragma unroll
            for (int k = 0; k < KW; ++k) {
                const uint64_t cw = row[lane + 64 * k], st = stat[lane + 64 * k];
                painted[k] = (painted[k] & ~cw) | st;                        // the status bit stays "byte == 255" (bpw:723-725)
                valid[k] |= cw & ~last[k];
                last[k] = cw;
            }
            __builtin_amdgcn_wave_barrier();
        } else {
            __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
            __builtin_amdgcn_wave_barrier();
            int beam_hits = 0;
            if (prefetched) {
#pragma unroll
                for (int q = 0; q < PRE; ++q) {
                    const int sidx = pre[shot][q];
                    if (sidx >= 0) atomicOr(reinterpret_cast<unsigned long long *>(&row[sidx >> 6]), 1ull << (sidx & 63));
                    beam_hits += __popcll(ballot64(sidx >= 0));
                }
            } else {
                for (int b = 0; b < P.n_beams; b += 64) {
                    const int sidx = b + lane < P.n_beams ? hits[shot * a.cone_nb + b + lane] : -1;
                    if (sidx >= 0) atomicOr(reinterpret_cast<unsigned long long *>(&row[sidx >> 6]), 1ull << (sidx & 63));
                    beam_hits += __popcll(ballot64(sidx >= 0));
                }
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
    }
    uint32_t pix_l = 0;
#pragma unroll
    for (int k = 0; k < KW; ++k) pix_l += __popcll(valid[k]);
    const uint64_t sums = wave_sum_u64(((uint64_t)n_succeeded_l << 32) | pix_l);
    const int succeeded = (int)(sums >> 32), pixel_counter = (int)(sums & 0xffffffffu);
    double succeeded_f = (double)succeeded;
    if constexpr (HSI) succeeded_f = wave_sum_d(succ_l);
    const WaveLds wl{nullptr, nullptr, s_cnt[GENSEC ? wave : 0], nullptr, nullptr, nullptr, nullptr, 0};
    PROF_BEGIN();
    const int dn = finish_step<KW, GENSEC, false, HSI>(P, C, part_id, env, lane, S, state_rec, masks, painted, last, succeeded_f,
                                                       pixel_counter, counter_before, new_angle, facet_hint, StepRows{&a}, wl, nullptr PROF_PASS);
    store_state_live(state_rec, S, lane, dn != 0);
}

}  // namespace

PRL_HIDDEN int KFN(cone)(const void *step_args, const PrlStepSel *sel, void *stream) {
    const StepArgs &a = *static_cast<const StepArgs *>(step_args);
    hipStream_t s = static_cast<hipStream_t>(stream);
    const dim3 grid((a.n_envs + 3) / 4), block(256);
    if (sel->hsi) {
        void (*k)(StepArgs) = sel->gensec ? cone_finish_kernel<PRL_KW, true, true, 4> : cone_finish_kernel<PRL_KW, false, true, 4>;
        const size_t lds = sizeof(int) * 4 * (size_t)a.cone_nb + 4 * sizeof(uint64_t) * HSI_HASH_WORDS;      // the four waves' hit lists, then their hashed sets
        if (const hipError_t e = prl_grant_dyn_lds(reinterpret_cast<const void *>(k), lds)) return (int)e;      // (once per device)
        hipLaunchKernelGGL(k, grid, block, lds, s, a);
    } else if (sel->gensec) {
        hipLaunchKernelGGL((cone_finish_kernel<PRL_KW, true, false, 4>), grid, block, 0, s, a);
    } else {
        hipLaunchKernelGGL((cone_finish_kernel<PRL_KW, false, false, 4>), grid, block, 0, s, a);
    }
    return (int)hipGetLastError();
}

#include "prl_diag_export.hpp"
