// k_big.hip -- parts with more than 16 384 samples (door_lf ... door_rr_big, Part_Dict rge:106-117: 18 000 - 71 000
// front samples) do not fit four mask words per lane.  The ball painter's step (step_kernel_big) leaves the env's mask rows
// where they are, in HBM / L2 (prl_step.hpp HbmMasks: the painter touches the words of its cell block in place, the
// observation streams the painted row) -- no LDS for masks, sixteen waves a CU like the small parts' kernel.  COLOR_MODE
// the cone beams' finish kernel still works on LDS copies of the rows (BigMasks), sized at launch.  Everything else
// is the same device code (prl_step.hpp with KW = 0).  See prl_launch.hpp for the translation-unit layout.
#define PRL_UNIT_STEP 1                    // (prl_step.hpp step_env: the part's table pointers re-read per sub-shot)
#include "prl_all.hpp"

#define PRL_KW 0

namespace {

// The observation a reset to start point s returns, for every s of one part (PartDev::reset_obs): one wave per start
// point, run once when a batch is created.  KW = 0: LDS-resident mask (large parts).
template <int KW, bool GENSEC>
__global__ __launch_bounds__(256) void reset_obs_kernel(const PartDev *part, const CfgDev *cfg, double *out) {
    const int lane = threadIdx.x & 63;
    const int s = rfl(blockIdx.x * (blockDim.x >> 6) + (threadIdx.x >> 6));
    PartRef P = *(const PartDev CAS *)part;
    CfgRef C = *(const CfgDev CAS *)cfg;
    if (s >= P.n_start) return;
    const double pose[3] = {P.start_pos[3 * s], P.start_pos[3 * s + 1], P.start_pos[3 * s + 2]};
    const int od = obs_dim_of(C.obs_mode, C.obs_grad);
    const bool hsi = C.color_mode == PRL_COLOR_HSI;           // thickness mode: every texel reads "painted" after a reset
    if constexpr (KW == 0) {
        extern __shared__ uint64_t big_lds[];
        uint64_t *m = big_lds + (size_t)rfl((int)(threadIdx.x >> 6)) * P.n_words;
        for (int w = lane; w < P.n_words; w += 64) m[w] = hsi ? ldg(P.word_valid, w) : 0;
        __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
        __builtin_amdgcn_wave_barrier();
        __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
        {
            const WaveLds wl_ = wave_lds<GENSEC>();
            observation_big<GENSEC>(P, C, pose, m, lane, out + (size_t)s * od, wl_.cnt, wl_.cand);
        }
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
                    base + (copies > 1 ? a.mask_stride : 0), base + (copies > 2 ? 2 * a.mask_stride : 0), n_words, lane,
                    base + (copies > 3 ? 3 * a.mask_stride : 0)};
}

// One launch = one batched step of a large part, COLOR_MODE 'RGB': one wavefront per env, WAVES envs per workgroup, four
// waves a SIMD -- the small parts' step_kernel with the masks left in HBM.
#ifndef PRL_BIG_OCC
#define PRL_BIG_OCC 4                  // waves a SIMD the large parts' step is compiled for (A/B: 3 = 168 registers)
#endif
template <bool GENSEC, bool KD, int WAVES, bool GRID = false>
__global__ __launch_bounds__(64 * WAVES, PRL_BIG_OCC) void step_kernel_big(StepArgs) {
    const StepArgs CAS &a = *(const StepArgs CAS *)__builtin_amdgcn_kernarg_segment_ptr();
    const int lane = threadIdx.x & 63;
    const int env = rfl(blockIdx.x * WAVES + (threadIdx.x >> 6));
    if (env >= a.n_envs) return;
    const WaveLds wl = wave_lds<GENSEC, KD, 0, WAVES>();
    const int part_id = a.env_part ? a.env_part[env] : 0;
    PartRef P = *(const PartDev CAS *)(a.parts + part_id);
    CfgRef C = *(const CfgDev CAS *)a.cfg;
    double *state_rec = a.state + (size_t)env * PRL_STATE_DOUBLES;
    EnvState S;
    TRACE_BEGIN();
    load_state_motion(state_rec, S);
    const HbmMasks masks = hbm_masks(a, env, P.n_words, lane);
    double delta1, delta2, new_angle;
    decode_action(C, a.actions, env, delta1, delta2, new_angle);
    PROF_BEGIN();                                    // (trace builds: tools/wave_trace.py with --diag-unit k_big)
    const int dn = step_env<0, GENSEC, true, false, KD, GRID ? 1 : 0>(P, C, part_id, env, lane, S, state_rec, masks, delta1, delta2,
                                                                    new_angle, StepRows{&a}, wl PROF_PASS);
    store_state_live(state_rec, S, lane, dn != 0);
    STAMP(PH_STORE);
    PROF_END();
    PROF_STORE(env);
    TRACE_END(env, dn);
}

// COLOR_MODE 'HSI' on a large part: the same with the thickness painter (paint_shots_hsi_words: the rows in HBM, every word of the
// shots' cell block visited once) -- until round 5 four LDS copies of the rows per env, one wave a SIMD, 755 us a step at 70 654 samples
template <bool GENSEC, bool KD>
__global__ __launch_bounds__(256, 4) void step_kernel_big_hsi(StepArgs) {
    const StepArgs CAS &a = *(const StepArgs CAS *)__builtin_amdgcn_kernarg_segment_ptr();
    const int lane = threadIdx.x & 63;
    const int env = rfl(blockIdx.x * (blockDim.x >> 6) + (threadIdx.x >> 6));
    if (env >= a.n_envs) return;
    const WaveLds wl = wave_lds<GENSEC, KD, 0, 4>();
    const int part_id = a.env_part ? a.env_part[env] : 0;
    PartRef P = *(const PartDev CAS *)(a.parts + part_id);
    CfgRef C = *(const CfgDev CAS *)a.cfg;
    double *state_rec = a.state + (size_t)env * PRL_STATE_DOUBLES;
    EnvState S;
    load_state_motion(state_rec, S);
    const HbmMasks masks = hbm_masks(a, env, P.n_words, lane);
    double delta1, delta2, new_angle;
    decode_action(C, a.actions, env, delta1, delta2, new_angle);
    PROF_BEGIN();
    const int dn = step_env<0, GENSEC, true, true, KD>(P, C, part_id, env, lane, S, state_rec, masks, delta1, delta2,
                                                     new_angle, StepRows{&a}, wl PROF_PASS);
    store_state_live(state_rec, S, lane, dn != 0);
}

template <bool GENSEC>
__global__ __launch_bounds__(256, 2) void reset_kernel_big(StepArgs a) {
    const int lane = threadIdx.x & 63;
    const int env = rfl(blockIdx.x * (blockDim.x >> 6) + (threadIdx.x >> 6));
    if (env >= a.n_envs) return;
    if (a.reset_mask && !a.reset_mask[env]) return;
    PartRef P = *(const PartDev CAS *)(a.parts + (a.env_part ? a.env_part[env] : 0));
    CfgRef C = *(const CfgDev CAS *)a.cfg;
    EnvState S = *reinterpret_cast<const EnvState *>(a.state + (size_t)env * PRL_STATE_DOUBLES);
    int start = a.start_idx ? a.start_idx[env] : draw_start(C.seed, env, S.episode, P.n_start);
    start = start < 0 ? 0 : (start >= P.n_start ? P.n_start - 1 : start);
    reset_state(P, S, start);
    const bool hsi = C.color_mode == PRL_COLOR_HSI;                    // every byte 255, every real sample reads "painted"
    for (int w = lane; w < P.n_words; w += 64) {
        a.painted[(size_t)env * a.mask_stride + w] = hsi ? ldg(P.word_valid, w) : 0;
        a.last[(size_t)env * a.mask_stride + w] = 0;
    }
    if (lane < a.nz_stride) a.last_nz[(size_t)env * a.nz_stride + lane] = 0;       // (the whole last-shot row is zero: HbmMasks)
    if (hsi) {
        uint64_t *t8 = reinterpret_cast<uint64_t *>(a.thick + (size_t)env * 64 * a.mask_stride);
        for (int i = lane; i < 8 * P.n_words; i += 64) t8[i] = ~0ull;
    }
    store_state(a.state + (size_t)env * PRL_STATE_DOUBLES, S, lane);
    if (a.obs) {
        const int od = obs_dim_of(C.obs_mode, C.obs_grad);
        for (int k = lane; k < od; k += 64) a.obs[(size_t)env * od + k] = ldg(P.reset_obs, start * od + k);
    }
}

template <bool GENSEC>
__global__ __launch_bounds__(256) void observe_kernel_big(StepArgs a) {
    const int lane = threadIdx.x & 63;
    const int env = rfl(blockIdx.x * (blockDim.x >> 6) + (threadIdx.x >> 6));
    if (env >= a.n_envs) return;
    PartRef P = *(const PartDev CAS *)(a.parts + (a.env_part ? a.env_part[env] : 0));
    CfgRef C = *(const CfgDev CAS *)a.cfg;
    const EnvState S = *reinterpret_cast<const EnvState *>(a.state + (size_t)env * PRL_STATE_DOUBLES);
    const uint64_t GAS *painted = (const uint64_t GAS *)(a.painted + (size_t)env * a.mask_stride);      // the row where it is
    const WaveLds wl = wave_lds<GENSEC>();
    observation_big<GENSEC>(P, C, S.pose, painted, lane, a.obs + (size_t)env * obs_dim_of(C.obs_mode, C.obs_grad), wl.cnt, wl.cand);
}

// PAINT_METHOD 'normal' on a large part, COLOR_MODE 'RGB': the finish kernel of k_cone.hip with the mask rows left in HBM
// (HbmMasks) and TWO rows in LDS -- the hit bits of the shot being folded and of the shot before it (first: the last-shot row).
// A shot's fold reads its hit row once: a word with hits fetches its painted word (newly painted = hits & ~painted, written back
// if any) and ORs its valid set (hits & ~previous shot's) into the env's last-shot row in HBM, which serves as the union's
// accumulator until the end.  The last shot's row goes back to HBM with its
// set of non-zero words (StepArgs::last_nz), so finish_step's auto-reset clears exactly those.  Until round 5 four LDS copies of
// the rows (painted, last, union, shot): one workgroup of four waves a CU at 70 654 samples, 254 us of the step's 1.45 ms.
template <bool GENSEC>
__global__ __launch_bounds__(256, 2) void cone_finish_kernel_rows(StepArgs, int) {
    extern __shared__ uint64_t big_lds[];
    const StepArgs CAS &a = *(const StepArgs CAS *)__builtin_amdgcn_kernarg_segment_ptr();
    const int lane = threadIdx.x & 63, wave = rfl((int)(threadIdx.x >> 6));
    const int env = rfl(blockIdx.x * (blockDim.x >> 6) + (threadIdx.x >> 6));
    if (env >= a.n_envs) return;
    const WaveLds wl = wave_lds<GENSEC, false>();
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
    const HbmMasks masks = hbm_masks(a, env, P.n_words, lane);
    masks.template load<0>(nullptr, nullptr);                        // (this lane's word of the non-zero set)
    uint64_t *cur = big_lds + (size_t)wave * 2 * a.mask_stride, *prev = cur + a.mask_stride;
    const int nw = P.n_words;
    auto sync = [] {
        __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
        __builtin_amdgcn_wave_barrier();
        __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
    };
    // the last-shot row to LDS (its non-zero words are read, the rest is zero); its place in HBM, zeroed, collects the union of the
    // shots' valid sets (rob:425: the penalty counts a sample once however many shots it was valid in) until the row is rewritten
    for (int w0 = 0; w0 < nw; w0 += 64) {
        const int w = w0 + lane;
        const uint64_t set = bcast_u64(masks.old_nz, w0 >> 6);
        const bool have = w < nw && ((set >> lane) & 1);
        if (w < nw) prev[w] = have ? masks.last[w] : 0;
        if (have) masks.last[w] = 0;
    }
    const int *hits = a.cone_hits + (size_t)env * PAINT_PER_ACTION * a.cone_nb;
    uint32_t n_succeeded_l = 0, pix_l = 0;
    for (int shot = 0; shot < PAINT_PER_ACTION; ++shot) {
        for (int w = lane; w < nw; w += 64) cur[w] = 0;
        sync();
        int beam_hits = 0;
        for (int b = 0; b < P.n_beams; b += 64) {
            const int sidx = b + lane < P.n_beams ? hits[shot * a.cone_nb + b + lane] : -1;
            if (sidx >= 0) atomicOr(reinterpret_cast<unsigned long long *>(&cur[sidx >> 6]), 1ull << (sidx & 63));
            beam_hits += __popcll(ballot64(sidx >= 0));
        }
        sync();
        if (beam_hits > 0) {                                          // (no hit: the reference returns early, the last-shot set stays)
            for (int w = lane; w < nw; w += 64) {
                const uint64_t cw = cur[w];
                if (cw) {
                    const uint64_t pw = masks.painted[w];
                    n_succeeded_l += __popcll(cw & ~pw);
                    if (cw & ~pw) masks.painted[w] = pw | cw;
                    const uint64_t vw = cw & ~prev[w];
                    if (vw) masks.last[w] |= vw;                       // (this lane's word in every shot)
                }
            }
            uint64_t *t = cur;
            cur = prev;
            prev = t;
        }
        sync();
    }
    // the last shot's row back to HBM: the words that are not zero now or were not before; the new set of non-zero words
    uint64_t nz_new = 0;
    for (int w0 = 0; w0 < nw; w0 += 64) {
        const int w = w0 + lane;
        const uint64_t lw = w < nw ? prev[w] : 0, uw = w < nw ? masks.last[w] : 0;
        pix_l += __popcll(uw);
        if (uw != lw) masks.last[w] = lw;
        const uint64_t is = ballot64(lw != 0);
        nz_new = lane == (w0 >> 6) ? is : nz_new;
    }
    if (lane < masks.nz_words() && nz_new != masks.old_nz) masks.nz[lane] = nz_new;
    masks.old_nz = nz_new;
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "workgroup");             // (the observation reads the painted row through other lanes)
    __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "workgroup");
    const uint64_t sums = wave_sum_u64(((uint64_t)n_succeeded_l << 32) | pix_l);
    const int pixel_counter = (int)(sums & 0xffffffffu);
    const double succeeded_f = (double)(int)(sums >> 32);
    uint64_t none[KW_MAX] = {0, 0, 0, 0}, none2[KW_MAX] = {0, 0, 0, 0};
    PROF_BEGIN();
    const int dn = finish_step<0, GENSEC, false, false>(P, C, part_id, env, lane, S, state_rec, masks, none, none2, succeeded_f,
                                                        pixel_counter, counter_before, new_angle, facet_hint, StepRows{&a}, wl, nullptr PROF_PASS);
    store_state_live(state_rec, S, lane, dn != 0);
}

// PAINT_METHOD 'normal' on a large part: the finish kernel of k_cone.hip with the masks in LDS (painted, last, the union of
// the shots' valid sets, the shot being folded, and with COLOR_MODE 'HSI' its status bits: five copies of n_words words).
template <bool GENSEC, bool HSI>
__global__ __launch_bounds__(256, 2) void cone_finish_kernel_big(StepArgs, int list_off) {
    extern __shared__ uint64_t big_lds[];
    const StepArgs CAS &a = *(const StepArgs CAS *)__builtin_amdgcn_kernarg_segment_ptr();
    const int lane = threadIdx.x & 63, wave = rfl((int)(threadIdx.x >> 6));
    const int env = rfl(blockIdx.x * (blockDim.x >> 6) + (threadIdx.x >> 6));
    if (env >= a.n_envs) return;
    const WaveLds wl = wave_lds<GENSEC, false>();
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
    constexpr int COPIES = HSI ? 5 : 4;
    uint64_t *base = big_lds + (size_t)wave * COPIES * a.mask_stride;
    uint64_t *painted = base, *last = base + a.mask_stride, *valid = base + 2 * a.mask_stride, *row = base + 3 * a.mask_stride;
    uint64_t *stat = base + (HSI ? 4 : 3) * a.mask_stride;
    uint64_t *g_painted = a.painted + (size_t)env * a.mask_stride, *g_last = a.last + (size_t)env * a.mask_stride;
    const int nw = P.n_words;
    for (int w = lane; w < nw; w += 64) {
        painted[w] = g_painted[w];
        last[w] = g_last[w];
        valid[w] = 0;
    }
    const int *hits = a.cone_hits + (size_t)env * PAINT_PER_ACTION * a.cone_nb;
    int *list = reinterpret_cast<int *>(big_lds) + list_off + (size_t)wave * a.cone_nb;       // HSI: the shot's hit list
    uint32_t n_succeeded_l = 0;
    double succ_l = 0.0;
    for (int shot = 0; shot < PAINT_PER_ACTION; ++shot) {
        for (int w = lane; w < nw; w += 64) {
            row[w] = 0;
            if constexpr (HSI) stat[w] = 0;
        }
        int beam_hits = 0;
        if constexpr (HSI) {                                            // (k_cone.hip cone_finish_kernel holds the commentary)
            uint8_t *thick = a.thick + (size_t)env * 64 * a.mask_stride;
            for (int b = lane; b < a.cone_nb; b += 64) list[b] = b < P.n_beams ? hits[shot * a.cone_nb + b] : -1;
            __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
            __builtin_amdgcn_wave_barrier();
            __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
            const double *sh = a.cone_shots + ((size_t)env * PAINT_PER_ACTION + shot) * 8;
            const double pos[3] = {uni_d(sh[0]), uni_d(sh[1]), uni_d(sh[2])};
            const double quat[4] = {uni_d(sh[3]), uni_d(sh[4]), uni_d(sh[5]), uni_d(sh[6])};
            double c[3];
            transform_point(pos, quat, 0.0, 0.0, SHOT_CENTRE_OFFSET, c);
            double dmax_l = -1.0;
            for (int b0 = 0; b0 < P.n_beams; b0 += 64) {
                const int sidx = list[b0 + lane];
                if (sidx >= 0) {
                    const double dx = c[0] - ldg(P.samp[0], sidx), dy = c[1] - ldg(P.samp[1], sidx), dz = c[2] - ldg(P.samp[2], sidx);
                    const double dd = (dx * dx + dy * dy) + dz * dz;
                    dmax_l = dd > dmax_l ? dd : dmax_l;
                }
                beam_hits += __popcll(ballot64(sidx >= 0));
            }
            if (beam_hits > 0) {
                const double rmax = sqrt(wave_max_d(dmax_l));
                if (P.n_beams > HSI_ROUNDS_FROM && P.n_beams <= 64 * 64) {                                   // (prl_paint.hpp: rounds over a hashed set, linear in the beams)
                    uint64_t *hash = reinterpret_cast<uint64_t *>(reinterpret_cast<int *>(big_lds) + list_off + (size_t)(blockDim.x >> 6) * a.cone_nb) +
                                     (size_t)wave * HSI_HASH_WORDS;
                    hsi_list_deposits(P, list, P.n_beams, lane, c, rmax, thick, row, stat, hash, succ_l);
                } else
                for (int b0 = 0; b0 < P.n_beams; b0 += 64) {
                    const int sidx = list[b0 + lane];
                    int mult = 0;
                    bool first = sidx >= 0;
                    for (int j = 0; j < P.n_beams; ++j) {
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
            }
        } else {
            __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
            __builtin_amdgcn_wave_barrier();
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
            for (int w = lane; w < nw; w += 64) {
                const uint64_t cw = row[w];
                if constexpr (HSI) {
                    painted[w] = (painted[w] & ~cw) | stat[w];
                } else {
                    n_succeeded_l += __popcll(cw & ~painted[w]);
                    painted[w] |= cw;
                }
                valid[w] |= cw & ~last[w];
                last[w] = cw;
            }
        }
        __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
        __builtin_amdgcn_wave_barrier();
        __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
    }
    uint32_t pix_l = 0;
    for (int w = lane; w < nw; w += 64) pix_l += __popcll(valid[w]);
    const uint64_t sums = wave_sum_u64(((uint64_t)n_succeeded_l << 32) | pix_l);
    const int pixel_counter = (int)(sums & 0xffffffffu);
    double succeeded_f = (double)(int)(sums >> 32);
    if constexpr (HSI) succeeded_f = wave_sum_d(succ_l);
    // finish_step's view of the masks: `last` doubles as the set this step's last shot affected (what store() writes back)
    const BigMasks masks{g_painted, g_last, painted, last, last, nw, lane, nullptr};
    uint64_t none[KW_MAX] = {0, 0, 0, 0}, none2[KW_MAX] = {0, 0, 0, 0};
    PROF_BEGIN();
    const int dn = finish_step<0, GENSEC, false, HSI>(P, C, part_id, env, lane, S, state_rec, masks, none, none2, succeeded_f,
                                                      pixel_counter, counter_before, new_angle, facet_hint, StepRows{&a}, wl, nullptr PROF_PASS);
    store_state_live(state_rec, S, lane, dn != 0);
}

// Large parts: dynamic LDS = waves x copies x mask_stride words (+ extra bytes per wave); as many waves per workgroup (at
// most four) as fit the CU's 160 KB NEXT TO the kernel's own static LDS (wave_lds: candidate list, shot centres, section
// counters, and with the stale tree 20 KB of kd-walk rows) -- asked of the runtime once per (device, kernel) and remembered,
// like the dynamic size already granted there (prl_dynlds.hpp).
size_t big_static_lds(const void *kernel) { return prl_static_lds(kernel); }
hipError_t big_grant_lds(const void *kernel, size_t lds) { return prl_grant_dyn_lds(kernel, lds); }

constexpr size_t LDS_PER_WORKGROUP = 160 * 1024;

template <typename K>
int big_waves(K kernel, const StepArgs &a, int copies, size_t extra_per_wave) {
    const size_t per_wave = (size_t)copies * a.mask_stride * sizeof(uint64_t) + extra_per_wave;
    const size_t fixed = big_static_lds(reinterpret_cast<const void *>(kernel));
    if (fixed >= LDS_PER_WORKGROUP) return 0;
    int waves = per_wave ? (int)((LDS_PER_WORKGROUP - fixed) / per_wave) : 4;
    return waves > 4 ? 4 : waves;
}

template <typename... Extra>
int launch_big(void (*kernel)(StepArgs, Extra...), const StepArgs &a, int copies, size_t extra_per_wave, hipStream_t s, Extra... extra) {
    const int waves = big_waves(kernel, a, copies, extra_per_wave);
    if (waves < 1) return (int)hipErrorInvalidValue;              // (the caller names the part's size in its message)
    const size_t lds = (size_t)waves * ((size_t)copies * a.mask_stride * sizeof(uint64_t) + extra_per_wave);
    if (const hipError_t e = big_grant_lds(reinterpret_cast<const void *>(kernel), lds)) return (int)e;
    hipLaunchKernelGGL(kernel, dim3((a.n_envs + waves - 1) / waves), dim3(64 * waves), lds, s, a, extra...);
    return (int)hipGetLastError();
}

typedef void (*BigStepFn)(StepArgs);
template <int WAVES>
BigStepFn pick_big_step(const PrlStepSel &sel) {           // COLOR_MODE 'RGB': rows in HBM
    const bool gs = sel.gensec != 0;
    if (sel.grid && !gs) return sel.kd ? step_kernel_big<false, true, WAVES, true> : step_kernel_big<false, false, WAVES, true>;      // (per observation family)
    return sel.kd ? (gs ? step_kernel_big<true, true, WAVES> : step_kernel_big<false, true, WAVES>)
                  : (gs ? step_kernel_big<true, false, WAVES> : step_kernel_big<false, false, WAVES>);
}
BigStepFn pick_big_step_hsi(const PrlStepSel &sel) {
    const bool gs = sel.gensec != 0;
    return sel.kd ? (gs ? step_kernel_big_hsi<true, true> : step_kernel_big_hsi<false, true>)
                  : (gs ? step_kernel_big_hsi<true, false> : step_kernel_big_hsi<false, false>);
}

}  // namespace

PRL_HIDDEN int KFN(step)(const void *step_args, const PrlStepSel *sel, void *stream) {
    const StepArgs &a = *static_cast<const StepArgs *>(step_args);
    hipStream_t s = static_cast<hipStream_t>(stream);
    if (sel->hsi) {
        hipLaunchKernelGGL(pick_big_step_hsi(*sel), dim3((a.n_envs + 3) / 4), dim3(256), 0, s, a);
        return (int)hipGetLastError();
    }
    if (sel->wide) hipLaunchKernelGGL(pick_big_step<STEP_WAVES_WIDE>(*sel), dim3((a.n_envs + STEP_WAVES_WIDE - 1) / STEP_WAVES_WIDE), dim3(64 * STEP_WAVES_WIDE), 0, s, a);
    else hipLaunchKernelGGL(pick_big_step<STEP_WAVES_NARROW>(*sel), dim3((a.n_envs + STEP_WAVES_NARROW - 1) / STEP_WAVES_NARROW), dim3(64 * STEP_WAVES_NARROW), 0, s, a);
    return (int)hipGetLastError();
}

// what prl_batch_step_occupancy reports: waves per workgroup, workgroups resident per CU, dynamic LDS bytes
PRL_HIDDEN int KFN(step_occupancy)(const void *step_args, const PrlStepSel *sel, int out[3]) {
    const StepArgs &a = *static_cast<const StepArgs *>(step_args);
    int nb = 0;
    if (sel->hsi) {
        const hipError_t e = hipOccupancyMaxActiveBlocksPerMultiprocessor(&nb, reinterpret_cast<const void *>(pick_big_step_hsi(*sel)), 256, 0);
        out[0] = 4;
        out[1] = nb;
        out[2] = 0;
        return (int)e;
    }
    const int waves = sel->wide ? STEP_WAVES_WIDE : STEP_WAVES_NARROW;
    const hipError_t e = hipOccupancyMaxActiveBlocksPerMultiprocessor(
        &nb, reinterpret_cast<const void *>(sel->wide ? pick_big_step<STEP_WAVES_WIDE>(*sel) : pick_big_step<STEP_WAVES_NARROW>(*sel)), 64 * waves, 0);
    out[0] = waves;
    out[1] = nb;
    out[2] = 0;
    return (int)e;
}

PRL_HIDDEN int KFN(reset)(const void *step_args, int gensec, void *stream) {
    const StepArgs &a = *static_cast<const StepArgs *>(step_args);
    return launch_big(gensec ? reset_kernel_big<true> : reset_kernel_big<false>, a, 0, 0, static_cast<hipStream_t>(stream));
}

PRL_HIDDEN int KFN(observe)(const void *step_args, int gensec, void *stream) {
    const StepArgs &a = *static_cast<const StepArgs *>(step_args);
    return launch_big(gensec ? observe_kernel_big<true> : observe_kernel_big<false>, a, 0, 0, static_cast<hipStream_t>(stream));
}

// the last launch of a cone-beam step (k_cone_beams.hip) for a large part
PRL_HIDDEN int KFN(cone)(const void *step_args, const PrlStepSel *sel, void *stream) {
    const StepArgs &a = *static_cast<const StepArgs *>(step_args);
    hipStream_t s = static_cast<hipStream_t>(stream);
    if (sel->hsi) {
        const size_t list_bytes = sizeof(int) * (size_t)a.cone_nb + sizeof(uint64_t) * HSI_HASH_WORDS;      // per wave, behind all the mask copies: the hit lists, then the hashed sets
        void (*k)(StepArgs, int) = sel->gensec ? cone_finish_kernel_big<true, true> : cone_finish_kernel_big<false, true>;
        const int waves = big_waves(k, a, 5, list_bytes);
        const int list_off = (int)((size_t)waves * 5 * a.mask_stride * 2);          // in ints
        return launch_big(k, a, 5, list_bytes, s, list_off);
    }
    return launch_big(sel->gensec ? cone_finish_kernel_rows<true> : cone_finish_kernel_rows<false>, a, 2, 0, s, 0);
}

PRL_HIDDEN int KFN(reset_obs)(const void *part_dev, const void *cfg_dev, double *out, int n_start, int n_words, int gensec) {
    void (*k)(const PartDev *, const CfgDev *, double *) = gensec ? reset_obs_kernel<0, true> : reset_obs_kernel<0, false>;
    const size_t lds = (size_t)4 * n_words * sizeof(uint64_t);
    if (const hipError_t e = big_grant_lds(reinterpret_cast<const void *>(k), lds)) return (int)e;
    hipLaunchKernelGGL(k, dim3((n_start + 3) / 4), dim3(256), lds, 0, static_cast<const PartDev *>(part_dev),
                       static_cast<const CfgDev *>(cfg_dev), out);
    return (int)hipGetLastError();
}

// (the fused rollout kernels of large parts: k_rollout.hip compiled with -DPRL_KW=0)

#include "prl_diag_export.hpp"
