// k_step.hip -- the per-step kernels of parts whose coverage masks fit PRL_KW 64-bit words per lane (PRL_KW = 1..4:
// up to 4 096 / 8 192 / 12 288 / 16 384 samples): reset, reset observations, observe, and step_kernel -- ONE launch
// per batched PaintGymEnv.step() (rge:349-368).  Compiled once per mask width (-DPRL_KW=...), see prl_launch.hpp.
//
// One wavefront (64 lanes) advances one environment by one step: five dependent sub-shots (tool move -> ray onto the
// collision triangles -> nearest vertex -> closest incident triangle -> hook pose), the ball paint of the five shots
// together, reward / termination, and the observation, all in ONE kernel (prl_step.hpp).
//
//  * the env's coverage state (painted mask, last-shot mask) lives in registers while it is worked on: 64-bit word w
//    of a mask is owned by lane (w & 63), slot (w >> 6); HBM traffic per env-step is one coalesced read and one
//    coalesced write of the two masks plus a 128-byte scalar record;
//  * static part tables are shared by all envs and stay L2-resident; samples and vertices are sorted by uniform-grid
//    cell so a sub-shot touches 3 short contiguous ranges (coalesced loads, one sample per lane, hit mask by ballot);
//  * collision triangles are culled with a 16-byte box per lane-triangle before the float64 Moller-Trumbore test;
//    closest hit by wave min-reduction;
//  * section / grid observations are popcounts over mask words; only words whose bounding box straddles the tool
//    position are classified per sample.
// No MFMA: this is gather / scan / bit work.
#define PRL_UNIT_STEP 1                    // (prl_step.hpp step_env: how this unit holds the part's table pointers)
#include "prl_all.hpp"

#ifdef PRL_FACET_TILE                      // (A/B switch, prl_ray.hpp: the ray's LDS facet tile; off)
#define PRL_TILE_ON true
#else
#define PRL_TILE_ON false
#endif

#ifdef PRL_RECORD_GATHER                   // (A/B switch, prl_search.hpp: candidates' records fetched by all lanes through LDS; off)
#define PRL_GATHER_ON true
#else
#define PRL_GATHER_ON false
#endif

#ifndef PRL_KW
#error "compile with -DPRL_KW=1..4 (paintrl_amd/build.py)"
#endif

namespace {

// ---------------------------------------------------------------- reset kernel (rge:370-387)
template <int KW, bool GENSEC>
__global__ __launch_bounds__(256) void reset_kernel(StepArgs a) {
    const int lane = threadIdx.x & 63;
    const int env = rfl(blockIdx.x * 4 + (threadIdx.x >> 6));
    if (env >= a.n_envs) return;
    if (a.reset_mask && !a.reset_mask[env]) return;
    PartRef P = *(const PartDev CAS *)(a.parts + (a.env_part ? a.env_part[env] : 0));
    CfgRef C = *(const CfgDev CAS *)a.cfg;
    EnvState S = *reinterpret_cast<const EnvState *>(a.state + (size_t)env * PRL_STATE_DOUBLES);
    int start = a.start_idx ? a.start_idx[env] : draw_start(C.seed, env, S.episode, P.n_start);
    start = start < 0 ? 0 : (start >= P.n_start ? P.n_start - 1 : start);
    reset_state(P, S, start);
    uint64_t painted[KW_MAX] = {0, 0, 0, 0}, last[KW_MAX] = {0, 0, 0, 0};
    if (C.color_mode == PRL_COLOR_HSI) reset_thickness<KW>(P, a.thick + (size_t)env * 64 * a.mask_stride, lane, painted);
    store_masks<KW>(a, env, P.n_words, lane, painted, last);
    store_state(a.state + (size_t)env * PRL_STATE_DOUBLES, S, lane);
    if (a.obs) {
        const int od = obs_dim_of(C.obs_mode, C.obs_grad);
        for (int k = lane; k < od; k += 64) a.obs[(size_t)env * od + k] = ldg(P.reset_obs, start * od + k);
    }
}

// The observation a reset to start point s returns, for every s of one part (PartDev::reset_obs): one wave per start
// point, run once when a batch is created.  KW = 0: LDS-resident mask (large parts).
template <int KW, bool GENSEC>
__global__ __launch_bounds__(256) void reset_obs_kernel(const PartDev *part, const CfgDev *cfg, double *out) {
    const int lane = threadIdx.x & 63;
    const int s = rfl(blockIdx.x * 4 + (threadIdx.x >> 6));
    PartRef P = *(const PartDev CAS *)part;
    CfgRef C = *(const CfgDev CAS *)cfg;
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

// ---------------------------------------------------------------- observation of the current state (rge:306-319)
template <int KW, bool GENSEC>
__global__ __launch_bounds__(256) void observe_kernel(StepArgs a) {
    const int lane = threadIdx.x & 63;
    const int env = rfl(blockIdx.x * 4 + (threadIdx.x >> 6));
    if (env >= a.n_envs) return;
    PartRef P = *(const PartDev CAS *)(a.parts + (a.env_part ? a.env_part[env] : 0));
    CfgRef C = *(const CfgDev CAS *)a.cfg;
    const EnvState S = *reinterpret_cast<const EnvState *>(a.state + (size_t)env * PRL_STATE_DOUBLES);
    uint64_t painted[KW_MAX] = {0, 0, 0, 0}, last[KW_MAX] = {0, 0, 0, 0};
    load_masks<KW>(a, env, P.n_words, lane, painted, last);
    observation_wave<KW, GENSEC>(P, C, S.pose, painted, lane, a.obs + (size_t)env * obs_dim_of(C.obs_mode, C.obs_grad), wave_lds<GENSEC>().cnt);
}


// ---------------------------------------------------------------- step kernel (rge:349-368)
// One launch = one batched step: one wavefront per env, WAVES envs per workgroup (prl_step.hpp holds the step).
template <int KW, bool GENSEC, bool HSI = false, bool KD = false, int WAVES = STEP_WAVES_NARROW, bool GRID = false>
__global__ __launch_bounds__(64 * WAVES, 4) void step_kernel(StepArgs) {
    // the one by-value argument, read in place (constant address space) wherever a field is needed
    const StepArgs CAS &a = *(const StepArgs CAS *)__builtin_amdgcn_kernarg_segment_ptr();
    const int lane = threadIdx.x & 63;
#ifdef PRL_GRID_LDS                        // (A/B switch: the part's cell-start tables in the workgroup's LDS)
    __shared__ int s_grid[2][GRID_LDS];
    bool grids_staged = false;
    if (!a.env_part) {                     // a batch of one part: every env of the workgroup reads the same two tables
        PartRef P0 = *(const PartDev CAS *)a.parts;
        const int n_vg = P0.vg_nx * P0.vg_ny + 1, n_sg = P0.sg_nx * P0.sg_ny + 1;
        if (n_vg <= GRID_LDS && n_sg <= GRID_LDS) {
            for (int i = threadIdx.x; i < n_vg; i += 64 * WAVES) s_grid[0][i] = ldg(P0.vg_start, i);
            for (int i = threadIdx.x; i < n_sg; i += 64 * WAVES) s_grid[1][i] = ldg(P0.sg_start, i);
            grids_staged = true;
            __syncthreads();               // (before any wave of the workgroup leaves: the last workgroup may hold waves without an env)
        }
    }
#endif
#ifdef PRL_ENV_PERM                        // (diagnostic build, tools/tail_experiment.py)
    const int slot_ = rfl(blockIdx.x * WAVES + (threadIdx.x >> 6));
    if (slot_ >= a.n_envs) return;
    const int env = rfl(g_env_perm ? g_env_perm[slot_] : slot_);
#else
    const int slot = rfl(blockIdx.x * WAVES + (threadIdx.x >> 6));
    if (slot >= a.n_envs) return;
    const int env = a.slot_env ? rfl(a.slot_env[slot]) : slot;      // (mixed batches: StepArgs::slot_env)
#endif
    if (env >= a.n_envs) return;
    // (round 3 parked the last-shot masks of the four-word kernels in LDS rows -- prl_paint.hpp RowWords, WaveLds::lastrow --
    // against 63 spilled vector registers on the reference's own sheet.  The registers were the observation's: four slots'
    // pivot probes at once, 64 of them (prl_observe.hpp section4_accumulate); taken two slots at a time no KW = 4 kernel
    // spills any more and the rows are not needed.)
    WaveLds wl = wave_lds<GENSEC, KD, 0, WAVES, PRL_TILE_ON, PRL_GATHER_ON>();
#ifdef PRL_GRID_LDS
    if (grids_staged) {
        wl.vg_lds = s_grid[0];
        wl.sg_lds = s_grid[1];
    }
#endif
    const int part_id = a.env_part ? a.env_part[env] : 0;
    PartRef P = *(const PartDev CAS *)(a.parts + part_id);
    CfgRef C = *(const CfgDev CAS *)a.cfg;
    double *state_rec = a.state + (size_t)env * PRL_STATE_DOUBLES;
    EnvState S;
    TRACE_BEGIN();
    load_state_motion(state_rec, S);
#ifdef PRL_STAGGER                         // A/B: the odd wave slots of a SIMD start PRL_STAGGER x 64 cycles late (de-phasing the
    if (__builtin_amdgcn_s_getreg(63492) & 1) __builtin_amdgcn_s_sleep(PRL_STAGGER);      // four envs that share it)
#endif
    PROF_BEGIN();
    const GlobalMasksT<!GENSEC> masks = global_masks<!GENSEC>(a, env, P.n_words, lane);
    masks.prefetch();
    if constexpr (GENSEC) last_row_untracked(a, env, lane);      // (the atan2-sector kernels write the rows whole)
    double delta1, delta2, new_angle;
    decode_action(C, a.actions, env, delta1, delta2, new_angle);
    const int dn = step_env<KW, GENSEC, true, HSI, KD, GRID ? 1 : 0>(P, C, part_id, env, lane, S, state_rec, masks, delta1, delta2, new_angle,
                                                      StepRows{&a}, wl PROF_PASS);
    store_state_live(state_rec, S, lane, dn != 0);
    STAMP(PH_STORE);
    PROF_END();
    PROF_STORE(env);
    TRACE_END(env, dn);
}

typedef void (*StepKernelFn)(StepArgs);
template <int WAVES>
StepKernelFn pick_step(const PrlStepSel &sel) {
    constexpr int KW = PRL_KW;
    if (sel.grid && !sel.gensec) {                 // OBS_MODE 'grid': the grid-only builds (atan2 sectors are a section / discrete matter)
        if (sel.kd && sel.hsi) return step_kernel<KW, false, true, true, WAVES, true>;
        if (sel.kd) return step_kernel<KW, false, false, true, WAVES, true>;
        if (sel.hsi) return step_kernel<KW, false, true, false, WAVES, true>;
        return step_kernel<KW, false, false, false, WAVES, true>;
    }
    if (sel.kd && sel.hsi && sel.gensec) return step_kernel<KW, true, true, true, WAVES>;
    if (sel.kd && sel.hsi) return step_kernel<KW, false, true, true, WAVES>;
    if (sel.kd && sel.gensec) return step_kernel<KW, true, false, true, WAVES>;
    if (sel.kd) return step_kernel<KW, false, false, true, WAVES>;
    if (sel.hsi && sel.gensec) return step_kernel<KW, true, true, false, WAVES>;
    if (sel.hsi) return step_kernel<KW, false, true, false, WAVES>;
    if (sel.gensec) return step_kernel<KW, true, false, false, WAVES>;
    return step_kernel<KW, false, false, false, WAVES>;
}

template <int WAVES>
void launch_step_w(const StepArgs &a, const PrlStepSel &sel, hipStream_t s) {
    const dim3 grid((a.n_envs + WAVES - 1) / WAVES), block(64 * WAVES);
    hipLaunchKernelGGL(pick_step<WAVES>(sel), grid, block, 0, s, a);
}

}  // namespace

PRL_HIDDEN int KFN(step)(const void *step_args, const PrlStepSel *sel, void *stream) {
    const StepArgs &a = *static_cast<const StepArgs *>(step_args);
    if (sel->wide) launch_step_w<STEP_WAVES_WIDE>(a, *sel, static_cast<hipStream_t>(stream));
    else launch_step_w<STEP_WAVES_NARROW>(a, *sel, static_cast<hipStream_t>(stream));
    return (int)hipGetLastError();
}

// what prl_batch_step_occupancy reports: waves per workgroup, workgroups resident per CU, dynamic LDS bytes
PRL_HIDDEN int KFN(step_occupancy)(const void *, const PrlStepSel *sel, int out[3]) {
    const int waves = sel->wide ? STEP_WAVES_WIDE : STEP_WAVES_NARROW;
    int nb = 0;
    const hipError_t e = hipOccupancyMaxActiveBlocksPerMultiprocessor(
        &nb, reinterpret_cast<const void *>(sel->wide ? pick_step<STEP_WAVES_WIDE>(*sel) : pick_step<STEP_WAVES_NARROW>(*sel)), 64 * waves, 0);
    out[0] = waves;
    out[1] = nb;
    out[2] = 0;
    return (int)e;
}

PRL_HIDDEN int KFN(reset)(const void *step_args, int gensec, void *stream) {
    const StepArgs &a = *static_cast<const StepArgs *>(step_args);
    const dim3 grid((a.n_envs + 3) / 4), block(256);
    if (gensec) hipLaunchKernelGGL((reset_kernel<PRL_KW, true>), grid, block, 0, static_cast<hipStream_t>(stream), a);
    else hipLaunchKernelGGL((reset_kernel<PRL_KW, false>), grid, block, 0, static_cast<hipStream_t>(stream), a);
    return (int)hipGetLastError();
}

PRL_HIDDEN int KFN(observe)(const void *step_args, int gensec, void *stream) {
    const StepArgs &a = *static_cast<const StepArgs *>(step_args);
    const dim3 grid((a.n_envs + 3) / 4), block(256);
    if (gensec) hipLaunchKernelGGL((observe_kernel<PRL_KW, true>), grid, block, 0, static_cast<hipStream_t>(stream), a);
    else hipLaunchKernelGGL((observe_kernel<PRL_KW, false>), grid, block, 0, static_cast<hipStream_t>(stream), a);
    return (int)hipGetLastError();
}

// The observation a reset to each start point of one part returns (PartDev::reset_obs), on the default stream.
PRL_HIDDEN int KFN(reset_obs)(const void *part_dev, const void *cfg_dev, double *out, int n_start, int /*n_words*/, int gensec) {
    const dim3 grid((n_start + 3) / 4), block(256);
    const PartDev *p = static_cast<const PartDev *>(part_dev);
    const CfgDev *c = static_cast<const CfgDev *>(cfg_dev);
    if (gensec) hipLaunchKernelGGL((reset_obs_kernel<PRL_KW, true>), grid, block, 0, 0, p, c, out);
    else hipLaunchKernelGGL((reset_obs_kernel<PRL_KW, false>), grid, block, 0, 0, p, c, out);
    return (int)hipGetLastError();
}

#include "prl_diag_export.hpp"
