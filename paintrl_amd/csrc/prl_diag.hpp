// prl_diag.hpp -- every diagnostic hook of the device code, in one place.
//
// The product build defines none of the PRL_* diagnostic macros: STAMP / WCNT / PROF_* expand to nothing and the
// kernels carry no instrumentation.  Diagnostic builds (tools/, tests/test_gpu_forced_paths.py) may define:
//   PRL_PHASE_TIMING=<k>       s_memtime deltas of phase k summed over all waves (tools/phase_timing.py)
//   PRL_WAVE_TRACE             per env and launch: start / end time and path counters of its wave (tools/wave_trace.py)
//   PRL_ENV_PERM               wave slot -> env through a host-set permutation (tools/tail_experiment.py)
//   PRL_CUT=<1|2|4|5|6>       instruction-count builds: no observation / nor painting / nor hook point / nor ray / no
//                              shots at all (wrong results;
//                              counter differences between them give each phase's instructions, tools/pmc_cuts.sh)
//   PRL_NO_PRIO, PRL_PRIO_SLOT=<s>  no s_setprio by progress / youngest-wave bias from hardware slot s (prl_step.hpp)
//   PRL_PRIO_HI=<n>, PRL_PRIO_C, PRL_PRIO_ROTATE   other splits of the four priority levels over a step's phases (A/B)
//   round-4 A/B switches, each measured in profiles/r04_ab_log.txt (the default is the faster form):
//     PRL_FACET_TILE, PRL_RECORD_GATHER, PRL_WIDE_FACET_LOAD, PRL_PAINT_PIPELINE, PRL_NT_MASKS, PRL_STAGGER, PRL_OBS_SCALAR_MASKS
//                              opt-in forms that lost (LDS tile, gathers through LDS, whole-record scalar test, pipelined
//                              painter, non-temporal masks, staggered starts, observation pass 3 on scalar lane masks)
//     PRL_STORE_ALL_WORDS, PRL_PLAIN_DIVISION, PRL_EXACT_OUTWARD, PRL_NO_OUTLINE_MISS
//                              the plain forms of shortcuts that are on (whole mask rows written back, the compiler's float64
//                              division, nextafterf for outward rounding, the general ray search without the outline's
//                              miss certificate); the last three are part of the forced-general parity variant
//   PRL_OLD_BALLOT             HIP's __ballot instead of the builtin (prl_device.hpp)
//   PRL_WALK_STEPS, PRL_CONE_JOINT_FROM, PRL_FINE_CELL, PRL_HG_CELL, PRL_BEAM_OCC, PRL_BEAM_WAVES, PRL_FAR_OCC, PRL_FAR_WGS, PRL_FAR_WAVES,
//   PRL_REST_WGS, PRL_BFS_LANES   tuning constants of the cone-beam painter (prl_cone.hpp, paintrl_hip.hip, k_cone_beams.hip);
//   PRL_CONE_F64_RECORDS, PRL_CONE_OUTLINE_IN_BEAMS: A/B switches; PRL_CONE_CUT_NN, PRL_CONE_CUT_FAR: timing builds
//                              (wrong results: the beams kernel without its search / without its list pushes)
//   PRL_CONE_TRACE             path counters of the cone-beam kernels (= 2: wave-time histograms alone; tools/cone_stats.py;
//                              = 4: per-wave phase times of the far kernel, plain stores, no atomics; tools/far_trace.py)
//   PRL_FRAG_TIMING            policy phase of act_step_kernel per barrier, rollout kernels' phases (tools/fragment_timing.py)
//   PRL_PAINT_ONE_ROW_PER_TRIP one sample-grid row per trip of the painter (multi-trip path)   } the general paths,
//   PRL_FORCE_FULL_SCANS       whole-table scans instead of the ring searches                } run by the forced-
//   PRL_FORCE_GENERAL_RAY      general two-stage ray search instead of the convex fast path  } path parity tests
//   PRL_FORCE_F64_PAINT        every paint word through the float64 distance test (no float pre-filter)
//   PRL_WIDE_PAINT_BAND        pre-filter band x 4096: the mixed float / float64 path runs constantly
// The switches are read where the fast path is chosen (prl_paint.hpp, prl_search.hpp, prl_ray.hpp).
#pragma once

namespace {

#ifdef PRL_WAVE_TRACE
// Per wave: [start, end] in s_memrealtime ticks (100 MHz, one clock for the whole device) and eight 8-bit path
// counters bumped where a slow path is entered (WCNT slots: 0 general ray search, 1 its second stage, 2 chunk
// visits, 3 vertex-ring trips, 4 neighbourhood ray rounds, 5 paint trips, 6 straddle / f64 trips, 7 single-facet hits).
#define PRL_TRACE_ENVS 8192
__device__ unsigned long long g_wave_trace[4 * PRL_TRACE_ENVS];
__shared__ unsigned long long g_wcnt[4], g_wcnt16[4];
__device__ unsigned long long g_wave_trace16[PRL_TRACE_ENVS];      // four 16-bit counters of the cone-beam painter
#define WCNT16(slot, v)                                                                            \
    do {                                                                                           \
        if ((threadIdx.x & 63) == 0) g_wcnt16[threadIdx.x >> 6] += (unsigned long long)(v) << (16 * (slot)); \
    } while (0)
#define WCNT(slot, v)                                                                              \
    do {                                                                                           \
        if ((threadIdx.x & 63) == 0) g_wcnt[threadIdx.x >> 6] += (unsigned long long)(v) << (8 * (slot)); \
    } while (0)
#define TRACE_BEGIN()                                                          \
    const unsigned long long trace_t0 = __builtin_amdgcn_s_memrealtime();      \
    if ((threadIdx.x & 63) == 0) g_wcnt[threadIdx.x >> 6] = g_wcnt16[threadIdx.x >> 6] = 0
#define TRACE_END(env, dn)                                                     \
    do {                                                                       \
        if ((threadIdx.x & 63) == 0 && (env) < PRL_TRACE_ENVS) {               \
            g_wave_trace[4 * (env)] = trace_t0;                                \
            g_wave_trace[4 * (env) + 1] = __builtin_amdgcn_s_memrealtime();    \
            g_wave_trace[4 * (env) + 2] = g_wcnt[threadIdx.x >> 6];            \
            g_wave_trace16[env] = g_wcnt16[threadIdx.x >> 6];                  \
            g_wave_trace[4 * (env) + 3] = (unsigned long long)((dn) != 0) | ((unsigned long long)__builtin_amdgcn_s_getreg(63492) << 1) | \
                                          ((unsigned long long)(__builtin_amdgcn_s_getreg(63508) & 15) << 40);   /* HW_ID, XCC_ID */ \
        }                                                                      \
    } while (0)
#else
#define WCNT(slot, v)
#define WCNT16(slot, v)
#define TRACE_BEGIN() \
    do {              \
    } while (0)
#define TRACE_END(env, dn) \
    do {                   \
    } while (0)
#endif

#ifdef PRL_ENV_PERM
// diagnostic build only (tools/tail_experiment.py): wave slot -> env through a permutation the host sets per launch, to measure
// what a launch gains when the envs of a SIMD are chosen by their expected cost
__device__ const int *g_env_perm;
#endif

#ifdef PRL_FRAG_TIMING
// Per wave of the rollout-fragment kernel, summed over waves and steps (s_memrealtime ticks, 10 ns): [0] policy
// phase incl. its barriers, [1] env step, [2] wait at the barrier after the step, [3] wave-steps counted.
__device__ unsigned long long g_frag_ticks[4];
__device__ unsigned long long g_pol_stamps[8];          // policy_forward of workgroup 0: time at each barrier
#define POL_STAMP(k)                                                                  \
    do {                                                                              \
        if (blockIdx.x == 0 && threadIdx.x == 0) g_pol_stamps[k] = __builtin_amdgcn_s_memrealtime(); \
    } while (0)
#define FRAG_T(var) const unsigned long long var = __builtin_amdgcn_s_memrealtime()
#define FRAG_DECL() unsigned long long frag_acc[3] = {0, 0, 0}, frag_n = 0
#define FRAG_ACC(slot, t0, t1) frag_acc[slot] += (t1) - (t0)
#define FRAG_COUNT() ++frag_n
#define FRAG_FLUSH()                                                           \
    do {                                                                       \
        if ((threadIdx.x & 63) == 0) {                                         \
            for (int k_ = 0; k_ < 3; ++k_) atomicAdd(&g_frag_ticks[k_], frag_acc[k_]); \
            atomicAdd(&g_frag_ticks[3], frag_n);                               \
        }                                                                      \
    } while (0)
#else
#define POL_STAMP(k) \
    do {            \
    } while (0)
#define FRAG_T(var) \
    do {            \
    } while (0)
#define FRAG_ACC(slot, t0, t1) \
    do {                       \
    } while (0)
#define FRAG_DECL() \
    do {            \
    } while (0)
#define FRAG_COUNT() \
    do {             \
    } while (0)
#define FRAG_FLUSH() \
    do {             \
    } while (0)
#endif

#ifdef PRL_NO_PRIO                       // A/B switches for the progress-based issue priority (prl_step.hpp)
#define PRIO_BY_PROGRESS(p)
#define PRIO_YOUNG_DECL()
#define PRIO_YOUNG_OLD(y, o, ph)
#else                                    // by progress; the two youngest waves of a SIMD (hardware wave slots 2, 3) one
#ifndef PRL_PRIO_SLOT                    // level up from shot 2 on: 41.31 -> 41.05 us
#define PRL_PRIO_SLOT 2
#endif
#define PRIO_BY_PROGRESS(p) __builtin_amdgcn_s_setprio(p)
#define PRIO_YOUNG_DECL() const int prio_slot = __builtin_amdgcn_s_getreg(63492) & 15
#ifdef PRL_PRIO_ROTATE                   // A/B: the favoured pair of slots rotates with the phase (ph = 2 .. 6: shots 2-4, painting, observation)
#define PRIO_IS_YOUNG(ph) (((prio_slot + (ph)) & 2) != 0)
#elif defined(PRL_PRIO_MASK)             // A/B: the favoured slots as a bit mask (0xa: slots 1 and 3, 0x3: slots 0 and 1, ...)
#define PRIO_IS_YOUNG(ph) (((PRL_PRIO_MASK) >> (prio_slot & 3)) & 1)
#else
#define PRIO_IS_YOUNG(ph) (prio_slot >= PRL_PRIO_SLOT)
#endif
#define PRIO_YOUNG_OLD(y, o, ph)                   \
    do {                                           \
        if (PRIO_IS_YOUNG(ph)) __builtin_amdgcn_s_setprio(y); \
        else __builtin_amdgcn_s_setprio(o);        \
    } while (0)
#endif

#if defined(PRL_WAVE_TRACE) && !defined(PRL_PHASE_TIMING)
// The trace build also splits each wave's life by phase: the STAMP points of the step add the s_memrealtime ticks (10 ns)
// since the previous stamp to lane <phase> of ONE vector register (a lane move each way, no memory, two scalar registers for
// the last stamp) -- the per-phase atomics of PRL_PHASE_TIMING tripled the kernel's run time.  Stored per env at PROF_END.
enum { PH_LOAD = 0, PH_RAY, PH_VERTEX, PH_BARY, PH_MATH, PH_BALL, PH_APPLY, PH_OBS, PH_STORE, PH_COUNT };
__device__ unsigned g_wave_phase[16 * PRL_TRACE_ENVS];
struct Prof {
    unsigned acc;       // lane k: ticks of phase k
    unsigned prev;      // wave-uniform
};
#define PROF_ARG , Prof &prof
#define PROF_PASS , prof
#define PROF_BEGIN()                                          \
    Prof prof;                                                \
    prof.acc = 0;                                             \
    prof.prev = (unsigned)__builtin_amdgcn_s_memrealtime()
#if defined(__HIP_DEVICE_COMPILE__)
#define STAMP(ph)                                                                                        \
    do {                                                                                                 \
        const unsigned now_ = (unsigned)__builtin_amdgcn_s_memrealtime();                                \
        const unsigned cur_ = (unsigned)__builtin_amdgcn_readlane((int)prof.acc, (ph));                  \
        const unsigned new_ = (unsigned)__builtin_amdgcn_readfirstlane((int)(cur_ + (now_ - prof.prev)));  \
        asm volatile("v_writelane_b32 %0, %1, %2" : "+v"(prof.acc) : "s"(new_), "n"(ph));                 \
        prof.prev = now_;                                                                                \
    } while (0)
#else                                    // (the host pass of the compiler only has to parse the device functions)
#define STAMP(ph) \
    do {          \
    } while (0)
#endif
#define PROF_END() \
    do {           \
    } while (0)
#define PROF_STORE(env)                                                                                  \
    do {                                                                                                 \
        if ((env) < PRL_TRACE_ENVS && (threadIdx.x & 63) < 16) g_wave_phase[16 * (env) + (threadIdx.x & 63)] = prof.acc; \
    } while (0)
#elif defined(PRL_PHASE_TIMING)
#define PROF_STORE(env) \
    do {                \
    } while (0)
// (cdna_hip_programming.md "In-kernel stamps"): the stamps go to a buffer nothing else reads.  ONE phase is
// accumulated per build (-DPRL_PHASE_TIMING=<phase index>, or -1 for the whole kernel): the accumulator and the
// previous stamp then fit in four scalar registers and the build keeps the product kernel's register allocation
// (nine accumulators pushed it into scratch and tripled its run time).
enum { PH_LOAD = 0, PH_RAY, PH_VERTEX, PH_BARY, PH_MATH, PH_BALL, PH_APPLY, PH_OBS, PH_STORE, PH_COUNT };
__device__ unsigned long long g_phase_cycles[16];
struct Prof {
    unsigned long long acc, prev, t0;
};
#define PROF_ARG , Prof &prof
#define PROF_PASS , prof
#define PROF_BEGIN()                               \
    Prof prof;                                     \
    prof.acc = 0;                                  \
    prof.prev = __builtin_amdgcn_s_memtime();      \
    prof.t0 = prof.prev
#define PROF_END()                                                                             \
    do {                                                                                       \
        if ((threadIdx.x & 63) == 0) {                                                         \
            atomicAdd(&g_phase_cycles[0], prof.acc);                                           \
            atomicAdd(&g_phase_cycles[1], (unsigned long long)__builtin_amdgcn_s_memtime() - prof.t0); \
        }                                                                                      \
    } while (0)
#define STAMP(ph)                                                         \
    do {                                                                  \
        const unsigned long long now_ = __builtin_amdgcn_s_memtime();     \
        if ((ph) == (PRL_PHASE_TIMING)) prof.acc += now_ - prof.prev;     \
        prof.prev = now_;                                                 \
    } while (0)
#else
#define PROF_ARG
#define PROF_PASS
#define PROF_BEGIN() \
    do {             \
    } while (0)
#define PROF_END() \
    do {           \
    } while (0)
#define PROF_STORE(env) \
    do {                \
    } while (0)
#define STAMP(ph) \
    do {          \
    } while (0)
#endif

}  // namespace
