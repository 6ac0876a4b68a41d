// prl_diag.hpp -- every diagnostic hook of the device code, in one place.
//
// The product build defines none of the PRL_* diagnostic macros: STAMP / WCNT / PROF_* expand to nothing and the
// kernels carry no instrumentation.  Diagnostic builds (tools/, tests/test_gpu_forced_paths.py) may define:
//   PRL_PHASE_TIMING=<k>       s_memtime deltas of phase k summed over all waves (tools/phase_timing.py)
//   PRL_PAINT_ONE_ROW_PER_TRIP one sample-grid row per trip of the painter (multi-trip path)   } the general paths,
//   PRL_FORCE_FULL_SCANS       whole-table scans instead of the ring searches                } run by the forced-
//   PRL_FORCE_GENERAL_RAY      general two-stage ray search instead of the convex fast path  } path parity tests
//   PRL_FORCE_F64_PAINT        every paint word through the float64 distance test (no float pre-filter)
//   PRL_WIDE_PAINT_BAND        pre-filter band x 4096: the mixed float / float64 path runs constantly
// The switches are read where the fast path is chosen (prl_paint.hpp, prl_search.hpp, prl_ray.hpp).
#pragma once

namespace {

#define WCNT(slot, v)

#ifdef PRL_PHASE_TIMING
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
#define STAMP(ph) \
    do {          \
    } while (0)
#endif

}  // namespace
