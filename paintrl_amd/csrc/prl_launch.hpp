// prl_launch.hpp -- how the host side of the C ABI (paintrl_hip.hip) reaches the kernels.
//
// The library is built from several translation units so that hipcc compiles them side by side (paintrl_amd/build.py):
// k_step.hip, k_cone.hip and k_rollout.hip are each compiled once per mask width (-DPRL_KW=1..4: 64-bit mask words
// per lane), k_big.hip holds the LDS-mask kernels of parts beyond 16 384 samples (KW = 0).  Every unit includes the
// same device headers (anonymous namespace: each gets its own copy of the device functions) and exports plain
// launchers named prl_k<KW>_<what>; argument structs travel as const void * (the struct types live in those anonymous
// namespaces, with one definition in the shared headers).  A launcher returns a hipError_t as int.
#pragma once

#include <cstddef>

#define PRL_HIDDEN extern "C" __attribute__((visibility("hidden")))

struct PrlStepSel {                  // which instantiation of the step kernel a batch takes
    int gensec;                      // atan2-sector observation (OBS_GRAD != 4 in section / discrete mode)
    int hsi;                         // COLOR_MODE 'HSI'
    int kd;                          // some part carries the reference's stale vertex kd-tree
    int wide;                        // eight envs per workgroup (the whole launch is resident at once)
    int grid;                        // OBS_MODE 'grid': the per-step kernel's grid-only build (the others carry every mode but grid)
};

#define PRL_K_PROTOS(KW)                                                                                               \
    PRL_HIDDEN int prl_k##KW##_step(const void *step_args, const PrlStepSel *sel, void *stream);                       \
    PRL_HIDDEN int prl_k##KW##_step_occupancy(const void *step_args, const PrlStepSel *sel, int out[3]);               \
    PRL_HIDDEN int prl_k##KW##_reset(const void *step_args, int gensec, void *stream);                                 \
    PRL_HIDDEN int prl_k##KW##_observe(const void *step_args, int gensec, void *stream);                               \
    PRL_HIDDEN int prl_k##KW##_reset_obs(const void *part_dev, const void *cfg_dev, double *out, int n_start,          \
                                         int n_words, int gensec);                                                     \
    PRL_HIDDEN int prl_k##KW##_cone(const void *step_args, const PrlStepSel *sel, void *stream);                       \
    PRL_HIDDEN int prl_k##KW##_act_step(const void *act_step_args, size_t policy_lds, int flags, void *stream);           \
    PRL_HIDDEN int prl_k##KW##_rollout_policy(const void *policy_fragment_args, size_t policy_lds, int flags, void *stream); \
    PRL_HIDDEN int prl_k##KW##_rollout_fragment(const void *fragment_args, int flags, void *stream);

// PAINT_METHOD 'normal': the tool path and the beams of a step (k_cone_beams.hip); prl_k<KW>_cone finishes it
PRL_HIDDEN int prl_kc_path(const void *step_args, int kd, int wide, void *stream);
PRL_HIDDEN int prl_kc_beams(const void *step_args, void *stream);

PRL_K_PROTOS(0)
PRL_K_PROTOS(1)
PRL_K_PROTOS(2)
PRL_K_PROTOS(3)
PRL_K_PROTOS(4)

// In a kernel unit: KFN(step) -> prl_k3_step for -DPRL_KW=3.
#define PRL_CAT3_(a, b, c) a##b##c
#define PRL_CAT3(a, b, c) PRL_CAT3_(a, b, c)
#define KFN(what) PRL_CAT3(prl_k, PRL_KW, _##what)
